#!/bin/bash
# Builds bevyray_amd/lib<NAME>.so from the current sources with extra compiler flags, for scripts/ab.sh:
#   scripts/build_variant.sh NAME "-DBRT_SHARED_RCP=0"      (objects under bevyray_amd/csrc/build_<NAME>/)
set -e
name="$1"; flags="$2"
cd "$(dirname "$0")/../bevyray_amd/csrc"
rm -rf "build_$name"; mkdir -p "build_$name"
for f in brt_api.cpp brt_host.cpp brt_kernels.hip brt_trace_prod.hip brt_trace_tune.hip brt_bvh.hip brt_order.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
      -fno-fast-math -fno-slp-vectorize $flags -c -o "build_$name/${f%.*}.o" "$f" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "../lib$name.so" build_$name/*.o
rm -rf "build_$name"
ls -la "../lib$name.so"
