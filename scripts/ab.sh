#!/bin/bash
# A/B timing of library builds on ONE box: scripts/ab.sh "<args for gpu_time.py>" ab/libA.so ab/libB.so ...
# (each build is copied over bevyray_amd/libbevyray_amd.so in turn, two passes, so that box-to-box variance cancels)
args="$1"; shift
cp bevyray_amd/libbevyray_amd.so /tmp/lib_orig.so
for pass in 1 2; do
  for lib in "$@"; do
    cp "$lib" bevyray_amd/libbevyray_amd.so
    echo "== $lib (pass $pass)"
    timeout 120 python scripts/gpu_time.py $args 2>&1 | grep "^rep" | tail -n +2 | awk '{s+=$4; n++; if(m==0||$4<m)m=$4} END{printf "   mean %.2f ms  min %.2f ms over %d reps\n", s/n, m, n}'
  done
done
cp /tmp/lib_orig.so bevyray_amd/libbevyray_amd.so
