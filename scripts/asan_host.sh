#!/bin/bash
# The library's HOST side (brt_api.cpp, brt_interop.cpp, brt_host.cpp: validation, encoders, both CPU BVH builders, tree reach rule,
# store-format tables, RCCL loader) compiled by g++ with AddressSanitizer + UndefinedBehaviorSanitizer, linked with the hipcc-built
# kernel objects, and the CPU test suite run against it (BRT_LIB_PATH).  CPU only: the GPU boxes run no sanitizers.
#   bash scripts/asan_host.sh            -> all passed (the address-space-cap test skips itself), 0 sanitizer reports expected
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
src="$root/bevyray_amd/csrc"
out="${TMPDIR:-/tmp}/brt_asan"
mkdir -p "$out"
make -s -C "$src" -j8
for f in brt_api brt_interop brt_host; do
    g++ -std=c++17 -O1 -g -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-omit-frame-pointer \
        -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Wall -Wextra -Wno-unused-parameter -c -o "$out/$f.o" "$src/$f.cpp"
done
g++ -shared -fPIC -fsanitize=address,undefined -o "$out/libbrt_asan.so" "$out"/brt_api.o "$out"/brt_interop.o "$out"/brt_host.o \
    "$src"/build/brt_kernels.o "$src"/build/brt_trace_prod.o "$src"/build/brt_trace_tune.o "$src"/build/brt_bvh.o "$src"/build/brt_sah.o \
    "$src"/build/brt_order.o -L/opt/rocm/lib -lamdhip64 -ldl
asan="$(g++ -print-file-name=libasan.so)"
ubsan="$(g++ -print-file-name=libubsan.so)"
cd "$root"
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD="$asan $ubsan" BRT_LIB_PATH="$out/libbrt_asan.so" \
    python -m pytest tests -q -s -m "not gpu" -p no:cacheprovider > "$out/run.log" 2>&1 || true
tail -n 1 "$out/run.log"
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' "$out/run.log" || true)"
# ... and the checker itself: the oracle under the same sanitizers (IEEE division by zero is part of the shader's arithmetic: 1 / d),
# on the golden fixture and on scripts/fuzz_parity.py's adversarial cases
gcc -O1 -g -std=c11 -ffp-contract=off -fno-fast-math -fPIC -pthread -fsanitize=address,undefined -fno-sanitize=float-divide-by-zero \
    -shared -o "$out/liboracle_asan.so" "$root/oracle/bevyray_oracle.c" -lm
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD="$asan $ubsan" ORACLE_ASAN="$out/liboracle_asan.so" python - > "$out/oracle.log" 2>&1 <<'PY' || true
import ctypes as C, os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "scripts")
import numpy as np, bevyray_amd as brt, oracle_loader, fuzz_parity as F
from helpers import fixture_buffers
o = oracle_loader.Oracle(C.CDLL(os.environ["ORACLE_ASAN"]))
b, lvl, cam, win, frame, counters = fixture_buffers()
got, cnt = o.render(b, lvl, cam, win, 64, 36)
assert np.array_equal(got.view(np.uint32), np.asarray(frame, np.float32).view(np.uint32))
rng, n = np.random.default_rng(11), 0
for i in range(150):
    c = F.random_case(rng)
    bb = c["buffers"]
    if len(bb.models) > 3000:
        continue
    if bb.bvh is None:
        bb = brt.Buffers(bb.models, bb.materials, brt.build_bvh_sah(bb.models))
    o.render(bb, c["level"], c["camera"], c["window"], c["w"], c["h"], raster_rgba=c["raster"], raster_depth=c["depth"])
    n += 1
x = np.random.default_rng(1).random((50, 50, 4)).astype(np.float32) * 2 - 0.5
for f in ("srgb8", "unorm8", "f16"):
    o.encode_frame(x, f)
print(f"oracle: golden fixture + {n} adversarial cases")
PY
tail -n 1 "$out/oracle.log"
echo "sanitizer reports (oracle): $(grep -c 'runtime error\|AddressSanitizer' "$out/oracle.log" || true)"
