#!/bin/bash
# A/B timing of library builds on ONE box: scripts/ab.sh "<scene w h spp bounces>" libA.so libB.so ...
# (BRT_LIB_PATH selects the build, bevyray_amd/_lib.py; two passes, so that clock ramp and box variance cancel)
args="${1:-0 1920 1080 64 8}"; shift
for pass in 1 2; do
  for lib in "$@"; do
    echo "== $lib (pass $pass)"
    BRT_LIB_PATH="$(realpath "$lib")" SWEEP_REPS=8 timeout 300 python scripts/sweep_env.py $args "" 2>&1 | tail -1
  done
done
