#!/bin/bash
# A/B of library builds on one box: lone-wave latency (1x1 and 8x8 frames) and the headline frame, two passes.
# usage: scripts/ab_walk.sh libA.so libB.so ...
for pass in 1 2; do
  for lib in "$@"; do
    echo "== $lib (pass $pass)"
    BRT_LIB_PATH="$(realpath "$lib")" timeout 300 python3 scripts/lone_wave_time.py 0 256 8 2>&1 | grep -E "^(1x1|8x8)" | cut -c1-150
    BRT_LIB_PATH="$(realpath "$lib")" SWEEP_REPS=8 timeout 300 python3 scripts/sweep_env.py 0 1920 1080 64 8 "" 2>&1 | tail -1
  done
done
