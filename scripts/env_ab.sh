#!/bin/bash
# Timing of env-knob variants on ONE box: scripts/env_ab.sh "<gpu_time args>" "A=1 B=2" "C=3" ...  ("" = defaults)
args="$1"; shift
for pass in 1 2; do
  for kv in "$@"; do
    echo "== [$kv] (pass $pass)"
    env $kv timeout 120 python scripts/gpu_time.py $args 2>&1 | grep "^rep" | tail -n +2 | awk '{s+=$4; n++; if(m==0||$4<m)m=$4; g=$14" "$15} END{printf "   mean %.2f ms  min %.2f ms over %d reps  %s\n", s/n, m, n, g}'
  done
done
