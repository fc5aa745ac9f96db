#!/bin/bash
# Round-3 measurement set on one MI355X (run through gpurun): scripts/r03_measure.sh TAG
#   bench line (no extras), strong-scaling emulation (one rank's share of configs 2 and 4 on one GPU),
#   configs 3 and 5, section profile of the headline frame.  Outputs under gpurun_out/TAG.
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-pmc --no-extras > $OUT/bench.json 2> $OUT/bench.err &&
python3 scripts/gpu_parts.py > $OUT/parts_emulation.txt 2>&1 &&
python3 scripts/run_configs.py 3 5 > $OUT/configs.jsonl 2> $OUT/configs.err &&
python3 scripts/gpu_time.py --reps 4 > $OUT/section_profile.txt 2>&1
cat $OUT/bench.json | cut -c1-600; cat $OUT/parts_emulation.txt; cat $OUT/configs.jsonl | cut -c1-300; tail -12 $OUT/section_profile.txt
