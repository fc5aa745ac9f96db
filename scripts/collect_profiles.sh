#!/bin/bash
# Collects the round's profile artefacts on the GPU box (run through gpurun): 
#   1. rocprofv3 --kernel-trace --stats of the bench command (per-kernel average durations)
#   2. PMC counters of the headline workload (separate --pmc passes) + the device-code hash they were taken on
#   3. the BASELINE.json config table (scripts/run_configs.py), section profile of the headline frame
# Outputs under gpurun_out/$1; copy the summaries into profiles/rNN afterwards.
TAG=${1:-profiles}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
find $OUT/ktrace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
cd $GRAFT_REPO_ROOT
python3 - <<PY
import json, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
from bevyray_amd import _lib
bench.PMC_PASSES = [
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES",
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM",
  "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE GRBM_COUNT",
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAVES SQ_IFETCH"]
got, why = bench.live_pmc(timeout_s=600.0)
rec = {"workload": bench.PMC_WORKLOAD_TAG, "n_gpus": 1, "kernel_code_hash": _lib.kernel_code_hash(), "counters": got, "note": why,
       "method": "rocprofv3 --pmc, one pass per counter set over scripts/pmc_frame.py (3 frames), values of the LAST k_trace_persistent "
                 "dispatch; FETCH_SIZE/WRITE_SIZE in KB (bench.py doubles FETCH_SIZE: gfx950 counts 64 B per 128-B request)"}
json.dump(rec, open("$OUT/pmc_summary.json", "w"), indent=1)
print(json.dumps(rec)[:400])
PY
python3 scripts/run_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
python3 scripts/gpu_time.py --reps 3 > $OUT/section_profile.txt 2>&1
python3 scripts/gpu_time.py --scene 2 --reps 3 > $OUT/section_profile_config5.txt 2>&1
tail -3 $OUT/configs.jsonl; cat $OUT/kernel_stats.csv | head -5
