#!/bin/bash
# Collects the round's profile artefacts on the GPU box (run through gpurun):  scripts/collect_profiles.sh TAG
#   1. rocprofv3 --kernel-trace --stats of the bench command (per-kernel average durations)
#   2. PMC counters of the headline workload and of config 4 (separate --pmc passes) + the device-code hash they were taken on
#   3. the BASELINE.json config table (scripts/run_configs.py), section profiles, strong-scaling emulation, lone-wave latency
# Outputs under gpurun_out/TAG; copy the summaries into profiles/rNN afterwards.
# A second argument selects a part (the whole takes ~20 minutes; a gpurun call is limited to 20):  pmc | rest | all (default)
TAG=${1:-profiles}
PART=${2:-all}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$PART" != "rest" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
find $OUT/ktrace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/ktrace
cd $GRAFT_REPO_ROOT
python3 - <<PY
import json, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
from bevyray_amd import _lib
SETS = [
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES",
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM",
  "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE GRBM_COUNT",
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAVES SQ_IFETCH"]
for tag, name, sets, budget in ((bench.PMC_WORKLOAD_TAG, "pmc_summary.json", SETS, 600.0), (bench.PMC_WORKLOAD4_TAG, "pmc_summary_config4.json", SETS[:2], 400.0),
                               (bench.PMC_WORKLOAD3_TAG, "pmc_summary_config3.json", SETS[:4], 300.0),
                               (bench.PMC_WORKLOAD5_TAG, "pmc_summary_config5.json", SETS[:4] + bench.PMC_PASSES_MEMORY[1:] + [SETS[5]], 400.0)):
    got, why = bench.live_pmc(timeout_s=budget, workload=tag, passes=sets)
    rec = {"workload": tag, "n_gpus": 1, "kernel_code_hash": _lib.kernel_code_hash(), "counters": got, "note": why,
           "method": "rocprofv3 --pmc, one pass per counter set over scripts/pmc_frame.py, values of the LAST k_trace_persistent "
                     "dispatch; FETCH_SIZE/WRITE_SIZE in KB (bench.py doubles FETCH_SIZE: gfx950 counts 64 B per 128-B request)"}
    json.dump(rec, open("$OUT/" + name, "w"), indent=1)
    print(json.dumps(rec)[:300])
PY
fi
cd $GRAFT_REPO_ROOT
if [ "$PART" = "pmc" ]; then exit 0; fi
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python3 scripts/run_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
python3 scripts/gpu_time.py --reps 4 > $OUT/section_profile.txt 2>&1
python3 scripts/gpu_time.py --scene 2 --reps 3 > $OUT/section_profile_config5.txt 2>&1
python3 scripts/gpu_time.py --scene 1 --camera rtiow --spp 256 --bounces 50 --reps 3 > $OUT/section_profile_config3.txt 2>&1
python3 scripts/gpu_parts.py > $OUT/parts_emulation.txt 2>&1
python3 scripts/lone_wave_time.py 0 512 8 > $OUT/lone_wave_cover.txt 2>&1
python3 scripts/lone_wave_time.py 1 512 50 > $OUT/lone_wave_rtiow.txt 2>&1
python3 scripts/sah_time.py > $OUT/sah_build_time.txt 2>&1
python3 scripts/moving_camera_time.py > $OUT/moving_camera.txt 2>&1
cut -c1-200 $OUT/configs.jsonl; head -5 $OUT/kernel_stats.csv; cat $OUT/parts_emulation.txt
