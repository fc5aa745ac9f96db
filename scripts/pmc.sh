#!/bin/bash
# Collects PMC counters for the trace kernel in separate passes (gpurun refuses --pmc combined
# with tracing flags).  Usage (on the GPU box, via gpurun): bash scripts/pmc.sh <tag> [gpu_time args]
TAG=${1:-pmc}
shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/scripts/gpu_time.py --reps 2 $@"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVE_DEP_WAIT SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- $CMD > $OUT/pass$i.log 2>&1
  tail -2 $OUT/pass$i.log
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        tot[k]["_n_"+r["Counter_Name"]] += 1
for k, d in tot.items():
    print(k)
    for c in sorted(d):
        if not c.startswith("_n_"):
            print("   %-26s %18.0f  (per dispatch %16.0f, n=%d)" % (c, d[c], d[c]/d["_n_"+c], d["_n_"+c]))
PY
