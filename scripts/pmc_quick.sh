#!/bin/bash
# One PMC pass with the instruction-mix counters of the trace kernel. Usage: bash scripts/pmc_quick.sh <tag> [env assignments...]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH --output-format csv -d $OUT/p -- python3 $GRAFT_REPO_ROOT/scripts/gpu_time.py --reps 1 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "persistent" in r["Kernel_Name"]:
            k = "counters" if "true>(" in r["Kernel_Name"].replace(" ", "").replace("ELb1EEE", "") and r["Kernel_Name"].rstrip(")").split("<")[1].split(">")[0].replace(" ", "").endswith("true") else "timing"
            tot[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in tot.items():
    print("$TAG", k, {c: "%.3e" % (sum(v)/len(v)) for c, v in sorted(d.items())})
PY
