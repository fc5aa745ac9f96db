#!/bin/bash
# HBM traffic + L2 hit rate of the trace kernel on config 5 (10k-sphere scene, scene NOT in LDS).
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_cfg5}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/scripts/run_configs.py 5 > $OUT/$tag.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
v = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_trace_persistent" in r["Kernel_Name"] and "false>(" in r["Kernel_Name"].replace(" ", "")[-200:]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(x)/len(x) for k, x in v.items()}
out["n_dispatches"] = {k: len(x) for k, x in v.items()}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["hbm_bytes_per_launch"] = (2*out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024
if "TCC_HIT_sum" in out:
    out["l2_hit_rate"] = out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"])
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out))
PY
