#!/bin/bash
# HBM traffic of the trace kernel for bench.py's workload, as MI355X_MICROARCH.md prescribes:
# separate --pmc passes for FETCH_SIZE and WRITE_SIZE (KiB units), read side doubled on gfx950.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-traffic}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/$C.log 2>&1
done
python3 - <<PY
import csv, glob, json
v = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    xs = []
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if "k_trace_persistent" in r["Kernel_Name"] and r["Counter_Name"] == c:
                xs.append(float(r["Counter_Value"]))
    xs = xs[1:] if len(xs) > 1 else xs      # drop the first (counters) launch
    v[c] = sum(xs) / max(1, len(xs))
out = {"workload": "cover_1920x1080_64spp_8b", "n_gpus": 1, "fetch_size_kib_per_launch": v["FETCH_SIZE"],
       "write_size_kib_per_launch": v["WRITE_SIZE"],
       "hbm_bytes_per_launch": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py --steps 5; mean over the timed "
                 "k_trace_persistent dispatches; KiB units; FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads)"}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
print(json.dumps(out))
PY
