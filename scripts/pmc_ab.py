#!/usr/bin/env python3
"""Instruction counters of the headline frame for several builds of the library on one box:
    python scripts/pmc_ab.py bevyray_amd/libA.so bevyray_amd/libB.so ...
One rocprofv3 --pmc pass per library (bench.live_pmc over scripts/pmc_frame.py; BRT_LIB_PATH selects the library the child
loads).  Prints VALU / SALU / LDS / branch instructions, average active lanes and wave cycles of the production kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SETS = ["SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES"]
if os.environ.get("PMC_AB_WAIT"):
    SETS.append("SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM")
workload = os.environ.get("PMC_AB_WORKLOAD", bench.PMC_WORKLOAD_TAG)
for lib in sys.argv[1:]:
    os.environ["BRT_LIB_PATH"] = os.path.realpath(lib)
    got, why = bench.live_pmc(timeout_s=300.0, workload=workload, passes=SETS)
    if not got:
        print(f"{lib}: no counters ({why})")
        continue
    v = got["SQ_INSTS_VALU"]
    lanes = got["SQ_THREAD_CYCLES_VALU"] / got["SQ_ACTIVE_INST_VALU"] * 1.0 if got.get("SQ_ACTIVE_INST_VALU") else float("nan")
    print(f"{lib}: VALU {v / 1e9:.3f} G  SALU {got['SQ_INSTS_SALU'] / 1e9:.3f} G  LDS {got['SQ_INSTS_LDS'] / 1e9:.3f} G  "
          f"branch {got['SQ_INSTS_BRANCH'] / 1e9:.3f} G  thread-cycles/active-inst {lanes:.2f}  wave cycles {got['SQ_WAVE_CYCLES'] / 1e9:.2f} G"
          + (f"  wait_any {got['SQ_WAIT_ANY'] / 1e9:.2f} G wait_inst_any {got['SQ_WAIT_INST_ANY'] / 1e9:.2f} G active_any {got['SQ_ACTIVE_INST_ANY'] / 1e9:.2f} G"
             if "SQ_WAIT_ANY" in got else "") + (f"  ({why})" if why else ""), flush=True)
