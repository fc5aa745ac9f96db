#!/usr/bin/env python3
"""rocprofv3 --pmc passes over scripts/pmc_frame.py for a list of counter sets (one child process per set; the
program stands directly behind `--`); prints {counter: value of the LAST k_trace_persistent dispatch} as JSON.
usage: pmc_sets.py [--scene N] [--env K=V ...] -- "SET ONE" "SET TWO" ...      (run on the GPU box)"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (live_pmc only; importing bench touches no GPU)


def main():
    argv = sys.argv[1:]
    env = {}
    while argv and argv[0] != "--":
        a = argv.pop(0)
        if a == "--scene":
            env["BRT_PMC_SCENE"] = argv.pop(0)
        elif a == "--env":
            k, v = argv.pop(0).split("=", 1)
            env[k] = v
    sets = argv[1:] if argv else []
    os.environ.update(env)
    if env:
        os.environ["BRT_ENABLE_TUNING"] = "1"      # knobs come from the environment once, at brt_create, on request
    bench.PMC_PASSES = sets or bench.PMC_PASSES
    got, why = bench.live_pmc(timeout_s=600.0)
    print(json.dumps({"env": env, "counters": got, "note": why}, indent=1), flush=True)


if __name__ == "__main__":
    main()
