import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time, os
import bevyray_amd as brt
rng=np.random.default_rng(1)
with brt.RaytracePlugin([0]) as p:
    for n in (506, 2000, 4096, 8192, 10004, 16384):
        m=np.zeros(n, brt.MODEL_DTYPE); m["position"]=rng.uniform(-50,50,(n,3)).astype(np.float32); m["radius"]=rng.uniform(0.05,0.6,n).astype(np.float32)
        res=[]
        for mode in ("100000000", "0"):
            p.set_tuning("BRT_PLOC_ONE_BLOCK_MAX", int(mode))
            p.build_bvh(m)
            best=min(p.build_bvh(m)[1] for _ in range(5))
            res.append(best)
        print(f"n={n}: one workgroup {res[0]:.3f} ms, grid {res[1]:.3f} ms", flush=True)
