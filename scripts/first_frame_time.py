import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt
W,H,spp,b=1920,1080,64,8
sc=brt.generate_scene(brt.SCENE_COVER,1); other=brt.generate_scene(brt.SCENE_COVER,2)
lvl,cam,win=brt.cover_camera(W,H,spp,b)
# kernel ms of the first frame of a view + its dispatch-order pre-pass, four fresh views per setting
SETTINGS = [("prepass 2 spp", {"BRT_PREPASS_SPP": "2"})]
for k in (3, 4, 5, 6, 8):
    SETTINGS.append((f"prepass {k} spp", {"BRT_PREPASS_SPP": str(k)}))
SETTINGS += [("no prepass (raster order)", {"BRT_PREPASS_SPP": "0"})]
for label, env in SETTINGS:
    with brt.RaytracePlugin([0]) as p:
        for k, v in env.items(): p.set_tuning(k, int(v))
        out=p.alloc_frame(W,H)
        res=[]
        for rep in range(4):
            p.node.write_buffers(brt.Buffers(other.models, other.materials, None)); p.node.write_buffers(brt.Buffers(sc.models, sc.materials, None)); p.set_tuning("BRT_LPT", 1)   # (setting a knob forgets the view's history)
            p.node.run(lvl,cam,win,W,H,out=out); s=p.node.last_stats
            res.append((s["kernel_ms"], s["prepass_ms"]))
        print(label, " ".join(f"{k:.2f}+{q:.2f}={k+q:.2f}" for k,q in res), flush=True)
