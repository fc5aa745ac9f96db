"""Half-sample jobs at the end of the dispatch order: kernel ms of 8 frames of config 2 per BRT_SPLIT_TAIL setting, the order's meta (tiles split,
second halves that took the state over / left the pixel to the first-half lane), and -- with a library built with -DBRT_LIFE_HIST (BRT_LIB_PATH) -- the span of
the last launch, the mean wave lifetime and the histogram of the lifetimes."""
import os, sys
sys.path.insert(0, os.getcwd())
import bevyray_amd as brt
scene = int(sys.argv[1]) if len(sys.argv) > 1 else 0         # usage: split_diag.py [scene w h spp bounces [settings ...]]
W, H, spp, b = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (1920, 1080, 64, 8)
sc=brt.generate_scene(scene,1)
lvl,cam,win=(brt.rtiow_camera if scene == 1 else brt.cover_camera)(W,H,spp,b)
SETTINGS = [{"BRT_SPLIT_TAIL": int(x)} if x != "default" else {} for x in sys.argv[6:]] or [{"BRT_SPLIT_TAIL": 0}, {"BRT_SPLIT_TAIL": 4}, {"BRT_SPLIT_TAIL": 8}, {"BRT_SPLIT_TAIL": 12}, {}, {"BRT_SPLIT_TAIL": 24}, {"BRT_SPLIT_TAIL": 32}]
for knobs in SETTINGS:
    with brt.RaytracePlugin([0]) as p:
        for k, v in knobs.items(): p.set_tuning(k, v)
        out=p.alloc_frame(W,H)
        p.node.write_buffers(brt.Buffers(sc.models, sc.materials, None))
        ks=[]
        for i in range(int(os.environ.get('DIAG_FRAMES', '8'))):
            p.node.run(lvl,cam,win,W,H,out=out); ks.append(round(p.node.last_stats["kernel_ms"],3))
        p.debug_profile()
        print(knobs, ks, p.last_order_meta, p.node.last_stats["rays"], p.node.last_stats.get("kernel_variant"), flush=True)
        tl = getattr(p, "last_timeline", None)
        if tl: print("   end_ms", tl["end_ms"], "mean_life", tl["mean_wave_life_ms"], "hist(0.082ms from 7.2ms)", tl["wave_life_hist_0.33ms"], flush=True)
