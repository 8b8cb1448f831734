import sys
sys.path.insert(0,'/root/repo')
import numpy as np
import rust_renderer_amd as rr
for cfg,W,H in ((1,1920,1080),(2,1920,1080),(4,1920,1080),(0,256,256),(1,3840,2160),(3,1920,1080),(1,2560,1440)):
    kw = dict(tex_size=256)
    scene = rr.scenes.scene_for_config(cfg, **kw)
    mask = rr.PASS_ALL if cfg == 2 else rr.PASS_REFERENCE_PT
    out=[]
    for fused in (-1,0):
        r = rr.Renderer(W,H); r.set_option("fused_bounces", fused); scene.upload(r)
        loop = rr.FrameLoop(r, scene.make_view(W,H))
        for _ in range(3):
            loop.frame(mask); r.synchronize()
        out.append((r.read_accumulation().view(np.uint32), r.read_output_bgra8(), list(r.get_stats().rays), r.get_stats().closest_hits, r.get_stats().misses))
        del r
    a,b=out
    ok = np.array_equal(a[0],b[0]) and np.array_equal(a[1],b[1]) and a[2:]==b[2:]
    print("config %d %dx%d: %s rays %s" % (cfg,W,H,"bit-identical" if ok else "MISMATCH", a[2]), flush=True)
