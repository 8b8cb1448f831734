import sys, time
sys.path.insert(0,'/root/repo')
import rust_renderer_amd as rr
W,H=1920,1080
scene = rr.scenes.scene_for_config(1, tex_size=1024)
r = rr.Renderer(W,H); scene.upload(r)
loop = rr.FrameLoop(r, scene.make_view(W,H))
for i in range(3):
    t=time.perf_counter(); loop.frame(rr.PASS_REFERENCE_PT); r.synchronize(); print("frame %d %.3f ms" % (i,(time.perf_counter()-t)*1e3), flush=True)
s = r.get_stats()
print("aborts", s.fused_aborts, "rays", list(s.rays), "hits", s.closest_hits, "misses", s.misses, flush=True)
