import sys
sys.path.insert(0,'/root/repo')
import rust_renderer_amd as rr
W,H=1920,1080
scene = rr.scenes.scene_for_config(1, tex_size=1024)
r = rr.Renderer(W,H); r.set_option("count_visits",1); scene.upload(r)
loop = rr.FrameLoop(r, scene.make_view(W,H))
for _ in range(3): loop.frame(rr.PASS_REFERENCE_PT); r.synchronize()
r.reset_stats()
for _ in range(4): loop.frame(rr.PASS_REFERENCE_PT); r.synchronize()
s = r.get_stats()
t,w,sh = s.light_nodes_visited, s.light_tris_tested, s.sun_covered_rays
tot=t+w+sh
print("wave clock: trace %.3f  barrier wait %.3f  shade %.3f  (ticks per wave per frame: %.0f)" % (t/tot, w/tot, sh/tot, tot/4/4096))
