"""Per-ray traversal statistics of the closest-hit kernel (diagnostic batch kernel: node / triangle visits per ray), config-1 scene."""
import sys
sys.path.insert(0, '/root/repo')
import rust_renderer_amd as rr
W, H = 1920, 1080
scene = rr.scenes.scene_for_config(1, tex_size=64)
r = rr.Renderer(W, H)
for kv in sys.argv[1:]:
    k, v = kv.split('=')
    r.set_option(k, int(v))
scene.upload(r)
r.set_option("count_visits", 1)
loop = rr.FrameLoop(r, scene.make_view(W, H))
loop.frame(rr.PASS_REFERENCE_PT)
s = r.get_stats()
cl = s.rays[0] + s.rays[1]
sh = s.rays[2] + s.rays[3]
print("closest rays", cl, "nodes/ray %.2f tris/ray %.2f" % (s.nodes_visited / cl, s.tris_tested / cl))
print("shadow rays", sh, "nodes/ray %.2f tris/ray %.2f" % (s.shadow_nodes_visited / sh, s.shadow_tris_tested / sh))
