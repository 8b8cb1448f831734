"""uh_refit_acceleration vs uh_build_acceleration on the config-2 scene, and what a refit after moving
the two spheres costs the traversal (nodes/ray)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import rust_renderer_amd as rr
W, H = 1920, 1080
scene = rr.scenes.scene_for_config(1, tex_size=64)
r = rr.Renderer(W, H)
scene.upload(r)
print("build_ms %.1f  tris %d nodes %d" % (r.get_stats().build_ms, r.get_stats().bvh_triangles, r.get_stats().bvh_nodes))
n = scene.num_meshes


def frame_stats(tag):
    r.set_option("count_visits", 1)
    r.reset_accumulation(); r.reset_stats()
    loop = rr.FrameLoop(r, scene.make_view(W, H))
    loop.frame(rr.PASS_REFERENCE_PT)
    s = r.get_stats()
    cl = s.rays[0] + s.rays[1]
    print("%s: closest nodes/ray %.2f tris/ray %.2f frame %.3f ms" % (tag, s.nodes_visited / cl, s.tris_tested / cl, s.last_frame_ms))
    r.set_option("count_visits", 0)


frame_stats("built")
for i in range(3):
    t0 = time.perf_counter(); r.rebuild_tlas(); dt = (time.perf_counter() - t0) * 1e3
    print("refit (nothing moved) %.3f ms wall, stats %.3f" % (dt, r.get_stats().build_ms))
frame_stats("refit in place")
r.set_instance_transform(n - 1, rr.transform3x4((1, 1, 1), (3.0, 0.5, 1.0)))
r.set_instance_transform(n - 2, rr.transform3x4((1, 1, 1), (-4.0, 1.5, -1.0)))
t0 = time.perf_counter(); r.rebuild_tlas(); dt = (time.perf_counter() - t0) * 1e3
print("refit (2 meshes moved) %.3f ms" % dt)
frame_stats("refit, spheres moved")
t0 = time.perf_counter(); r.initialize_raytracing(); dt = (time.perf_counter() - t0) * 1e3
print("full rebuild %.1f ms" % dt)
frame_stats("rebuilt, spheres moved")
