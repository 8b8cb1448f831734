"""PLOC search radius against tree quality and build time (config-1 scene, 1080p)."""
import sys, time
sys.path.insert(0, '/root/repo')
import rust_renderer_amd as rr
W, H = 1920, 1080
scene = rr.scenes.scene_for_config(1, tex_size=64)
for radius in (4, 8, 16, 32, 64):
    r = rr.Renderer(W, H)
    r.set_option("device_build", 1)
    r.set_option("ploc_radius", radius)
    scene.upload(r)
    t0 = time.perf_counter(); r.set_instance_transform(0, rr.identity3x4()); r.initialize_raytracing(); again = (time.perf_counter() - t0) * 1e3
    r.set_option("count_visits", 1)
    loop = rr.FrameLoop(r, scene.make_view(W, H))
    loop.frame(rr.PASS_REFERENCE_PT)
    c = r.get_stats()
    cl = c.rays[0] + c.rays[1]; sh = c.rays[2] + c.rays[3]
    print("radius %2d: rebuild %6.1f ms | closest nodes/ray %.2f tris/ray %.2f shadow %.2f / %.2f" % (radius, again, c.nodes_visited / cl, c.tris_tested / cl, c.shadow_nodes_visited / sh, c.shadow_tris_tested / sh), flush=True)
