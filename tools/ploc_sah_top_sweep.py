"""Option "ploc_sah_top": PLOC rounds stop at K clusters and a host SAH tree over them is the top - rebuild time, traversal quality, frame time."""
import sys, time
sys.path.insert(0, '/root/repo')
import rust_renderer_amd as rr
W, H = 1920, 1080
for cfg, tops in ((1, (0, 1024, 4096, 8192, 16384, 32768, 131072)), (3, (0, 8192, 65536))):
    scene = rr.scenes.scene_for_config(cfg, tex_size=64)
    for top in tops:
        r = rr.Renderer(W, H)
        r.set_option("device_build", 1)
        r.set_option("ploc_sah_top", top)
        scene.upload(r)
        t0 = time.perf_counter(); r.set_instance_transform(0, rr.identity3x4()); r.initialize_raytracing(); again = (time.perf_counter() - t0) * 1e3
        r.set_option("count_visits", 1)
        loop = rr.FrameLoop(r, scene.make_view(W, H, use_ris_light_sampling=0))
        loop.frame(rr.PASS_REFERENCE_PT)
        c = r.get_stats()
        cl = c.rays[0] + c.rays[1]; sh = c.rays[2] + c.rays[3]
        r.set_option("count_visits", 0)
        r.reset_accumulation(); r.reset_stats()
        loop = rr.FrameLoop(r, scene.make_view(W, H, use_ris_light_sampling=0))
        loop.frames(8, rr.PASS_REFERENCE_PT); r.synchronize()
        t0 = time.perf_counter(); loop.frames(32, rr.PASS_REFERENCE_PT); r.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / 32
        print("config %d sah_top %7d nodes %8d levels? | rebuild %6.1f ms | closest nodes/ray %.2f tris/ray %.2f shadow %.2f / %.2f | %.3f ms/frame" % (
            cfg, top, r.get_stats().bvh_nodes, again, c.nodes_visited / cl, c.tris_tested / cl, c.shadow_nodes_visited / max(sh, 1), c.shadow_tris_tested / max(sh, 1), ms), flush=True)
        del loop, r
