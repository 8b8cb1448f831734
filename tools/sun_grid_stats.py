import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rust_renderer_amd as rr
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
scene = rr.scenes.scene_for_config(cfg, tex_size=64)
W, H = (256, 256) if cfg == 0 else (1920, 1080)
sweeps = ([("sun_grid", 0)], [], [("sun_grid_density", 12)], [("sun_grid_density", 200), ("sun_grid_max_mb", 2048)])
if len(sys.argv) > 2 and sys.argv[2] == "lists":
    sweeps = [[("sun_grid", 0)], [("sun_grid_force", 1)], [("sun_grid_force", 1), ("sun_grid_density", 96), ("sun_grid_max_mb", 2048)]]
elif len(sys.argv) > 2 and sys.argv[2] == "fallback":
    sweeps = [[("sun_grid", 0)], [("sun_grid_force", 1)]]
elif len(sys.argv) > 2 and sys.argv[2] == "walk":
    sweeps = [[("sun_grid_max_walk", w)] for w in (8, 16, 32, 64, 128)]
for opts in sweeps:
    r = rr.Renderer(W, H)
    for k, v in opts: r.set_option(k, v)
    scene.upload(r)
    r.set_option("count_visits", 1)
    loop = rr.FrameLoop(r, scene.make_view(W, H))
    loop.frame(rr.PASS_REFERENCE_PT)
    s = r.get_stats()
    print(opts, "cells", s.sun_grid_cells, "entries", s.sun_grid_entries, "build ms", s.sun_grid_build_ms, "mean list", s.sun_grid_mean_list,
          "tests/ray", s.shadow_tris_tested / s.rays[2], "rays", s.rays[2], "handed to the tree", s.sun_tree_rays)
    r.set_option("count_visits", 0)
    r.set_option("time_kernels", 1)
    for k, v in (("frames_in_flight", 1), ("overlap", 0)): r.set_option(k, v)
    loop.frames(16, rr.PASS_REFERENCE_PT)
    r.reset_stats()
    loop.frames(16, rr.PASS_REFERENCE_PT)
    s = r.get_stats()
    print("   serial ms/frame: closest %.3f shadow %.3f shade %.3f" % (s.trace_closest_ms / 16, s.trace_shadow_ms / 16, s.shade_ms / 16))
    r.close()
