#!/usr/bin/env python3
"""Where the waves of k_path_fused spend their clock (a library built with -DUH_FUSED_PROFILE: tools/build_variant.py prof -DUH_FUSED_PROFILE;
UTOPIAN_HIP_LIB=rust-renderer_amd/libuh_prof.so python3 tools/fused_phase_profile.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_renderer_amd as rr  # noqa: E402

W, H = 1920, 1080
scene = rr.scenes.scene_for_config(int(sys.argv[1]) if len(sys.argv) > 1 else 1, tex_size=1024)
r = rr.Renderer(W, H)
r.set_option("count_visits", 1)
r.set_option("camera_grid", 0)  # (its counters carry the shading wave's clock here)
scene.upload(r)
loop = rr.FrameLoop(r, scene.make_view(W, H))
for _ in range(3):
    loop.frame(rr.PASS_REFERENCE_PT)
    r.synchronize()
r.reset_stats()
N = 4
for _ in range(N):
    loop.frame(rr.PASS_REFERENCE_PT)
    r.synchronize()
s = r.get_stats()
walk, work, room = s.light_nodes_visited, s.light_tris_tested, s.sun_covered_rays
shade_all, shade_idle = s.camera_grid_tris_tested, s.camera_tree_rays
print("walking waves: waiting for work %.3f, for room in their rings %.3f of their clock (%.0f ticks per wave per frame at 100 MHz)" % (work / walk, room / walk, walk / N / 3072))
print("shading waves: idle %.3f of their clock (%.0f ticks per wave per frame)" % (shade_idle / shade_all, shade_all / N / 1024))
