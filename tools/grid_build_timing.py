#!/usr/bin/env python3
"""The device builds of the two grids (config 1 at 1080p): milliseconds as the library reports them (UhStats), five sun directions /
cameras each - run once per HIP runtime (UH_HIP_RUNTIME=torch for the PyTorch wheel's) to see what the runtime's allocator costs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_renderer_amd as rr  # noqa: E402

W, H = 1920, 1080
scene = rr.scenes.scene_for_config(1, tex_size=64)
r = rr.Renderer(W, H)
scene.upload(r)
print(rr.version() if hasattr(rr, "version") else "", rr.hip_versions())
import numpy as np
for k in range(5):
    v = scene.make_view(W, H)
    d = np.array([0.3 + 0.1 * k, 0.8, 0.2 - 0.05 * k]); d /= np.linalg.norm(d)
    v.sun_dir[0], v.sun_dir[1], v.sun_dir[2] = d
    loop = rr.FrameLoop(r, v)
    loop.frames(8, rr.PASS_REFERENCE_PT)
    r.synchronize()
    loop.frames(8, rr.PASS_REFERENCE_PT)
    r.synchronize()
    s = r.get_stats()
    print("sun grid %.2f ms (%d entries, %d cells)  camera grid %.2f ms" % (s.sun_grid_build_ms, s.sun_grid_entries, s.sun_grid_cells, s.camera_grid_build_ms))
