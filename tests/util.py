"""Shared helpers for the parity tests: drive the HIP path (through the C ABI) and the CPU oracle
through the same host calls on the same seeded scene."""
import numpy as np

import oracle_api as oa
import rust_renderer_amd as rr

L2_TOL = 1e-3  # BASELINE.json north_star: per-pixel L2 <= 1e-3 on linear radiance


def make_pair(scene, W, H, **oracle_kw):
    gpu = scene.upload(rr.Renderer(W, H, device=0))
    cpu = scene.upload(oa.OracleRenderer(W, H, **oracle_kw))
    return gpu, cpu


def run_frames(renderer, scene, W, H, frames, pass_mask=rr.PASS_ALL, **view_overrides):
    loop = rr.FrameLoop(renderer, scene.make_view(W, H, **view_overrides))
    for _ in range(frames):
        loop.frame(pass_mask)
    return loop


def per_pixel_l2(a, b):
    """sqrt(mean over pixels of |rgb_a - rgb_b|^2) on linear radiance."""
    return float(np.sqrt(np.mean(np.sum((a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)) ** 2, axis=-1))))


def random_rays(scene_bounds, n, seed, tmin=0.001, tmax=10000.0):
    lo, hi = (np.asarray(x, dtype=np.float32) for x in scene_bounds)
    u = rr.scenes.hash_floats(seed, 6 * n).reshape(n, 6)
    o = lo + u[:, :3] * (hi - lo)
    d = u[:, 3:] * 2.0 - 1.0
    d[np.abs(d).sum(axis=1) == 0] = 1.0
    rays = np.empty((n, 8), dtype=np.float32)
    rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, tmin, d, tmax
    return rays
