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
    """MAX over pixels of |rgb_a - rgb_b|_2 on linear radiance: the bound holds for every pixel, so a handful of
    wrong pixels cannot hide in a frame-wide average (image_rms is that average, reported beside it)."""
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    return float(np.sqrt(np.sum(d * d, axis=-1)).max()) if d.size else 0.0


def image_rms(a, b):
    """sqrt(mean over pixels of |rgb_a - rgb_b|^2): the frame-wide figure, never the pass criterion."""
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    return float(np.sqrt(np.mean(np.sum(d * d, axis=-1)))) if d.size else 0.0


def random_rays(scene_bounds, n, seed, tmin=0.001, tmax=10000.0):
    lo, hi = (np.asarray(x, dtype=np.float32) for x in scene_bounds)
    u = rr.scenes.hash_floats(seed, 6 * n).reshape(n, 6)
    o = lo + u[:, :3] * (hi - lo)
    d = u[:, 3:] * 2.0 - 1.0
    d[np.abs(d).sum(axis=1) == 0] = 1.0
    rays = np.empty((n, 8), dtype=np.float32)
    rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, tmin, d, tmax
    return rays


def torture_scene():
    """coincident quads in different meshes (exact t ties), zero-area triangles, a very large and very
    small triangle, a fan sharing one vertex, triangles under scaled / mirrored instance transforms"""
    from rust_renderer_amd.scenes import Mesh, Model, Scene, _pack_vertices

    def mesh(pos, idx, transform=None, mtype=0):
        pos = np.asarray(pos, dtype=np.float32)
        nrm = np.tile(np.float32([0, 0, 1]), (len(pos), 1))
        uv = np.zeros((len(pos), 2), dtype=np.float32)
        m = Mesh(_pack_vertices(pos, nrm, uv), np.asarray(idx, dtype=np.uint32).reshape(-1), mtype)
        if transform is not None:
            m.transform = np.asarray(transform, dtype=np.float32).reshape(12)
        return m

    quad = [[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]]
    qi = [0, 1, 2, 0, 2, 3]
    meshes = [
        mesh(quad, qi),                                                   # mesh 0
        mesh(quad, qi),                                                   # mesh 1: coincident with mesh 0
        mesh(quad, [0, 2, 3, 0, 1, 2]),                                   # mesh 2: coincident, different diagonal order
        mesh([[0, 0, 1], [0, 0, 1], [0, 0, 1], [1, 0, 1], [2, 0, 1]], [0, 1, 2, 0, 3, 4]),  # zero-area (point, line)
        mesh([[-1e5, -1e5, -3], [1e5, -1e5, -3], [0, 1e5, -3]], [0, 1, 2]),               # huge
        mesh([[0.25, 0.25, 0.5], [0.250001, 0.25, 0.5], [0.25, 0.250001, 0.5]], [0, 1, 2]),  # tiny
        mesh([[0, 0, 2]] + [[np.cos(a), np.sin(a), 2] for a in np.linspace(0, 2 * np.pi, 13)], sum(([0, k, k + 1] for k in range(1, 13)), [])),  # fan
        mesh(quad, qi, rr.transform3x4((0.5, -2.0, 1.0), (0.1, 0.2, -1.0))),           # mirrored + scaled instance
        mesh(quad, qi, rr.transform3x4((1e-3, 1e-3, 1.0), (0.3, -0.3, 0.75))),         # strongly scaled instance
    ]
    from rust_renderer_amd.camera import Camera
    return Scene("torture", [(Model(meshes, []), None)], [(0.0, 0.0, 5.0)], Camera((0, 0, 6), (0, 0, 0), 60.0, 1.0, 0.01, 1000.0))
