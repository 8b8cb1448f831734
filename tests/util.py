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


class DeviceBuffer:
    """a plain device allocation for tests that hand the library raw device pointers (uh_pack_tiles / uh_compose_tiles): hipMalloc
    through ctypes on the HIP runtime the library itself bound (soname libamdhip64.so.7: the object already in the process) - no
    torch in a GPU test process"""

    _hip = None

    @classmethod
    def hip(cls):
        if cls._hip is None:
            import ctypes as C

            rr.load_library()
            cls._hip = C.CDLL("libamdhip64.so.7")
            cls._hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            cls._hip.hipFree.argtypes = [C.c_void_p]
            cls._hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
            cls._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        return cls._hip

    def __init__(self, nbytes):
        import ctypes as C

        p = C.c_void_p()
        assert self.hip().hipMalloc(C.byref(p), max(int(nbytes), 1)) == 0
        self.ptr, self.nbytes = p.value, int(nbytes)
        assert self.hip().hipMemset(self.ptr, 0, max(self.nbytes, 1)) == 0 and self.hip().hipDeviceSynchronize() == 0

    def copy_from(self, other, nbytes, dst_offset=0, src_offset=0):
        assert self.hip().hipMemcpy(self.ptr + dst_offset, other.ptr + src_offset, nbytes, 3) == 0  # hipMemcpyDeviceToDevice
        assert self.hip().hipDeviceSynchronize() == 0

    def zero(self, offset, nbytes):
        assert self.hip().hipMemset(self.ptr + offset, 0, nbytes) == 0 and self.hip().hipDeviceSynchronize() == 0

    def free(self):
        if self.ptr:
            self.hip().hipFree(self.ptr)
            self.ptr = None

    __del__ = free


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


# ---- independent float64 numpy restatements for the iso-surface checks (test infrastructure; they used to live in the product
# package as its host fallback): the reference's density field and a marching-TETRAHEDRA extraction of its zero set - a
# different algorithm than the marching cubes of csrc/isosurface.hip and of the oracle (oracle.cpp orc_marching_cubes), so the
# three can be held against each other
def reference_density(p):
    """marching_cubes.comp:83-103 at view.time = 0: density = max(-1, -sdTorus, -sdBox, -sdSphere(radius 0)) with the torus (R 5, r 3,
    axis y) above the box (half size 5); positive inside. The zero-radius sphere's term -|p - c| only wins within 1 of its centre
    and reaches 0 at the centre point alone."""
    q = p - np.array([16.0, 20.0, 16.0])
    torus = np.sqrt((np.sqrt(q[..., 0] ** 2 + q[..., 2] ** 2) - 5.0) ** 2 + q[..., 1] ** 2) - 3.0
    d = np.abs(p - np.array([16.0, 10.0, 16.0])) - 5.0
    box = np.minimum(np.maximum(d[..., 0], np.maximum(d[..., 1], d[..., 2])), 0.0) + np.sqrt((np.maximum(d, 0.0) ** 2).sum(-1))
    sphere = np.sqrt(((p - np.array([16.0, 26.0, 16.0])) ** 2).sum(-1))
    return np.maximum(np.maximum(np.maximum(-torus, -box), -1.0), -sphere)


# the 6 tetrahedra of a cube around its 0-6 diagonal (corner numbering of marching_cubes.rs:23-32)
_CUBE_CORNERS = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]])
_TETS = np.array([[0, 5, 1, 6], [0, 1, 2, 6], [0, 2, 3, 6], [0, 3, 7, 6], [0, 7, 4, 6], [0, 4, 5, 6]])


def extract_isosurface(density, lo, hi, resolution, slab=16):
    """triangle soup of {density = 0} on a resolution^3 grid over [lo, hi]^3 by marching tetrahedra
    (host-side stand-in for the GPU marching-cubes extraction of SURVEY.md section 8f N3)."""
    h = (hi - lo) / resolution
    tris = []
    ax = lo + h * np.arange(resolution + 1)
    for z0 in range(0, resolution, slab):
        z1 = min(z0 + slab, resolution)
        X, Y, Z = np.meshgrid(ax, ax, ax[z0 : z1 + 1], indexing="ij")
        P = np.stack([X, Y, Z], -1)
        D = density(P)
        inside = D > 0
        c = inside[:-1, :-1, :-1]
        mixed = np.zeros_like(c)
        cnt = np.zeros(c.shape, dtype=np.int8)
        for dx, dy, dz in _CUBE_CORNERS:
            cnt += inside[dx : dx + resolution, dy : dy + resolution, dz : dz + (z1 - z0)]
        mixed = (cnt > 0) & (cnt < 8)
        ix, iy, iz = np.nonzero(mixed)
        if len(ix) == 0:
            continue
        cp = np.stack([P[ix + dx, iy + dy, iz + dz] for dx, dy, dz in _CUBE_CORNERS], 1)  # (n, 8, 3)
        cv = np.stack([D[ix + dx, iy + dy, iz + dz] for dx, dy, dz in _CUBE_CORNERS], 1)  # (n, 8)
        for tet in _TETS:
            p, v = cp[:, tet], cv[:, tet]
            ins = v > 0
            k = ins.sum(1)
            order = np.argsort(~ins, axis=1, kind="stable")  # inside vertices first
            p = np.take_along_axis(p, order[..., None], 1)
            v = np.take_along_axis(v, order[:, :], 1)

            def cut(a, b, sel):
                t = (v[sel, a] / (v[sel, a] - v[sel, b]))[:, None]
                return p[sel, a] + t * (p[sel, b] - p[sel, a])

            s1, s2, s3 = k == 1, k == 2, k == 3
            if s1.any():
                tris.append(np.stack([cut(0, 1, s1), cut(0, 2, s1), cut(0, 3, s1)], 1))
            if s3.any():
                tris.append(np.stack([cut(0, 3, s3), cut(1, 3, s3), cut(2, 3, s3)], 1))
            if s2.any():
                a, b, c2, d2 = cut(0, 2, s2), cut(0, 3, s2), cut(1, 3, s2), cut(1, 2, s2)
                tris.append(np.stack([a, b, c2], 1))
                tris.append(np.stack([a, c2, d2], 1))
    return np.concatenate(tris) if tris else np.zeros((0, 3, 3))
