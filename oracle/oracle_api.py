"""ctypes binding of liboracle.so — TEST INFRASTRUCTURE ONLY (see oracle/oracle.cpp header).
Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the package.
Reuses the product's host-side Renderer class over the orc_* entry points so that parity tests
drive both implementations through the same calls."""
import ctypes as C
import os
import subprocess

import numpy as np

import rust_renderer_amd as rr
from rust_renderer_amd.api import CApi, Renderer
from rust_renderer_amd.types import ViewUniformData

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build_oracle(force=False):
    src = os.path.join(_HERE, "oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "utopian_hip.h")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def build_native():
    """-O3 -march=native build for the machine this runs on (bench.py's cpu_baseline leg); falls back to the portable
    library when no compiler is there. Must be called before the first lib()."""
    global LIB_PATH
    try:
        subprocess.run(["make", "-C", _HERE, "-B", "native"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        LIB_PATH = os.path.join(_HERE, "_native", "liboracle.so")
        return True
    except Exception:
        return False


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.endswith(os.path.join("_native", "liboracle.so")):
            build_oracle()
        _lib = C.CDLL(LIB_PATH)
        L = _lib
        L.orc_create.argtypes, L.orc_create.restype = [C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)], C.c_int
        L.orc_jenkins_hash.argtypes, L.orc_jenkins_hash.restype = [C.c_uint32], C.c_uint32
        L.orc_init_rng.argtypes, L.orc_init_rng.restype = [C.c_uint32] * 4, C.c_uint32
        L.orc_random_float.argtypes, L.orc_random_float.restype = [C.POINTER(C.c_uint32)], C.c_float
        L.orc_random_point_in_unit_sphere.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        L.orc_frame_number.argtypes, L.orc_frame_number.restype = [C.POINTER(ViewUniformData)], C.c_uint32
        L.orc_offset_ray.argtypes = [C.POINTER(C.c_float)] * 3
        L.orc_linear_to_srgb.argtypes, L.orc_linear_to_srgb.restype = [C.c_float], C.c_float
        L.orc_luminance.argtypes, L.orc_luminance.restype = [C.POINTER(C.c_float)], C.c_float
        L.orc_sky.argtypes = [C.POINTER(C.c_float)] * 4
        L.orc_target_function.argtypes, L.orc_target_function.restype = [C.c_void_p, C.c_int, C.POINTER(C.c_float)], C.c_float
        L.orc_primary_ray.argtypes = [C.POINTER(ViewUniformData), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.orc_sample_texture.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.orc_closest_hit_shader.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        L.orc_hardware_threads.restype = C.c_int
        L.orc_marching_cubes.argtypes = [C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_marching_cubes.restype = C.c_uint64
        L.orc_mc_density.argtypes, L.orc_mc_density.restype = [C.c_float] * 4, C.c_float
        L.orc_mc_normal.argtypes = [C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float)]
    return _lib


_MC_TABLES = None


def mc_reference_tables():
    """edgeTable[256], triangleTable[256][16] of the reference's tables.glsl, as data (tests/golden/mc_reference_tables.npz)"""
    global _MC_TABLES
    if _MC_TABLES is None:
        t = np.load(os.path.join(_HERE, "..", "tests", "golden", "mc_reference_tables.npz"))
        _MC_TABLES = (np.ascontiguousarray(t["edge_table"], dtype=np.int32), np.ascontiguousarray(t["triangle_table"], dtype=np.int32))
    return _MC_TABLES


def marching_cubes(resolution, lo, hi, time=0.0, order=0, positions=True):
    """the oracle's restatement of marching_cubes.comp:179-254 on the reference's tables:
    -> dict(cube_index (res^3,) uint8, tri_count (res^3,) uint8, positions (T, 3, 3) float32 in voxel order, x fastest)"""
    edge, tri = mc_reference_tables()
    cells = int(resolution) ** 3
    cube, count = np.zeros(cells, dtype=np.uint8), np.zeros(cells, dtype=np.uint8)
    L = lib()
    total = L.orc_marching_cubes(resolution, lo, hi, time, edge.ctypes.data, tri.ctypes.data, order, cube.ctypes.data, count.ctypes.data, None, 0)
    pos = np.zeros((total, 3, 3), dtype=np.float32)
    if positions and total:
        L.orc_marching_cubes(resolution, lo, hi, time, edge.ctypes.data, tri.ctypes.data, order, None, None, pos.ctypes.data, total)
    return dict(cube_index=cube, tri_count=count, positions=pos, triangles=int(total))


def mc_density(p, time=0.0):
    return lib().orc_mc_density(float(p[0]), float(p[1]), float(p[2]), float(time))


def _fv(values):
    return (C.c_float * len(values))(*[float(x) for x in values])


class OracleRenderer(Renderer):
    """Same host interface as rust_renderer_amd.Renderer, backed by the CPU oracle."""

    backend = "oracle"

    def __init__(self, width, height, threads=0, brute_force=False):
        L = lib()

        def factory(w, h):
            ctx = C.c_void_p()
            assert L.orc_create(w, h, C.byref(ctx)) == 0
            return ctx

        super().__init__(width, height, _api=CApi(L, "orc_"), _ctx_factory=factory)
        self._num_lights = 0
        if threads:
            self.set_option("threads", threads)
        if brute_force:
            self.set_option("brute_force", 1)

    def add_light(self, position, color=(1, 1, 1), range_=1.0):
        self._num_lights += 1
        return super().add_light(position, color, range_)

    def add_gpu_light(self, light):
        self._num_lights += 1
        return super().add_gpu_light(light)

    def add_isosurface_mesh(self, resolution, lo, hi, time=0.0, material=None, world3x4=None):
        """the counterpart of uh_add_isosurface_mesh on the checker's side: the mesh of the reference's marching cubes as the
        oracle restates it (marching_cubes.comp:179-254 on the reference's tables, normals by generateNormal :160-177; nothing
        dropped), added as one mesh. Returns (mesh index or None, triangle count)."""
        mc = marching_cubes(resolution, lo, hi, time)
        if mc["triangles"] == 0:
            return None, 0
        pos = mc["positions"].reshape(-1, 3)
        nrm = np.zeros_like(pos)
        L = lib()
        fp = C.POINTER(C.c_float)
        for i in range(len(pos)):  # small grids only: the parity tests hand the oracle the device's own mesh for image comparisons
            L.orc_mc_normal(pos[i].ctypes.data_as(fp), time, nrm[i].ctypes.data_as(fp))
        v = np.zeros(len(pos), dtype=rr.types.VERTEX_DTYPE)
        v["pos"][:, :3], v["pos"][:, 3] = pos, 1.0
        v["normal"][:, :3] = nrm
        v["uv"] = pos[:, [0, 2]] / np.float32(hi - lo)
        v["color"] = 1.0
        if material is None:
            material = rr.make_material(base_color=(0.8, 0.8, 0.8, 1.0), diffuse_map=self.default_diffuse_map())
        return self.add_mesh(v, np.arange(len(pos), dtype=np.uint32), material, world3x4), mc["triangles"]

    # unit entry points -------------------------------------------------------------------
    def target_function(self, light_index, p):
        return lib().orc_target_function(self._ctx, light_index, _fv(p))

    def sample_texture(self, tex, u, v):
        out = (C.c_float * 3)()
        lib().orc_sample_texture(self._ctx, tex, u, v, out)
        return np.array(out[:], dtype=np.float32)

    def closest_hit_shader(self, mesh, prim, t, u, v, direction, seed):
        s = C.c_uint32(seed)
        out = (C.c_float * 11)()
        lib().orc_closest_hit_shader(self._ctx, mesh, prim, t, u, v, _fv(direction), C.byref(s), out)
        return np.array(out[:], dtype=np.float32), s.value


def jenkins_hash(x):
    return lib().orc_jenkins_hash(x)


def init_rng(px, py, resx, frame):
    return lib().orc_init_rng(px, py, resx, frame)


def random_floats(state, n):
    s = C.c_uint32(state)
    vals = [lib().orc_random_float(C.byref(s)) for _ in range(n)]
    return np.array(vals, dtype=np.float32), s.value


def random_point_in_unit_sphere(state):
    s = C.c_uint32(state)
    out = (C.c_float * 3)()
    lib().orc_random_point_in_unit_sphere(C.byref(s), out)
    return np.array(out[:], dtype=np.float32), s.value


def frame_number(view):
    return lib().orc_frame_number(C.byref(view))


def offset_ray(p, n):
    out = (C.c_float * 3)()
    lib().orc_offset_ray(_fv(p), _fv(n), out)
    return np.array(out[:], dtype=np.float32)


def linear_to_srgb(x):
    return lib().orc_linear_to_srgb(x)


def luminance(rgb):
    return lib().orc_luminance(_fv(rgb))


def sky(origin, direction, sun):
    out = (C.c_float * 3)()
    lib().orc_sky(_fv(origin), _fv(direction), _fv(sun), out)
    return np.array(out[:], dtype=np.float32)


def primary_ray(view, W, H, px, py, jx, jy):
    out = (C.c_float * 6)()
    lib().orc_primary_ray(C.byref(view), W, H, px, py, jx, jy, out)
    return np.array(out[:], dtype=np.float32)


def hardware_threads():
    return lib().orc_hardware_threads()
