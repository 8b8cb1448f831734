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
    return _lib


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
