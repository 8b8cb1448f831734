"""Host-side mirror of the reference's plugin surface for the path-tracing + ReSTIR path, over
the C ABI of include/utopian_hip.h (ctypes; no torch types cross the boundary).

Reference verbs mirrored (same names / argument meaning):
  Renderer.add_model / add_light / get_num_lights      utopian/src/renderer.rs:222,391,412
  Renderer.initialize_raytracing                       Raytracing::initialize, utopian/src/raytracing.rs:89
  build_path_tracing_render_graph pass order           utopian/src/renderers/mod.rs:246-358
  FrameLoop (total_samples / prev_frame_projection_view protocol)   prototype/src/main.rs:460-471,545-546
Errors: the reference panics on every failure (graph.rs:253, raytracing.rs:178); here every
non-zero status raises UtopianError carrying uh_last_error().
"""
import ctypes as C
import os

import numpy as np

from . import camera as cam
from .types import (
    ERR_NAMES,
    PASS_ALL,
    RESERVOIR_DTYPE,
    RESTIR_EXCHANGE_FN,
    VERTEX_DTYPE,
    GpuLight,
    GpuMaterial,
    Reservoir,
    RestirRows,
    Stats,
    ViewUniformData,
)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UTOPIAN_HIP_LIB") or os.path.join(_HERE, "libutopian_hip.so")  # UTOPIAN_HIP_LIB: an alternative build of the same library (experiments)


class UtopianError(RuntimeError):
    pass


class CApi:
    """Typed ctypes view of a library exporting the utopian_hip.h entry points under `prefix`."""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix
        p, u32, vp = C.POINTER, C.c_uint32, C.c_void_p
        sig = {
            "add_texture_rgba8": [vp, vp, u32, u32, p(u32)],
            "add_mesh": [vp, vp, u32, vp, u32, p(GpuMaterial), p(C.c_float), p(u32)],
            "add_light": [vp, p(GpuLight), p(u32)],
            "set_instance_transform": [vp, u32, p(C.c_float)],
            "build_acceleration": [vp],
            "refit_acceleration": [vp],
            "render_frame": [vp, p(ViewUniformData), u32],
            "render_frames": [vp, p(ViewUniformData), u32, u32],
            "reset_accumulation": [vp],
            "read_accumulation": [vp, vp],
            "read_output_bgra8": [vp, vp],
            "read_reservoirs": [vp, C.c_int, vp],
            "write_reservoirs": [vp, C.c_int, vp],
            "write_gbuffer_position": [vp, vp],
            "read_gbuffer_position": [vp, vp],
            "trace_closest": [vp, vp, u32, vp, vp, vp],
            "trace_any": [vp, vp, u32, vp],
            "get_stats": [vp, p(Stats)],
            "reset_stats": [vp],
            "set_option": [vp, C.c_char_p, C.c_int],
            "set_tile_partition": [vp, u32, u32, u32],
            "set_restir_partition": [vp, u32, u32, vp, vp],
            "get_restir_rows": [vp, p(RestirRows)],
            "resolve_output": [vp, u32, u32],
        }
        for name, argtypes in sig.items():
            if not hasattr(lib, prefix + name) and (name == "render_frames" or prefix == "uh_mgpu_"):
                continue  # the oracle renders frame by frame; the GPU group has no per-context queries
            fn = getattr(lib, prefix + name)
            fn.argtypes, fn.restype = argtypes, C.c_int
            setattr(self, name, fn)
        self.destroy = getattr(lib, prefix + "destroy")
        self.destroy.argtypes, self.destroy.restype = [vp], None


_lib_cache = {}


def _preload_hip_runtime():
    """One HIP runtime per process, and by default the one the library was BUILT with: libutopian_hip.so carries a RUNPATH to
    /opt/rocm's lib directory and binds libamdhip64.so.7 there - what a C / C++ / Rust host links. Nothing in a GPU
    process of this package needs torch (the multi-GPU composition is RCCL inside the library, the launcher's rendezvous a TCP socket).
    UH_HIP_RUNTIME=torch is the opt-in for a process that must share the GPU with PyTorch: the wheel bundles its own libamdhip64.so
    (same soname, ROCm 7.0) and, loaded second, would see no devices - so its copy is loaded first, by path, without importing
    torch, and the library binds to it by soname. That is a 7.2-built library on a 7.0 runtime: uh_version() / uh_last_error(NULL)
    say so, and round 4's context-churn soaks corrupted the host heap in that configuration only (profiles/README.md)."""
    import importlib.util

    if os.environ.get("UH_HIP_RUNTIME", "system") != "torch":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library(path=LIB_PATH):
    """Load libutopian_hip.so. Fails loudly if it has not been built: there is no fallback."""
    if path in _lib_cache:
        return _lib_cache[path]
    if not os.path.exists(path):
        raise UtopianError(f"{path} not built - run `python -c 'import __graft_entry__ as g; g.build()'`")
    _preload_hip_runtime()
    lib = C.CDLL(path)
    _lib_cache[path] = lib
    lib.uh_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.uh_create.restype = C.c_int
    lib.uh_last_error.argtypes, lib.uh_last_error.restype = [C.c_void_p], C.c_char_p
    lib.uh_version.restype = C.c_char_p
    lib.uh_hip_versions.argtypes, lib.uh_hip_versions.restype = [C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int
    for name, extra in (
        ("uh_get_num_lights", [C.POINTER(C.c_uint32)]),
        ("uh_synchronize", []),
        ("uh_tile_pack_count", [C.c_uint32, C.POINTER(C.c_uint64)]),
        ("uh_pack_tiles", [C.c_void_p, C.c_uint64]),
        ("uh_unpack_tiles", [C.c_uint32, C.c_void_p, C.c_uint64]),
        ("uh_compose_tiles", [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]),
        ("uh_device_pointer", [C.c_int, C.POINTER(C.c_void_p)]),
        ("uh_stream", [C.POINTER(C.c_void_p)]),
    ):
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = [C.c_void_p] + extra, C.c_int
    return lib


def hip_versions():
    """(built_with, runtime) as 'major.minor.patch' strings: the HIP release that compiled libutopian_hip.so and the one this
    process runs it on (uh_hip_versions); they differ when another libamdhip64.so.7 was loaded first (UH_HIP_RUNTIME=torch)"""
    lib = load_library()
    b, r = C.c_int(0), C.c_int(0)
    lib.uh_hip_versions(C.byref(b), C.byref(r))
    fmt = lambda v: f"{v // 10000000}.{v // 100000 % 100}.{v % 100000}"
    return fmt(b.value), fmt(r.value)


def identity3x4():
    return np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float32)


def transform3x4(scale=(1, 1, 1), translation=(0, 0, 0), rotation=None):
    """row-major 3x4 from scale / rotation(3x3) / translation: the recomposed
    VkTransformMatrixKHR of Raytracing::fill_instance_array (raytracing.rs:229-248)."""
    r = np.eye(3, dtype=np.float32) if rotation is None else np.asarray(rotation, dtype=np.float32)
    m = np.zeros((3, 4), dtype=np.float32)
    m[:, :3] = r * np.asarray(scale, dtype=np.float32)[None, :]
    m[:, 3] = np.asarray(translation, dtype=np.float32)
    return m.reshape(12)


def make_material(material_type=0, material_property=0.0, base_color=(1, 1, 1, 1), diffuse_map=0, metallic=1.0, roughness=1.0):
    """GpuMaterial as Renderer::add_model fills it (renderer.rs:266-281)."""
    m = GpuMaterial()
    m.diffuse_map = diffuse_map
    m.base_color_factor[:] = [float(x) for x in base_color]
    m.metallic_factor, m.roughness_factor = metallic, roughness
    m.raytrace_properties[:] = [float(material_type), float(material_property), 0.0, 0.0]
    return m


def make_light(position, color=(1, 1, 1), range_=1.0, intensity=(1, 1, 1)):
    """GpuLight as Renderer::add_light fills it (renderer.rs:391-404)."""
    l = GpuLight()
    l.color[:] = [color[0], color[1], color[2], 0.0]
    l.position[:] = [float(x) for x in position]
    l.range = range_
    l.attenuation[:] = [0.0, 0.0, 0.1]
    l.light_type = 1.0
    l.intensity[:] = [float(x) for x in intensity]
    return l


class Renderer:
    """utopian::Renderer + Raytracing + the path-tracing graph resources, on one GPU.

    `_api`/`_ctx_factory` let the test-only oracle binding reuse this class over liboracle.so
    (oracle/oracle_api.py); the product never passes them.
    """

    backend = "hip"

    def __init__(self, width, height, device=0, _api=None, _ctx_factory=None):
        self.width, self.height = int(width), int(height)
        if _api is None:
            lib = load_library()
            self._lib = lib
            ctx = C.c_void_p()
            st = lib.uh_create(int(device), self.width, self.height, C.byref(ctx))
            if st != 0:
                msg = lib.uh_last_error(None)
                raise UtopianError(f"uh_create failed: {ERR_NAMES.get(st, st)}: {msg.decode() if msg else ''}")
            self._ctx, self._api = ctx, CApi(lib, "uh_")
        else:
            self._lib, self._api, self._ctx = _api.lib, _api, _ctx_factory(self.width, self.height)
        self._keep = []

    # -- helpers --------------------------------------------------------------------------
    def _check(self, st):
        if st != 0:
            msg = b""
            if self.backend == "hip":
                msg = self._lib.uh_last_error(self._ctx) or b""
            raise UtopianError(f"{ERR_NAMES.get(st, st)}: {msg.decode()}")

    def close(self):
        if getattr(self, "_ctx", None):
            self._api.destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene (Renderer::add_model / add_light) ------------------------------------------
    def add_texture(self, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert rgba.ndim == 3 and rgba.shape[2] == 4
        out = C.c_uint32()
        self._check(self._api.add_texture_rgba8(self._ctx, rgba.ctypes.data, rgba.shape[1], rgba.shape[0], C.byref(out)))
        return out.value

    def add_mesh(self, vertices, indices, material, world3x4=None):
        vertices = np.ascontiguousarray(vertices, dtype=VERTEX_DTYPE)
        indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        w = identity3x4() if world3x4 is None else np.ascontiguousarray(world3x4, dtype=np.float32).reshape(12)
        out = C.c_uint32()
        self._check(
            self._api.add_mesh(
                self._ctx, vertices.ctypes.data, len(vertices), indices.ctypes.data, len(indices), C.byref(material),
                w.ctypes.data_as(C.POINTER(C.c_float)), C.byref(out),
            )
        )
        return out.value

    def initialize(self, default_textures=None):
        """Renderer::initialize (renderer.rs:202-220): the four default maps get bindless indices 0..3 - diffuse (white),
        normal (flat), occlusion (white), metallic-roughness - before any model texture. `default_textures`: four
        (H, W, 4) uint8 arrays (the files of utopian/data/textures/defaults/); without them 1x1 texels of the same
        values are used (the files are constant images except for +-2 LSB dither in the normal map)."""
        if hasattr(self, "_defaults"):
            return self._defaults
        if default_textures is None:
            texel = lambda *rgba: np.array(rgba, dtype=np.uint8).reshape(1, 1, 4)
            default_textures = [texel(255, 255, 255, 255), texel(127, 127, 255, 255), texel(255, 255, 255, 255), texel(0, 255, 0, 255)]
        diffuse, normal, occlusion, mr = (self.add_texture(t) for t in default_textures)
        self._defaults = dict(diffuse=diffuse, normal=normal, occlusion=occlusion, metallic_roughness=mr)
        self._default_diffuse = diffuse
        return self._defaults

    def add_model(self, model, transform=None):
        """model: scenes.Model (textures + meshes). Texture indices are remapped to bindless indices exactly like
        Renderer::add_model does (renderer.rs:222-262): a map that is DEFAULT_TEXTURE_MAP (None here) takes the
        renderer's default of its kind, every other one is a fresh bindless texture."""
        tex_map = {}
        mesh_ids = []
        defaults = getattr(self, "_defaults", None)
        for mesh in model.meshes:
            mat = mesh.material_struct()
            if defaults:  # the reference assigns all four maps; the path tracer reads diffuse_map only (rchit:40)
                mat.normal_map, mat.metallic_roughness_map, mat.occlusion_map = defaults["normal"], defaults["metallic_roughness"], defaults["occlusion"]
            if mesh.texture is not None:
                if mesh.texture not in tex_map:
                    tex_map[mesh.texture] = self.add_texture(model.textures[mesh.texture])
                mat.diffuse_map = tex_map[mesh.texture]
            else:
                mat.diffuse_map = self.default_diffuse_map()
            w = mesh.transform if transform is None else compose3x4(transform, mesh.transform)
            mesh_ids.append(self.add_mesh(mesh.vertices, mesh.indices, mat, w))
        return mesh_ids

    def add_isosurface_mesh(self, resolution, lo, hi, time=0.0, material=None, world3x4=None):
        """uh_add_isosurface_mesh: the reference's marching-cubes density field, extracted on the GPU.
        Returns (mesh index or None, triangle count)."""
        if material is None:
            material = make_material(base_color=(0.8, 0.8, 0.8, 1.0), diffuse_map=self.default_diffuse_map())
        w = identity3x4() if world3x4 is None else np.ascontiguousarray(world3x4, dtype=np.float32).reshape(12)
        fn = self._lib.uh_add_isosurface_mesh
        fn.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.POINTER(GpuMaterial), C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        fn.restype = C.c_int
        mesh, tris = C.c_uint32(), C.c_uint32()
        self._check(fn(self._ctx, int(resolution), float(lo), float(hi), float(time), C.byref(material), w.ctypes.data_as(C.POINTER(C.c_float)), C.byref(mesh), C.byref(tris)))
        return (None if mesh.value == 0xFFFFFFFF else mesh.value), tris.value

    def isosurface_cells(self, resolution, lo, hi, time=0.0):
        """uh_isosurface_cells: per cell of the extraction grid (x fastest) its marching-cubes case index and the triangles it keeps"""
        n = int(resolution) ** 3
        cube, kept = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        fn = self._lib.uh_isosurface_cells
        fn.argtypes, fn.restype = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p], C.c_int
        self._check(fn(self._ctx, int(resolution), float(lo), float(hi), float(time), cube.ctypes.data, kept.ctypes.data))
        return cube, kept

    def read_mesh(self, mesh_index):
        """the context's host copy of a mesh: (vertices as VERTEX_DTYPE, indices)"""
        lib = self._lib
        lib.uh_mesh_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.uh_read_mesh.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.uh_mesh_info.restype = lib.uh_read_mesh.restype = C.c_int
        nv, ni = C.c_uint32(), C.c_uint32()
        self._check(lib.uh_mesh_info(self._ctx, mesh_index, C.byref(nv), C.byref(ni)))
        v = np.zeros(nv.value, dtype=VERTEX_DTYPE)
        idx = np.zeros(ni.value, dtype=np.uint32)
        self._check(lib.uh_read_mesh(self._ctx, mesh_index, v.ctypes.data, idx.ctypes.data))
        return v, idx

    def default_diffuse_map(self):
        """Renderer::initialize's default_diffuse_map (renderer.rs:202-220): a white texel."""
        if not hasattr(self, "_default_diffuse"):
            self._default_diffuse = self.add_texture(np.full((1, 1, 4), 255, dtype=np.uint8))
        return self._default_diffuse

    def add_light(self, position, color=(1, 1, 1), range_=1.0):
        out = C.c_uint32()
        l = make_light(position, color, range_)
        self._check(self._api.add_light(self._ctx, C.byref(l), C.byref(out)))
        return out.value

    def add_gpu_light(self, light):
        out = C.c_uint32()
        self._check(self._api.add_light(self._ctx, C.byref(light), C.byref(out)))
        return out.value

    def get_num_lights(self):
        if self.backend == "hip":
            out = C.c_uint32()
            self._check(self._lib.uh_get_num_lights(self._ctx, C.byref(out)))
            return out.value
        return self._num_lights

    def set_instance_transform(self, mesh_index, world3x4):
        w = np.ascontiguousarray(world3x4, dtype=np.float32).reshape(12)
        self._check(self._api.set_instance_transform(self._ctx, mesh_index, w.ctypes.data_as(C.POINTER(C.c_float))))

    def initialize_raytracing(self):
        """Raytracing::initialize (raytracing.rs:89): build the acceleration structure."""
        self._check(self._api.build_acceleration(self._ctx))

    build_acceleration = initialize_raytracing

    def rebuild_tlas(self):
        """Raytracing::rebuild_tlas (raytracing.rs:400): on-device refit after set_instance_transform."""
        self._check(self._api.refit_acceleration(self._ctx))

    refit_acceleration = rebuild_tlas

    # -- per frame ------------------------------------------------------------------------
    def render_frame(self, view, pass_mask=PASS_ALL):
        self._check(self._api.render_frame(self._ctx, C.byref(view), pass_mask))

    def render_frames(self, view, pass_mask, count):
        """`count` consecutive path-tracing frames of a static camera (uh_render_frames)."""
        self._check(self._api.render_frames(self._ctx, C.byref(view), pass_mask, count))

    def reset_accumulation(self):
        self._check(self._api.reset_accumulation(self._ctx))

    def synchronize(self):
        if self.backend == "hip":
            self._check(self._lib.uh_synchronize(self._ctx))

    # -- read-back ------------------------------------------------------------------------
    def read_accumulation(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._api.read_accumulation(self._ctx, out.ctypes.data))
        return out

    def read_output_bgra8(self):
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._check(self._api.read_output_bgra8(self._ctx, out.ctypes.data))
        return out

    def read_reservoirs(self, which):
        out = np.empty((self.height, self.width), dtype=RESERVOIR_DTYPE)
        self._check(self._api.read_reservoirs(self._ctx, which, out.ctypes.data))
        return out

    def write_reservoirs(self, which, data):
        data = np.ascontiguousarray(data, dtype=RESERVOIR_DTYPE)
        assert data.size == self.width * self.height
        self._check(self._api.write_reservoirs(self._ctx, which, data.ctypes.data))

    def write_gbuffer_position(self, rgba32f):
        """uh_write_gbuffer_position: frames rendered without PASS_GBUFFER read these positions (known-answer tests)"""
        data = np.ascontiguousarray(rgba32f, dtype=np.float32)
        assert data.size == self.width * self.height * 4
        self._check(self._api.write_gbuffer_position(self._ctx, data.ctypes.data))

    def read_gbuffer_position(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._api.read_gbuffer_position(self._ctx, out.ctypes.data))
        return out

    # -- stand-alone ray queries ----------------------------------------------------------
    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = len(rays)
        tuv = np.empty((n, 3), dtype=np.float32)
        mesh = np.empty(n, dtype=np.uint32)
        prim = np.empty(n, dtype=np.uint32)
        self._check(self._api.trace_closest(self._ctx, rays.ctypes.data, n, tuv.ctypes.data, mesh.ctypes.data, prim.ctypes.data))
        return tuv, mesh, prim

    def trace_any(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        occ = np.empty(len(rays), dtype=np.uint8)
        self._check(self._api.trace_any(self._ctx, rays.ctypes.data, len(rays), occ.ctypes.data))
        return occ

    # -- stats / options ------------------------------------------------------------------
    def get_stats(self):
        s = Stats()
        self._check(self._api.get_stats(self._ctx, C.byref(s)))
        return s

    def reset_stats(self):
        self._check(self._api.reset_stats(self._ctx))

    def set_option(self, name, value):
        self._check(self._api.set_option(self._ctx, name.encode(), int(value)))

    # -- multi-GPU tile partition ---------------------------------------------------------
    def set_tile_partition(self, rank, world, tile_size=64):
        self._check(self._api.set_tile_partition(self._ctx, rank, world, tile_size))

    def resolve_output(self, total_samples, accumulation_limit=999999):
        self._check(self._api.resolve_output(self._ctx, total_samples, accumulation_limit))

    # -- multi-GPU, the reservoir passes by bands of rows (uh_set_restir_partition) --------
    def set_restir_partition(self, rank, world, exchange=None):
        """`exchange(stream, spatial_base, band_bytes, rank, world) -> int`: called after every spatial pass, while the frame is
        enqueued; it must enqueue on `stream` (a hipStream_t as int; None on the CPU oracle, which calls it synchronously)
        whatever fills the other ranks' bands of the buffer at `spatial_base`. None with world > 1: the other bands stay stale."""
        fn = None
        if exchange is not None:
            fn = RESTIR_EXCHANGE_FN(lambda user, stream, base, band_bytes, r, w: int(exchange(stream, base, band_bytes, r, w) or 0))
        self._check(self._api.set_restir_partition(self._ctx, rank, world, C.cast(fn, C.c_void_p) if fn else None, None))
        self._restir_exchange = fn  # the library keeps the pointer: the trampoline must live as long as the partition

    def restir_rows(self):
        out = RestirRows()
        self._check(self._api.get_restir_rows(self._ctx, C.byref(out)))
        return out

    @staticmethod
    def rccl_unique_id():
        lib = load_library()
        lib.uh_rccl_unique_id.argtypes, lib.uh_rccl_unique_id.restype = [C.c_void_p], C.c_int
        buf = (C.c_uint8 * 128)()
        st = lib.uh_rccl_unique_id(buf)
        if st != 0:
            raise UtopianError(f"uh_rccl_unique_id failed: {ERR_NAMES.get(st, st)} (is librccl loadable?)")
        return bytes(buf)

    def rccl_attach(self, rank, world, unique_id):
        """ncclCommInitRank on this context's device + the band partition with ncclAllGather as its exchange (collective:
        every rank of the job calls it with the id rank 0 made)"""
        self._lib.uh_rccl_attach.argtypes, self._lib.uh_rccl_attach.restype = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p], C.c_int
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._check(self._lib.uh_rccl_attach(self._ctx, rank, world, buf))

    def sun_grid_compare_builders(self):
        """uh_sun_grid_compare_builders: the device-built grid in use against the host builder on the same raster"""
        out = (C.c_uint64 * 8)()
        self._lib.uh_sun_grid_compare_builders.argtypes, self._lib.uh_sun_grid_compare_builders.restype = [C.c_void_p, C.POINTER(C.c_uint64)], C.c_int
        self._check(self._lib.uh_sun_grid_compare_builders(self._ctx, out))
        keys = ("cells", "entries_device", "entries_host", "cells_length_differs", "cells_list_differs", "cells_cover_differs", "walkable_cells", "host_build_us")
        return dict(zip(keys, (int(x) for x in out)))

    def rccl_comm_count(self):
        """ranks of the attached RCCL communicator as ncclCommCount reports them (0: none attached)"""
        n = C.c_uint32(0)
        self._lib.uh_rccl_comm_count.argtypes, self._lib.uh_rccl_comm_count.restype = [C.c_void_p, C.POINTER(C.c_uint32)], C.c_int
        self._check(self._lib.uh_rccl_comm_count(self._ctx, C.byref(n)))
        return n.value

    def rccl_gather_tiles(self, root, total_samples, accumulation_limit=999999):
        """uh_rccl_gather_tiles: every rank's tiles to `root` over the attached communicator, composed there; enqueued (no host wait)"""
        fn = self._lib.uh_rccl_gather_tiles
        fn.argtypes, fn.restype = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32], C.c_int
        self._check(fn(self._ctx, root, total_samples, accumulation_limit))

    def rccl_detach(self):
        self._lib.uh_rccl_detach.argtypes, self._lib.uh_rccl_detach.restype = [C.c_void_p], C.c_int
        self._check(self._lib.uh_rccl_detach(self._ctx))

    def tile_pack_count(self, rank):
        out = C.c_uint64()
        self._check(self._lib.uh_tile_pack_count(self._ctx, rank, C.byref(out)))
        return out.value

    def pack_tiles(self, device_ptr, capacity_pixels):
        self._check(self._lib.uh_pack_tiles(self._ctx, C.c_void_p(device_ptr), capacity_pixels))

    def unpack_tiles(self, from_rank, device_ptr, num_pixels):
        self._check(self._lib.uh_unpack_tiles(self._ctx, from_rank, C.c_void_p(device_ptr), num_pixels))

    def compose_tiles(self, device_ptr_all, stride_pixels, total_samples, accumulation_limit=999999):
        """every other rank's packed tiles (world buffers of stride_pixels each) into the accumulation image + the output resolve, one launch"""
        self._check(self._lib.uh_compose_tiles(self._ctx, C.c_void_p(device_ptr_all), stride_pixels, total_samples, accumulation_limit))

    def device_pointer(self, which):
        out = C.c_void_p()
        self._check(self._lib.uh_device_pointer(self._ctx, which, C.byref(out)))
        return out.value

    def stream(self):
        out = C.c_void_p()
        self._check(self._lib.uh_stream(self._ctx, C.byref(out)))
        return out.value


class MultiGpuRenderer(Renderer):
    """uh_mgpu_*: the same host interface over several GPUs driven by ONE process (tiles t % N == i
    per GPU, composition on GPU 0 at read-back). `devices` may repeat an ordinal."""

    backend = "hip-group"

    def __init__(self, width, height, devices, tile_size=64):
        lib = load_library()
        lib.uh_mgpu_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        lib.uh_mgpu_create.restype = C.c_int
        lib.uh_mgpu_last_error.argtypes, lib.uh_mgpu_last_error.restype = [C.c_void_p], C.c_char_p
        lib.uh_mgpu_context.argtypes, lib.uh_mgpu_context.restype = [C.c_void_p, C.c_int], C.c_void_p
        lib.uh_mgpu_num_devices.argtypes, lib.uh_mgpu_num_devices.restype = [C.c_void_p], C.c_int
        for name, extra in (("get_num_lights", [C.POINTER(C.c_uint32)]), ("synchronize", []), ("compose", []), ("refit_acceleration", [])):
            fn = getattr(lib, "uh_mgpu_" + name)
            fn.argtypes, fn.restype = [C.c_void_p] + extra, C.c_int
        devices = [int(d) for d in devices]
        arr = (C.c_int * len(devices))(*devices)

        def factory(w, h):
            ctx = C.c_void_p()
            st = lib.uh_mgpu_create(len(devices), arr, w, h, int(tile_size), C.byref(ctx))
            if st != 0:
                raise UtopianError(f"uh_mgpu_create failed: {ERR_NAMES.get(st, st)}: {(lib.uh_mgpu_last_error(None) or b'').decode()}")
            return ctx

        super().__init__(width, height, _api=CApi(lib, "uh_mgpu_"), _ctx_factory=factory)
        self.devices = devices

    def _check(self, st):
        if st != 0:
            raise UtopianError(f"{ERR_NAMES.get(st, st)}: {(self._lib.uh_mgpu_last_error(self._ctx) or b'').decode()}")

    def get_num_lights(self):
        out = C.c_uint32()
        self._check(self._lib.uh_mgpu_get_num_lights(self._ctx, C.byref(out)))
        return out.value

    def synchronize(self):
        self._check(self._lib.uh_mgpu_synchronize(self._ctx))

    def compose(self):
        self._check(self._lib.uh_mgpu_compose(self._ctx))


def compose3x4(a, b):
    """a * b for row-major 3x4 affine matrices (instance.transform * model.transforms[i])."""
    A = np.vstack([np.asarray(a, dtype=np.float32).reshape(3, 4), [0, 0, 0, 1]]).astype(np.float32)
    B = np.vstack([np.asarray(b, dtype=np.float32).reshape(3, 4), [0, 0, 0, 1]]).astype(np.float32)
    return (A @ B)[:3].astype(np.float32).reshape(12)


def default_view(camera, width, height, num_lights=0):
    """ViewUniformData with the reference's defaults (prototype/src/main.rs:55-86), time = 0."""
    v = ViewUniformData()
    view, proj = camera.get_view(), camera.get_projection()
    v.view[:] = cam.to_glam(view).tolist()
    v.projection[:] = cam.to_glam(proj).tolist()
    v.inverse_view[:] = cam.to_glam(cam.inverse(view)).tolist()
    v.inverse_projection[:] = cam.to_glam(cam.inverse(proj)).tolist()
    v.prev_frame_projection_view[:] = cam.to_glam(np.diag(np.float32([-1, -1, -1, -1]))).tolist()
    v.eye_pos[:] = camera.get_position().tolist()
    v.samples_per_frame, v.total_samples, v.num_bounces = 1, 0, 5
    v.viewport_width, v.viewport_height = width, height
    v.time = 0.0
    v.num_lights = num_lights
    sun = np.float32([0.0, 0.9, 0.15])
    sun = sun / np.sqrt(np.dot(sun, sun), dtype=np.float32)
    v.sun_dir[:] = sun.tolist()
    v.shadows_enabled = v.ssao_enabled = v.fxaa_enabled = v.cubemap_enabled = v.ibl_enabled = 1
    v.sky_enabled = v.sun_shadow_enabled = v.lights_enabled = 1
    v.max_num_lights_used = 10000
    v.marching_cubes_enabled = 0
    v.temporal_reuse_enabled = v.spatial_reuse_enabled = 1
    v.rebuild_tlas = 0
    v.accumulation_limit = 999999
    v.use_ris_light_sampling = 1
    v.raytracing_supported = 1
    return v


class FrameLoop:
    """The per-frame protocol of Application::run (prototype/src/main.rs:460-471, 545-546):
    total_samples += samples_per_frame BEFORE the frame is rendered; prev_frame_projection_view =
    projection * view AFTER it was recorded."""

    def __init__(self, renderer, view):
        self.renderer, self.view = renderer, view

    def frame(self, pass_mask=PASS_ALL):
        v = self.view
        v.total_samples += v.samples_per_frame
        v.num_lights = self.renderer.get_num_lights()
        self.renderer.render_frame(v, pass_mask)
        self.end_frame()

    def end_frame(self):
        """main.rs:545-546: prev_frame_projection_view = projection * view, after the frame."""
        v = self.view
        proj = np.array(v.projection[:], dtype=np.float32).reshape(4, 4).T
        view = np.array(v.view[:], dtype=np.float32).reshape(4, 4).T
        v.prev_frame_projection_view[:] = cam.to_glam((proj @ view).astype(np.float32)).tolist()

    def frames(self, count, pass_mask=PASS_ALL):
        """`count` frames of the loop with the camera at rest. Everything that includes the path-tracing pass is handed to
        uh_render_frames in one call (the library batches frames into shared wavefronts of about 32 M paths: 16 frames at
        1080p, at most 16 when the reservoir passes run too). The first frame of a run that includes the temporal pass goes alone:
        it is the one that may still see another prev_frame_projection_view."""
        from .types import PASS_REFERENCE_PT, PASS_TEMPORAL_REUSE

        v = self.view
        batchable = self.renderer.backend.startswith("hip") and count > 1 and (pass_mask & PASS_REFERENCE_PT)
        if batchable and (pass_mask & PASS_TEMPORAL_REUSE):
            self.frame(pass_mask)
            count -= 1
            batchable = count > 1
        if not batchable:
            for _ in range(count):
                self.frame(pass_mask)
            return
        v.num_lights = self.renderer.get_num_lights()
        v.total_samples += v.samples_per_frame  # first frame of the run
        self.renderer.render_frames(v, pass_mask, count)
        v.total_samples += (count - 1) * v.samples_per_frame
        proj = np.array(v.projection[:], dtype=np.float32).reshape(4, 4).T
        view = np.array(v.view[:], dtype=np.float32).reshape(4, 4).T
        v.prev_frame_projection_view[:] = cam.to_glam((proj @ view).astype(np.float32)).tolist()

    def reset(self):
        """camera / setting change: total_samples = 0 (main.rs:400-413)."""
        self.view.total_samples = 0
        self.renderer.reset_accumulation()
