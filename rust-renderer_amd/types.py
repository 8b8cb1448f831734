"""ctypes / numpy mirrors of the POD structs in include/utopian_hip.h.

Each struct is byte-identical to the reference's GPU-side struct it replaces:
Vertex (utopian/src/primitive.rs:9-17), GpuMaterial / GpuMesh / GpuLight
(utopian/src/renderer.rs:20-59), ViewUniformData (utopian/src/renderer.rs:84-120),
Reservoir (utopian/shaders/include/restir_sampling.glsl:51-57).
"""
import ctypes as C

import numpy as np

VERTEX_DTYPE = np.dtype(
    [("pos", "<f4", 4), ("normal", "<f4", 4), ("uv", "<f4", 2), ("_pad", "<f4", 2), ("color", "<f4", 4), ("tangent", "<f4", 4)]
)
assert VERTEX_DTYPE.itemsize == 80

RESERVOIR_DTYPE = np.dtype([("Y", "<i4"), ("W_sum", "<f4"), ("W_X", "<f4"), ("M", "<i4")])
assert RESERVOIR_DTYPE.itemsize == 16

# MaterialType (utopian/src/gltf_loader.rs:11-17)
LAMBERTIAN, METAL, DIELECTRIC, DIFFUSE_LIGHT = 0, 1, 2, 3
PBR = 4  # extension: Cook-Torrance from metallic_factor / roughness_factor; no reference scene uses it


class GpuMaterial(C.Structure):
    _fields_ = [
        ("diffuse_map", C.c_uint32),
        ("normal_map", C.c_uint32),
        ("metallic_roughness_map", C.c_uint32),
        ("occlusion_map", C.c_uint32),
        ("base_color_factor", C.c_float * 4),
        ("metallic_factor", C.c_float),
        ("roughness_factor", C.c_float),
        ("padding", C.c_float * 2),
        ("raytrace_properties", C.c_float * 4),
    ]


class GpuLight(C.Structure):
    _fields_ = [
        ("color", C.c_float * 4),
        ("position", C.c_float * 3),
        ("range", C.c_float),
        ("direction", C.c_float * 3),
        ("spot", C.c_float),
        ("attenuation", C.c_float * 3),
        ("light_type", C.c_float),
        ("intensity", C.c_float * 3),
        ("id", C.c_float),
        ("padding", C.c_float * 4),
    ]


class ViewUniformData(C.Structure):
    _fields_ = [
        ("view", C.c_float * 16),
        ("projection", C.c_float * 16),
        ("inverse_view", C.c_float * 16),
        ("inverse_projection", C.c_float * 16),
        ("prev_frame_projection_view", C.c_float * 16),
        ("eye_pos", C.c_float * 3),
        ("samples_per_frame", C.c_uint32),
        ("sun_dir", C.c_float * 3),
        ("total_samples", C.c_uint32),
        ("num_bounces", C.c_uint32),
        ("viewport_width", C.c_uint32),
        ("viewport_height", C.c_uint32),
        ("time", C.c_float),
        ("num_lights", C.c_uint32),
        ("shadows_enabled", C.c_uint32),
        ("ssao_enabled", C.c_uint32),
        ("fxaa_enabled", C.c_uint32),
        ("cubemap_enabled", C.c_uint32),
        ("ibl_enabled", C.c_uint32),
        ("sky_enabled", C.c_uint32),
        ("sun_shadow_enabled", C.c_uint32),
        ("lights_enabled", C.c_uint32),
        ("max_num_lights_used", C.c_uint32),
        ("marching_cubes_enabled", C.c_uint32),
        ("temporal_reuse_enabled", C.c_uint32),
        ("spatial_reuse_enabled", C.c_uint32),
        ("rebuild_tlas", C.c_uint32),
        ("accumulation_limit", C.c_uint32),
        ("use_ris_light_sampling", C.c_uint32),
        ("raytracing_supported", C.c_uint32),
        ("_tail_pad", C.c_uint32 * 3),
    ]


class Reservoir(C.Structure):
    _fields_ = [("Y", C.c_int32), ("W_sum", C.c_float), ("W_X", C.c_float), ("M", C.c_int32)]


RAY_KINDS = 5
RAY_PRIMARY, RAY_BOUNCE, RAY_SUN_SHADOW, RAY_LIGHT_SHADOW, RAY_GBUFFER = range(5)


class RestirRows(C.Structure):
    """UhRestirRows: the rows a context's reservoir passes cover under uh_set_restir_partition"""
    _fields_ = [(n, C.c_uint32) for n in ("band_row0", "band_rows", "reuse_row0", "reuse_rows", "reuse_extra_row0", "reuse_extra_rows",
                                          "cast_row0", "cast_rows", "cast_extra_row0", "cast_extra_rows", "rows_per_band")]


# int exchange(void* user, void* hip_stream, void* spatial_base, uint64_t band_bytes, uint32_t rank, uint32_t world)
RESTIR_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32)


class Stats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64 * RAY_KINDS),
        ("nodes_visited", C.c_uint64),
        ("tris_tested", C.c_uint64),
        ("shadow_nodes_visited", C.c_uint64),
        ("shadow_tris_tested", C.c_uint64),
        ("closest_hits", C.c_uint64),
        ("misses", C.c_uint64),
        ("frames", C.c_uint64),
        ("bvh_nodes", C.c_uint32),
        ("bvh_triangles", C.c_uint32),
        ("build_ms", C.c_float),
        ("last_frame_ms", C.c_float),
        ("trace_closest_ms", C.c_float),
        ("trace_shadow_ms", C.c_float),
        ("shade_ms", C.c_float),
        ("trace_closest_launches", C.c_uint32),
        ("sun_grid_cells", C.c_uint32),
        ("sun_grid_entries", C.c_uint32),
        ("sun_grid_build_ms", C.c_float),
        ("sun_grid_mean_list", C.c_float),
        ("sun_tree_rays", C.c_uint64),
        ("camera_grid_cells", C.c_uint32),
        ("camera_grid_entries", C.c_uint32),
        ("camera_grid_build_ms", C.c_float),
        ("camera_grid_mean_list", C.c_float),
        ("camera_tree_rays", C.c_uint64),
        ("camera_grid_tris_tested", C.c_uint64),
        ("camera_grid_ms", C.c_float),
        ("trace_light_ms", C.c_float),
        ("sun_covered_rays", C.c_uint64),
        ("sun_grid_bytes", C.c_uint64),
        ("camera_grid_bytes", C.c_uint64),
        ("light_nodes_visited", C.c_uint64),
        ("light_tris_tested", C.c_uint64),
        ("trace_light_launches", C.c_uint32),
        ("reserved1", C.c_uint32),
    ]

    @property
    def path_rays(self):
        """rays the metric counts: primary + bounce + sun-shadow + light-shadow (not G-buffer)."""
        return sum(self.rays[i] for i in range(4))


assert C.sizeof(GpuMaterial) == 64
assert C.sizeof(GpuLight) == 96
assert C.sizeof(ViewUniformData) == 448
assert C.sizeof(Reservoir) == 16
assert ViewUniformData.eye_pos.offset == 320 and ViewUniformData.samples_per_frame.offset == 332
assert ViewUniformData.sun_dir.offset == 336 and ViewUniformData.total_samples.offset == 348
assert ViewUniformData.num_bounces.offset == 352 and ViewUniformData.sky_enabled.offset == 392
assert ViewUniformData.accumulation_limit.offset == 424 and ViewUniformData.raytracing_supported.offset == 432

PASS_GBUFFER, PASS_RESET_RESERVOIRS, PASS_INITIAL_RIS, PASS_TEMPORAL_REUSE, PASS_SPATIAL_REUSE, PASS_REFERENCE_PT = (1 << i for i in range(6))
PASS_RESTIR = 0x1F
PASS_ALL = 0x3F

UH_OK = 0
ERR_NAMES = {1: "INVALID_ARGUMENT", 2: "NO_DEVICE", 3: "HIP", 4: "CAPACITY", 5: "NOT_BUILT", 6: "OUT_OF_MEMORY"}
