/*
 * utopian_hip.h — C ABI of libutopian_hip.so: the MI355X (gfx950) replacement for the
 * reference's path-tracing + ReSTIR render-graph nodes.
 *
 * Every entry point below replaces one verb of the reference's Rust/Vulkan surface for this
 * path (citations are relative to the reference checkout):
 *
 *   uh_create                 Renderer::new + Raytracing::new + the graph resources of
 *                             build_path_tracing_render_graph   (utopian/src/renderer.rs:123,
 *                             utopian/src/raytracing.rs:36, utopian/src/renderers/mod.rs:199-244)
 *   uh_add_texture_rgba8      Renderer::add_bindless_texture    (utopian/src/renderer.rs:301)
 *   uh_add_mesh               Renderer::add_model, per mesh     (utopian/src/renderer.rs:222-299)
 *                             + one row of fill_instance_array  (utopian/src/raytracing.rs:218-277)
 *   uh_add_light              Renderer::add_light               (utopian/src/renderer.rs:391-410)
 *   uh_set_instance_transform gizmo edit + rebuild_tlas         (utopian/src/raytracing.rs:400-459)
 *   uh_build_acceleration     Raytracing::initialize            (utopian/src/raytracing.rs:89-111)
 *   uh_render_frame           the 6 graph passes gbuffer→reset→initial_ris→temporal→spatial→pt
 *                             (utopian/src/renderers/mod.rs:246-358), one ViewUniformData memcpy
 *                             per frame (prototype/src/main.rs:477-478)
 *   uh_render_frames          the same node for N consecutive frames of a static camera (batched)
 *   uh_reset_accumulation     total_samples = 0 semantics       (prototype/src/main.rs:400-413)
 *   uh_read_*                 pt_accumulation_image / pt_output_image / reservoir SSBO read-back
 *   uh_set_tile_partition,
 *   uh_pack_tiles, uh_unpack_tiles, uh_resolve_output
 *                             multi-GPU framebuffer tile partition; no reference counterpart
 *                             (the reference is single-device, utopian/src/device.rs:45)
 *   uh_set_restir_partition, uh_rccl_attach
 *                             multi-GPU partition of the reservoir passes by bands of rows with one
 *                             all-gather of spatial_reuse_reservoirs per frame (temporal_reuse.rgen:90-99,
 *                             spatial_reuse.rgen:40-60 read across any pixel partition)
 *
 * Contract: plain C, POD in / status out, no exceptions cross the boundary. One context per
 * GPU; all calls on one context are serialised by the caller (the reference has a single render
 * thread, utopian/src/graph.rs:1004-1007). The library fails loudly (UH_ERR_NO_DEVICE) when no
 * HIP device is present: there is no CPU fallback.
 */
#ifndef UTOPIAN_HIP_H
#define UTOPIAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- POD structs, byte-identical to the reference's GPU structs ------------------------ */

/* utopian/src/primitive.rs:9-17 == shaders/include/bindless.glsl:4-11 (std430, 80 B) */
typedef struct UhVertex {
   float pos[4];     /* @0  */
   float normal[4];  /* @16 */
   float uv[2];      /* @32 */
   float _pad[2];    /* @40 */
   float color[4];   /* @48 */
   float tangent[4]; /* @64 */
} UhVertex;

/* utopian/src/renderer.rs:20-36 == bindless.glsl:13-28 (scalar layout, 64 B) */
typedef struct UhGpuMaterial {
   uint32_t diffuse_map;
   uint32_t normal_map;
   uint32_t metallic_roughness_map;
   uint32_t occlusion_map;
   float base_color_factor[4];   /* @16 */
   float metallic_factor;        /* @32 */
   float roughness_factor;       /* @36 */
   float padding[2];             /* @40 */
   float raytrace_properties[4]; /* @48: x = 0 lambertian,1 metal,2 dielectric,3 diffuse light (reference.rchit:47-89), 4 = Cook-Torrance from
                                  * metallic_factor / roughness_factor (extension, SURVEY 8f N2: no reference scene uses it); y = fuzz | ior */
} UhGpuMaterial;

/* utopian/src/renderer.rs:38-44 (12 B) */
typedef struct UhGpuMesh {
   uint32_t vertex_buffer;
   uint32_t index_buffer;
   uint32_t material;
} UhGpuMesh;

/* utopian/src/renderer.rs:46-59 == bindless.glsl:37-49 (scalar layout, 96 B) */
typedef struct UhGpuLight {
   float color[4];       /* @0  */
   float position[3];    /* @16 */
   float range;          /* @28 */
   float direction[3];   /* @32 */
   float spot;           /* @44 */
   float attenuation[3]; /* @48 */
   float light_type;     /* @60 */
   float intensity[3];   /* @64 */
   float id;             /* @76 */
   float padding[4];     /* @80 */
} UhGpuLight;

/* utopian/src/renderer.rs:84-120 == shaders/include/view.glsl:1-35 (std140, 448 B).
 * Matrices are column-major (glam::Mat4): m[c*4 + r]. */
typedef struct UhViewUniformData {
   float view[16];                       /* @0   */
   float projection[16];                 /* @64  */
   float inverse_view[16];               /* @128 */
   float inverse_projection[16];         /* @192 */
   float prev_frame_projection_view[16]; /* @256 */
   float eye_pos[3];                     /* @320 */
   uint32_t samples_per_frame;           /* @332 */
   float sun_dir[3];                     /* @336 */
   uint32_t total_samples;               /* @348 */
   uint32_t num_bounces;                 /* @352 */
   uint32_t viewport_width;              /* @356 */
   uint32_t viewport_height;             /* @360 */
   float time;                           /* @364 */
   uint32_t num_lights;                  /* @368 */
   uint32_t shadows_enabled;             /* @372 */
   uint32_t ssao_enabled;                /* @376 */
   uint32_t fxaa_enabled;                /* @380 */
   uint32_t cubemap_enabled;             /* @384 */
   uint32_t ibl_enabled;                 /* @388 */
   uint32_t sky_enabled;                 /* @392 */
   uint32_t sun_shadow_enabled;          /* @396 */
   uint32_t lights_enabled;              /* @400 */
   uint32_t max_num_lights_used;         /* @404 */
   uint32_t marching_cubes_enabled;      /* @408 */
   uint32_t temporal_reuse_enabled;      /* @412 */
   uint32_t spatial_reuse_enabled;       /* @416 */
   uint32_t rebuild_tlas;                /* @420 */
   uint32_t accumulation_limit;          /* @424 */
   uint32_t use_ris_light_sampling;      /* @428 */
   uint32_t raytracing_supported;        /* @432 */
   uint32_t _tail_pad[3];                /* @436 → 448 */
} UhViewUniformData;

/* shaders/include/restir_sampling.glsl:51-57 (16 B) */
typedef struct UhReservoir {
   int32_t Y;
   float W_sum;
   float W_X;
   int32_t M;
} UhReservoir;

/* ---- compile-time layout guard: a C11 / C++11 consumer (or a bindgen run over this header)
 * learns of a packing mismatch when it compiles, not at run time. Sizes and offsets are the reference's. */
#include <stddef.h>
#if defined(__cplusplus)
#define UH_LAYOUT_ASSERT(cond, msg) static_assert(cond, msg)
#elif defined(__STDC_VERSION__) && __STDC_VERSION__ >= 201112L
#define UH_LAYOUT_ASSERT(cond, msg) _Static_assert(cond, msg)
#else
#define UH_LAYOUT_ASSERT(cond, msg) typedef char uh_layout_assert_[(cond) ? 1 : -1]
#endif
UH_LAYOUT_ASSERT(sizeof(UhVertex) == 80 && offsetof(UhVertex, normal) == 16 && offsetof(UhVertex, uv) == 32 && offsetof(UhVertex, color) == 48 &&
                    offsetof(UhVertex, tangent) == 64, "UhVertex: std430 Vertex of bindless.glsl:4-11 (80 B)");
UH_LAYOUT_ASSERT(sizeof(UhGpuMaterial) == 64 && offsetof(UhGpuMaterial, base_color_factor) == 16 && offsetof(UhGpuMaterial, metallic_factor) == 32 &&
                    offsetof(UhGpuMaterial, roughness_factor) == 36 && offsetof(UhGpuMaterial, raytrace_properties) == 48, "UhGpuMaterial: renderer.rs:20-36 (64 B)");
UH_LAYOUT_ASSERT(sizeof(UhGpuMesh) == 12, "UhGpuMesh: renderer.rs:38-44 (12 B)");
UH_LAYOUT_ASSERT(sizeof(UhGpuLight) == 96 && offsetof(UhGpuLight, position) == 16 && offsetof(UhGpuLight, range) == 28 && offsetof(UhGpuLight, direction) == 32 &&
                    offsetof(UhGpuLight, attenuation) == 48 && offsetof(UhGpuLight, light_type) == 60 && offsetof(UhGpuLight, intensity) == 64 &&
                    offsetof(UhGpuLight, id) == 76, "UhGpuLight: renderer.rs:46-59 (96 B)");
UH_LAYOUT_ASSERT(sizeof(UhViewUniformData) == 448 && offsetof(UhViewUniformData, inverse_view) == 128 && offsetof(UhViewUniformData, prev_frame_projection_view) == 256 &&
                    offsetof(UhViewUniformData, eye_pos) == 320 && offsetof(UhViewUniformData, samples_per_frame) == 332 && offsetof(UhViewUniformData, sun_dir) == 336 &&
                    offsetof(UhViewUniformData, total_samples) == 348 && offsetof(UhViewUniformData, num_bounces) == 352 && offsetof(UhViewUniformData, time) == 364 &&
                    offsetof(UhViewUniformData, sky_enabled) == 392 && offsetof(UhViewUniformData, sun_shadow_enabled) == 396 &&
                    offsetof(UhViewUniformData, lights_enabled) == 400 && offsetof(UhViewUniformData, max_num_lights_used) == 404 &&
                    offsetof(UhViewUniformData, temporal_reuse_enabled) == 412 && offsetof(UhViewUniformData, spatial_reuse_enabled) == 416 &&
                    offsetof(UhViewUniformData, accumulation_limit) == 424 && offsetof(UhViewUniformData, use_ris_light_sampling) == 428 &&
                    offsetof(UhViewUniformData, raytracing_supported) == 432, "UhViewUniformData: std140 UBO_view of view.glsl:1-35 (448 B)");
UH_LAYOUT_ASSERT(sizeof(UhReservoir) == 16 && offsetof(UhReservoir, W_sum) == 4 && offsetof(UhReservoir, W_X) == 8 && offsetof(UhReservoir, M) == 12,
                 "UhReservoir: restir_sampling.glsl:51-57 (16 B)");

/* ---- pass mask for uh_render_frame (pass order of renderers/mod.rs:246-358) -------------- */
enum {
   UH_PASS_GBUFFER = 1u << 0,        /* gbuffer_pass (position only; produced by primary-ray cast) */
   UH_PASS_RESET_RESERVOIRS = 1u << 1,
   UH_PASS_INITIAL_RIS = 1u << 2,
   UH_PASS_TEMPORAL_REUSE = 1u << 3,
   UH_PASS_SPATIAL_REUSE = 1u << 4,
   UH_PASS_REFERENCE_PT = 1u << 5,
   UH_PASS_RESTIR = (1u << 0) | (1u << 1) | (1u << 2) | (1u << 3) | (1u << 4),
   UH_PASS_ALL = 0x3f
};

/* ---- status codes ---------------------------------------------------------------------- */
enum {
   UH_OK = 0,
   UH_ERR_INVALID_ARGUMENT = 1,
   UH_ERR_NO_DEVICE = 2,
   UH_ERR_HIP = 3,
   UH_ERR_CAPACITY = 4,   /* > 1024 materials/meshes/lights (utopian/src/renderer.rs:5-7) */
   UH_ERR_NOT_BUILT = 5,  /* render before uh_build_acceleration */
   UH_ERR_OUT_OF_MEMORY = 6
};

enum { UH_MAX_GPU_MATERIALS = 1024, UH_MAX_GPU_MESHES = 1024, UH_MAX_GPU_LIGHTS = 1024 };

/* ray kinds counted in UhStats.rays[] ("ray" = one traceRayEXT-equivalent query) */
enum { UH_RAY_PRIMARY = 0, UH_RAY_BOUNCE = 1, UH_RAY_SUN_SHADOW = 2, UH_RAY_LIGHT_SHADOW = 3, UH_RAY_GBUFFER = 4, UH_RAY_KINDS = 5 };

typedef struct UhStats {
   uint64_t rays[UH_RAY_KINDS]; /* since the last uh_reset_stats */
   uint64_t nodes_visited;      /* BVH4 nodes fetched by closest-hit traversals (only with uh_set_option("count_visits",1)) */
   uint64_t tris_tested;        /* triangle packets tested by closest-hit traversals (same option) */
   uint64_t shadow_nodes_visited;
   uint64_t shadow_tris_tested;
   uint64_t closest_hits;       /* closest-hit shader invocations for path rays */
   uint64_t misses;             /* miss (sky) evaluations for path rays */
   uint64_t frames;
   uint32_t bvh_nodes;          /* BVH4 node count */
   uint32_t bvh_triangles;
   float build_ms;              /* last uh_build_acceleration / uh_refit_acceleration, host wall time */
   float last_frame_ms;         /* hipEvent time of the last uh_render_frame (all passes) */
   float trace_closest_ms;      /* summed hipEvent time of closest-hit traversal launches since reset (option "time_kernels") */
   float trace_shadow_ms;       /* the sun shadow rays: grid kernel + the tree walk of what it hands over */
   float shade_ms;
   uint32_t trace_closest_launches;
   uint32_t sun_grid_cells;     /* the sun-direction visibility grid in use (0 = none: the sun shadow rays walk the tree) */
   uint32_t sun_grid_entries;   /* (triangle, cell) pairs it holds */
   float sun_grid_build_ms;     /* host time of its last build (once per sun direction and geometry) */
   float sun_grid_mean_list;    /* entries per occupied cell */
   uint64_t sun_tree_rays;      /* sun shadow rays the grid handed to the tree walk (border cells, long lists); part of rays[UH_RAY_SUN_SHADOW] */
   uint32_t camera_grid_cells;  /* the per-camera grid the primary rays go through (pixels; 0 = none: they walk the tree) */
   uint32_t camera_grid_entries;
   float camera_grid_build_ms;  /* host wall time of its last build (on the device; once per camera at rest and geometry) */
   float camera_grid_mean_list; /* entries per occupied pixel */
   uint64_t camera_tree_rays;   /* primary rays the grid handed to the tree walk (pixels with long lists); part of rays[UH_RAY_PRIMARY] */
   uint64_t camera_grid_tris_tested; /* triangle packets tested by the grid walk of the primary rays (option "count_visits") */
   float camera_grid_ms;        /* summed hipEvent time of the primary rays' launches when they go through the grid (option "time_kernels") */
   float trace_light_ms;        /* summed hipEvent time of the light shadow rays' traversal launches (reference.rgen:106-124; option "time_kernels"); not part of trace_shadow_ms */
   uint64_t sun_covered_rays;   /* sun shadow rays answered by their cell's cover depth alone (option "count_visits") */
   uint64_t sun_grid_bytes;     /* device memory of the sun grid in use: cell records + entry lists + coarse cover + the lists as 64-byte records (when within "sun_grid_inline_max_mb") */
   uint64_t camera_grid_bytes;  /* device memory of the camera grid in use: cell offsets + entry lists */
   uint64_t light_nodes_visited; /* BVH4 nodes fetched / triangle packets tested by the light shadow rays' walks (option "count_visits"); */
   uint64_t light_tris_tested;   /* not part of shadow_nodes_visited / shadow_tris_tested, which count the sun rays */
   uint32_t trace_light_launches;
   uint32_t reserved1;
} UhStats;

typedef struct uh_ctx uh_ctx;

/* ---- Stream ordering (what a caller may rely on; tests/test_gpu_stream_order.py holds the library to it) ----------------
 * uh_render_frame / uh_render_frames ENQUEUE and return; up to "frames_in_flight" frames run on streams of their own, the
 * reservoir passes on another. Every other verb that reads or writes device state WAITS for all frames in flight first and
 * is COMPLETE when it returns (its own copies and clears are waited for: the next frame may run on any of the library's
 * streams, none of which is ordered against the null stream):
 *   waits + complete on return:  uh_reset_stats, uh_get_stats, uh_reset_accumulation, uh_synchronize, every uh_read_*,
 *        uh_write_reservoirs, uh_write_gbuffer_position, uh_build_acceleration, uh_refit_acceleration (also when uh_render_frame calls it for
 *        view->rebuild_tlas), uh_set_tile_partition, uh_set_restir_partition, uh_rccl_attach / uh_rccl_detach, uh_pack_tiles,
 *        uh_unpack_tiles, uh_compose_tiles, uh_resolve_output, uh_add_isosurface_mesh, uh_destroy;
 *   enqueues like a frame, ordered behind the frames in flight and before those that follow:  uh_rccl_gather_tiles, uh_mgpu_compose;
 *        uh_set_option for "frames_in_flight" and for "time_kernels" 1 -> 0 (the others only change what the NEXT enqueued
 *        frame does: "furnace", "sun_grid*", "camera_grid*", "overlap", "batch_frames", "trace_blocks_per_cu", "count_visits",
 *        "full_frame_restir", "primary_implicit"; "device_build", "ploc_sah_top" invalidate the tree: the next frame
 *        needs uh_build_acceleration, which waits);
 *   host state only (no device access, nothing to wait for):  uh_add_mesh, uh_add_light, uh_set_instance_transform,
 *        uh_get_num_lights, uh_mesh_info, uh_read_mesh, uh_get_restir_rows, uh_tile_pack_count, uh_last_error;
 *   uh_add_texture_rgba8 uploads into a fresh allocation no frame in flight can reference (textures enter a frame's tables at the
 *        next uh_build_acceleration) and is complete on return;
 *   uh_trace_closest / uh_trace_any run on the context's first stream, in order with the frames of that stream, read the scene
 *        only, and are complete on return.
 * The sun-direction grid and the camera grid are (re)built inside the first frame call that wants them, after a wait for the frames in flight. */

/* ---- lifetime -------------------------------------------------------------------------- */
int uh_create(int device_ordinal, uint32_t width, uint32_t height, uh_ctx** out);
void uh_destroy(uh_ctx* ctx);
const char* uh_last_error(uh_ctx* ctx); /* ctx may be NULL: the last creation error - or, after a SUCCESSFUL uh_create, "" or a
                                         * "warning: ..." when the HIP runtime of the process is of another release (major.minor)
                                         * than the one the library was built with (also printed to stderr once per process) */
const char* uh_version(void);           /* "utopian-hip <v> (gfx950; built with HIP a.b.c; HIP runtime x.y.z)"; needs no GPU */
/* the same two releases as numbers, HIP_VERSION style (major * 10000000 + minor * 100000 + patch): the hipcc that compiled the
 * library, and hipRuntimeGetVersion() of the libamdhip64.so.7 this process bound (the soname covers every 7.x: a host that loaded
 * another copy first - the PyTorch wheel bundles one - hands it to this library as well). Either pointer may be NULL. */
int uh_hip_versions(int* built_with, int* runtime);

/* ---- scene ----------------------------------------------------------------------------- */
int uh_add_texture_rgba8(uh_ctx* ctx, const uint8_t* pixels, uint32_t w, uint32_t h, uint32_t* out_index);
/* material->diffuse_map must be an index returned by uh_add_texture_rgba8.
 * world3x4: row-major 3x4 object-to-world (VkTransformMatrixKHR layout, raytracing.rs:233-248). */
int uh_add_mesh(uh_ctx* ctx, const UhVertex* vertices, uint32_t num_vertices, const uint32_t* indices,
                uint32_t num_indices, const UhGpuMaterial* material, const float world3x4[12],
                uint32_t* out_mesh_index);
int uh_add_light(uh_ctx* ctx, const UhGpuLight* light, uint32_t* out_index);
int uh_get_num_lights(uh_ctx* ctx, uint32_t* out); /* Renderer::get_num_lights (renderer.rs:412) */
int uh_set_instance_transform(uh_ctx* ctx, uint32_t mesh_index, const float world3x4[12]);
int uh_build_acceleration(uh_ctx* ctx);
/* Raytracing::rebuild_tlas (raytracing.rs:400-459): after uh_set_instance_transform calls, re-bakes the
 * triangles and recomputes every box of the existing tree ON THE DEVICE (no host rebuild; topology kept).
 * Results equal a full uh_build_acceleration bit for bit - hits do not depend on the boxes - only the
 * traversal cost grows while instances drift from where the tree was built. UH_ERR_NOT_BUILT when meshes
 * or lights were added since the last build. uh_render_frame calls it by itself when transforms are
 * pending and view->rebuild_tlas == 1 (the flag the application sets, main.rs:392,526). */
int uh_refit_acceleration(uh_ctx* ctx);

/* ---- per frame ------------------------------------------------------------------------- */
/* UH_ERR_INVALID_ARGUMENT for view->num_bounces > 64, view->samples_per_frame > 4096, view->num_lights beyond the
 * lights added; UH_ERR_NOT_BUILT before uh_build_acceleration (or after moved instances without view->rebuild_tlas). */
int uh_render_frame(uh_ctx* ctx, const UhViewUniformData* view, uint32_t pass_mask);
/* `count` consecutive frames of the reference_pt pass with an unchanged camera: frame i is rendered
 * with total_samples = view->total_samples + i * samples_per_frame, exactly what `count` calls of
 * uh_render_frame under the application's frame protocol (prototype/src/main.rs:467-469) produce,
 * bit for bit; internally up to option "batch_frames" frames share one wavefront (path id =
 * frame * W*H + pixel) so that a rank owning few pixels still launches full-size kernels.
 * Only UH_PASS_REFERENCE_PT without reservoir light sampling can be batched (each ReSTIR frame
 * depends on the previous one); otherwise UH_ERR_INVALID_ARGUMENT. */
int uh_render_frames(uh_ctx* ctx, const UhViewUniformData* view, uint32_t pass_mask, uint32_t count);
int uh_reset_accumulation(uh_ctx* ctx);
int uh_synchronize(uh_ctx* ctx);

/* ---- read-back (host pointers; each call synchronises the context's stream) ------------ */
int uh_read_accumulation(uh_ctx* ctx, float* rgba32f /* W*H*4 */);
int uh_read_output_bgra8(uh_ctx* ctx, uint8_t* bgra /* W*H*4 */);
int uh_read_reservoirs(uh_ctx* ctx, int which /* 0 initial, 1 temporal, 2 spatial */, UhReservoir* out /* W*H */);
int uh_read_gbuffer_position(uh_ctx* ctx, float* rgba32f /* W*H*4, un-filtered texels */);
/* upload a reservoir buffer (tests seed the temporal history with it) */
int uh_write_reservoirs(uh_ctx* ctx, int which, const UhReservoir* in /* W*H */);
/* upload gbuffer_position (tests run the reservoir passes on given positions: frames without UH_PASS_GBUFFER read what is there) */
int uh_write_gbuffer_position(uh_ctx* ctx, const float* rgba32f /* W*H*4 */);

/* ---- stand-alone ray queries through the same traversal kernels (parity tests) ---------- */
/* rays: n * 8 floats (ox,oy,oz,tmin,dx,dy,dz,tmax); hits: n * 4 words (t,u,v as f32, then
 * (mesh_index << 22 | primitive) as u32, 0xffffffff on miss... see DESIGN.md "hit record") */
int uh_trace_closest(uh_ctx* ctx, const float* rays, uint32_t n, float* out_tuv /* n*3 */,
                     uint32_t* out_mesh /* n */, uint32_t* out_prim /* n */);
int uh_trace_any(uh_ctx* ctx, const float* rays, uint32_t n, uint8_t* out_occluded /* n */);

/* ---- stats / options ------------------------------------------------------------------- */
int uh_get_stats(uh_ctx* ctx, UhStats* out);
int uh_reset_stats(uh_ctx* ctx);
/* The 25 options (DESIGN.md section 7 has the defaults and what was measured); unknown names return UH_ERR_INVALID_ARGUMENT.
 *  diagnostics   "count_visits" (0/1: UhStats' node / triangle / cover counters), "time_kernels" (0/1: hipEvent time per kernel kind)
 *  results       "full_frame_restir" (0/1; 1 = documented divergence: the reservoir for every pixel instead of the reference's
 *                x > W/2 split), "furnace" (0/1: the reference's FURNACE_TEST build of the miss shader, reference.rmiss:14-28 - a path
 *                ray that leaves the scene returns white whatever view->sky_enabled says), "iso_reference_triangulation" (0/1,
 *                default 1: see uh_add_isosurface_mesh)
 *  the tree      "device_build" (0/1/2; 1 or 2 = uh_build_acceleration builds the tree ON THE DEVICE in a few ms instead of the host
 *                SAH tree in tens to hundreds: same hits bit for bit, about 10 % (1: clusters under a SAH top) or 30 % (2: radix tree)
 *                more traversal work per ray - for geometry that changes every few frames), "ploc_sah_top" (clusters the PLOC rounds
 *                stop at; 0 = PLOC to the root)
 *  sun grid      "sun_grid" (0/1, default 1: sun shadow rays through a per-direction visibility grid once the direction has settled;
 *                same images), "sun_grid_build" (0/1, default 1: built on the device in a few milliseconds; 0: by the host builder,
 *                the reference implementation, in 130-550 ms), "sun_grid_density" (entries per triangle the cell size aims at),
 *                "sun_grid_max_mb" (budget of the entry lists), "sun_grid_max_walk" (longest list a ray tests itself),
 *                "sun_grid_force" (0/1: 1 = never refused for its worth - long lists, much of the surface handed to the tree), "sun_grid_inline_max_mb" (the lists a
 *                second time as 64-byte records that carry their triangle packet - one round trip per triangle test instead of two:
 *                budget in MB, -1 = default = four times the packet array, 0 = never), "sun_grid_coarse" (0..6, default 2: a cover
 *                depth per block of 4 x 4 cells, small enough to stay in the L2, asked before the cell's own record; 0: none)
 *  camera grid   "camera_grid" (0/1, default 1: the primary rays of a camera that has been the same for two consecutive frame calls -
 *                or for a call of 8 or more frames - go through a per-camera grid of packet lists, one cell per pixel, instead of
 *                the tree; same hit records bit for bit), "camera_grid_max_walk", "camera_grid_walk_whole",
 *                "camera_grid_max_mean_list_x10", "primary_implicit" (0/1, default 1: with that grid in use and one sample per frame,
 *                the primary rays' state is not stored - the kernels of the first bounce compute it from the path id; same images)
 *  scheduling    "frames_in_flight" (1..8, default 4), "batch_frames" (0 = auto), "overlap" (0/1, default 1: the miss shader and the
 *                shadow traversals on a second stream beside the next bounce's traversal), "trace_blocks_per_cu" (1..8: persistent
 *                grid of the traversal kernels), "fused_bounces" (default 1: a frame that goes alone - uh_render_frame, or a call
 *                of one frame - and finds the GPU idle, i.e. a caller that waits for its frames, runs its bounces 1 .. inside one
 *                persistent kernel, a wavefront per block, instead of four launches per bounce: same images, 2.5 ms against 2.85
 *                for a 1080p frame, 0.99 against 1.57 at 960 x 540; with frames in flight the launches interleave better and are
 *                kept, and so they are for frames of more than 4 M paths, whose launches are large already. 0: never; -1: always;
 *                2..8: as 1, and the kernel's blocks per CU)
 * Removed in round 5 with the measured-negative variants they selected: "closest_variant" / "shadow_variant" / "trace_variant" (batch
 * traversal kernels), "primary_tiles", "interleave", "sun_grid_fused", "sun_leftover_batch", "sun_grid_async", "sun_grid_inline",
 * "spatial_splits", "bvh_optimise", "raw_visit_counts", "ploc_radius", "overlap_miss" / "overlap_shadow" (now "overlap"),
 * "sun_grid_max_mean_list_x10" / "sun_grid_max_fallback_pct" (now "sun_grid_force"),
 * "closest_blocks_per_cu" / "shadow_blocks_per_cu" (now "trace_blocks_per_cu"), "single_frame_blocks_per_cu", "miss_blocks_per_cu". */
int uh_set_option(uh_ctx* ctx, const char* name, int value);

/* diagnostics: the sun-direction grid in use (built on the device, option "sun_grid_build" = 1) read back and held against the host
 * builder - the reference implementation whose margins tests/cpp/sun_grid_check.cpp checks against brute force - run on the same
 * packets and the same raster. out[0] cells, out[1] / out[2] entries of the device / host grid, out[3] cells whose list length
 * differs, out[4] cells whose list differs (element by element where a ray may walk it - interior cells of at most
 * "sun_grid_max_walk" entries -, as a set elsewhere), out[5] cells whose cover depth differs in any bit, out[6] walkable cells
 * compared, out[7] the host builder's time in microseconds. UH_ERR_INVALID_ARGUMENT when no grid is in use. */
int uh_sun_grid_compare_builders(uh_ctx* ctx, uint64_t out[8]);

/* ---- multi-GPU framebuffer tile partition (one process per GPU) ------------------------ */
/* After this call uh_render_frame path-traces only pixels of tiles t with t % world == rank
 * (tile_size x tile_size tiles, row-major tile ids). ReSTIR passes stay full-frame unless uh_set_restir_partition says otherwise. */
int uh_set_tile_partition(uh_ctx* ctx, uint32_t rank, uint32_t world, uint32_t tile_size);
/* number of float4 pixels uh_pack_tiles writes for `rank` (padded: whole tiles) */
int uh_tile_pack_count(uh_ctx* ctx, uint32_t rank, uint64_t* out_pixels);
/* pack this rank's owned tiles of the RGBA32F accumulation into a contiguous DEVICE buffer */
int uh_pack_tiles(uh_ctx* ctx, void* device_out, uint64_t capacity_pixels);
/* scatter `from_rank`'s packed tiles (DEVICE buffer) into this context's accumulation image */
int uh_unpack_tiles(uh_ctx* ctx, uint32_t from_rank, const void* device_in, uint64_t num_pixels);
/* the root's whole composition in one launch: `device_all` holds `world` packed buffers (rank r's at r * stride_pixels
 * float4 pixels, as uh_pack_tiles wrote them; the root's own slot is not read); scatters every other rank's tiles into the
 * accumulation image and recomputes pt_output_image (= uh_unpack_tiles for every rank + uh_resolve_output) */
int uh_compose_tiles(uh_ctx* ctx, const void* device_all, uint64_t stride_pixels, uint32_t total_samples, uint32_t accumulation_limit);
/* recompute pt_output_image from the accumulation image (after uh_unpack_tiles on the root) */
int uh_resolve_output(uh_ctx* ctx, uint32_t total_samples, uint32_t accumulation_limit);
/* ---- multi-GPU, the reservoir passes: a band of rows per rank + one exchange per frame ------ */
/* The path tracer needs spatial_reuse_reservoirs only at its own pixels, but temporal_reuse reads last frame's buffer at a
 * reprojected pixel (restir/temporal_reuse.rgen:90-99) and spatial_reuse gathers from a 30-pixel neighbourhood
 * (restir/spatial_reuse.rgen:40-60), so run full-frame on every rank these passes do not scale (alternative).
 * After uh_set_restir_partition(rank, world) the G-buffer cast and the reservoir passes of this context cover
 *   spatial_reuse      rows [rank * B, min(H, (rank + 1) * B)),  B = ceil(H / world)        (the band)
 *   reset / initial / temporal   the band +- 30 rows, plus row H - 1 for the first band (spatial_reuse.rgen:54: a row
 *                      offset below zero wraps and is clamped to the last row)
 *   G-buffer cast      those rows and the row above each (the 2 x 2 corner filter of initial_ris.rgen:22-23)
 * and every spatial pass is followed by ONE call of `exchange`, which must enqueue on `hip_stream` whatever makes
 * spatial_base[k * band_bytes, (k + 1) * band_bytes) hold rank k's band for every k (an in-place all-gather: this rank's band
 * is already where it belongs). It is called while the frame is ENQUEUED, not when it runs: it must not wait for the GPU.
 * Each spatial_reuse buffer is world * B rows long (the frame, padded to equal bands). With the exchange in place every
 * rank holds the whole spatial_reuse_reservoirs of every frame - bit for bit the single-GPU buffer - while buffers 0 and 1
 * (uh_read_reservoirs) are valid on the rank's own rows only. world = 1 (the default) restores full-frame passes (an
 * exchange given with world = 1 is still called: a one-rank all-gather, for rehearsals). exchange == NULL with world > 1
 * leaves the other bands stale: for timing one rank's share only. The call waits for the frames in flight and keeps the
 * temporal history. */
typedef int (*UhRestirExchangeFn)(void* user, void* hip_stream, void* spatial_base, uint64_t band_bytes, uint32_t rank, uint32_t world);
int uh_set_restir_partition(uh_ctx* ctx, uint32_t rank, uint32_t world, UhRestirExchangeFn exchange, void* user);
/* the rows of this context's band and of its reservoir / G-buffer passes (counts of rows; *_extra_row0 is the first row of the second interval or 0 with *_extra_rows 0) */
typedef struct UhRestirRows {
   uint32_t band_row0, band_rows;        /* spatial_reuse */
   uint32_t reuse_row0, reuse_rows, reuse_extra_row0, reuse_extra_rows;   /* reset, initial_ris, temporal_reuse */
   uint32_t cast_row0, cast_rows, cast_extra_row0, cast_extra_rows;       /* G-buffer cast */
   uint32_t rows_per_band;               /* B */
} UhRestirRows;
int uh_get_restir_rows(uh_ctx* ctx, UhRestirRows* out);
/* RCCL, built in (one process per GPU): librccl is opened at run time (the library does not link it). Rank 0 makes an id, the
 * job's launcher hands the 128 bytes to every rank (rust-renderer_amd/launch.py: a TCP socket on 127.0.0.1 - no torch in a GPU
 * process), every rank attaches: ncclCommInitRank + uh_set_restir_partition(rank, world, <ncclAllGather on the reservoir stream>);
 * a job that wants full-frame reservoir passes calls uh_set_restir_partition(ctx, 0, 1, NULL, NULL) afterwards (the communicator
 * stays for uh_rccl_gather_tiles). */
int uh_rccl_unique_id(uint8_t out_id[128]);
int uh_rccl_attach(uh_ctx* ctx, uint32_t rank, uint32_t world, const uint8_t id[128]);
int uh_rccl_detach(uh_ctx* ctx);
/* ranks of the communicator uh_rccl_attach made, as RCCL itself counts them (ncclCommCount); 0 when none is attached */
int uh_rccl_comm_count(uh_ctx* ctx, uint32_t* out_ranks);
/* The composition of a tile-partitioned frame over that communicator (ONE gather per composed image): every rank packs
 * its tiles of pt_accumulation_image (uh_pack_tiles' layout) and sends them to `root` - grouped ncclSend / ncclRecv, the peers' tiles
 * arrive on distinct xGMI links at once -, the root scatters them into its accumulation image and recomputes pt_output_image with
 * (total_samples, accumulation_limit) in one launch (the two images of renderers/mod.rs:199-214,354-358, which the reference's
 * single device holds whole). Collective: every rank of the communicator calls it, with the same root, after
 * uh_set_tile_partition(rank, world, tile) with the communicator's rank and size. ENQUEUED on the context's stream behind the frames
 * in flight - no host wait, no staging through the host; the root's uh_read_* (or uh_synchronize) waits for it, and frames enqueued
 * after it accumulate behind it. */
int uh_rccl_gather_tiles(uh_ctx* ctx, uint32_t root, uint32_t total_samples, uint32_t accumulation_limit);
/* raw device pointers (zero-copy wrap by the caller, e.g. for RCCL): 0 accumulation RGBA32F,
 * 1 output BGRA8 */
int uh_device_pointer(uh_ctx* ctx, int which, void** out);
/* the HIP stream all work of this context is enqueued on (hipStream_t as void*); also makes the context's device
 * the calling thread's current device */
int uh_stream(uh_ctx* ctx, void** out);

/* ---- GPU extraction of the reference's marching-cubes density field (BASELINE configs[4]) --------
 * Adds the iso-surface {density = 0} of utopian/shaders/marching_cubes/marching_cubes.comp:83-103 (torus over a box,
 * plus the sphere of radius 8 |sin(0.3 time)|; shapes placed in a 32-unit domain) sampled on a resolution^3 grid over
 * [lo, hi]^3 as one mesh with uh_add_mesh semantics. Extraction runs on the device: marching cubes on the reference's case table,
 * cell by cell the triangles marching_cubes.comp:231-251 emits - same vertices (vertexInterp in the shader's corner order), same
 * order within a cell, zero-area triangles included; cells in x-fastest order (the reference's order is whatever its atomics
 * give). Option "iso_reference_triangulation" = 0 selects this repository's own tables (other interior diagonals, slivers
 * dropped). Vertex normals come from the density gradient (generateNormal), uv = position.xz / (hi - lo). *out_triangles receives the triangle count; when nothing
 * crosses the iso value no mesh is added and *out_mesh_index is 0xffffffff. UH_ERR_CAPACITY above 4 Mi triangles. */
int uh_add_isosurface_mesh(uh_ctx* ctx, uint32_t resolution, float lo, float hi, float time, const UhGpuMaterial* material,
                           const float world3x4[12], uint32_t* out_mesh_index, uint32_t* out_triangles);
/* diagnostics of the extraction: per cell (x fastest, resolution^3 of them) the marching-cubes case index (bit i set when corner i
 * is outside, marching_cubes.comp:185-190) and the number of triangles the cell contributes (zero-area ones dropped); either
 * pointer may be NULL. The oracle's restatement of the shader is compared with these cell by cell. */
int uh_isosurface_cells(uh_ctx* ctx, uint32_t resolution, float lo, float hi, float time, uint8_t* out_cube_index, uint8_t* out_triangle_count);
/* the context's host copy of a mesh (Model keeps CPU copies, primitive.rs:19-24): sizes, then the data */
int uh_mesh_info(uh_ctx* ctx, uint32_t mesh_index, uint32_t* num_vertices, uint32_t* num_indices);
int uh_read_mesh(uh_ctx* ctx, uint32_t mesh_index, UhVertex* vertices, uint32_t* indices);

/* ---- several GPUs behind ONE application process (SURVEY.md section 8b "multi-GPU", 8e) ------------
 * The reference application is a single process with one render thread (prototype/src/main.rs:86-570);
 * a maintainer who wants N GPUs behind it binds this group instead of one uh_ctx. Every verb above has a
 * uh_mgpu_ twin with the same meaning: scene verbs replicate the scene on every GPU, frame verbs make GPU i
 * path-trace the tiles t % N == i (tile_size x tile_size, row-major ids; ReSTIR / G-buffer passes run
 * full-frame on every GPU, identical results), nothing is exchanged per frame, and the read-backs (or
 * uh_mgpu_compose) gather the packed RGBA32F tiles onto GPU 0 with peer copies over xGMI and recompute
 * pt_output_image there. Pixels are bit-identical to a single-GPU render (RNG keyed on absolute pixel
 * coordinates, random.glsl:14-18). device_ordinals == NULL means GPUs 0..ngpus-1; the same ordinal may
 * appear several times (how the 1-GPU tests exercise this layer). */
typedef struct uh_mgpu uh_mgpu;
int uh_mgpu_create(int ngpus, const int* device_ordinals, uint32_t width, uint32_t height, uint32_t tile_size, uh_mgpu** out);
void uh_mgpu_destroy(uh_mgpu* group);
const char* uh_mgpu_last_error(uh_mgpu* group); /* NULL: the last creation error */
int uh_mgpu_num_devices(uh_mgpu* group);
uh_ctx* uh_mgpu_context(uh_mgpu* group, int index); /* GPU i's context (options, stats, queries); owned by the group */
int uh_mgpu_add_texture_rgba8(uh_mgpu* group, const uint8_t* pixels, uint32_t w, uint32_t h, uint32_t* out_index);
int uh_mgpu_add_mesh(uh_mgpu* group, const UhVertex* vertices, uint32_t num_vertices, const uint32_t* indices, uint32_t num_indices,
                     const UhGpuMaterial* material, const float world3x4[12], uint32_t* out_mesh_index);
int uh_mgpu_add_light(uh_mgpu* group, const UhGpuLight* light, uint32_t* out_index);
int uh_mgpu_get_num_lights(uh_mgpu* group, uint32_t* out);
int uh_mgpu_set_instance_transform(uh_mgpu* group, uint32_t mesh_index, const float world3x4[12]);
int uh_mgpu_build_acceleration(uh_mgpu* group);  /* the N host builds run concurrently */
int uh_mgpu_refit_acceleration(uh_mgpu* group);
int uh_mgpu_render_frame(uh_mgpu* group, const UhViewUniformData* view, uint32_t pass_mask);   /* enqueues on every GPU, does not wait */
int uh_mgpu_render_frames(uh_mgpu* group, const UhViewUniformData* view, uint32_t pass_mask, uint32_t count);
int uh_mgpu_reset_accumulation(uh_mgpu* group);
int uh_mgpu_synchronize(uh_mgpu* group);
int uh_mgpu_compose(uh_mgpu* group);             /* gather tiles to GPU 0 + resolve; a no-op until the next frame */
int uh_mgpu_read_accumulation(uh_mgpu* group, float* rgba32f /* W*H*4 */);  /* compose, then read GPU 0 */
int uh_mgpu_read_output_bgra8(uh_mgpu* group, uint8_t* bgra /* W*H*4 */);
/* reservoir buffers of the whole frame (0 initial, 1 temporal: each GPU's band of rows; 2 spatial: complete on every GPU) */
int uh_mgpu_read_reservoirs(uh_mgpu* group, int which, UhReservoir* out /* W*H */);
int uh_mgpu_get_stats(uh_mgpu* group, UhStats* out); /* counters summed over GPUs, times = slowest GPU */
int uh_mgpu_reset_stats(uh_mgpu* group);
/* every context's options, plus "restir_partition" (default 1 for more than one GPU): the G-buffer cast and the reservoir
 * passes by bands of rows, one band per GPU (uh_set_restir_partition), the bands exchanged by peer copies after every
 * spatial pass; 0 = every GPU runs them for the whole frame */
int uh_mgpu_set_option(uh_mgpu* group, const char* name, int value);

#ifdef __cplusplus
}
#endif

#endif /* UTOPIAN_HIP_H */
