// device_types.h — structs passed between the host context (context.hip) and the kernels
// (kernels.hip). Device pointers only; everything here is plain data.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "bvh.h"
#include "sun_grid.h"
#include "utopian_hip.h"

namespace uh {

constexpr uint32_t kMaxBounces = 64;
// per bounce: RAY (paths whose ray the bounce traces; the shading kernels tell hits from misses by the hit
// record, so no hit / miss queues exist) and LIGHT (scattered paths that carry a light sample)
constexpr uint32_t kQueueKinds = 5;
// Q_SUN_TREE: sun rays the grid kernel hands to the tree walk (border cells, long lists). Q_MISS: the paths of a bounce whose ray
// left the scene - k_shade_hit meets them while it classifies the bounce's RAY queue and hands their ids to k_shade_miss
// Q_CAM_TREE (bounce 0 only): primary rays the camera grid hands to the tree walk (pixels with long lists); shares queue 3 with
// Q_SUN_TREE, which the same bounce's sun rays fill only after the shading kernel
enum { Q_RAY = 0, Q_LIGHT = 1, Q_SUN_TREE = 2, Q_MISS = 3, Q_CAM_TREE = 4 };
constexpr uint32_t kLaunchSlots = kMaxBounces * 4 + 8;  // per bounce: closest, sun (grid), sun leftovers (tree), light; bounce 0: + camera grid, its leftovers

// Queues are sharded: path p lives in shard shard_of_run(p / 64) for its whole life, every queue
// has one segment (capacity PathState::shard_cap) and one counter per shard, and the blocks of a
// launch are bound to shards by blockIdx % kShards. A single device-wide counter saturates at
// ~88 atomics/us on MI355X (MI355X_MICROARCH.md "dequeue"), which capped every queue-building
// kernel of the unsharded first version (profiles/r01a_*): 32 shards lift that ceiling 32x - provided
// the 32 counters live in 32 different 128-byte lines (see Control below).
// Blocks b and b + 8 share an XCD under the observed round-robin dispatch, so shard s is served by
// XCD s % 8 and its queue segments stay in that XCD's L2 between producer and consumer launches.
constexpr uint32_t kShards = 32;
constexpr uint32_t kShardBits = 5;
static_assert((1u << kShardBits) == kShards, "kShards is a power of two");
// Fibonacci hash of the 64-path run index (top 5 bits). A plain `run % 32` aliases with the tile
// partition: with 64-pixel-wide tiles every run a rank owns is even, which left half of the shards
// (and half of the traversal blocks) empty on every rank of a multi-GPU job.
__host__ __device__ inline uint32_t shard_of_run(uint32_t run) { return (run * 0x9E3779B1u) >> (32 - kShardBits); }

// Zeroed once per sample pass by one hipMemsetAsync.
// Counter layout: the words that one launch hammers concurrently (the same bounce and queue kind, or
// the same launch slot, of all 32 shards) must not share a 128-byte line - atomics to one LINE
// serialise in its L2 channel just like atomics to one word. So the shard (and queue kind) select
// the line and the bounce / launch slot the word inside it.
constexpr uint32_t kBounceStride = ((kMaxBounces + 1 + 31) / 32) * 32;  // words, whole lines
constexpr uint32_t kSlotStride = ((kLaunchSlots + 31) / 32) * 32;
struct Control {
   uint32_t q_count[kShards * kQueueKinds * kBounceStride];
   uint32_t cursor[kShards * kSlotStride];  // persistent-thread work cursors, one per (shard, launch of the pass)
};
__host__ __device__ inline uint32_t qc_index(uint32_t bounce, uint32_t kind, uint32_t shard) { return (shard * kQueueKinds + kind) * kBounceStride + bounce; }
__host__ __device__ inline uint32_t cursor_index(uint32_t slot, uint32_t shard) { return shard * kSlotStride + slot; }

// Persistent across frames; read back by uh_get_stats.
struct DeviceStats {
   unsigned long long rays[UH_RAY_KINDS];
   unsigned long long nodes_visited, tris_tested, shadow_nodes_visited, shadow_tris_tested;
   unsigned long long closest_hits, misses;
   unsigned long long sun_tree_rays;  // sun rays k_trace_sun_grid handed to the tree walk
   unsigned long long cam_tree_rays;  // primary rays k_trace_camera_grid handed to the tree walk
   unsigned long long cam_tris_tested;  // triangle packets k_trace_camera_grid tested (count_visits)
   unsigned long long sun_covered_rays; // sun rays k_trace_sun_grid answered from the cell's cover depth alone (count_visits)
   unsigned long long light_nodes_visited, light_tris_tested;  // the light shadow rays' walks (count_visits)
};

// per-mesh shading record (80 B): inverse instance rotation/scale + the material fields the
// closest-hit shader reads (reference.rchit:22-23,32,40-41,47-89)
struct alignas(16) MeshShade {
   float w2o[9];  // row-major inverse of the object-to-world upper 3x3
   uint32_t diffuse_map;
   float base_color[3];
   float type;      // raytrace_properties.x
   float property;  // raytrace_properties.y
   uint32_t pad;
   float metallic, roughness;  // metallic_factor / roughness_factor: read by material type 4 only (Cook-Torrance extension)
   float pad2[2];
};
static_assert(sizeof(MeshShade) == 80, "mesh shading record");

struct TexInfo {
   const uchar4* texels;
   uint32_t w, h;
   // tiles_x > 0: texels are stored in 8x8-texel tiles (256 B, row-major inside the tile, tiles
   // row-major), so the 2x2 bilinear footprint usually sits in one 128-B line instead of two rows.
   // 0: plain row-major (sizes that are not multiples of 8).
   uint32_t tiles_x;
   uint32_t pad;
};

struct SceneDev {
   const uint4* nodes;    // 3 uint4 per Node4C (quantised BVH4 node with implicit child addresses, 48 B)
   const float4* tris;    // 3 float4 per TriPacket
   const float4* shade;   // 4 float4 per ShadePacket
   const MeshShade* meshes;
   const TexInfo* textures;
   const float4* lights;  // 2 float4 per light: (pos, 0), (intensity, 0)
   const float* unorm_lut;  // 256 entries: c / 255.0f (host-computed, exact)
   uint32_t num_nodes, num_tris, num_meshes, num_textures, num_lights;
};

constexpr uint32_t kMaxBatchFrames = 32;

struct FrameParams {
   // A launch chain may carry `batch_frames` consecutive frames of the path-tracing pass as one
   // wavefront: path id = f * (W*H) + pixel. Frames differ only in their RNG frame number and in
   // total_samples (prototype/src/main.rs:467-469 adds samples_per_frame per frame); the
   // accumulate / store tail applies them in frame order, so results equal one-by-one rendering.
   uint32_t batch_frames;
   uint32_t frame_numbers[kMaxBatchFrames];
   uint32_t total_samples_of[kMaxBatchFrames];
   // spatial_reuse_reservoirs of each frame of the batch (rgen:98; a ring in the context: context.hip render_batch)
   const UhReservoir* spatial_of[kMaxBatchFrames];
   float inv_view[16], inv_proj[16], prev_pv[16];
   float sun_dir[3];  // normalize(view.sun_dir), computed on the host with the contract's normalize
   uint32_t W, H, frame_number;
   uint32_t samples_per_frame, total_samples, num_bounces, accumulation_limit;
   uint32_t sky_enabled, sun_shadow_enabled, lights_enabled, use_ris, full_frame_restir;
   // option "primary_implicit" (camera grid in use, one sample per frame): the state planes of bounce 0 are not materialised -
   // k_generate only fills the ray queue with path ids, and every kernel of bounce 0 computes a path's origin, direction,
   // throughput (1) and RNG words from its id (device_math.h primary_state): 48 bytes less written and 96 less read per path
   uint32_t primary_implicit;
   uint32_t furnace;  // option "furnace": the reference's FURNACE_TEST build of the miss shader (reference.rmiss:14-28): a miss returns white
   uint32_t num_lights_used;  // min(view.num_lights, view.max_num_lights_used)
   uint32_t temporal_enabled, spatial_enabled;
   uint32_t tp_rank, tp_world, tp_tile, tiles_x;
   // pixels this rank owns under the tile partition, ascending; nullptr = all W*H pixels. The
   // per-pixel kernels (generate, finish_sample) walk this list, so their cost shrinks with 1/world.
   const uint32_t* owned_pixels;
   uint32_t n_owned;
};

// Path state: four 16-byte quads per path, each quad in a PLANE of its own (SoA), and - round 4 - indexed by the path's POSITION
// IN ITS BOUNCE'S RAY QUEUE (shard segment + position), not by path id. Two SETS of planes ping-pong: set b & 1 holds the paths of
// bounce b's queue; k_shade_hit(b) reads a path's state at its position in queue b and writes the scattered path's new state at
// the position it gets in queue b + 1 (in set (b + 1) & 1). Why: every kernel of the path walks a queue, so state by position is
// read and written as contiguous streams (a wave's scattered paths get consecutive positions from one wave-level append), where
// state by path id was one 64-byte sector per 16-byte record once the first bounce had thinned the ids out (rounds 1-3: the
// shading and sun-ray kernels ran at the memory system's random-sector rate, profiles/r03_counters.json). The price: the
// radiance travels with the path (plane 3: k_shade_hit copies it forward, +32 dense bytes per scattered hit) instead of resting
// in a per-id array, and a path's final radiance is written where the path ends - miss, absorption, or the flush of the paths
// still alive after the last bounce - into `radf`, by path id, once.
// The RNG words ride in the w components the constant ray range (rgen:45-47: 0.001, 10000) does not need:
//   plane 0  ray origin.xyz            | raygen rngState (bits)            reference.rgen:24,31
//   plane 1  ray direction.xyz         | rayPayload.randomSeed (bits)      un-normalised direction (rgen:61); seed: rgen:30, rchit:91
//   plane 2  throughput.rgb            | light weight f                    radiance += throughput * f when the light is visible (rgen:121)
//   plane 3  radiance.rgb              | light index (bits)                rgen:69-78, :118-122 add to it; not materialised for bounce 0 (it is zero)
// The hit record of a bounce's ray (t, u, v | packet index, 0xffffffff = miss) lies in a plane of its own at the same position.
constexpr uint32_t kRecQuads = 4;
enum { REC_ORIGIN = 0, REC_DIR = 1, REC_THR = 2, REC_RAD = 3 };
struct PathRecs {
   float4* base;
   size_t plane;  // float4 between two planes (the queue capacity plus a stagger: the planes must not alias in the caches)
};
struct PathState {
   PathRecs set[2];  // set b & 1: state of the paths of bounce b's RAY queue, by queue position
   float4* hit;      // hit record of the bounce being traced, by queue position
   float4* radf;     // by path id: a finished path's radiance.rgb | its raygen rngState (the frame's next sample starts from it, rgen:28-31)
   float4* pixcol;   // by path id: sum over the frame's samples
   // 0,1 = ray ping-pong: path ids; 2 = light, 3 = sun rays for the tree walk: positions in the NEXT bounce's ray queue; 4 = misses:
   // (position in the CURRENT one, id) pairs; each kShards * shard_cap entries
   uint32_t* queue[5];
   uint32_t shard_cap;  // entries per shard segment = pixels a shard can own (multiple of 64)
};
__host__ __device__ inline float4* rec_quad(const PathRecs& rec, uint32_t pos, int quad) { return rec.base + rec.plane * (size_t)quad + pos; }

// rays of the stand-alone queries (uh_trace_closest, the G-buffer cast): record i = ray i, tmin / tmax in the w components
struct RawRays {
   float4* ray_o;   // origin.xyz, tmin
   float4* ray_d;   // direction.xyz, tmax
   float4* hit;     // t, u, v, packet index (bits)
};

struct Images {
   float4* accumulation;  // pt_accumulation_image RGBA32F
   uchar4* output;        // pt_output_image B8G8R8A8_UNORM
   float4* gbuffer_pos;   // gbuffer_position RGBA32F, un-filtered texels
   UhReservoir* reservoirs[3];          // initial, temporal, spatial (the buffer this frame's spatial pass writes / the path tracer reads)
   const UhReservoir* prev_spatial;     // last frame's spatial_reuse_reservoirs, read by the temporal pass (renderers/mod.rs:294)
};

// Rows of the frame a reservoir-pass launch covers: the whole frame on one GPU; on a rank of an N-rank job (DESIGN.md
// section 5, "band partition") the rank's band of rows plus what its spatial pass gathers from - at most two row intervals
// (spatial_reuse.rgen:54's uvec2 wrap sends the first 30 rows to row H - 1). Work item j of a launch is pixel pixel_of(j).
struct RowSpans {
   uint32_t row0[2], rows[2];  // interval k covers rows [row0[k], row0[k] + rows[k]); rows[1] == 0: one interval
   uint32_t W;
   __host__ __device__ uint32_t total() const { return (rows[0] + rows[1]) * W; }
   __host__ __device__ uint32_t pixel_of(uint32_t j) const {
      const uint32_t first = rows[0] * W;
      return j < first ? row0[0] * W + j : row0[1] * W + (j - first);
   }
};
inline RowSpans whole_frame(uint32_t W, uint32_t H) { return RowSpans{{0, 0}, {H, 0}, W}; }

// launch wrappers implemented in kernels.hip --------------------------------------------------
struct LaunchCfg {
   hipStream_t stream;
   uint32_t num_cus;
   uint32_t closest_blocks_per_cu, shadow_blocks_per_cu;
   bool count_visits;
   uint32_t fused_blocks_per_cu = 4;  // k_path_fused (kernels.hip UH_FUSED_BLOCKS)
};

void launch_generate(const LaunchCfg&, const FrameParams&, const PathState&, Control*, uint32_t sample);
void launch_trace_closest(const LaunchCfg&, const SceneDev&, const PathState&, Control*, DeviceStats*, uint32_t bounce, uint32_t cursor_slot,
                          int ray_kind);
void launch_trace_camera_grid(const LaunchCfg&, const FrameParams&, const SceneDev&, const PathState&, Control*, DeviceStats*, uint32_t cursor_slot_grid, uint32_t cursor_slot_tree,
                              const SunGridDev&, bool leftovers_possible = true);  // false: the grid's longest list is one its kernel walks itself - no tree-walk launch behind it
void launch_shade_miss(const LaunchCfg&, const FrameParams&, const PathState&, Control*, DeviceStats*, uint32_t bounce);
void launch_shade_hit(const LaunchCfg&, const FrameParams&, const SceneDev&, const PathState&, const Images&, Control*, DeviceStats*, uint32_t bounce);
// the paths still alive after the last bounce hand their radiance to the per-id array k_finish_sample reads
void launch_flush_survivors(const LaunchCfg&, const FrameParams&, const PathState&, Control*);
void launch_path_fused(const LaunchCfg&, const FrameParams&, const SceneDev&, const PathState&, Control*, DeviceStats*, const SunGridDev& g, bool use_grid, bool sun_of_bounce0);
void launch_trace_shadow(const LaunchCfg&, const FrameParams&, const SceneDev&, const PathState&, Control*, DeviceStats*, uint32_t bounce,
                         uint32_t cursor_slot, bool light, bool sun_leftovers = false);
// sun shadow rays through the per-direction grid (sun_grid.h) instead of the tree
void launch_trace_sun_grid(const LaunchCfg&, const FrameParams&, const SceneDev&, const PathState&, Control*, DeviceStats*, uint32_t bounce,
                           uint32_t cursor_slot, const SunGridDev&);
void launch_finish_sample(const LaunchCfg&, const FrameParams&, const PathState&, const Images&, uint32_t sample, bool last);
void launch_resolve(const LaunchCfg&, const Images&, uint32_t W, uint32_t H, uint32_t total_samples, uint32_t limit);
// G-buffer + ReSTIR
// the reservoir passes and the G-buffer cast over the rows of `spans`; `counted` = the G-buffer rays this launch adds to the
// statistics (a rank counts its own band, not the halo it casts again)
void launch_gbuffer(const LaunchCfg&, const FrameParams&, const SceneDev&, const RawRays&, const Images&, DeviceStats*, const RowSpans& spans, uint32_t counted,
                    const SunGridDev* camera_grid = nullptr);  // camera_grid: the cast walks the per-camera grid instead of the tree
void launch_reset_reservoirs(const LaunchCfg&, const FrameParams&, const Images&, const RowSpans& spans);
void launch_initial_ris(const LaunchCfg&, const FrameParams&, const SceneDev&, const Images&, const RowSpans& spans);
void launch_temporal_reuse(const LaunchCfg&, const FrameParams&, const SceneDev&, const Images&, const RowSpans& spans);
void launch_spatial_reuse(const LaunchCfg&, const FrameParams&, const SceneDev&, const Images&, const RowSpans& spans);
// stand-alone queries (n rays in ray_o/ray_d[0..n), identity queue)
void launch_trace_closest_raw(const LaunchCfg&, const SceneDev&, const float4* ray_o, const float4* ray_d, float4* hit, uint32_t n);
void launch_trace_any_raw(const LaunchCfg&, const SceneDev&, const float4* ray_o, const float4* ray_d, uint32_t* occluded, uint32_t n);
// tiles
// on-device refit (refit.hip): per-mesh object->world rows, and what one refit pass touches
struct RefitMesh {
   float o2w[12];
   uint32_t identity;
   uint32_t pad[3];
};
struct RefitArgs {
   const float* obj_corners;   // 9 floats per triangle packet, object space, leaf order
   const RefitMesh* meshes;
   float4* tris;               // TriPacket array, rewritten
   float* world_corners;       // scratch, 9 floats per packet
   uint4* nodes;               // Node4C array: origin, step exponents and planes rewritten; counts and bases kept
   float* node_box;            // scratch, 6 floats per node (unpadded)
   const uint32_t* level_start;  // HOST array: BFS level l = nodes [level_start[l], level_start[l+1])
   uint32_t num_levels;
   uint32_t num_tris;
};
void launch_refit(const LaunchCfg&, const RefitArgs&);

// on-device LBVH build (lbvh.hip): topology + packets in Morton order; boxes come from launch_refit afterwards
struct LbvhArgs {
   const float* src_corners;     // 9 floats per triangle, object space, mesh order
   const uint32_t* src_keys;     // mesh << 22 | primitive, mesh order
   const float4* src_shade;      // ShadePacket (4 float4) per triangle, mesh order
   const RefitMesh* meshes;
   float bounds_lo[3], bounds_hi[3];  // world-space box containing every centroid (Morton normalisation)
   uint32_t num_tris;
   uint32_t kind;                // binary tree under the 4-wide collapse: 1 = PLOC (default), 2 = radix tree (Karras)
   uint32_t ploc_radius;         // PLOC: places searched to either side for the nearest neighbour (1..64)
   uint32_t sah_top;             // PLOC: the rounds stop at this many clusters and a binned-SAH tree over them (host, bvh_build.cpp) is the top; <= 1: PLOC to the root
   uint4* nodes;                 // out: Node4C array (child counts and bases valid, boxes to be refitted)
   uint32_t node_capacity;       // nodes the array can hold
   float4* tris;                 // out: TriPacket array in leaf order (keys; refit writes the geometry)
   float4* shade;                // out: ShadePacket array in leaf order
   float* obj_corners;           // out: object-space corners in leaf order (the refit input)
};
hipError_t lbvh_build(const LbvhArgs&, hipStream_t, std::vector<uint32_t>& level_start, uint32_t* out_nodes);

void launch_pack_tiles(const LaunchCfg&, const float4* acc, float4* out, uint32_t W, uint32_t H, uint32_t rank, uint32_t world, uint32_t tile);
void launch_unpack_tiles(const LaunchCfg&, float4* acc, const float4* in, uint32_t W, uint32_t H, uint32_t rank, uint32_t world, uint32_t tile);
void launch_compose_tiles(const LaunchCfg&, const Images&, const float4* all, uint64_t stride, uint32_t W, uint32_t H, uint32_t rank, uint32_t world, uint32_t tile,
                          uint32_t total_samples, uint32_t limit);
uint32_t query_trace_occupancy();

}  // namespace uh
