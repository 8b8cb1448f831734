// context.hip — the C ABI of include/utopian_hip.h: context lifetime, scene upload, acceleration
// structure build, the per-frame pass chain and read-backs. Host-side counterpart of
// utopian::Renderer (utopian/src/renderer.rs), utopian::Raytracing (utopian/src/raytracing.rs) and
// build_path_tracing_render_graph (utopian/src/renderers/mod.rs:189-375) for this path only.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library opens librccl at run time (uh_rccl_attach)
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "bvh.h"
#include "device_types.h"
#include "utopian_hip.h"

using namespace uh;

namespace {

std::string g_create_error;

struct HostMesh {
   std::vector<UhVertex> vertices;
   std::vector<uint32_t> indices;
   UhGpuMaterial material;
   float o2w[12];
   float w2o[9];
};

struct EventPair {
   hipEvent_t start, stop;
   int kind;  // 0 trace_closest, 1 trace_shadow, 2 shade
};

template <typename T>
struct DevBuf {
   T* p = nullptr;
   void* base = nullptr;
   size_t n = 0;
   // stagger_bytes shifts the array inside its allocation: the per-pixel SoA arrays are all the same
   // size, and kernels touch the same index of several of them at once; without a stagger those
   // accesses are an exact multiple of the array size apart
   hipError_t alloc(size_t count, size_t stagger_bytes = 0) {
      release();
      n = count;
      if (count == 0) return hipSuccess;
      hipError_t e = hipMalloc(&base, count * sizeof(T) + stagger_bytes);
      if (e == hipSuccess) p = reinterpret_cast<T*>(static_cast<char*>(base) + stagger_bytes);
      return e;
   }
   void release() {
      if (base) (void)hipFree(base);
      p = nullptr;
      base = nullptr;
      n = 0;
   }
};

}  // namespace

// One frame in flight: its own stream pair, hazard events, path state and queue control block.
// Frames of the path-tracing pass are independent except for the order of the accumulation
// read-modify-write (reference.rgen:131-143), so up to `frames_in_flight` of them overlap on the
// GPU: one frame's memory-bound shading and kernel tails are filled by another frame's traversal,
// and a rank that owns only 1/N of the pixels still keeps the chip busy.
constexpr uint32_t kMaxSlots = 8;
struct Slot {
   hipStream_t stream = nullptr;
   // second stream: shade_miss (pure VALU, touches only paths that left the scene) and the shadow
   // traversals overlap the main stream's shade_hit / next closest-hit traversal
   hipStream_t side = nullptr;
   hipEvent_t ev_traced = nullptr, ev_missed = nullptr, ev_shaded = nullptr, ev_shadowed = nullptr, ev_side_done = nullptr;
   hipEvent_t ev_acc = nullptr;  // recorded after the frame's accumulate / store tail
   hipEvent_t frame_start = nullptr, frame_stop = nullptr;
   DevBuf<float4> rec, radf, pixcol;  // rec: two sets of four path-state planes + the hit plane (device_types.h PathState)
   DevBuf<uint32_t> queues[5];
   DevBuf<Control> control;
   PathState ps{};
   bool ready = false;
   size_t capacity = 0;  // path ids this slot can hold (pixels x frames per batch)

   hipError_t create(size_t n) {
      capacity = n;
      uint32_t shard_cap = 0;
      {  // exact: the largest number of 64-path runs one shard receives (shard_of_run)
         const uint32_t runs = (uint32_t)((n + 63) / 64);
         uint32_t per_shard[kShards] = {0};
         for (uint32_t r = 0; r < runs; r++) per_shard[shard_of_run(r)]++;
         for (uint32_t s = 0; s < kShards; s++) shard_cap = per_shard[s] * 64 > shard_cap ? per_shard[s] * 64 : shard_cap;
      }
      hipError_t e;
#define SLOT_TRY(expr)                 \
   if ((e = (expr)) != hipSuccess) return e
      SLOT_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      SLOT_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
      for (hipEvent_t* ev : {&ev_traced, &ev_missed, &ev_shaded, &ev_shadowed, &ev_side_done, &ev_acc}) SLOT_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
      SLOT_TRY(hipEventCreate(&frame_start));
      SLOT_TRY(hipEventCreate(&frame_stop));
      const size_t stagger = 4352;  // 4 KiB + 256 B per array slot
      // planes staggered like the arrays: the same index of two planes must not alias. The hit plane is indexed by queue position
      // (shard segment + position in the shard's queue), which runs to kShards * shard_cap >= n
      const size_t cap_q = (size_t)shard_cap * kShards;
      const size_t plane = (n > cap_q ? n : cap_q) + stagger / sizeof(float4);
      SLOT_TRY(rec.alloc(plane * (2 * kRecQuads + 1), 0 * stagger));
      SLOT_TRY(radf.alloc(n, 1 * stagger));
      SLOT_TRY(pixcol.alloc(n, 2 * stagger));
      // sharded queues: capacity per shard = the pixels (64-pixel runs) a shard can own
      // (the miss queue holds (position, id) pairs: twice the words)
      for (int qi = 0; qi < 5; qi++) SLOT_TRY(queues[qi].alloc((size_t)shard_cap * kShards * (qi == 4 ? 2 : 1)));
      SLOT_TRY(control.alloc(1));
      SLOT_TRY(hipMemsetAsync(control.p, 0, sizeof(Control), stream));
      SLOT_TRY(hipStreamSynchronize(stream));
#undef SLOT_TRY
      ps.set[0] = PathRecs{rec.p, plane};
      ps.set[1] = PathRecs{rec.p + plane * kRecQuads, plane};
      ps.hit = rec.p + plane * 2 * kRecQuads;
      ps.radf = radf.p;
      ps.pixcol = pixcol.p;
      for (int i = 0; i < 5; i++) ps.queue[i] = queues[i].p;
      ps.shard_cap = shard_cap;
      ready = true;
      return hipSuccess;
   }
   void destroy() {
      if (stream) (void)hipStreamSynchronize(stream);
      if (side) (void)hipStreamSynchronize(side);
      rec.release();
      radf.release();
      pixcol.release();
      for (auto& q : queues) q.release();
      control.release();
      for (hipEvent_t* ev : {&ev_traced, &ev_missed, &ev_shaded, &ev_shadowed, &ev_side_done, &ev_acc, &frame_start, &frame_stop}) {
         if (*ev) (void)hipEventDestroy(*ev);
         *ev = nullptr;  // (a create() that fails half-way must not leave handles for the next destroy())
      }
      if (side) (void)hipStreamDestroy(side);
      if (stream) (void)hipStreamDestroy(stream);
      stream = side = nullptr;
      ready = false;
   }
};

// frames of the reservoir passes one uh_render_frames wavefront carries at most, and the ring of spatial buffers that
// lets the next batch's chains run beside the current wavefront (two batches + the history slot)
constexpr uint32_t kRestirBatch = 16;
constexpr int kSpatialRing = 2 * (int)kRestirBatch + 1;

struct uh_ctx {
   int device = 0;
   Slot slots[kMaxSlots];
   uint32_t frames_in_flight = 4;     // slots used round-robin by path-tracing frames (swept: profiles/README.md)
   uint32_t batch_frames = 0;         // frames one uh_render_frames launch chain carries (option "batch_frames"); 0 = auto
   uint32_t next_slot = 0;
   uint32_t shard_cap = 0;
   hipEvent_t last_acc = nullptr;     // ev_acc of the most recent frame (accumulation order)
   // G-buffer cast + reservoir passes run in call order on their own stream, beside path-tracing frames in flight.
   // spatial_reuse_reservoirs is a RING of kSpatialRing buffers: the path tracer of frame f reads slot `spatial_cur` while
   // frame f+1's passes already run - its temporal pass reads the same slot and its spatial pass writes the next one, after
   // the last path-tracing wavefront that read THAT one has finished (spatial_reader[]). A batch of B static-camera frames
   // runs its B reservoir chains back to back (slots cur+1 .. cur+B) and then ONE path-tracing wavefront in which the
   // paths of frame f sample from slot cur+1+f (FrameParams::spatial_of): the ring holds two batches and the history.
   hipStream_t restir_stream = nullptr;
   hipEvent_t ev_restir = nullptr, rs_start = nullptr, rs_stop = nullptr;
   bool restir_recorded = false;
   int spatial_cur = 0;
   hipEvent_t spatial_reader[kSpatialRing] = {};
   // the reservoir passes by bands of rows over the ranks of a job (uh_set_restir_partition)
   uint32_t rp_rank = 0, rp_world = 1, rp_band_rows = 0;
   size_t res_stride = 0;  // reservoirs per spatial_reuse buffer: the frame, padded to rp_world equal bands
   UhRestirExchangeFn rp_exchange = nullptr;
   void* rp_user = nullptr;
   hipEvent_t ev_band[kSpatialRing] = {};  // "this context's band of ring slot k is written" (in-process groups pull on it)
   void* rccl = nullptr;                   // RcclLink (uh_rccl_attach)
   hipEvent_t t_start = nullptr, t_stop = nullptr;  // bracket of the last uh_render_frame call (last_frame_ms)
   Slot* last_slot = nullptr;
   hipStream_t& stream = slots[0].stream;  // slot 0 also serves every non-frame operation
   PathState& ps = slots[0].ps;
   DevBuf<Control>& control = slots[0].control;
   bool overlap_miss = true, overlap_shadow = true;
   uint32_t W = 0, H = 0;
   uint32_t num_cus = 256;
   // persistent grids of the traversal kernels, blocks per CU (the refill kernels' LDS - stacks + ray pool - admits 6 / 5). Round 4,
   // closest / shadow = 6/5, 5/5, 5/4, 4/4, 4/3, 3/4: a 16-frame wavefront 1.774 / 1.777 / 1.764 / 1.780 / 1.773 / 1.799 ms per frame,
   // one frame per call with a wait after it 2.95 / 2.88 / 2.88 / 2.84 / 2.85 / 2.89 ms: fewer waves finish a small launch's tail sooner
   uint32_t closest_blocks_per_cu = 5, shadow_blocks_per_cu = 5;  // (config 2, whose light shadow rays are a third of the frame: 6/5, 5/5, 5/4, 6/4 = 8,230 / 8,266 / 7,997 / 7,950 Mrays/s)
   uint32_t cam_walk_whole = 512;     // option "camera_grid_walk_whole" (sun_grid.h SunGridDev::walk_whole)
   // one frame per call: bounces 1 .. of a lone frame inside one persistent kernel (k_path_fused) instead of four launches per bounce
   bool fused_bounces = true;  // option "fused_bounces"
   bool fused_always = false;  // fused_bounces = -1: also with frames in flight and for frames of any size (tests)
   static constexpr uint32_t kFusedMaxPaths = 4u << 20;
   uint32_t fused_blocks_per_cu = 4;
   static constexpr uint32_t kSingleFrameBlocksPerCu = 4;  // the cap on both for a wavefront of one frame (fewer persistent waves reach the end of a small launch's tail sooner: round 4's sweep)
   std::string err;

   // host scene
   std::vector<HostMesh> meshes;
   std::vector<UhGpuLight> lights;
   struct HostTex {
      uint32_t w, h;
      uchar4* dev;
      uint32_t tiles_x;  // 0 = row-major
   };
   std::vector<HostTex> textures;
   bool built = false;

   // device scene
   DevBuf<float4> d_nodes, d_tris, d_shade, d_lights;
   DevBuf<MeshShade> d_meshes;
   // on-device refit (refit.hip), allocated by the first uh_refit_acceleration
   DevBuf<float> d_obj_corners, d_world_corners, d_node_box;
   DevBuf<RefitMesh> d_refit_meshes;
   std::vector<uint32_t> packet_keys;  // key of triangle packet i (leaf order)
   std::vector<uint32_t> level_start;  // BFS levels of the node array
   bool topology_valid = false;        // the device tree matches the mesh list (transforms may differ)
   // on-device build (lbvh.hip, option "device_build"): per-triangle sources in mesh order, kept on the device
   // until a mesh is added, so that a rebuild after moved instances or changed parameters uploads nothing
   bool device_build = false, src_valid = false;
   uint32_t device_build_kind = 1;  // 1 = PLOC, 2 = radix tree (lbvh.hip)
   // PLOC rounds stop at this many clusters; a host SAH tree over them is the top (option "ploc_sah_top", 0 = PLOC to the root).
   // Config-1 scene: 0 / 1,024 / 8,192 / 131,072 clusters = 21.7 / 20.3 / 20.1 / 19.0 nodes per ray, rebuild 7.7 / 6.0 / 9.4 / 66 ms (host tree: 18.8)
   uint32_t ploc_sah_top = 1024;
   static constexpr uint32_t kPlocRadius = 8;  // swept 4..64 in round 3: tree quality flat (21.7-23.0 nodes/ray), build time grows with it (profiles/README.md)
   DevBuf<float> d_src_corners;
   DevBuf<uint32_t> d_src_keys;
   DevBuf<float4> d_src_shade;
   float refit_ms = 0.0f;
   DevBuf<TexInfo> d_tex;
   DevBuf<float> d_lut;
   SceneDev scene{};

   // frame-persistent per-pixel images (graph resources of renderers/mod.rs:199-244)
   DevBuf<float4> accumulation, gbuffer;
   DevBuf<uchar4> output;
   DevBuf<UhReservoir> reservoirs[3], spatial_ring;  // ring slot 0 = reservoirs[2], slots 1.. = spatial_ring (allocated by the first reservoir pass)
   DevBuf<float4> gb_ray_o, gb_ray_d, gb_hit;  // scratch of the G-buffer cast (allocated by the first G-buffer pass)
   DevBuf<DeviceStats> dstats;
   Images im{};

   // options / stats
   bool count_visits = false, time_kernels = false, full_frame_restir = false;
   bool iso_reference = true;  // option "iso_reference_triangulation": uh_add_isosurface_mesh emits the reference's triangles (isosurface.hip)
   bool furnace = false;  // option "furnace": reference.rmiss compiled with FURNACE_TEST (a miss returns white whatever view.sky_enabled says)
   uint64_t frames = 0;
   float build_ms = 0.0f, last_frame_ms = 0.0f;
   float ms_by_kind[5] = {0, 0, 0, 0, 0};  // trace_closest, sun shadow rays (grid + tree), shade, camera grid (bounce 0 through the grid + its leftovers), light shadow rays
   uint32_t trace_closest_launches = 0, trace_light_launches = 0;
   bool frame_timed = false;
   std::vector<EventPair> pending, free_events;
   uint32_t bvh_nodes = 0, bvh_tris = 0;

   // sun shadow rays through a per-direction grid instead of the tree (sun_grid.h; option "sun_grid"). The grid belongs to one
   // (geometry, sun direction) pair: it is built on the first frame that traces sun rays and again when the direction or the
   // geometry has changed and then stayed put for two consecutive frames - a sun or an instance that moves every frame keeps
   // the tree walk.
   bool sun_grid_enabled = true;
   bool sun_valid = false;          // d_sun_* hold a usable grid for (sun_geom, sun_dir_built)
   bool sun_attempted = false;      // a build for (sun_geom, sun_dir_built) was tried (it may have been refused: sun_why)
   bool sun_have_pending = false;
   uint64_t geom_version = 1, sun_geom = 0, sun_geom_pending = 0;
   float sun_dir_built[3] = {0, 0, 0}, sun_dir_pending[3] = {0, 0, 0};
   DevBuf<uint32_t> d_sun_cells;
   DevBuf<SunGridEntry> d_sun_entries;
   DevBuf<float4> d_sun_recs;       // the entries with their packets inline (SunGridDev::recs; option "sun_grid_inline")
   DevBuf<float> d_sun_coarse;      // the coarse cover (SunGridDev::coarse; option "sun_grid_coarse")
   uint32_t sun_coarse_shift = 2;   // blocks of 4 x 4 cells; 0: no coarse cover
   // the lists a second time as 64-byte records that carry their packet (SunGridDev::recs): by default only while they stay within
   // four times the packet array (a grid of 96 entries per triangle repeats every packet 96 times: 1.4 GB for the 17 MB of the
   // config-1 scene); option "sun_grid_inline_max_mb" raises the budget (0: never)
   int64_t sun_inline_max_mb = -1;  // -1: auto = 4 x the packet array
   SunGridDev sun_dev{};
   SunGridLimits sun_limits;
   std::string sun_why;
   float sun_build_ms = 0.0f, sun_mean_list = 0.0f;
   float sun_fallback_area = 1.0f;  // share of the scene's surface whose cell hands its sun rays to the tree (the builders' figure)
   uint32_t sun_cells = 0, sun_entries = 0, sun_max_list = 0;
   bool sun_this_frame = false;     // set by render_batch for the frame being enqueued
   bool primary_implicit = true;    // option "primary_implicit" (FrameParams::primary_implicit)
   bool sun_device_build = true;    // option "sun_grid_build": 1 = on the device (sun_grid_build.hip: a few ms), 0 = the host builder (sun_grid.cpp)

   // the primary rays through a per-camera grid instead of the tree (sun_grid.h "camera grid"; option "camera_grid"). The grid belongs
   // to one (geometry, inverse_view, inverse_projection, frame size): it is built - on the device, a few milliseconds - when the
   // same camera has been asked for in two consecutive frame calls, or at once by a call that carries several frames of it; a
   // camera that moves every frame keeps the tree walk.
   bool cam_grid_enabled = true;
   bool cam_valid = false, cam_attempted = false, cam_have_pending = false;
   uint64_t cam_geom = 0, cam_geom_pending = 0;
   float cam_mats[32] = {0}, cam_mats_pending[32] = {0};  // inverse_view, inverse_projection of the grid / of the last request
   DevBuf<uint32_t> d_cam_cells;
   DevBuf<SunGridEntry> d_cam_entries;
   SunGridDev cam_dev{};
   SunGridLimits cam_limits;
   std::string cam_why;
   float cam_build_ms = 0.0f, cam_mean_list = 0.0f;
   uint32_t cam_cells = 0, cam_entries = 0, cam_max_list = 0, cam_max_list_interior = 0;
   bool cam_this_frame = false;

   // tile partition
   uint32_t tp_rank = 0, tp_world = 1, tp_tile = 64;
   DevBuf<uint32_t> owned_pixels;  // ascending pixel ids this rank owns (empty = the whole frame)
   uint32_t n_owned = 0;
   // composition of a partitioned frame without a host wait (uh_rccl_gather_tiles; the in-process group's uh_mgpu_compose): this
   // rank's packed tiles, on the root every rank's, and the event behind the last pack / composition - the next frame's accumulate
   // tail waits for it like for a frame's (last_acc)
   DevBuf<float4> tile_send, tile_recv;
   hipEvent_t ev_compose = nullptr;
};

namespace {

int fail(uh_ctx* c, int code, const std::string& msg) {
   if (c)
      c->err = msg;
   else
      g_create_error = msg;
   return code;
}

#define HIP_TRY(ctx, expr)                                                                                   \
   do {                                                                                                      \
      hipError_t e_ = (expr);                                                                                \
      if (e_ != hipSuccess) return fail(ctx, e_ == hipErrorOutOfMemory ? UH_ERR_OUT_OF_MEMORY : UH_ERR_HIP, \
                                        std::string(#expr) + ": " + hipGetErrorString(e_));                 \
   } while (0)

bool is_identity3x4(const float* m) {
   static const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
   return std::memcmp(m, I, sizeof(I)) == 0;
}

// inverse of the upper 3x3 of a row-major 3x4 by cofactors (arithmetic contract: cofactor * (1/det))
void invert3x3(const float* m, float* inv) {
   float a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
   float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
   float det = (a * A + b * B) + c * C;
   float id = 1.0f / det;
   inv[0] = A * id;
   inv[1] = -(b * i - c * h) * id;
   inv[2] = (b * f - c * e) * id;
   inv[3] = B * id;
   inv[4] = (a * i - c * g) * id;
   inv[5] = -(a * f - c * d) * id;
   inv[6] = C * id;
   inv[7] = -(a * h - b * g) * id;
   inv[8] = (a * e - b * d) * id;
}

void set_transform(HostMesh& m, const float* w) {
   std::memcpy(m.o2w, w, sizeof(m.o2w));
   if (is_identity3x4(w)) {
      static const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      std::memcpy(m.w2o, I, sizeof(I));
   } else {
      invert3x3(w, m.w2o);
   }
}

LaunchCfg cfg(uh_ctx* c) {
   return LaunchCfg{c->stream, c->num_cus, c->closest_blocks_per_cu, c->shadow_blocks_per_cu, c->count_visits, c->fused_blocks_per_cu};
}

void begin_timed(uh_ctx* c, int kind, hipStream_t stream = nullptr) {
   if (!stream) stream = c->stream;
   if (!c->time_kernels) return;
   EventPair ep;
   if (!c->free_events.empty()) {
      ep = c->free_events.back();
      c->free_events.pop_back();
   } else {
      (void)hipEventCreate(&ep.start);
      (void)hipEventCreate(&ep.stop);
   }
   ep.kind = kind;
   (void)hipEventRecord(ep.start, stream);
   c->pending.push_back(ep);
}
void end_timed(uh_ctx* c, hipStream_t stream = nullptr) {
   if (!c->time_kernels) return;
   (void)hipEventRecord(c->pending.back().stop, stream ? stream : c->stream);
}
void drain_timed(uh_ctx* c) {
   for (EventPair& ep : c->pending) {
      float ms = 0.0f;
      if (hipEventSynchronize(ep.stop) == hipSuccess && hipEventElapsedTime(&ms, ep.start, ep.stop) == hipSuccess) {
         c->ms_by_kind[ep.kind] += ms;
         if (ep.kind == 0) c->trace_closest_launches++;
         if (ep.kind == 4) c->trace_light_launches++;
      }
      c->free_events.push_back(ep);
   }
   c->pending.clear();
}

}  // namespace

namespace {
std::string hip_version_string(int v) { return std::to_string(v / 10000000) + "." + std::to_string(v / 100000 % 100) + "." + std::to_string(v % 100000); }
// "" when the runtime's major.minor is the one this library was built with
std::string hip_skew_warning() {
   int built = 0, rt = 0;
   if (uh_hip_versions(&built, &rt) != UH_OK || rt / 100000 == built / 100000) return "";
   return "warning: libutopian_hip.so was built with HIP " + hip_version_string(built) + " but this process runs it on HIP runtime " + hip_version_string(rt) +
          " (another libamdhip64.so.7 was loaded first - e.g. the copy bundled with a PyTorch wheel); link or preload the ROCm release the library was built with";
}
}  // namespace

extern "C" {

// The HIP release this library was compiled by (the hipcc whose headers and code objects are in it) and the one it RUNS on (whichever
// libamdhip64.so.7 the process bound first). They can differ: the soname covers every 7.x, and a host that has another copy loaded -
// the PyTorch wheel bundles ROCm 7.0's - hands that copy to this library too. Both go on record, and a differing major.minor is
// said out loud (round 4's heap corruption under context churn showed with a 7.2-built library on the wheel's 7.0 runtime only).
int uh_hip_versions(int* built, int* runtime) {
   if (built) *built = HIP_VERSION;
   int rt = 0;
   const hipError_t e = hipRuntimeGetVersion(&rt);
   if (runtime) *runtime = e == hipSuccess ? rt : 0;
   return e == hipSuccess ? UH_OK : UH_ERR_HIP;
}

const char* uh_version(void) {
   static const std::string s = [] {
      int built = 0, rt = 0;
      (void)uh_hip_versions(&built, &rt);
      return "utopian-hip 0.5 (gfx950; built with HIP " + hip_version_string(built) + "; HIP runtime " + (rt ? hip_version_string(rt) : std::string("unknown")) + ")";
   }();
   return s.c_str();
}

const char* uh_last_error(uh_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int uh_create(int device_ordinal, uint32_t width, uint32_t height, uh_ctx** out) {
   if (!out || width == 0 || height == 0 || (uint64_t)width * height > (1ull << 26)) return fail(nullptr, UH_ERR_INVALID_ARGUMENT, "bad size");
   int ndev = 0;
   hipError_t e = hipGetDeviceCount(&ndev);
   if (e != hipSuccess || ndev == 0)
      return fail(nullptr, UH_ERR_NO_DEVICE, std::string("no HIP device visible (") + hipGetErrorString(e) + "); this library has no CPU fallback");
   if (device_ordinal < 0 || device_ordinal >= ndev) return fail(nullptr, UH_ERR_INVALID_ARGUMENT, "device ordinal out of range");
   uh_ctx* c = new uh_ctx();
   c->device = device_ordinal;
   c->W = width;
   c->H = height;
   auto bail = [&](int code) {
      g_create_error = c->err;
      uh_destroy(c);
      return code;
   };
#define CREATE_TRY(expr)                                                                      \
   do {                                                                                       \
      hipError_t e_ = (expr);                                                                 \
      if (e_ != hipSuccess) {                                                                 \
         c->err = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
         return bail(e_ == hipErrorOutOfMemory ? UH_ERR_OUT_OF_MEMORY : UH_ERR_HIP);          \
      }                                                                                       \
   } while (0)
   CREATE_TRY(hipSetDevice(device_ordinal));
   hipDeviceProp_t prop;
   CREATE_TRY(hipGetDeviceProperties(&prop, device_ordinal));
   c->num_cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256;
   {
      uint32_t occ = query_trace_occupancy();  // blocks/CU the traversal kernels can keep resident
      if (c->closest_blocks_per_cu > occ) c->closest_blocks_per_cu = occ;
      if (c->shadow_blocks_per_cu > occ) c->shadow_blocks_per_cu = occ;
   }
   const size_t n = (size_t)width * height;
   CREATE_TRY(c->slots[0].create(n));
   CREATE_TRY(c->accumulation.alloc(n));
   CREATE_TRY(c->gbuffer.alloc(n));
   CREATE_TRY(c->output.alloc(n));
   for (auto& r : c->reservoirs) CREATE_TRY(r.alloc(n));
   CREATE_TRY(c->dstats.alloc(1));
   CREATE_TRY(hipMemsetAsync(c->accumulation.p, 0, n * sizeof(float4), c->stream));
   CREATE_TRY(hipMemsetAsync(c->output.p, 0, n * sizeof(uchar4), c->stream));
   CREATE_TRY(hipMemsetAsync(c->gbuffer.p, 0, n * sizeof(float4), c->stream));
   for (auto& r : c->reservoirs) CREATE_TRY(hipMemsetAsync(r.p, 0, n * sizeof(UhReservoir), c->stream));
   CREATE_TRY(hipMemsetAsync(c->dstats.p, 0, sizeof(DeviceStats), c->stream));
   // c / 255.0f table (exact host division; replaces 12 IEEE divides per bilinear fetch)
   float lut[256];
   for (int i = 0; i < 256; i++) lut[i] = (float)i / 255.0f;
   CREATE_TRY(c->d_lut.alloc(256));
   CREATE_TRY(hipMemcpy(c->d_lut.p, lut, sizeof(lut), hipMemcpyHostToDevice));
   CREATE_TRY(hipStreamSynchronize(c->stream));
#undef CREATE_TRY
   c->im.accumulation = c->accumulation.p;
   c->im.output = c->output.p;
   c->im.gbuffer_pos = c->gbuffer.p;
   for (int i = 0; i < 3; i++) c->im.reservoirs[i] = c->reservoirs[i].p;
   c->im.prev_spatial = c->reservoirs[2].p;
   c->res_stride = n;
   c->rp_band_rows = height;
   c->cam_limits.max_walk = 48;          // a pixel listing more packets than this hands its ray to the tree walk
   c->cam_limits.max_mean_list = 24.0;   // entries per occupied pixel beyond which the grid is refused (the tree walk costs about 20 records per ray)
   c->cam_limits.max_fallback_area = 2.0;
   // a runtime of another release than the one the library was built with: uh_last_error(NULL) says so after a SUCCESSFUL create too,
   // and stderr once per process
   g_create_error = hip_skew_warning();
   if (!g_create_error.empty()) {
      static bool said = false;
      if (!said) std::fprintf(stderr, "[libutopian_hip] %s\n", g_create_error.c_str());
      said = true;
   }
   *out = c;
   return UH_OK;
}

void uh_destroy(uh_ctx* c) {
   if (!c) return;
   (void)hipSetDevice(c->device);
   // the communicator's collectives were enqueued on the reservoir stream: it goes (and every stream is drained) while that
   // stream still exists
   uh_rccl_detach(c);
   // every stream idle before anything goes; the streams themselves go LAST, after every event that was recorded on or waited for by
   // one of them (slot streams wait for the reservoir stream's event and the other way round)
   (void)hipDeviceSynchronize();
   c->spatial_ring.release();
   c->gb_ray_o.release();
   c->gb_ray_d.release();
   c->gb_hit.release();
   for (auto& s : c->slots) {
      if (s.stream) (void)hipStreamSynchronize(s.stream);
      if (s.side) (void)hipStreamSynchronize(s.side);
   }
   for (auto& t : c->textures)
      if (t.dev) (void)hipFree(t.dev);
   for (auto& ep : c->pending) {
      (void)hipEventDestroy(ep.start);
      (void)hipEventDestroy(ep.stop);
   }
   for (auto& ep : c->free_events) {
      (void)hipEventDestroy(ep.start);
      (void)hipEventDestroy(ep.stop);
   }
   c->d_nodes.release();
   c->d_tris.release();
   c->d_shade.release();
   c->d_lights.release();
   c->d_meshes.release();
   c->d_obj_corners.release();
   c->d_world_corners.release();
   c->d_node_box.release();
   c->d_refit_meshes.release();
   c->d_src_corners.release();
   c->d_src_keys.release();
   c->d_src_shade.release();
   c->d_tex.release();
   c->d_lut.release();
   c->d_sun_cells.release();
   c->d_sun_entries.release();
   c->d_sun_recs.release();
   c->d_sun_coarse.release();
   c->d_cam_cells.release();
   c->d_cam_entries.release();
   c->accumulation.release();
   c->gbuffer.release();
   c->output.release();
   for (auto& r : c->reservoirs) r.release();
   c->dstats.release();
   for (auto& s : c->slots) s.destroy();
   for (hipEvent_t ev : c->ev_band)
      if (ev) (void)hipEventDestroy(ev);
   if (c->restir_stream) {
      for (hipEvent_t ev : {c->ev_restir, c->rs_start, c->rs_stop})
         if (ev) (void)hipEventDestroy(ev);
      (void)hipStreamDestroy(c->restir_stream);
      c->restir_stream = nullptr;
   }
   c->owned_pixels.release();
   c->tile_send.release();
   c->tile_recv.release();
   if (c->ev_compose) (void)hipEventDestroy(c->ev_compose);
   delete c;
}

int uh_add_texture_rgba8(uh_ctx* c, const uint8_t* pixels, uint32_t w, uint32_t h, uint32_t* out_index) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!pixels || !w || !h) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_add_texture_rgba8: null or empty texture");
   HIP_TRY(c, hipSetDevice(c->device));
   uchar4* dev = nullptr;
   HIP_TRY(c, hipMalloc((void**)&dev, (size_t)w * h * 4));
   // 8x8-texel tiles when the size allows (device_types.h TexInfo)
   const uint32_t tiles_x = (w % 8 == 0 && h % 8 == 0) ? w / 8 : 0;
   std::vector<uint8_t> tiled;
   const uint8_t* src = pixels;
   if (tiles_x) {
      tiled.resize((size_t)w * h * 4);
      for (uint32_t y = 0; y < h; y++)
         for (uint32_t x = 0; x < w; x++) {
            size_t at = ((size_t)((y >> 3) * tiles_x + (x >> 3)) << 6) + ((y & 7) << 3) + (x & 7);
            std::memcpy(&tiled[at * 4], &pixels[((size_t)y * w + x) * 4], 4);
         }
      src = tiled.data();
   }
   hipError_t e = hipMemcpy(dev, src, (size_t)w * h * 4, hipMemcpyHostToDevice);
   if (e != hipSuccess) {
      (void)hipFree(dev);
      return fail(c, UH_ERR_HIP, std::string("texture upload: ") + hipGetErrorString(e));
   }
   c->textures.push_back(uh_ctx::HostTex{w, h, dev, tiles_x});
   c->built = false;
   if (out_index) *out_index = (uint32_t)c->textures.size() - 1;
   return UH_OK;
}

int uh_add_mesh(uh_ctx* c, const UhVertex* vertices, uint32_t num_vertices, const uint32_t* indices, uint32_t num_indices,
                const UhGpuMaterial* material, const float world3x4[12], uint32_t* out_mesh_index) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!vertices || !indices || !material || !world3x4 || num_indices % 3 != 0)
      return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_add_mesh: null argument or index count not a multiple of 3");
   if (c->meshes.size() >= UH_MAX_GPU_MESHES) return fail(c, UH_ERR_CAPACITY, "uh_add_mesh: more than 1024 meshes (MAX_NUM_GPU_MESHES)");
   if (num_indices / 3 > (1u << kPrimBits)) return fail(c, UH_ERR_CAPACITY, "uh_add_mesh: more than 4 Mi triangles in one mesh");
   for (uint32_t i = 0; i < num_indices; i++)
      if (indices[i] >= num_vertices) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_add_mesh: index out of range");
   for (uint32_t i = 0; i < num_vertices; i++)
      if (!std::isfinite(vertices[i].pos[0]) || !std::isfinite(vertices[i].pos[1]) || !std::isfinite(vertices[i].pos[2]))
         return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_add_mesh: vertex position is not finite");
   for (int i = 0; i < 12; i++)
      if (!std::isfinite(world3x4[i])) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_add_mesh: transform is not finite");
   HostMesh m;
   m.vertices.assign(vertices, vertices + num_vertices);
   m.indices.assign(indices, indices + num_indices);
   m.material = *material;
   set_transform(m, world3x4);
   c->meshes.push_back(std::move(m));
   c->built = false;
   c->topology_valid = false;
   c->src_valid = false;
   if (out_mesh_index) *out_mesh_index = (uint32_t)c->meshes.size() - 1;
   return UH_OK;
}

int uh_add_light(uh_ctx* c, const UhGpuLight* light, uint32_t* out_index) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!light) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_add_light: null light");
   if (c->lights.size() >= UH_MAX_GPU_LIGHTS) return fail(c, UH_ERR_CAPACITY, "uh_add_light: more than 1024 lights (MAX_NUM_GPU_LIGHTS)");
   c->lights.push_back(*light);
   c->built = false;
   c->topology_valid = false;
   if (out_index) *out_index = (uint32_t)c->lights.size() - 1;
   return UH_OK;
}

int uh_get_num_lights(uh_ctx* c, uint32_t* out) {
   if (!c || !out) return UH_ERR_INVALID_ARGUMENT;
   *out = (uint32_t)c->lights.size();
   return UH_OK;
}

int uh_set_instance_transform(uh_ctx* c, uint32_t mesh_index, const float world3x4[12]) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (mesh_index >= c->meshes.size() || !world3x4) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_set_instance_transform: bad mesh index");
   for (int i = 0; i < 12; i++)
      if (!std::isfinite(world3x4[i])) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_set_instance_transform: transform is not finite");
   set_transform(c->meshes[mesh_index], world3x4);
   c->built = false;
   return UH_OK;
}

static int sync_all(uh_ctx* c);

// per-mesh shading records, light table, texture descriptors: everything of the scene except the geometry
static int upload_scene_tables(uh_ctx* c) {
   std::vector<MeshShade> ms(c->meshes.size());
   for (size_t i = 0; i < c->meshes.size(); i++) {
      const HostMesh& m = c->meshes[i];
      std::memcpy(ms[i].w2o, m.w2o, sizeof(m.w2o));
      ms[i].diffuse_map = m.material.diffuse_map;
      for (int a = 0; a < 3; a++) ms[i].base_color[a] = m.material.base_color_factor[a];
      ms[i].type = m.material.raytrace_properties[0];
      ms[i].property = m.material.raytrace_properties[1];
      ms[i].pad = 0;
      ms[i].metallic = m.material.metallic_factor;
      ms[i].roughness = m.material.roughness_factor;
      ms[i].pad2[0] = ms[i].pad2[1] = 0.0f;
   }
   std::vector<float4> lights(2 * c->lights.size());
   for (size_t i = 0; i < c->lights.size(); i++) {
      const UhGpuLight& l = c->lights[i];
      lights[2 * i] = make_float4(l.position[0], l.position[1], l.position[2], 0.0f);
      lights[2 * i + 1] = make_float4(l.intensity[0], l.intensity[1], l.intensity[2], 0.0f);
   }
   std::vector<TexInfo> tex(c->textures.size());
   for (size_t i = 0; i < tex.size(); i++) tex[i] = TexInfo{c->textures[i].dev, c->textures[i].w, c->textures[i].h, c->textures[i].tiles_x, 0};

   HIP_TRY(c, c->d_meshes.alloc(ms.size()));
   HIP_TRY(c, c->d_lights.alloc(lights.size()));
   HIP_TRY(c, c->d_tex.alloc(tex.size()));
   if (!ms.empty()) HIP_TRY(c, hipMemcpy(c->d_meshes.p, ms.data(), ms.size() * sizeof(MeshShade), hipMemcpyHostToDevice));
   if (!lights.empty()) HIP_TRY(c, hipMemcpy(c->d_lights.p, lights.data(), lights.size() * sizeof(float4), hipMemcpyHostToDevice));
   if (!tex.empty()) HIP_TRY(c, hipMemcpy(c->d_tex.p, tex.data(), tex.size() * sizeof(TexInfo), hipMemcpyHostToDevice));
   c->scene.meshes = c->d_meshes.p;
   c->scene.textures = c->d_tex.p;
   c->scene.lights = c->d_lights.p;
   c->scene.unorm_lut = c->d_lut.p;
   c->scene.num_meshes = (uint32_t)ms.size();
   c->scene.num_textures = (uint32_t)tex.size();
   c->scene.num_lights = (uint32_t)c->lights.size();
   return UH_OK;
}

static int build_on_device(uh_ctx* c);

int uh_build_acceleration(uh_ctx* c) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   HIP_TRY(c, hipSetDevice(c->device));
   if (c->device_build) return build_on_device(c);
   auto t0 = std::chrono::steady_clock::now();
   // bake instance transforms: world = ((m0*x + m1*y) + m2*z) + m3 per row (identity: verbatim)
   size_t total = 0;
   for (const HostMesh& m : c->meshes) total += m.indices.size() / 3;
   if (total > kMaxTriangles) return fail(c, UH_ERR_CAPACITY, "scene has more than 2^31 - 2 triangles");
   std::vector<float> corners(9 * total);
   std::vector<uint32_t> keys(total);
   size_t t = 0;
   for (uint32_t mi = 0; mi < c->meshes.size(); mi++) {
      const HostMesh& m = c->meshes[mi];
      const bool ident = is_identity3x4(m.o2w);
      const float* w = m.o2w;
      const uint32_t nt = (uint32_t)m.indices.size() / 3;
      for (uint32_t p = 0; p < nt; p++, t++) {
         for (int k = 0; k < 3; k++) {
            const UhVertex& vx = m.vertices[m.indices[3 * (size_t)p + k]];
            float x = vx.pos[0], y = vx.pos[1], z = vx.pos[2];
            float* o = &corners[9 * t + 3 * k];
            if (ident) {
               o[0] = x;
               o[1] = y;
               o[2] = z;
            } else {
               o[0] = ((w[0] * x + w[1] * y) + w[2] * z) + w[3];
               o[1] = ((w[4] * x + w[5] * y) + w[6] * z) + w[7];
               o[2] = ((w[8] * x + w[9] * y) + w[10] * z) + w[11];
            }
         }
         keys[t] = (mi << kPrimBits) | p;
      }
   }
   BuildInput in{corners.data(), keys.data(), (uint32_t)total};
   BuildOutput bo;
   int threads = (int)std::thread::hardware_concurrency();
   if (threads < 1) threads = 1;
   if (threads > 32) threads = 32;
   build_bvh4(in, bo, threads);
   if (bo.level_start.size() - 1 > kMaxTreeLevels) {
      // a tree deeper than the traversal stack holds (clustered / exponentially scaled geometry) would drop subtrees
      // silently: rebuild it balanced (median splits, depth ceil(log2 n) / 1..2 per 4-wide level)
      build_bvh4(in, bo, threads, true);
      if (bo.level_start.size() - 1 > kMaxTreeLevels) return fail(c, UH_ERR_CAPACITY, "internal: balanced BVH deeper than the traversal stack");
   }

   // packets in leaf order
   std::vector<TriPacket> tp(total);
   std::vector<ShadePacket> sp(total);
   for (size_t i = 0; i < total; i++) {
      uint32_t src = bo.tri_order[i];
      const float* cr = &corners[9 * (size_t)src];
      TriPacket& q = tp[i];
      q.v0[0] = cr[0];
      q.v0[1] = cr[1];
      q.v0[2] = cr[2];
      q.e1x = cr[3] - cr[0];
      q.e1yz[0] = cr[4] - cr[1];
      q.e1yz[1] = cr[5] - cr[2];
      q.e2[0] = cr[6] - cr[0];
      q.e2[1] = cr[7] - cr[1];
      q.e2z = cr[8] - cr[2];
      q.key = keys[src];
      q.pad[0] = q.pad[1] = 0;
      uint32_t mi = keys[src] >> kPrimBits, p = keys[src] & kPrimMask;
      const HostMesh& m = c->meshes[mi];
      const UhVertex* v[3] = {&m.vertices[m.indices[3 * (size_t)p]], &m.vertices[m.indices[3 * (size_t)p + 1]], &m.vertices[m.indices[3 * (size_t)p + 2]]};
      ShadePacket& s = sp[i];
      for (int a = 0; a < 3; a++) {
         s.n0[a] = v[0]->normal[a];
         s.n1[a] = v[1]->normal[a];
         s.n2[a] = v[2]->normal[a];
      }
      for (int a = 0; a < 2; a++) {
         s.uv0[a] = v[0]->uv[a];
         s.uv1[a] = v[1]->uv[a];
         s.uv2[a] = v[2]->uv[a];
      }
      s.mesh = mi;
   }
   if (bo.cnodes.empty()) return fail(c, UH_ERR_INVALID_ARGUMENT, "internal: BVH builder produced no root node");
   if (int st = sync_all(c)) return st;
   if (int st = upload_scene_tables(c)) return st;
   HIP_TRY(c, c->d_nodes.alloc(bo.cnodes.size() * kNodeStride16));
   HIP_TRY(c, c->d_tris.alloc(total * kTriStride16));
   HIP_TRY(c, c->d_shade.alloc(total * 4));
   // 48-byte records into arrays of stride kNodeStride16 / kTriStride16 x 16 bytes
   HIP_TRY(c, hipMemcpy2D(c->d_nodes.p, 16 * kNodeStride16, bo.cnodes.data(), sizeof(Node4C), sizeof(Node4C), bo.cnodes.size(), hipMemcpyHostToDevice));
   if (total) {
      HIP_TRY(c, hipMemcpy2D(c->d_tris.p, 16 * kTriStride16, tp.data(), sizeof(TriPacket), sizeof(TriPacket), total, hipMemcpyHostToDevice));
      HIP_TRY(c, hipMemcpy(c->d_shade.p, sp.data(), total * sizeof(ShadePacket), hipMemcpyHostToDevice));
   }
   c->scene.nodes = reinterpret_cast<const uint4*>(c->d_nodes.p);
   c->scene.tris = c->d_tris.p;
   c->scene.shade = c->d_shade.p;
   c->scene.num_nodes = (uint32_t)bo.nodes.size();
   c->scene.num_tris = (uint32_t)total;
   c->bvh_nodes = c->scene.num_nodes;
   c->bvh_tris = c->scene.num_tris;
   c->packet_keys.resize(total);
   for (size_t i = 0; i < total; i++) c->packet_keys[i] = tp[i].key;
   c->level_start = bo.level_start;
   c->d_obj_corners.release();  // leaf order changed: the next refit re-creates its inputs
   c->geom_version++;
   c->topology_valid = true;
   c->built = true;
   c->build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
   return UH_OK;
}

// Raytracing::rebuild_tlas (raytracing.rs:400-459) for a flattened tree: see refit.hip
int uh_refit_acceleration(uh_ctx* c) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!c->topology_valid)
      return fail(c, UH_ERR_NOT_BUILT, "uh_refit_acceleration: meshes or lights were added since the last uh_build_acceleration (or it never ran)");
   HIP_TRY(c, hipSetDevice(c->device));
   auto t0 = std::chrono::steady_clock::now();
   if (int st = sync_all(c)) return st;  // frames in flight still traverse the old boxes
   const size_t total = c->packet_keys.size();
   if (total && !c->d_obj_corners.p) {
      std::vector<float> oc(9 * total);
      for (size_t i = 0; i < total; i++) {
         const HostMesh& m = c->meshes[c->packet_keys[i] >> kPrimBits];
         const uint32_t p = c->packet_keys[i] & kPrimMask;
         for (int k = 0; k < 3; k++) {
            const UhVertex& vx = m.vertices[m.indices[3 * (size_t)p + k]];
            for (int a = 0; a < 3; a++) oc[9 * i + 3 * k + a] = vx.pos[a];
         }
      }
      HIP_TRY(c, c->d_obj_corners.alloc(9 * total));
      HIP_TRY(c, c->d_world_corners.alloc(9 * total));
      HIP_TRY(c, c->d_node_box.alloc(6 * (size_t)c->scene.num_nodes));
      HIP_TRY(c, c->d_refit_meshes.alloc(c->meshes.size()));
      HIP_TRY(c, hipMemcpy(c->d_obj_corners.p, oc.data(), oc.size() * sizeof(float), hipMemcpyHostToDevice));
   }
   std::vector<RefitMesh> rm(c->meshes.size());
   std::vector<MeshShade> ms(c->meshes.size());
   HIP_TRY(c, hipMemcpy(ms.data(), c->d_meshes.p, ms.size() * sizeof(MeshShade), hipMemcpyDeviceToHost));
   for (size_t i = 0; i < c->meshes.size(); i++) {
      std::memset(&rm[i], 0, sizeof(RefitMesh));
      std::memcpy(rm[i].o2w, c->meshes[i].o2w, sizeof(rm[i].o2w));
      rm[i].identity = is_identity3x4(c->meshes[i].o2w) ? 1u : 0u;
      std::memcpy(ms[i].w2o, c->meshes[i].w2o, sizeof(ms[i].w2o));
   }
   if (!ms.empty()) HIP_TRY(c, hipMemcpy(c->d_meshes.p, ms.data(), ms.size() * sizeof(MeshShade), hipMemcpyHostToDevice));
   if (total) {
      HIP_TRY(c, hipMemcpy(c->d_refit_meshes.p, rm.data(), rm.size() * sizeof(RefitMesh), hipMemcpyHostToDevice));
      RefitArgs a;
      a.obj_corners = c->d_obj_corners.p;
      a.meshes = c->d_refit_meshes.p;
      a.tris = c->d_tris.p;
      a.world_corners = c->d_world_corners.p;
      a.nodes = reinterpret_cast<uint4*>(c->d_nodes.p);
      a.node_box = c->d_node_box.p;
      a.level_start = c->level_start.data();
      a.num_levels = (uint32_t)c->level_start.size() - 1;
      a.num_tris = (uint32_t)total;
      launch_refit(cfg(c), a);
      HIP_TRY(c, hipGetLastError());
      HIP_TRY(c, hipStreamSynchronize(c->stream));
   }
   c->built = true;
   c->geom_version++;
   c->build_ms = c->refit_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
   return UH_OK;
}

// uh_build_acceleration with option "device_build": Morton-order tree built by lbvh.hip, boxes by refit.hip
static int build_on_device(uh_ctx* c) {
   auto t0 = std::chrono::steady_clock::now();
   size_t total = 0;
   for (const HostMesh& m : c->meshes) total += m.indices.size() / 3;
   if (total > kMaxTriangles) return fail(c, UH_ERR_CAPACITY, "scene has more than 2^31 - 2 triangles");
   if (int st = sync_all(c)) return st;
   if (int st = upload_scene_tables(c)) return st;
   if (!c->src_valid) {
      std::vector<float> corners(9 * total);
      std::vector<uint32_t> keys(total);
      std::vector<ShadePacket> sp(total);
      size_t t = 0;
      for (uint32_t mi = 0; mi < c->meshes.size(); mi++) {
         const HostMesh& m = c->meshes[mi];
         const uint32_t nt = (uint32_t)m.indices.size() / 3;
         for (uint32_t p = 0; p < nt; p++, t++) {
            const UhVertex* v[3] = {&m.vertices[m.indices[3 * (size_t)p]], &m.vertices[m.indices[3 * (size_t)p + 1]], &m.vertices[m.indices[3 * (size_t)p + 2]]};
            ShadePacket& s = sp[t];
            for (int k = 0; k < 3; k++)
               for (int a = 0; a < 3; a++) corners[9 * t + 3 * k + a] = v[k]->pos[a];
            for (int a = 0; a < 3; a++) {
               s.n0[a] = v[0]->normal[a];
               s.n1[a] = v[1]->normal[a];
               s.n2[a] = v[2]->normal[a];
            }
            for (int a = 0; a < 2; a++) {
               s.uv0[a] = v[0]->uv[a];
               s.uv1[a] = v[1]->uv[a];
               s.uv2[a] = v[2]->uv[a];
            }
            s.mesh = mi;
            keys[t] = (mi << kPrimBits) | p;
         }
      }
      HIP_TRY(c, c->d_src_corners.alloc(9 * total));
      HIP_TRY(c, c->d_src_keys.alloc(total));
      HIP_TRY(c, c->d_src_shade.alloc(4 * total));
      if (total) {
         HIP_TRY(c, hipMemcpy(c->d_src_corners.p, corners.data(), corners.size() * sizeof(float), hipMemcpyHostToDevice));
         HIP_TRY(c, hipMemcpy(c->d_src_keys.p, keys.data(), keys.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
         HIP_TRY(c, hipMemcpy(c->d_src_shade.p, sp.data(), sp.size() * sizeof(ShadePacket), hipMemcpyHostToDevice));
      }
      c->src_valid = true;
   }
   // per-mesh rows + a box that holds every centroid: the 8 corners of each mesh's object-space box, transformed
   std::vector<RefitMesh> rm(c->meshes.size());
   float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
   for (size_t i = 0; i < c->meshes.size(); i++) {
      const HostMesh& m = c->meshes[i];
      std::memset(&rm[i], 0, sizeof(RefitMesh));
      std::memcpy(rm[i].o2w, m.o2w, sizeof(rm[i].o2w));
      rm[i].identity = is_identity3x4(m.o2w) ? 1u : 0u;
      float olo[3] = {INFINITY, INFINITY, INFINITY}, ohi[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (const UhVertex& v : m.vertices)
         for (int a = 0; a < 3; a++) {
            olo[a] = std::fmin(olo[a], v.pos[a]);
            ohi[a] = std::fmax(ohi[a], v.pos[a]);
         }
      if (m.vertices.empty()) continue;
      for (int k = 0; k < 8; k++) {
         const float x = (k & 1) ? ohi[0] : olo[0], y = (k & 2) ? ohi[1] : olo[1], z = (k & 4) ? ohi[2] : olo[2];
         for (int a = 0; a < 3; a++) {
            const float w = m.o2w[4 * a] * x + m.o2w[4 * a + 1] * y + m.o2w[4 * a + 2] * z + m.o2w[4 * a + 3];
            lo[a] = std::fmin(lo[a], w);
            hi[a] = std::fmax(hi[a], w);
         }
      }
   }
   for (int a = 0; a < 3; a++)
      if (!(lo[a] <= hi[a]) || !std::isfinite(lo[a]) || !std::isfinite(hi[a])) lo[a] = hi[a] = 0.0f;
   const size_t node_cap = total > 1 ? total : 1;
   HIP_TRY(c, c->d_nodes.alloc(node_cap * kNodeStride16));
   HIP_TRY(c, c->d_tris.alloc(total * kTriStride16));
   HIP_TRY(c, c->d_shade.alloc(total * 4));
   HIP_TRY(c, c->d_obj_corners.alloc(9 * total));
   HIP_TRY(c, c->d_world_corners.alloc(9 * total));
   HIP_TRY(c, c->d_node_box.alloc(6 * node_cap));
   HIP_TRY(c, c->d_refit_meshes.alloc(rm.size()));
   if (!rm.empty()) HIP_TRY(c, hipMemcpy(c->d_refit_meshes.p, rm.data(), rm.size() * sizeof(RefitMesh), hipMemcpyHostToDevice));
   LbvhArgs la;
   la.src_corners = c->d_src_corners.p;
   la.src_keys = c->d_src_keys.p;
   la.src_shade = c->d_src_shade.p;
   la.meshes = c->d_refit_meshes.p;
   for (int a = 0; a < 3; a++) {
      la.bounds_lo[a] = lo[a];
      la.bounds_hi[a] = hi[a];
   }
   la.num_tris = (uint32_t)total;
   la.kind = c->device_build_kind;
   la.ploc_radius = uh_ctx::kPlocRadius;
   la.sah_top = c->ploc_sah_top;
   la.nodes = reinterpret_cast<uint4*>(c->d_nodes.p);
   la.node_capacity = (uint32_t)node_cap;
   la.tris = c->d_tris.p;
   la.shade = c->d_shade.p;
   la.obj_corners = c->d_obj_corners.p;
   uint32_t num_nodes = 1;
   hipError_t e = lbvh_build(la, c->stream, c->level_start, &num_nodes);
   if (e == hipErrorUnknown) {
      // the builder gave up on this geometry (a PLOC round that merges nothing: every union area inf / NaN, e.g. coordinates
      // around 1e19 whose area products overflow): like a tree that came out too deep, such a scene gets the host builder
      (void)hipGetLastError();
      c->device_build = false;
      const int st = uh_build_acceleration(c);
      c->device_build = true;
      return st;
   }
   if (e != hipSuccess) return fail(c, UH_ERR_HIP, std::string("device BVH build: ") + hipGetErrorString(e));
   if (c->level_start.size() - 1 > kMaxTreeLevels) {
      // a Morton tree over clustered geometry can be a long chain; deeper than the traversal stack it would drop
      // subtrees silently: this scene gets the host builder (which has a balanced fallback of its own)
      c->device_build = false;
      const int st = uh_build_acceleration(c);
      c->device_build = true;
      return st;
   }
   if (total) {
      RefitArgs a;
      a.obj_corners = c->d_obj_corners.p;
      a.meshes = c->d_refit_meshes.p;
      a.tris = c->d_tris.p;
      a.world_corners = c->d_world_corners.p;
      a.nodes = reinterpret_cast<uint4*>(c->d_nodes.p);
      a.node_box = c->d_node_box.p;
      a.level_start = c->level_start.data();
      a.num_levels = (uint32_t)c->level_start.size() - 1;
      a.num_tris = (uint32_t)total;
      launch_refit(cfg(c), a);
      HIP_TRY(c, hipGetLastError());
   }
   HIP_TRY(c, hipStreamSynchronize(c->stream));
   c->scene.nodes = reinterpret_cast<const uint4*>(c->d_nodes.p);
   c->scene.tris = c->d_tris.p;
   c->scene.shade = c->d_shade.p;
   c->scene.num_nodes = num_nodes;
   c->scene.num_tris = (uint32_t)total;
   c->bvh_nodes = num_nodes;
   c->bvh_tris = (uint32_t)total;
   c->packet_keys.assign(total, 0u);  // only its size is used once d_obj_corners exists (uh_refit_acceleration)
   c->geom_version++;
   c->topology_valid = true;
   c->built = true;
   c->build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
   return UH_OK;
}

static FrameParams make_params(uh_ctx* c, const UhViewUniformData& v) {
   FrameParams fp;
   std::memset(&fp, 0, sizeof(fp));
   std::memcpy(fp.inv_view, v.inverse_view, sizeof(fp.inv_view));
   std::memcpy(fp.inv_proj, v.inverse_projection, sizeof(fp.inv_proj));
   std::memcpy(fp.prev_pv, v.prev_frame_projection_view, sizeof(fp.prev_pv));
   // normalize(view.sun_dir) (reference.rgen:64, reference.rmiss:18): v * (1 / sqrt(dot))
   float d = (v.sun_dir[0] * v.sun_dir[0] + v.sun_dir[1] * v.sun_dir[1]) + v.sun_dir[2] * v.sun_dir[2];
   float inv = 1.0f / std::sqrt(d);
   for (int a = 0; a < 3; a++) fp.sun_dir[a] = v.sun_dir[a] * inv;
   fp.W = c->W;
   fp.H = c->H;
   // reference.rgen:24: int(float(total_samples) + time * 10000.0)
   fp.frame_number = (uint32_t)(int32_t)((float)v.total_samples + v.time * 10000.0f);
   fp.batch_frames = 1;
   fp.frame_numbers[0] = fp.frame_number;
   fp.total_samples_of[0] = v.total_samples;
   fp.samples_per_frame = v.samples_per_frame;
   fp.total_samples = v.total_samples;
   fp.num_bounces = v.num_bounces;
   fp.accumulation_limit = v.accumulation_limit;
   fp.sky_enabled = v.sky_enabled;
   fp.sun_shadow_enabled = v.sun_shadow_enabled;
   fp.lights_enabled = v.lights_enabled;
   fp.use_ris = v.use_ris_light_sampling;
   fp.full_frame_restir = c->full_frame_restir ? 1u : 0u;
   fp.furnace = c->furnace ? 1u : 0u;
   fp.num_lights_used = v.num_lights < v.max_num_lights_used ? v.num_lights : v.max_num_lights_used;
   fp.temporal_enabled = v.temporal_reuse_enabled;
   fp.spatial_enabled = v.spatial_reuse_enabled;
   fp.tp_rank = c->tp_rank;
   fp.tp_world = c->tp_world;
   fp.tp_tile = c->tp_tile;
   fp.tiles_x = (c->W + c->tp_tile - 1) / c->tp_tile;
   fp.owned_pixels = c->tp_world > 1 ? c->owned_pixels.p : nullptr;
   fp.n_owned = c->tp_world > 1 ? c->n_owned : c->W * c->H;
   return fp;
}

// every stream of every slot idle (read-backs, scene rebuilds, stats)
static int sync_all(uh_ctx* c) {
   if (c->restir_stream) HIP_TRY(c, hipStreamSynchronize(c->restir_stream));
   for (auto& s : c->slots) {
      if (!s.ready) continue;
      HIP_TRY(c, hipStreamSynchronize(s.stream));
      HIP_TRY(c, hipStreamSynchronize(s.side));
   }
   return UH_OK;
}

// slot i exists and can hold `batch` frames worth of paths
static int ensure_slot(uh_ctx* c, uint32_t i, uint32_t batch = 1) {
   size_t need = (size_t)(c->tp_world > 1 ? c->n_owned : c->W * c->H) * batch;  // path ids are dense over the rank's owned pixels
   if (need < 64) need = 64;  // a rank that owns no pixel (more ranks than tiles) still gets a well-formed, empty slot
   Slot& s = c->slots[i];
   if (s.ready && s.capacity >= need) return UH_OK;
   if (s.ready) {
      HIP_TRY(c, hipStreamSynchronize(s.stream));
      HIP_TRY(c, hipStreamSynchronize(s.side));
      if (c->last_acc == s.ev_acc) c->last_acc = nullptr;
      for (hipEvent_t& r : c->spatial_reader)
         if (r == s.ev_acc) r = nullptr;
      if (c->t_start == s.frame_start) c->t_start = nullptr;
      if (c->t_stop == s.frame_stop) c->t_stop = nullptr;
      if (c->last_slot == &s) c->last_slot = nullptr;
      s.destroy();
   }
   HIP_TRY(c, s.create(need));
   return UH_OK;
}

static int attach_sun_inline_records(uh_ctx* c);
// the grid `g` (or its refusal) becomes the context's grid for (geom, dir). Frames in flight may still read the old buffers.
static int adopt_sun_grid(uh_ctx* c, const SunGridHost& g, bool ok, const float dir[3], uint64_t geom, float build_ms) {
   if (int st = sync_all(c)) return st;
   c->sun_attempted = true;
   c->sun_have_pending = false;
   c->sun_geom = geom;
   std::memcpy(c->sun_dir_built, dir, sizeof(float) * 3);
   c->sun_valid = false;
   c->sun_why = g.why_not;
   c->sun_mean_list = (float)g.mean_list;
   c->sun_max_list = g.max_list;
   c->sun_fallback_area = (float)g.fallback_area;
   c->sun_cells = c->sun_entries = 0;
   if (ok) {
      // per cell four words: offset into the entries | cover depth | the first entry (packet, far depth) (sun_grid.h kSunCellWords)
      std::vector<uint32_t> cells(kSunCellWords * g.cell_start.size());
      for (size_t k = 0; k < g.cell_start.size(); k++) {
         const float cover = k < g.cell_cover.size() ? g.cell_cover[k] : -INFINITY;
         cells[kSunCellWords * k] = g.cell_start[k];
         std::memcpy(&cells[kSunCellWords * k + 1], &cover, 4);
         SunGridEntry first{kEmptyRef, 0.0f};
         if (k + 1 < g.cell_start.size() && g.cell_start[k + 1] > g.cell_start[k]) first = g.entries[g.cell_start[k]];
         cells[kSunCellWords * k + 2] = first.packet;
         std::memcpy(&cells[kSunCellWords * k + 3], &first.wmax, 4);
      }
      HIP_TRY(c, c->d_sun_cells.alloc(cells.size()));
      HIP_TRY(c, c->d_sun_entries.alloc(g.entries.size() ? g.entries.size() : 1));
      HIP_TRY(c, hipMemcpy(c->d_sun_cells.p, cells.data(), cells.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      if (!g.entries.empty()) HIP_TRY(c, hipMemcpy(c->d_sun_entries.p, g.entries.data(), g.entries.size() * sizeof(SunGridEntry), hipMemcpyHostToDevice));
      SunGridDev& d = c->sun_dev;
      std::memcpy(d.U, g.U, sizeof(d.U));
      std::memcpy(d.V, g.V, sizeof(d.V));
      std::memcpy(d.W, g.W, sizeof(d.W));
      d.u0 = g.u0;
      d.v0 = g.v0;
      d.inv_cell = g.inv_cell;
      d.nx = g.nx;
      d.ny = g.ny;
      d.max_walk = c->sun_limits.max_walk;
      d.cell_start = c->d_sun_cells.p;
      d.entries = c->d_sun_entries.p;
      c->sun_cells = g.nx * g.ny;
      c->sun_entries = (uint32_t)g.entries.size();
      c->sun_valid = true;
   } else {
      c->d_sun_cells.release();
      c->d_sun_entries.release();
   }
   c->sun_build_ms = build_ms;
   return attach_sun_inline_records(c);
}

// the device builder's result becomes the context's grid (its buffers change owner)
static int adopt_sun_grid_device(uh_ctx* c, SunGridDevice& g, bool ok, const float dir[3], uint64_t geom) {
   c->sun_attempted = true;
   c->sun_have_pending = false;
   c->sun_geom = geom;
   std::memcpy(c->sun_dir_built, dir, sizeof(float) * 3);
   c->sun_valid = false;
   c->sun_why = g.why_not;
   c->sun_mean_list = (float)g.mean_list;
   c->sun_max_list = g.max_list;
   c->sun_fallback_area = (float)g.fallback_area;
   c->sun_cells = c->sun_entries = 0;
   c->d_sun_cells.release();
   c->d_sun_entries.release();
   if (ok) {
      const size_t ncell = (size_t)g.params.nx * g.params.ny;
      c->d_sun_cells.p = g.cells;
      c->d_sun_cells.base = g.cells;
      c->d_sun_cells.n = kSunCellWords * (ncell + 1);
      c->d_sun_entries.p = g.entries;
      c->d_sun_entries.base = g.entries;
      c->d_sun_entries.n = (size_t)g.num_entries;
      g.cells = nullptr;
      g.entries = nullptr;
      SunGridDev& d = c->sun_dev;
      std::memcpy(d.U, g.params.U, sizeof(d.U));
      std::memcpy(d.V, g.params.V, sizeof(d.V));
      std::memcpy(d.W, g.params.W, sizeof(d.W));
      d.u0 = g.params.u0;
      d.v0 = g.params.v0;
      d.inv_cell = g.params.inv_cell;
      d.nx = g.params.nx;
      d.ny = g.params.ny;
      d.max_walk = c->sun_limits.max_walk;
      d.cell_start = c->d_sun_cells.p;
      d.entries = c->d_sun_entries.p;
      c->sun_cells = g.params.nx * g.params.ny;
      c->sun_entries = (uint32_t)g.num_entries;
      c->sun_valid = true;
   }
   g.release();
   return attach_sun_inline_records(c);
}

// the adopted grid's lists once more with the packets inline (sun_grid.h SunGridDev::recs): 64 bytes per entry
static int attach_sun_inline_records(uh_ctx* c) {
   c->d_sun_recs.release();
   c->sun_dev.recs = nullptr;
   c->d_sun_coarse.release();
   c->sun_dev.coarse = nullptr;
   c->sun_dev.coarse_shift = c->sun_dev.coarse_nx = 0;
   if (c->sun_valid && c->sun_coarse_shift) {  // the coarse cover (sun_grid.h)
      const uint32_t sh = c->sun_coarse_shift, b = 1u << sh, cnx = (c->sun_dev.nx + b - 1) / b, cny = (c->sun_dev.ny + b - 1) / b;
      HIP_TRY(c, c->d_sun_coarse.alloc((size_t)cnx * cny));
      HIP_TRY(c, (hipError_t)build_sun_coarse_cover((void*)c->stream, c->d_sun_cells.p, c->sun_dev.nx, c->sun_dev.ny, sh, c->d_sun_coarse.p));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      c->sun_dev.coarse = c->d_sun_coarse.p;
      c->sun_dev.coarse_shift = sh;
      c->sun_dev.coarse_nx = cnx;
   }
   const uint64_t n = c->sun_entries;
   const uint64_t budget = c->sun_inline_max_mb < 0 ? 4ull * 16 * kTriStride16 * c->scene.num_tris : (uint64_t)c->sun_inline_max_mb << 20;
   if (!c->sun_valid || n == 0 || n * 64ull > budget) return UH_OK;
   if (c->d_sun_recs.alloc((size_t)n * 4) != hipSuccess) {  // no room: the plain lists serve
      (void)hipGetLastError();
      return UH_OK;
   }
   HIP_TRY(c, (hipError_t)build_sun_inline_records((void*)c->stream, c->d_tris.p, c->d_sun_entries.p, n, c->d_sun_recs.p));
   HIP_TRY(c, hipStreamSynchronize(c->stream));
   c->sun_dev.recs = reinterpret_cast<const float*>(c->d_sun_recs.p);
   return UH_OK;
}

// the sun grid for this frame's direction, if there is (or now should be) one: see uh_ctx::sun_*
static int ensure_sun_grid(uh_ctx* c, const float dir[3]) {
   c->sun_this_frame = false;
   if (!c->sun_grid_enabled || c->scene.num_tris == 0) return UH_OK;
   const bool same_geom = c->sun_geom == c->geom_version;
   const bool same_dir = std::memcmp(dir, c->sun_dir_built, sizeof(float) * 3) == 0;
   if (same_geom && c->sun_attempted && same_dir) {
      c->sun_have_pending = false;
      c->sun_this_frame = c->sun_valid;
      return UH_OK;
   }
   if (c->sun_attempted) {
      // another direction or other geometry than the grid's (a sun dragged in the UI, instances moved with rebuild_tlas every
      // frame): rebuild once the same pair has been asked for twice in a row, walk the tree meanwhile - a build costs as much
      // as a thousand frames' worth of what the grid saves
      const bool settled = c->sun_have_pending && c->sun_geom_pending == c->geom_version && std::memcmp(dir, c->sun_dir_pending, sizeof(float) * 3) == 0;
      std::memcpy(c->sun_dir_pending, dir, sizeof(float) * 3);
      c->sun_geom_pending = c->geom_version;
      c->sun_have_pending = true;
      if (!settled) return UH_OK;
   }
   // build: the packets as the device holds them (leaf order; host build, device build and refit all end there)
   const uint32_t n = c->scene.num_tris;
   if (c->sun_device_build) {
      // on the device, from the packets where they lie: a few milliseconds, inside this frame call; the frames in flight may still
      // read the grid this one replaces
      if (int st = sync_all(c)) return st;
      const auto t0 = std::chrono::steady_clock::now();
      SunGridDevice g;
      const bool ok = build_sun_grid_device((void*)c->stream, c->d_tris.p, n, dir, c->sun_limits, nullptr, g);
      if (!ok && g.why_not.rfind("device build:", 0) == 0) return fail(c, UH_ERR_HIP, "sun grid: " + g.why_not);
      if (int st = adopt_sun_grid_device(c, g, ok, dir, c->geom_version)) return st;
      c->sun_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
      c->sun_this_frame = c->sun_valid;
      return UH_OK;
   }
   int threads = (int)std::thread::hardware_concurrency();
   threads = threads < 1 ? 1 : (threads > 32 ? 32 : threads);
   if (int st = sync_all(c)) return st;
   const auto t0 = std::chrono::steady_clock::now();
   std::vector<float> packets(12 * (size_t)n);
   HIP_TRY(c, hipMemcpy2D(packets.data(), sizeof(TriPacket), c->d_tris.p, 16 * kTriStride16, sizeof(TriPacket), n, hipMemcpyDeviceToHost));
   SunGridHost g;
   const bool ok = build_sun_grid(packets.data(), n, dir, c->sun_limits, threads, g);
   if (int st = adopt_sun_grid(c, g, ok, dir, c->geom_version, 0.0f)) return st;
   c->sun_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
   c->sun_this_frame = c->sun_valid;
   return UH_OK;
}

// the camera grid for this call's camera, if there is (or now should be) one: see uh_ctx::cam_*
static int ensure_camera_grid(uh_ctx* c, const FrameParams& fp, uint32_t batch) {
   c->cam_this_frame = false;
   if (!c->cam_grid_enabled || c->scene.num_tris == 0) return UH_OK;
   float mats[32];
   std::memcpy(mats, fp.inv_view, sizeof(float) * 16);
   std::memcpy(mats + 16, fp.inv_proj, sizeof(float) * 16);
   const bool same = c->cam_attempted && c->cam_geom == c->geom_version && std::memcmp(mats, c->cam_mats, sizeof(mats)) == 0;
   if (same) {
      c->cam_have_pending = false;
      c->cam_this_frame = c->cam_valid;
      return UH_OK;
   }
   // another camera or other geometry than the grid's: build once the same pair has been asked for twice in a row - or at once
   // when this call alone carries enough frames of it to repay the build
   const bool settled = c->cam_have_pending && c->cam_geom_pending == c->geom_version && std::memcmp(mats, c->cam_mats_pending, sizeof(mats)) == 0;
   std::memcpy(c->cam_mats_pending, mats, sizeof(mats));
   c->cam_geom_pending = c->geom_version;
   c->cam_have_pending = true;
   if (!settled && batch < 8) return UH_OK;
   if (int st = sync_all(c)) return st;
   const auto t0 = std::chrono::steady_clock::now();
   SunGridDevice g;
   const bool ok = build_camera_grid_device((void*)c->stream, c->d_tris.p, c->scene.num_tris, fp.inv_view, fp.inv_proj, c->W, c->H, c->cam_limits, g);
   if (!ok && g.why_not.rfind("device build:", 0) == 0) return fail(c, UH_ERR_HIP, "camera grid: " + g.why_not);
   c->cam_attempted = true;
   c->cam_have_pending = false;
   c->cam_geom = c->geom_version;
   std::memcpy(c->cam_mats, mats, sizeof(mats));
   c->cam_valid = false;
   c->cam_why = g.why_not;
   c->cam_mean_list = (float)g.mean_list;
   c->cam_max_list = g.max_list;
   c->cam_max_list_interior = g.max_list_interior;
   c->cam_cells = c->cam_entries = 0;
   c->d_cam_cells.release();
   c->d_cam_entries.release();
   if (ok) {
      const size_t ncell = (size_t)g.params.nx * g.params.ny;
      c->d_cam_cells.p = g.cells;
      c->d_cam_cells.base = g.cells;
      c->d_cam_cells.n = ncell + 1;
      c->d_cam_entries.p = g.entries;
      c->d_cam_entries.base = g.entries;
      c->d_cam_entries.n = (size_t)g.num_entries;
      g.cells = nullptr;
      g.entries = nullptr;
      SunGridDev& d = c->cam_dev;
      d = SunGridDev{};
      d.nx = g.params.nx;
      d.ny = g.params.ny;
      d.max_walk = c->cam_limits.max_walk;
      d.walk_whole = c->cam_walk_whole;
      d.cell_start = c->d_cam_cells.p;
      d.entries = c->d_cam_entries.p;
      c->cam_cells = c->W * c->H;
      c->cam_entries = (uint32_t)g.num_entries;
      c->cam_valid = true;
   }
   g.release();
   c->cam_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
   c->cam_this_frame = c->cam_valid;
   return UH_OK;
}

// reference_pt_pass of one frame on one slot (reference.rgen:22-145 as a kernel chain)
static int enqueue_path_trace(uh_ctx* c, Slot& s, const FrameParams& fp) {
   LaunchCfg lc = cfg(c);
   lc.stream = s.stream;
   if (fp.batch_frames == 1) {  // a lone frame's launches are small: fewer persistent waves reach the end of their tails sooner
      lc.closest_blocks_per_cu = std::min(lc.closest_blocks_per_cu, uh_ctx::kSingleFrameBlocksPerCu);
      lc.shadow_blocks_per_cu = std::min(lc.shadow_blocks_per_cu, uh_ctx::kSingleFrameBlocksPerCu);
   }
   Control* ctl = s.control.p;
   DeviceStats* st = c->dstats.p;
   // reference.rgen:28: samples of one frame run back to back (the raygen RNG state carries over)
   for (uint32_t smp = 0; smp < fp.samples_per_frame; smp++) {
      HIP_TRY(c, hipMemsetAsync(ctl, 0, sizeof(Control), s.stream));
      uint32_t slot = 0;
      launch_generate(lc, fp, s.ps, ctl, smp);
      // a lone frame: bounce 0 as a wavefront (its rays are coherent, the camera grid serves them), the others inside k_path_fused
      // ... when the caller waits for its frames (the frame before this one has left the GPU): the fused kernel fills the chip by
      // itself, so with frames in flight - a caller that does not wait - the wavefront's small launches interleave better (2.2 against
      // 2.5 ms per frame). Both give the same image.
      // ... and when the frame is not so large that its launches are wavefronts of a batch's size already (a 4K frame: 8.3 M paths -
      // the fused kernel 2.5 % behind on config 1, 12 % on config 3; at 1080p 12 % ahead, at 960 x 540 37 %, profiles/README.md)
      const bool gpu_idle = !c->last_acc || hipEventQuery(c->last_acc) == hipSuccess;
      const bool fused = c->fused_bounces && fp.batch_frames == 1 && fp.num_bounces >= 2 && fp.num_bounces <= 64 && s.ps.shard_cap < (1u << 23) &&
                         ((gpu_idle && fp.n_owned <= uh_ctx::kFusedMaxPaths) || c->fused_always);
      const bool fused_sun0 = fused && fp.sun_shadow_enabled == 1 && fp.lights_enabled != 1;  // bounce 0's sun rays inside the fused kernel too
      for (uint32_t b = 0; b < (fused ? 1u : fp.num_bounces); b++) {
         begin_timed(c, (b == 0 && c->cam_this_frame) ? 3 : 0, s.stream);
         if (b == 0 && c->cam_this_frame) {
            // no tree walk for the primary rays of a camera at rest (and no launch for one when no pixel's list is too long for the grid kernel)
            launch_trace_camera_grid(lc, fp, c->scene, s.ps, ctl, st, slot, slot + 1, c->cam_dev, c->cam_max_list_interior > std::max(c->cam_dev.walk_whole, c->cam_dev.max_walk));
            slot += 2;
         } else
            launch_trace_closest(lc, c->scene, s.ps, ctl, st, b, slot++, b == 0 ? UH_RAY_PRIMARY : UH_RAY_BOUNCE);
         end_timed(c, s.stream);
         const bool side_shadow = c->overlap_shadow && (fp.sun_shadow_enabled == 1 || fp.lights_enabled == 1);
         const bool side_used = side_shadow || c->overlap_miss;
         // shade_hit(b) rewrites ray_o / thr / rad of the paths shadow(b-1) still reads on the side stream, and refills the
         // miss queue shade_miss(b-1) reads there
         if (side_used && b > 0) HIP_TRY(c, hipStreamWaitEvent(s.stream, s.ev_shadowed, 0));
         begin_timed(c, 2, s.stream);
         launch_shade_hit(lc, fp, c->scene, s.ps, c->im, ctl, st, b);  // also hands the bounce's misses to shade_miss (Q_MISS)
         if (!c->overlap_miss) launch_shade_miss(lc, fp, s.ps, ctl, st, b);
         end_timed(c, s.stream);
         // shade_miss(b) and the shadow queries of bounce b are independent of trace_closest(b+1) (they only read what
         // shade_hit(b) wrote): on the side stream their blocks fill the tail of the other kernel
         LaunchCfg lsh = lc;
         hipStream_t sh_stream = s.stream;
         if (side_used) {
            HIP_TRY(c, hipEventRecord(s.ev_shaded, s.stream));
            HIP_TRY(c, hipStreamWaitEvent(s.side, s.ev_shaded, 0));
         }
         if (c->overlap_miss) {
            LaunchCfg ls = lc;
            ls.stream = s.side;
            begin_timed(c, 2, s.side);
            launch_shade_miss(ls, fp, s.ps, ctl, st, b);
            end_timed(c, s.side);
         }
         if (side_shadow) {
            lsh.stream = s.side;
            sh_stream = s.side;
         }
         if (fp.sun_shadow_enabled == 1 && !fused_sun0) {
            begin_timed(c, 1, sh_stream);
            if (c->sun_this_frame) {
               launch_trace_sun_grid(lsh, fp, c->scene, s.ps, ctl, st, b, slot++, c->sun_dev);
               launch_trace_shadow(lsh, fp, c->scene, s.ps, ctl, st, b, slot++, false, true);  // the rays the grid handed over (border cells, long lists)
            } else
               launch_trace_shadow(lsh, fp, c->scene, s.ps, ctl, st, b, slot++, false);
            end_timed(c, sh_stream);
         }
         if (fp.lights_enabled == 1) {
            begin_timed(c, 4, sh_stream);
            launch_trace_shadow(lsh, fp, c->scene, s.ps, ctl, st, b, slot++, true);
            end_timed(c, sh_stream);
         }
         if (side_used) HIP_TRY(c, hipEventRecord(s.ev_shadowed, s.side));
      }
      const bool join_side = (c->overlap_miss || c->overlap_shadow) && fp.num_bounces > 0;
      // the side stream's work: the sky integrals, and - unless the fused kernel asks them itself - bounce 0's sun and light rays, whose
      // results the fused kernel starts from
      if (join_side && !fused_sun0) {
         HIP_TRY(c, hipEventRecord(s.ev_side_done, s.side));
         HIP_TRY(c, hipStreamWaitEvent(s.stream, s.ev_side_done, 0));
      }
      if (fused) {
         begin_timed(c, 0, s.stream);
         launch_path_fused(lc, fp, c->scene, s.ps, ctl, st, c->sun_dev, c->sun_this_frame, fused_sun0);
         end_timed(c, s.stream);
      }
      if (join_side && fused_sun0) {  // (bounce 0's sky integrals ran beside the fused kernel)
         HIP_TRY(c, hipEventRecord(s.ev_side_done, s.side));
         HIP_TRY(c, hipStreamWaitEvent(s.stream, s.ev_side_done, 0));
      }
      // the paths still alive after the last bounce: their radiance (with what the last bounce's shadow rays added) goes to the
      // per-id array the tail reads
      if (fp.num_bounces > 0 && !fused) launch_flush_survivors(lc, fp, s.ps, ctl);
      const bool last = smp + 1 == fp.samples_per_frame;
      // the accumulate / store tail (rgen:130-144) is a read-modify-write on the accumulation image:
      // frames must apply it in order, everything before it may overlap with other frames in flight
      if (last && c->last_acc && c->last_acc != s.ev_acc) HIP_TRY(c, hipStreamWaitEvent(s.stream, c->last_acc, 0));
      launch_finish_sample(lc, fp, s.ps, c->im, smp, last);
   }
   if (fp.samples_per_frame == 0) {
      // zero samples: the raygen still runs its accumulate / store tail
      if (c->last_acc && c->last_acc != s.ev_acc) HIP_TRY(c, hipStreamWaitEvent(s.stream, c->last_acc, 0));
      launch_resolve(lc, c->im, c->W, c->H, fp.total_samples, fp.accumulation_limit);
   }
   HIP_TRY(c, hipEventRecord(s.ev_acc, s.stream));
   c->last_acc = s.ev_acc;
   return UH_OK;
}

// ---- one batch of frames, in phases: begin | per frame: reservoir chain, exchange | end (the path-tracing wavefront).
// uh_render_frame(s) runs the phases back to back; an in-process group (mgpu.hip) interleaves them over its contexts,
// because frame f + 1's temporal pass of one GPU reads the bands every other GPU wrote in frame f.
struct uh_batch {
   FrameParams fp;
   uint32_t pass_mask = 0, batch = 0;
   bool restir_frame = false, reads_reservoirs = false;
   int read_slot[kMaxBatchFrames];
   LaunchCfg rc;
};

static UhReservoir* spatial_buf(uh_ctx* c, int slot) { return slot ? c->spatial_ring.p + (size_t)(slot - 1) * c->res_stride : c->reservoirs[2].p; }

// rows of this context's reservoir passes (uh_set_restir_partition): its band, the band's 30-row neighbourhood (plus the last
// row for the first band: spatial_reuse.rgen:54's wrapped offset is clamped to size - 1), and for the G-buffer cast the row
// above each of those (the 2 x 2 corner filter)
static void restir_rows(const uh_ctx* c, UhRestirRows& r) {
   const uint32_t H = c->H, B = c->rp_world > 1 ? c->rp_band_rows : H;
   r = UhRestirRows{};
   r.rows_per_band = B;
   const uint32_t b0 = std::min<uint64_t>((uint64_t)c->rp_rank * B, H), b1 = std::min<uint64_t>((uint64_t)b0 + B, H);
   r.band_row0 = b0;
   r.band_rows = b1 - b0;
   if (b1 == b0) return;  // more ranks than bands: nothing to do here
   const uint32_t halo = 30;
   const uint32_t t0 = b0 > halo ? b0 - halo : 0, t1 = std::min<uint64_t>((uint64_t)b1 + halo, H);
   r.reuse_row0 = t0;
   r.reuse_rows = t1 - t0;
   if (b0 < halo && t1 < H) {
      r.reuse_extra_row0 = H - 1;
      r.reuse_extra_rows = 1;
   }
   const uint32_t g0 = t0 > 0 ? t0 - 1 : 0;
   r.cast_row0 = g0;
   r.cast_rows = t1 - g0;
   if (r.reuse_extra_rows) {
      const uint32_t e0 = std::max(H >= 2 ? H - 2 : 0u, t1);  // rows H - 2 and H - 1, without what the first interval already holds
      r.cast_extra_row0 = e0;
      r.cast_extra_rows = H - e0;
   }
}
static RowSpans spans_of(const uh_ctx* c, uint32_t row0, uint32_t rows, uint32_t extra0, uint32_t extra_rows) { return RowSpans{{row0, extra0}, {rows, extra_rows}, c->W}; }

static int batch_begin(uh_ctx* c, const UhViewUniformData* view, uint32_t pass_mask, uint32_t batch, uh_batch& bs) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!view) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_render_frame: null view");
   if (!c->built && c->topology_valid && view->rebuild_tlas == 1) {
      // the application moved instances and asks for the per-frame rebuild (main.rs:392,526; graph.rs:715-741)
      if (int st = uh_refit_acceleration(c)) return st;
   }
   if (!c->built) return fail(c, UH_ERR_NOT_BUILT, "uh_render_frame before uh_build_acceleration");
   if (view->num_bounces > kMaxBounces) return fail(c, UH_ERR_INVALID_ARGUMENT, "num_bounces > 64");
   if (view->samples_per_frame > 4096) return fail(c, UH_ERR_INVALID_ARGUMENT, "samples_per_frame > 4096 (the reference UI stops at 10)");
   if (view->num_lights > c->lights.size() && (view->lights_enabled == 1 || (pass_mask & UH_PASS_RESTIR)))
      return fail(c, UH_ERR_INVALID_ARGUMENT, "view.num_lights exceeds the lights added with uh_add_light");
   HIP_TRY(c, hipSetDevice(c->device));
   FrameParams& fp = bs.fp;
   fp = make_params(c, *view);
   fp.batch_frames = batch;
   for (uint32_t f = 0; f < batch; f++) {
      // frame f of the batch: total_samples advanced by the application once per frame (main.rs:467-469)
      const uint32_t total = view->total_samples + f * view->samples_per_frame;
      fp.total_samples_of[f] = total;
      fp.frame_numbers[f] = (uint32_t)(int32_t)((float)total + view->time * 10000.0f);
   }
   // the per-camera grid serves the primary rays of the path tracer and the G-buffer cast of this call
   c->cam_this_frame = false;
   const bool primary_rays = ((pass_mask & UH_PASS_REFERENCE_PT) && fp.num_bounces > 0 && fp.samples_per_frame > 0) || (pass_mask & UH_PASS_GBUFFER);
   if (primary_rays)
      if (int st = ensure_camera_grid(c, fp, batch)) return st;
   // with the camera grid nothing but bounce 0's own kernels reads a primary ray's state: it is not stored
   fp.primary_implicit = (c->primary_implicit && c->cam_this_frame && fp.samples_per_frame == 1) ? 1u : 0u;
   bs.pass_mask = pass_mask;
   bs.batch = batch;
   bs.restir_frame = (pass_mask & UH_PASS_RESTIR) != 0;
   // the path tracer reads spatial_reuse_reservoirs when it samples lights from them (rgen:98)
   bs.reads_reservoirs = (pass_mask & UH_PASS_REFERENCE_PT) && fp.lights_enabled == 1 && fp.use_ris == 1;

   if (batch > 1 && !(pass_mask & UH_PASS_REFERENCE_PT)) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_render_frames: a batch needs the path-tracing pass");
   if (batch > kRestirBatch && bs.restir_frame) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_render_frames: more reservoir-pass frames in one batch than the spatial ring holds");
   c->t_start = c->t_stop = nullptr;
   // the slot each frame of the batch samples lights from (rgen:98): without reservoir passes in this call, the current one
   for (uint32_t f = 0; f < kMaxBatchFrames; f++) bs.read_slot[f] = c->spatial_cur;

   if (bs.restir_frame) {
      if (!c->restir_stream) {
         HIP_TRY(c, hipStreamCreateWithFlags(&c->restir_stream, hipStreamNonBlocking));
         HIP_TRY(c, hipEventCreateWithFlags(&c->ev_restir, hipEventDisableTiming));
         HIP_TRY(c, hipEventCreate(&c->rs_start));
         HIP_TRY(c, hipEventCreate(&c->rs_stop));
      }
      if (!c->spatial_ring.p) {
         HIP_TRY(c, c->spatial_ring.alloc((size_t)(kSpatialRing - 1) * c->res_stride));
         // every slot is written before it is read - by this context's passes; under a row partition other ranks' bands arrive by
         // the exchange, and a caller that times one rank without one (exchange == NULL) must still read defined reservoirs
         if (c->rp_world > 1) HIP_TRY(c, hipMemsetAsync(c->spatial_ring.p, 0, c->spatial_ring.n * sizeof(UhReservoir), c->restir_stream));
         HIP_TRY(c, hipStreamSynchronize(c->stream));  // slot 0 (zeroed at creation or uh_write_reservoirs) is the history
      }
      bs.rc = cfg(c);
      bs.rc.stream = c->restir_stream;
      HIP_TRY(c, hipEventRecord(c->rs_start, c->restir_stream));
      c->t_start = c->rs_start;
   }
   return UH_OK;
}

// G-buffer cast and reservoir chain of frame f of the batch, on the reservoir stream
static int batch_restir_frame(uh_ctx* c, uh_batch& bs, uint32_t f) {
   if (!bs.restir_frame) return UH_OK;
   HIP_TRY(c, hipSetDevice(c->device));
   const uint32_t pass_mask = bs.pass_mask;
   const size_t npix = (size_t)c->W * c->H;
   FrameParams ff = bs.fp;  // frame f of the batch: its own RNG frame number (the reservoir kernels read nothing else per frame)
   ff.frame_number = bs.fp.frame_numbers[f];
   ff.total_samples = bs.fp.total_samples_of[f];
   UhRestirRows rows;
   restir_rows(c, rows);
   const RowSpans band = spans_of(c, rows.band_row0, rows.band_rows, 0, 0);
   const RowSpans reuse = spans_of(c, rows.reuse_row0, rows.reuse_rows, rows.reuse_extra_row0, rows.reuse_extra_rows);
   const RowSpans cast = spans_of(c, rows.cast_row0, rows.cast_rows, rows.cast_extra_row0, rows.cast_extra_rows);
   Images im = c->im;
   im.prev_spatial = spatial_buf(c, c->spatial_cur);
   im.reservoirs[2] = spatial_buf(c, c->spatial_cur);
   if (pass_mask & UH_PASS_GBUFFER) {
      if (!c->gb_hit.p) {
         HIP_TRY(c, c->gb_ray_o.alloc(npix));
         HIP_TRY(c, c->gb_ray_d.alloc(npix));
         HIP_TRY(c, c->gb_hit.alloc(npix));
      }
      const RawRays gps{c->gb_ray_o.p, c->gb_ray_d.p, c->gb_hit.p};
      launch_gbuffer(bs.rc, ff, c->scene, gps, im, c->dstats.p, cast, band.total(), c->cam_this_frame ? &c->cam_dev : nullptr);
   }
   if (pass_mask & UH_PASS_RESET_RESERVOIRS) launch_reset_reservoirs(bs.rc, ff, im, reuse);
   if (pass_mask & UH_PASS_INITIAL_RIS) launch_initial_ris(bs.rc, ff, c->scene, im, reuse);
   if (pass_mask & UH_PASS_TEMPORAL_REUSE) launch_temporal_reuse(bs.rc, ff, c->scene, im, reuse);
   if (pass_mask & UH_PASS_SPATIAL_REUSE) {
      // writes the NEXT slot of the ring: the path-tracing wavefronts that still read the current one keep going
      const int nxt = (c->spatial_cur + 1) % kSpatialRing;
      if (c->spatial_reader[nxt]) HIP_TRY(c, hipStreamWaitEvent(c->restir_stream, c->spatial_reader[nxt], 0));
      im.reservoirs[2] = spatial_buf(c, nxt);
      launch_spatial_reuse(bs.rc, ff, c->scene, im, band);
      c->spatial_cur = nxt;
      c->im.reservoirs[2] = spatial_buf(c, nxt);
      if (c->rp_world > 1) {
         if (!c->ev_band[nxt]) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_band[nxt], hipEventDisableTiming));
         HIP_TRY(c, hipEventRecord(c->ev_band[nxt], c->restir_stream));
      }
   }
   bs.read_slot[f] = c->spatial_cur;
   return UH_OK;
}

// the other ranks' bands of the buffer frame f's spatial pass wrote (uh_set_restir_partition)
static int batch_exchange(uh_ctx* c, uh_batch& bs, uint32_t f) {
   (void)f;
   if (!bs.restir_frame || !c->rp_exchange || !(bs.pass_mask & UH_PASS_SPATIAL_REUSE)) return UH_OK;  // world 1 with an exchange: a one-rank all-gather (rehearsals)
   HIP_TRY(c, hipSetDevice(c->device));
   const uint64_t band_bytes = (uint64_t)c->rp_band_rows * c->W * sizeof(UhReservoir);
   if (int st = c->rp_exchange(c->rp_user, (void*)c->restir_stream, (void*)c->im.reservoirs[2], band_bytes, c->rp_rank, c->rp_world))
      return fail(c, UH_ERR_HIP, std::string("the reservoir exchange reported error ") + std::to_string(st));
   return UH_OK;
}

static int batch_end(uh_ctx* c, uh_batch& bs) {
   HIP_TRY(c, hipSetDevice(c->device));
   FrameParams& fp = bs.fp;
   const uint32_t batch = bs.batch;
   if (bs.restir_frame) {
      HIP_TRY(c, hipEventRecord(c->ev_restir, c->restir_stream));
      c->restir_recorded = true;
      HIP_TRY(c, hipEventRecord(c->rs_stop, c->restir_stream));
      c->t_stop = c->rs_stop;
   }
   for (uint32_t f = 0; f < kMaxBatchFrames; f++) fp.spatial_of[f] = spatial_buf(c, bs.read_slot[f < batch ? f : 0]);

   if (bs.pass_mask & UH_PASS_REFERENCE_PT) {
      c->sun_this_frame = false;
      if (fp.sun_shadow_enabled == 1 && fp.num_bounces > 0 && fp.samples_per_frame > 0)
         if (int st = ensure_sun_grid(c, fp.sun_dir)) return st;
      const uint32_t in_flight = c->frames_in_flight ? c->frames_in_flight : 1;
      {
         const uint32_t si = c->next_slot;
         c->next_slot = (c->next_slot + 1) % in_flight;
         int st = ensure_slot(c, si, batch);
         if (st != UH_OK) return st;
         Slot& s = c->slots[si];
         const FrameParams& fk = fp;
         // rgen:98 reads this frame's spatial_reuse_reservoirs
         if (bs.reads_reservoirs && c->restir_recorded) HIP_TRY(c, hipStreamWaitEvent(s.stream, c->ev_restir, 0));
         HIP_TRY(c, hipEventRecord(s.frame_start, s.stream));
         if (!c->t_start) c->t_start = s.frame_start;
         st = enqueue_path_trace(c, s, fk);
         if (st != UH_OK) return st;
         if (bs.reads_reservoirs)
            for (uint32_t f = 0; f < batch; f++) c->spatial_reader[bs.read_slot[f]] = s.ev_acc;
         HIP_TRY(c, hipEventRecord(s.frame_stop, s.stream));
         c->t_stop = s.frame_stop;
         c->last_slot = &s;
      }
   }
   c->frame_timed = true;
   c->frames += batch;
   HIP_TRY(c, hipGetLastError());
   return UH_OK;
}

static int render_batch(uh_ctx* c, const UhViewUniformData* view, uint32_t pass_mask, uint32_t batch) {
   uh_batch bs;
   if (int st = batch_begin(c, view, pass_mask, batch, bs)) return st;
   for (uint32_t f = 0; f < batch; f++) {
      if (int st = batch_restir_frame(c, bs, f)) return st;
      if (int st = batch_exchange(c, bs, f)) return st;
   }
   return batch_end(c, bs);
}

// frames one wavefront carries for this pass mask (option "batch_frames", 0 = auto), and the slots it will rotate through
static int plan_batch(uh_ctx* c, uint32_t pass_mask, uint32_t* out_batch) {
   uint32_t batch = c->batch_frames;
   if (!batch) {
      // auto: about 32 M paths per wavefront (1080p: 16 frames, 4K: 4, a rank's eighth of 1080p: 32, 256 x 256: 32) - what
      // the launches need to fill the chip and amortise their tails; 1080p 4 / 8 / 12 / 16 frames = 6,271 / 6,341 / 6,388 /
      // 6,398 Mrays/s, the short paths of the iso-surface scene 4,974 / 5,393 / - / 6,223 (tools/sweep_batch.sh,
      // profiles/README.md) - within 33 M path-state records per slot (3.7 GB; path ids are dense over the pixels a rank
      // owns, so its wavefronts carry more frames than a whole frame's would)
      const uint64_t pixels = (uint64_t)c->W * c->H, owned = c->n_owned ? c->n_owned : pixels;
      uint64_t b = (32u << 20) / (owned ? owned : 1), cap = (33u << 20) / (owned ? owned : 1);
      if (b > cap) b = cap;
      batch = (uint32_t)(b < 1 ? 1 : b);
   }
   if (batch > kMaxBatchFrames) batch = kMaxBatchFrames;
   if ((pass_mask & UH_PASS_RESTIR) && batch > kRestirBatch) batch = kRestirBatch;
   if (!(pass_mask & UH_PASS_REFERENCE_PT)) batch = 1;  // reservoir passes alone: frame by frame
   // create every frames-in-flight slot now: the first call (an application's first frames, a benchmark's
   // warm-up) pays for the allocations and stream creation, not whichever later wavefront first reaches a slot
   if (pass_mask & UH_PASS_REFERENCE_PT)
      for (uint32_t i = 0; i < (c->frames_in_flight ? c->frames_in_flight : 1); i++)
         if (int st = ensure_slot(c, i, batch)) return st;
   *out_batch = batch;
   return UH_OK;
}

// ---- the phases for an in-process group (csrc/context_internal.h; mgpu.hip) ----
int uhi_plan_batch(uh_ctx* c, uint32_t pass_mask, uint32_t* out_batch) {
   if (!c || !out_batch) return UH_ERR_INVALID_ARGUMENT;
   HIP_TRY(c, hipSetDevice(c->device));
   return plan_batch(c, pass_mask, out_batch);
}
int uhi_batch_begin(uh_ctx* c, const UhViewUniformData* view, uint32_t pass_mask, uint32_t batch, uh_batch** out) {
   if (!c || !out) return UH_ERR_INVALID_ARGUMENT;
   uh_batch* bs = new uh_batch();
   int st = batch_begin(c, view, pass_mask, batch, *bs);
   if (st != UH_OK) {
      delete bs;
      bs = nullptr;
   }
   *out = bs;
   return st;
}
int uhi_batch_restir_frame(uh_ctx* c, uh_batch* bs, uint32_t f) { return (c && bs) ? batch_restir_frame(c, *bs, f) : UH_ERR_INVALID_ARGUMENT; }
int uhi_batch_end(uh_ctx* c, uh_batch* bs) {
   if (!c || !bs) return UH_ERR_INVALID_ARGUMENT;
   int st = batch_end(c, *bs);
   delete bs;
   return st;
}
void uhi_batch_abandon(uh_batch* bs) { delete bs; }
// what a peer pulls a band from / into: the buffer the last spatial pass wrote, the event recorded behind that pass, the
// stream the passes run on
int uhi_exchange_endpoints(uh_ctx* c, void** spatial_base, void** band_event, void** stream, uint64_t* band_bytes) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (spatial_base) *spatial_base = (void*)c->im.reservoirs[2];
   if (band_event) *band_event = (void*)c->ev_band[c->spatial_cur];
   if (stream) *stream = (void*)c->restir_stream;
   if (band_bytes) *band_bytes = (uint64_t)c->rp_band_rows * c->W * sizeof(UhReservoir);
   return UH_OK;
}

int uhi_iso_reference_triangulation(uh_ctx* c) { return c && c->iso_reference ? 1 : 0; }

int uh_render_frame(uh_ctx* c, const UhViewUniformData* view, uint32_t pass_mask) { return render_batch(c, view, pass_mask, 1); }

int uh_render_frames(uh_ctx* c, const UhViewUniformData* view, uint32_t pass_mask, uint32_t count) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!view || count == 0) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_render_frames: null view or zero frames");
   UhViewUniformData v = *view;
   uint32_t done = 0, batch = 1;
   if (int st = plan_batch(c, pass_mask, &batch)) return st;
   // wavefronts of equal size: 20 frames at 16 per wavefront go as 10 + 10, not 16 + 4 (a short wavefront's launches are mostly tail)
   const uint32_t n_batches = (count + batch - 1) / batch, even = (count + n_batches - 1) / n_batches;
   while (done < count) {
      uint32_t b = count - done < even ? count - done : even;
      int st = render_batch(c, &v, pass_mask, b);
      if (st != UH_OK) return st;
      done += b;
      v.total_samples += b * v.samples_per_frame;
   }
   return UH_OK;
}

int uh_reset_accumulation(uh_ctx* c) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   HIP_TRY(c, hipSetDevice(c->device));
   if (int st = sync_all(c)) return st;
   HIP_TRY(c, hipMemsetAsync(c->accumulation.p, 0, c->accumulation.n * sizeof(float4), c->stream));
   HIP_TRY(c, hipMemsetAsync(c->output.p, 0, c->output.n * sizeof(uchar4), c->stream));
   HIP_TRY(c, hipStreamSynchronize(c->stream));  // the next frame may run on another slot's stream, which does not wait for this one
   return UH_OK;
}

int uh_synchronize(uh_ctx* c) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   HIP_TRY(c, hipSetDevice(c->device));
   return sync_all(c);
}

// Device -> host, BLOCKING, behind a wait for the context's stream: every read-back of the library goes through here or is a
// plain hipMemcpy. (The round-4 soaks' heap corruption - profiles/README.md "the soak crash" - sits inside the HIP runtime copy
// bundled with the PyTorch wheel and does not show against /opt/rocm's; under the bundled one, asynchronous copies into locals
// followed by a wait for their stream made it twenty times as frequent, and a pinned staging buffer - per context or one for the
// process - made it worse again. This form is the one that ran cleanest under both.)
static int staged_read(uh_ctx* c, void* dst, const void* src, size_t bytes) {
   if (bytes == 0) return UH_OK;
   HIP_TRY(c, hipStreamSynchronize(c->stream));
   HIP_TRY(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
   return UH_OK;
}

static int read_back(uh_ctx* c, void* dst, const void* src, size_t bytes) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!dst) return fail(c, UH_ERR_INVALID_ARGUMENT, "null destination");
   HIP_TRY(c, hipSetDevice(c->device));
   if (int st = sync_all(c)) return st;
   return staged_read(c, dst, src, bytes);
}

int uh_read_accumulation(uh_ctx* c, float* out) { return read_back(c, out, c ? c->accumulation.p : nullptr, c ? c->accumulation.n * sizeof(float4) : 0); }
int uh_read_output_bgra8(uh_ctx* c, uint8_t* out) { return read_back(c, out, c ? c->output.p : nullptr, c ? c->output.n * sizeof(uchar4) : 0); }
int uh_read_gbuffer_position(uh_ctx* c, float* out) { return read_back(c, out, c ? c->gbuffer.p : nullptr, c ? c->gbuffer.n * sizeof(float4) : 0); }
int uh_read_reservoirs(uh_ctx* c, int which, UhReservoir* out) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (which < 0 || which > 2) return fail(c, UH_ERR_INVALID_ARGUMENT, "reservoir buffer index must be 0..2");
   // spatial_reuse_reservoirs is double-buffered (render_batch): the current one is what the reference's single buffer holds
   const UhReservoir* src = which == 2 ? c->im.reservoirs[2] : c->reservoirs[which].p;
   return read_back(c, out, src, (size_t)c->W * c->H * sizeof(UhReservoir));
}
int uh_write_reservoirs(uh_ctx* c, int which, const UhReservoir* in) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (which < 0 || which > 2 || !in) return fail(c, UH_ERR_INVALID_ARGUMENT, "reservoir buffer index must be 0..2");
   HIP_TRY(c, hipSetDevice(c->device));
   if (int st = sync_all(c)) return st;
   UhReservoir* dst = which == 2 ? c->im.reservoirs[2] : c->reservoirs[which].p;
   HIP_TRY(c, hipMemcpyAsync(dst, in, (size_t)c->W * c->H * sizeof(UhReservoir), hipMemcpyHostToDevice, c->stream));
   HIP_TRY(c, hipStreamSynchronize(c->stream));
   return UH_OK;
}

int uh_write_gbuffer_position(uh_ctx* c, const float* in) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!in) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_write_gbuffer_position: null source");
   HIP_TRY(c, hipSetDevice(c->device));
   if (int st = sync_all(c)) return st;
   HIP_TRY(c, hipMemcpyAsync(c->gbuffer.p, in, c->gbuffer.n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
   HIP_TRY(c, hipStreamSynchronize(c->stream));
   return UH_OK;
}

int uh_trace_closest(uh_ctx* c, const float* rays, uint32_t n, float* out_tuv, uint32_t* out_mesh, uint32_t* out_prim) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!c->built) return fail(c, UH_ERR_NOT_BUILT, "uh_trace_closest before uh_build_acceleration");
   if (n == 0) return UH_OK;
   if (!rays || !out_tuv || !out_mesh || !out_prim) return fail(c, UH_ERR_INVALID_ARGUMENT, "null argument");
   HIP_TRY(c, hipSetDevice(c->device));
   std::vector<float4> o(n), d(n);
   for (uint32_t i = 0; i < n; i++) {
      const float* r = rays + 8 * (size_t)i;
      o[i] = make_float4(r[0], r[1], r[2], r[3]);
      d[i] = make_float4(r[4], r[5], r[6], r[7]);
   }
   DevBuf<float4> d_o, dd, dh;
   HIP_TRY(c, d_o.alloc(n));
   HIP_TRY(c, dd.alloc(n));
   HIP_TRY(c, dh.alloc(n));
   HIP_TRY(c, hipMemcpyAsync(d_o.p, o.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
   HIP_TRY(c, hipMemcpyAsync(dd.p, d.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
   begin_timed(c, 0);
   launch_trace_closest_raw(cfg(c), c->scene, d_o.p, dd.p, dh.p, n);
   end_timed(c);
   std::vector<float4> h(n);
   if (int st = staged_read(c, h.data(), dh.p, n * sizeof(float4))) return st;
   // packet index -> key needs the packet table: read keys back once
   std::vector<TriPacket> tp(c->scene.num_tris);
   if (!tp.empty()) HIP_TRY(c, hipMemcpy2D(tp.data(), sizeof(TriPacket), c->d_tris.p, 16 * kTriStride16, sizeof(TriPacket), tp.size(), hipMemcpyDeviceToHost));
   for (uint32_t i = 0; i < n; i++) {
      uint32_t idx;
      std::memcpy(&idx, &h[i].w, 4);
      if (idx == kEmptyRef || idx >= tp.size()) {
         out_tuv[3 * i] = -1.0f;
         out_tuv[3 * i + 1] = out_tuv[3 * i + 2] = 0.0f;
         out_mesh[i] = out_prim[i] = 0xffffffffu;
      } else {
         out_tuv[3 * i] = h[i].x;
         out_tuv[3 * i + 1] = h[i].y;
         out_tuv[3 * i + 2] = h[i].z;
         out_mesh[i] = tp[idx].key >> kPrimBits;
         out_prim[i] = tp[idx].key & kPrimMask;
      }
   }
   d_o.release();
   dd.release();
   dh.release();
   return UH_OK;
}

int uh_trace_any(uh_ctx* c, const float* rays, uint32_t n, uint8_t* out_occluded) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!c->built) return fail(c, UH_ERR_NOT_BUILT, "uh_trace_any before uh_build_acceleration");
   if (n == 0) return UH_OK;
   if (!rays || !out_occluded) return fail(c, UH_ERR_INVALID_ARGUMENT, "null argument");
   HIP_TRY(c, hipSetDevice(c->device));
   std::vector<float4> o(n), d(n);
   for (uint32_t i = 0; i < n; i++) {
      const float* r = rays + 8 * (size_t)i;
      o[i] = make_float4(r[0], r[1], r[2], r[3]);
      d[i] = make_float4(r[4], r[5], r[6], r[7]);
   }
   DevBuf<float4> d_o, dd;
   DevBuf<uint32_t> occ;
   HIP_TRY(c, d_o.alloc(n));
   HIP_TRY(c, dd.alloc(n));
   HIP_TRY(c, occ.alloc(n));
   HIP_TRY(c, hipMemcpyAsync(d_o.p, o.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
   HIP_TRY(c, hipMemcpyAsync(dd.p, d.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
   begin_timed(c, 1);
   launch_trace_any_raw(cfg(c), c->scene, d_o.p, dd.p, occ.p, n);
   end_timed(c);
   std::vector<uint32_t> h(n);
   if (int st = staged_read(c, h.data(), occ.p, n * sizeof(uint32_t))) return st;
   for (uint32_t i = 0; i < n; i++) out_occluded[i] = h[i] ? 1 : 0;
   d_o.release();
   dd.release();
   occ.release();
   return UH_OK;
}

int uh_get_stats(uh_ctx* c, UhStats* out) {
   if (!c || !out) return UH_ERR_INVALID_ARGUMENT;
   HIP_TRY(c, hipSetDevice(c->device));
   if (int st = sync_all(c)) return st;
   DeviceStats ds;
   if (int st = staged_read(c, &ds, c->dstats.p, sizeof(ds))) return st;
   drain_timed(c);
   if (c->frame_timed) {
      float ms = 0.0f;
      if (c->t_start && c->t_stop && hipEventElapsedTime(&ms, c->t_start, c->t_stop) == hipSuccess) c->last_frame_ms = ms;
   }
   std::memset(out, 0, sizeof(*out));
   for (int i = 0; i < UH_RAY_KINDS; i++) out->rays[i] = ds.rays[i];
   out->nodes_visited = ds.nodes_visited;
   out->tris_tested = ds.tris_tested;
   out->shadow_nodes_visited = ds.shadow_nodes_visited;
   out->shadow_tris_tested = ds.shadow_tris_tested;
   out->closest_hits = ds.closest_hits;
   out->misses = ds.misses;
   out->frames = c->frames;
   out->bvh_nodes = c->bvh_nodes;
   out->bvh_triangles = c->bvh_tris;
   out->build_ms = c->build_ms;
   out->last_frame_ms = c->last_frame_ms;
   out->trace_closest_ms = c->ms_by_kind[0];
   out->trace_shadow_ms = c->ms_by_kind[1];
   out->shade_ms = c->ms_by_kind[2];
   out->camera_grid_ms = c->ms_by_kind[3];
   out->trace_light_ms = c->ms_by_kind[4];
   out->trace_light_launches = c->trace_light_launches;
   out->light_nodes_visited = ds.light_nodes_visited;
   out->light_tris_tested = ds.light_tris_tested;
   out->trace_closest_launches = c->trace_closest_launches;
   out->sun_grid_cells = c->sun_valid ? c->sun_cells : 0;
   out->sun_grid_entries = c->sun_valid ? c->sun_entries : 0;
   out->sun_grid_build_ms = c->sun_build_ms;
   out->sun_grid_mean_list = c->sun_mean_list;
   out->sun_tree_rays = ds.sun_tree_rays;
   out->camera_grid_cells = c->cam_valid && c->cam_this_frame ? c->cam_cells : 0;  // in use by the last frame call
   out->camera_grid_entries = c->cam_valid && c->cam_this_frame ? c->cam_entries : 0;
   out->camera_grid_build_ms = c->cam_build_ms;
   out->camera_grid_mean_list = c->cam_mean_list;
   out->camera_tree_rays = ds.cam_tree_rays;
   out->camera_grid_tris_tested = ds.cam_tris_tested;
   out->sun_covered_rays = ds.sun_covered_rays;
   if (c->sun_valid) out->sun_grid_bytes = c->d_sun_cells.n * sizeof(uint32_t) + c->d_sun_entries.n * sizeof(SunGridEntry) + c->d_sun_recs.n * sizeof(float4) + c->d_sun_coarse.n * sizeof(float);
   if (c->cam_valid && c->cam_this_frame) out->camera_grid_bytes = c->d_cam_cells.n * sizeof(uint32_t) + c->d_cam_entries.n * sizeof(SunGridEntry);
   return UH_OK;
}

int uh_reset_stats(uh_ctx* c) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   HIP_TRY(c, hipSetDevice(c->device));
   if (int st = sync_all(c)) return st;
   drain_timed(c);
   // on the context's stream and waited for: a hipMemset on the null stream may still be in flight when the call returns, and
   // the frames' non-blocking streams do not wait for it - the next frame's first counters could be wiped
   HIP_TRY(c, hipMemsetAsync(c->dstats.p, 0, sizeof(DeviceStats), c->stream));
   HIP_TRY(c, hipStreamSynchronize(c->stream));
   c->frames = 0;
   for (float& ms : c->ms_by_kind) ms = 0.0f;
   c->trace_closest_launches = c->trace_light_launches = 0;
   return UH_OK;
}

// The options (24): DESIGN.md section 7 has the table with defaults and what was measured. Variants that two rounds of measurements
// left behind (batch traversal kernels, wave-per-tile primary rays, sub-frame interleave, the fused coarse-cover look-up, spatial
// splits, insertion-based tree optimisation, the host-thread grid build) were removed in round 5 together with their options.
int uh_set_option(uh_ctx* c, const char* name, int value) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!name) return fail(c, UH_ERR_INVALID_ARGUMENT, "null option name");
   const std::string n(name);
   auto range = [&](int lo, int hi) { return value >= lo && value <= hi; };
   auto bad = [&](const char* what) { return fail(c, UH_ERR_INVALID_ARGUMENT, n + " " + what); };
   // ---- diagnostics
   if (n == "count_visits")
      c->count_visits = value != 0;
   else if (n == "time_kernels") {
      if (c->time_kernels && !value) {
         (void)sync_all(c);
         drain_timed(c);
      }
      c->time_kernels = value != 0;
   }
   // ---- the tree
   else if (n == "device_build") {
      // 0 = host SAH builder; 1 = device build, PLOC under a host SAH top; 2 = device build, radix tree
      if (!range(0, 2)) return bad("must be 0, 1 (PLOC) or 2 (radix tree)");
      if (c->device_build != (value != 0) || (value && c->device_build_kind != (uint32_t)value)) c->built = c->topology_valid = false;
      c->device_build = value != 0;
      if (value) c->device_build_kind = (uint32_t)value;
   } else if (n == "ploc_sah_top") {
      if (!range(0, 1 << 20)) return bad("must be 0..1048576");
      if (c->ploc_sah_top != (uint32_t)value && c->device_build) c->built = c->topology_valid = false;
      c->ploc_sah_top = (uint32_t)value;
   }
   // ---- the sun grid (sun_grid.h): every change takes effect with the next grid that is built
   else if (n == "sun_grid")
      c->sun_grid_enabled = value != 0;  // 0: the sun shadow rays always walk the tree
   else if (n == "sun_grid_build") {
      c->sun_device_build = value != 0;  // 1: built on the device (sun_grid_build.hip); 0: by the host builder (sun_grid.cpp, the reference implementation)
      c->sun_attempted = false;
   } else if (n == "sun_grid_density") {
      if (!range(1, 4096)) return bad("(entries per triangle) must be 1..4096");
      c->sun_limits.entries_per_triangle = (double)value;
      c->sun_attempted = false;
   } else if (n == "sun_grid_max_walk") {
      if (!range(1, 4096)) return bad("(longest list a ray tests itself) must be 1..4096");
      c->sun_limits.max_walk = (uint32_t)value;
      c->sun_attempted = false;
   } else if (n == "sun_grid_max_mb") {
      if (!range(1, 65536)) return bad("must be 1..65536");
      c->sun_limits.max_entries = ((uint64_t)value << 20) / sizeof(SunGridEntry);
      c->sun_attempted = false;
   } else if (n == "sun_grid_force") {
      // 1: the grid is never refused for what it would be worth (more than 12 entries per occupied cell, more than a fifth of the scene's
      // surface handed to the tree): tests and studies of scenes the defaults would refuse; 0: the defaults
      const SunGridLimits defaults;
      c->sun_limits.max_mean_list = value ? 1e30 : defaults.max_mean_list;
      c->sun_limits.max_fallback_area = value ? 2.0 : defaults.max_fallback_area;
      c->sun_attempted = false;
   } else if (n == "sun_grid_inline_max_mb") {
      // the lists a second time as 64-byte records that carry their packet (one round trip per triangle test instead of two): memory budget
      // in MB; -1 (default) = four times the packet array; 0 = never
      if (!range(-1, 1 << 20)) return bad("must be -1 (auto: 4 x the packet array), 0 (off) .. 1048576");
      c->sun_inline_max_mb = value;
      c->sun_attempted = false;
   } else if (n == "sun_grid_coarse") {
      // the coarse cover (sun_grid.h): one depth per block of 2^value x 2^value cells, asked before the cell's own record; 0: none
      if (!range(0, 6)) return bad("(log2 of the block edge in cells) must be 0..6");
      c->sun_coarse_shift = (uint32_t)value;
      c->sun_attempted = false;
   }
   // ---- the camera grid
   else if (n == "camera_grid")
      c->cam_grid_enabled = value != 0;  // 0: the primary rays always walk the tree
   else if (n == "camera_grid_max_walk") {
      if (!range(1, 4096)) return bad("must be 1..4096");
      c->cam_limits.max_walk = (uint32_t)value;
      c->cam_attempted = false;
   } else if (n == "camera_grid_walk_whole") {
      // lists of up to this many packets are walked whole by the grid kernel (those beyond camera_grid_max_walk are not sorted: no early
      // exit) instead of handing the ray to the tree; 0: every list beyond camera_grid_max_walk goes to the tree
      if (!range(0, 65536)) return bad("must be 0..65536");
      c->cam_walk_whole = (uint32_t)value;
      c->cam_attempted = false;
   } else if (n == "camera_grid_max_mean_list_x10") {
      if (!range(1, 100000)) return bad("must be 1..100000");
      c->cam_limits.max_mean_list = value / 10.0;
      c->cam_attempted = false;
   } else if (n == "primary_implicit")
      c->primary_implicit = value != 0;  // 0: bounce 0's state planes are stored by k_generate even when the camera grid is in use
   // ---- what the frames compute
   else if (n == "full_frame_restir")
      c->full_frame_restir = value != 0;
   else if (n == "iso_reference_triangulation")
      c->iso_reference = value != 0;
   else if (n == "furnace")
      c->furnace = value != 0;  // applies to the frames enqueued from now on
   // ---- how the frames are scheduled
   else if (n == "overlap")
      c->overlap_miss = c->overlap_shadow = value != 0;  // k_shade_miss and the shadow traversals on the slot's side stream
   else if (n == "batch_frames") {
      if (!range(0, (int)kMaxBatchFrames)) return bad("must be 0 (auto) .. 32");
      c->batch_frames = (uint32_t)value;
   } else if (n == "frames_in_flight") {
      if (!range(1, (int)kMaxSlots)) return bad("must be 1..8");
      (void)sync_all(c);
      c->frames_in_flight = (uint32_t)value;
      c->next_slot = 0;
   } else if (n == "fused_bounces") {
      // 0: a lone frame runs the wavefront of a batch (four launches per bounce); 1: its bounces 1 .. in one persistent kernel; 2..8: that
      // kernel's blocks per CU
      if (!range(-1, 8)) return bad("must be 0 (off), 1 (on when no frame is in flight), -1 (always on) or 2..8 (as 1, and blocks per CU of its grid)");
      c->fused_bounces = value != 0;
      c->fused_always = value == -1;
      if (value >= 2) c->fused_blocks_per_cu = (uint32_t)value;
   } else if (n == "trace_blocks_per_cu") {
      if (!range(1, 8)) return bad("must be 1..8");
      c->closest_blocks_per_cu = c->shadow_blocks_per_cu = (uint32_t)value;  // persistent grids of the traversal kernels
   } else
      return fail(c, UH_ERR_INVALID_ARGUMENT, "unknown option: " + n);
   return UH_OK;
}

int uh_set_tile_partition(uh_ctx* c, uint32_t rank, uint32_t world, uint32_t tile_size) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (world == 0 || rank >= world || tile_size == 0) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_set_tile_partition: need rank < world, tile_size > 0");
   HIP_TRY(c, hipSetDevice(c->device));
   if (int st = sync_all(c)) return st;
   c->tp_rank = rank;
   c->tp_world = world;
   c->tp_tile = tile_size;
   c->n_owned = c->W * c->H;
   c->owned_pixels.release();
   if (world > 1) {
      const uint32_t tiles_x = (c->W + tile_size - 1) / tile_size;
      std::vector<uint32_t> own;
      own.reserve((size_t)c->W * c->H / world + 1);
      for (uint32_t y = 0; y < c->H; y++)
         for (uint32_t x = 0; x < c->W; x++)
            if (((y / tile_size) * tiles_x + x / tile_size) % world == rank) own.push_back(y * c->W + x);
      c->n_owned = (uint32_t)own.size();
      HIP_TRY(c, c->owned_pixels.alloc(own.size() ? own.size() : 1));
      if (!own.empty()) HIP_TRY(c, hipMemcpy(c->owned_pixels.p, own.data(), own.size() * 4, hipMemcpyHostToDevice));
   }
   return UH_OK;
}

int uh_set_restir_partition(uh_ctx* c, uint32_t rank, uint32_t world, UhRestirExchangeFn exchange, void* user) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (world == 0 || rank >= world) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_set_restir_partition: need rank < world");
   HIP_TRY(c, hipSetDevice(c->device));
   if (int st = sync_all(c)) return st;
   const size_t npix = (size_t)c->W * c->H;
   const uint32_t band_rows = (c->H + world - 1) / world;
   const size_t stride = std::max(npix, (size_t)band_rows * world * c->W);
   if (stride != c->res_stride) {
      // the history (the current slot of the ring) moves into a buffer of the new length, which becomes slot 0
      DevBuf<UhReservoir> fresh;
      HIP_TRY(c, fresh.alloc(stride));
      HIP_TRY(c, hipMemsetAsync(fresh.p, 0, stride * sizeof(UhReservoir), c->stream));
      HIP_TRY(c, hipMemcpyAsync(fresh.p, c->im.reservoirs[2], npix * sizeof(UhReservoir), hipMemcpyDeviceToDevice, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      c->reservoirs[2].release();
      c->reservoirs[2] = fresh;
      fresh.p = nullptr;
      fresh.n = 0;
      c->spatial_ring.release();
      c->res_stride = stride;
   } else if (c->spatial_cur != 0) {
      HIP_TRY(c, hipMemcpyAsync(c->reservoirs[2].p, c->im.reservoirs[2], npix * sizeof(UhReservoir), hipMemcpyDeviceToDevice, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
   }
   c->spatial_cur = 0;
   for (auto& r : c->spatial_reader) r = nullptr;
   c->im.reservoirs[2] = c->reservoirs[2].p;
   c->im.prev_spatial = c->reservoirs[2].p;
   c->rp_rank = rank;
   c->rp_world = world;
   c->rp_band_rows = world > 1 ? band_rows : c->H;
   c->rp_exchange = exchange;
   c->rp_user = user;
   return UH_OK;
}

int uh_get_restir_rows(uh_ctx* c, UhRestirRows* out) {
   if (!c || !out) return UH_ERR_INVALID_ARGUMENT;
   restir_rows(c, *out);
   return UH_OK;
}

// ---- the exchange over RCCL. librccl is opened at run time (types from its header, no link dependency): a process that
// never attaches needs no RCCL at all, and one that has torch's copy loaded gets that same copy (same soname). ----
namespace {
struct RcclApi {
   void* so = nullptr;
   decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
   decltype(&ncclCommInitRank) CommInitRank = nullptr;
   decltype(&ncclCommDestroy) CommDestroy = nullptr;
   decltype(&ncclAllGather) AllGather = nullptr;
   decltype(&ncclSend) Send = nullptr;
   decltype(&ncclRecv) Recv = nullptr;
   decltype(&ncclGroupStart) GroupStart = nullptr;
   decltype(&ncclGroupEnd) GroupEnd = nullptr;
   decltype(&ncclCommCount) CommCount = nullptr;
   decltype(&ncclGetErrorString) GetErrorString = nullptr;
   std::string why;
};
RcclApi* rccl_api() {
   static RcclApi api;
   static bool tried = false;
   if (tried) return &api;
   tried = true;
   for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.so) break;
   }
   if (!api.so) {
      api.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "");
      return &api;
   }
   api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.so, "ncclGetUniqueId");
   api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.so, "ncclCommInitRank");
   api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.so, "ncclCommDestroy");
   api.AllGather = (decltype(api.AllGather))dlsym(api.so, "ncclAllGather");
   api.Send = (decltype(api.Send))dlsym(api.so, "ncclSend");
   api.Recv = (decltype(api.Recv))dlsym(api.so, "ncclRecv");
   api.GroupStart = (decltype(api.GroupStart))dlsym(api.so, "ncclGroupStart");
   api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.so, "ncclGroupEnd");
   api.CommCount = (decltype(api.CommCount))dlsym(api.so, "ncclCommCount");
   api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.so, "ncclGetErrorString");
   if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd) {
      api.why = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd";
      api.so = nullptr;
   }
   return &api;
}
struct RcclLink {
   ncclComm_t comm = nullptr;
   int last = 0;
   uint32_t rank = 0, world = 1;
};
// UhRestirExchangeFn: in-place all-gather of the bands, enqueued on the reservoir stream
int rccl_exchange(void* user, void* stream, void* base, uint64_t band_bytes, uint32_t rank, uint32_t) {
   RcclLink* l = (RcclLink*)user;
   l->last = (int)rccl_api()->AllGather((const char*)base + (size_t)rank * band_bytes, base, (size_t)band_bytes, ncclInt8, l->comm, (hipStream_t)stream);
   return l->last;
}
}  // namespace

int uh_rccl_unique_id(uint8_t out_id[128]) {
   static_assert(sizeof(ncclUniqueId) == 128, "the C ABI hands the id over as 128 bytes");
   if (!out_id) return UH_ERR_INVALID_ARGUMENT;
   RcclApi* api = rccl_api();
   if (!api->so) return UH_ERR_HIP;
   ncclUniqueId id;
   if (api->GetUniqueId(&id) != ncclSuccess) return UH_ERR_HIP;
   memcpy(out_id, &id, 128);
   return UH_OK;
}

int uh_rccl_attach(uh_ctx* c, uint32_t rank, uint32_t world, const uint8_t id[128]) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!id || world == 0 || rank >= world) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_rccl_attach: need an id and rank < world");
   RcclApi* api = rccl_api();
   if (!api->so) return fail(c, UH_ERR_HIP, "uh_rccl_attach: " + api->why);
   HIP_TRY(c, hipSetDevice(c->device));
   uh_rccl_detach(c);
   RcclLink* l = new RcclLink();
   ncclUniqueId nid;
   memcpy(&nid, id, 128);
   const ncclResult_t r = api->CommInitRank(&l->comm, (int)world, nid, (int)rank);
   if (r != ncclSuccess) {
      delete l;
      return fail(c, UH_ERR_HIP, std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(r) : "failed"));
   }
   l->rank = rank;
   l->world = world;
   c->rccl = l;
   return uh_set_restir_partition(c, rank, world, rccl_exchange, l);
}

int uh_rccl_detach(uh_ctx* c) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!c->rccl) return UH_OK;
   (void)hipSetDevice(c->device);
   (void)sync_all(c);
   RcclLink* l = (RcclLink*)c->rccl;
   if (l->comm) (void)rccl_api()->CommDestroy(l->comm);
   delete l;
   c->rccl = nullptr;
   if (c->rp_exchange == rccl_exchange) {
      c->rp_exchange = nullptr;
      c->rp_user = nullptr;
   }
   return UH_OK;
}

int uh_rccl_comm_count(uh_ctx* c, uint32_t* out_ranks) {
   if (!c || !out_ranks) return UH_ERR_INVALID_ARGUMENT;
   *out_ranks = 0;
   if (!c->rccl) return UH_OK;
   RcclLink* l = (RcclLink*)c->rccl;
   int n = 0;
   if (!rccl_api()->CommCount || rccl_api()->CommCount(l->comm, &n) != ncclSuccess) return fail(c, UH_ERR_HIP, "ncclCommCount failed");
   *out_ranks = (uint32_t)n;
   return UH_OK;
}

// ---- composition of a tile-partitioned frame, ENQUEUED: nothing below waits on the host ----
namespace {
uint64_t max_pack_pixels(uh_ctx* c) {
   uint64_t need = 0;
   for (uint32_t r = 0; r < c->tp_world; r++) {
      uint64_t n = 0;
      uh_tile_pack_count(c, r, &n);
      need = n > need ? n : need;
   }
   return need;
}
// what was just enqueued on the context's stream read or wrote the accumulation image: the next frame's accumulate tail
// (enqueue_path_trace) and the next composition wait for it
int mark_composed(uh_ctx* c) {
   if (!c->ev_compose) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_compose, hipEventDisableTiming));
   HIP_TRY(c, hipEventRecord(c->ev_compose, c->stream));
   c->last_acc = c->ev_compose;
   return UH_OK;
}
}  // namespace

// this context's owned tiles, packed into `device_out` (>= uh_tile_pack_count pixels), on the context's stream behind the accumulate
// tail of every frame in flight
int uhi_enqueue_pack_tiles(uh_ctx* c, void* device_out, void** out_stream) {
   if (!c || !device_out) return UH_ERR_INVALID_ARGUMENT;
   HIP_TRY(c, hipSetDevice(c->device));
   if (c->last_acc) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->last_acc, 0));
   launch_pack_tiles(cfg(c), c->accumulation.p, (float4*)device_out, c->W, c->H, c->tp_rank, c->tp_world, c->tp_tile);
   HIP_TRY(c, hipGetLastError());
   if (out_stream) *out_stream = (void*)c->stream;
   return mark_composed(c);
}
// the root's composition (k_compose_tiles over `device_all`: world buffers of stride_pixels, uh_compose_tiles' layout) on the context's
// stream, behind its own frames in flight and behind the `n_waits` events (hipEvent_t) that say the other ranks' tiles have landed
int uhi_enqueue_compose_tiles(uh_ctx* c, const void* device_all, uint64_t stride_pixels, uint32_t total_samples, uint32_t accumulation_limit, void* const* wait_events,
                              int n_waits) {
   if (!c || !device_all || stride_pixels < max_pack_pixels(c)) return UH_ERR_INVALID_ARGUMENT;
   HIP_TRY(c, hipSetDevice(c->device));
   if (c->last_acc) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->last_acc, 0));
   for (int k = 0; k < n_waits; k++)
      if (wait_events[k]) HIP_TRY(c, hipStreamWaitEvent(c->stream, (hipEvent_t)wait_events[k], 0));
   launch_compose_tiles(cfg(c), c->im, (const float4*)device_all, stride_pixels, c->W, c->H, c->tp_rank, c->tp_world, c->tp_tile, total_samples, accumulation_limit);
   HIP_TRY(c, hipGetLastError());
   return mark_composed(c);
}
// the event behind this context's last pack / composition (hipEvent_t; null before the first)
void* uhi_composed_event(uh_ctx* c) { return c ? (void*)c->ev_compose : nullptr; }

// One process per GPU: the ONE collective of the path tracer's data path. Every rank packs its tiles of
// pt_accumulation_image and sends them to `root` over the communicator uh_rccl_attach made (grouped ncclSend / ncclRecv: the peers'
// buffers arrive on distinct xGMI links at once; the root's own tiles take the same way, a local copy); the root scatters them and
// recomputes pt_output_image in one launch (renderers/mod.rs:199-214,354-358: the two images a single device holds). All of it on
// the context's stream, behind the frames in flight; the call returns at once and the next uh_read_* / uh_synchronize waits.
int uh_rccl_gather_tiles(uh_ctx* c, uint32_t root, uint32_t total_samples, uint32_t accumulation_limit) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   RcclLink* l = (RcclLink*)c->rccl;
   if (!l) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_rccl_gather_tiles: no communicator attached (uh_rccl_attach)");
   if (c->tp_world != l->world || c->tp_rank != l->rank)
      return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_rccl_gather_tiles: uh_set_tile_partition(rank, world) must be the communicator's rank and size");
   if (root >= l->world) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_rccl_gather_tiles: root >= world");
   HIP_TRY(c, hipSetDevice(c->device));
   const uint64_t stride = max_pack_pixels(c);
   const bool is_root = l->rank == root;
   if (c->tile_send.n < stride || (is_root && c->tile_recv.n < stride * l->world)) {
      if (int st = sync_all(c)) return st;  // (first call, or another partition: the old buffers may still be read)
      HIP_TRY(c, c->tile_send.alloc(stride ? stride : 1));
      if (is_root) HIP_TRY(c, c->tile_recv.alloc((stride ? stride : 1) * l->world));
   }
   if (int st = uhi_enqueue_pack_tiles(c, c->tile_send.p, nullptr)) return st;
   RcclApi* api = rccl_api();
   ncclResult_t r = api->GroupStart();
   uint64_t mine = 0;
   uh_tile_pack_count(c, l->rank, &mine);
   if (r == ncclSuccess && mine) r = api->Send(c->tile_send.p, (size_t)mine * 4, ncclFloat, (int)root, l->comm, c->stream);
   if (is_root)
      for (uint32_t k = 0; k < l->world && r == ncclSuccess; k++) {
         uint64_t n = 0;
         uh_tile_pack_count(c, k, &n);
         if (n) r = api->Recv(c->tile_recv.p + (size_t)k * stride, (size_t)n * 4, ncclFloat, (int)k, l->comm, c->stream);
      }
   const ncclResult_t e = api->GroupEnd();
   if (r == ncclSuccess) r = e;
   l->last = (int)r;
   if (r != ncclSuccess) return fail(c, UH_ERR_HIP, std::string("uh_rccl_gather_tiles: ") + (api->GetErrorString ? api->GetErrorString(r) : "RCCL error"));
   if (is_root) return uhi_enqueue_compose_tiles(c, c->tile_recv.p, stride, total_samples, accumulation_limit, nullptr, 0);
   return mark_composed(c);
}

// diagnostics: the grid in use (built on the device) against the host builder on the same raster - see utopian_hip.h
int uh_sun_grid_compare_builders(uh_ctx* c, uint64_t out[8]) {
   if (!c || !out) return UH_ERR_INVALID_ARGUMENT;
   for (int k = 0; k < 8; k++) out[k] = 0;
   HIP_TRY(c, hipSetDevice(c->device));
   if (int st = sync_all(c)) return st;
   if (!c->sun_valid) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_sun_grid_compare_builders: no sun grid in use (" + c->sun_why + ")");
   const uint32_t n = c->scene.num_tris;
   const size_t ncell = (size_t)c->sun_dev.nx * c->sun_dev.ny;
   std::vector<float> packets(12 * (size_t)n);
   HIP_TRY(c, hipMemcpy2D(packets.data(), sizeof(TriPacket), c->d_tris.p, 16 * kTriStride16, sizeof(TriPacket), n, hipMemcpyDeviceToHost));
   std::vector<uint32_t> cells(kSunCellWords * (ncell + 1));
   std::vector<SunGridEntry> entries(c->sun_entries);
   HIP_TRY(c, hipMemcpy(cells.data(), c->d_sun_cells.p, cells.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
   if (!entries.empty()) HIP_TRY(c, hipMemcpy(entries.data(), c->d_sun_entries.p, entries.size() * sizeof(SunGridEntry), hipMemcpyDeviceToHost));
   SunGridParams prm;
   std::memcpy(prm.U, c->sun_dev.U, sizeof(prm.U));
   std::memcpy(prm.V, c->sun_dev.V, sizeof(prm.V));
   std::memcpy(prm.W, c->sun_dev.W, sizeof(prm.W));
   prm.u0 = c->sun_dev.u0;
   prm.v0 = c->sun_dev.v0;
   prm.inv_cell = c->sun_dev.inv_cell;
   prm.nx = c->sun_dev.nx;
   prm.ny = c->sun_dev.ny;
   SunGridLimits lim = c->sun_limits;
   lim.max_mean_list = 1e30;  // the comparison wants the host grid whatever the host builder thinks of its worth
   lim.max_fallback_area = 2.0;
   SunGridHost h;
   int threads = (int)std::thread::hardware_concurrency();
   threads = threads < 1 ? 1 : (threads > 32 ? 32 : threads);
   if (!build_sun_grid(packets.data(), n, c->sun_dir_built, lim, threads, h, &prm)) return fail(c, UH_ERR_INVALID_ARGUMENT, "host builder refused: " + h.why_not);
   out[0] = ncell;
   out[1] = entries.size();
   out[2] = h.entries.size();
   if (h.cell_start.size() != ncell + 1) return fail(c, UH_ERR_INVALID_ARGUMENT, "host grid has another raster");
   std::vector<uint32_t> a, b;
   for (size_t k = 0; k < ncell; k++) {
      const uint32_t d0 = cells[kSunCellWords * k], d1 = cells[kSunCellWords * (k + 1)], h0 = h.cell_start[k], h1 = h.cell_start[k + 1];
      uint32_t hc;
      std::memcpy(&hc, &h.cell_cover[k], 4);
      if (cells[kSunCellWords * k + 1] != hc) out[5]++;  // cover depth: bit for bit
      if (d1 - d0 != h1 - h0 || d0 != h0) {
         out[3]++;  // another list length (or offset)
         continue;
      }
      const uint32_t ix = (uint32_t)(k % prm.nx), iy = (uint32_t)(k / prm.nx), len = d1 - d0;
      const bool walkable = !(ix == 0 || iy == 0 || ix == prm.nx - 1 || iy == prm.ny - 1) && len <= lim.max_walk;
      bool same = true;
      if (walkable) {
         out[6]++;
         for (uint32_t e = 0; e < len && same; e++) same = entries[d0 + e].packet == h.entries[h0 + e].packet && std::memcmp(&entries[d0 + e].wmax, &h.entries[h0 + e].wmax, 4) == 0;
      } else {
         // lists no ray walks are left in arrival order on the device: the same packets, as a set
         a.clear();
         b.clear();
         for (uint32_t e = 0; e < len; e++) {
            a.push_back(entries[d0 + e].packet);
            b.push_back(h.entries[h0 + e].packet);
         }
         std::sort(a.begin(), a.end());
         std::sort(b.begin(), b.end());
         same = a == b;
      }
      if (!same) out[4]++;
   }
   out[7] = (uint64_t)(h.build_ms * 1000.0);  // the host builder's time on this box, microseconds
   return UH_OK;
}

int uh_tile_pack_count(uh_ctx* c, uint32_t rank, uint64_t* out_pixels) {
   if (!c || !out_pixels) return UH_ERR_INVALID_ARGUMENT;
   if (rank >= c->tp_world) return fail(c, UH_ERR_INVALID_ARGUMENT, "rank >= world");
   uint32_t tiles = ((c->W + c->tp_tile - 1) / c->tp_tile) * ((c->H + c->tp_tile - 1) / c->tp_tile);
   uint32_t owned = tiles > rank ? (tiles - rank + c->tp_world - 1) / c->tp_world : 0;
   *out_pixels = (uint64_t)owned * c->tp_tile * c->tp_tile;
   return UH_OK;
}

int uh_pack_tiles(uh_ctx* c, void* device_out, uint64_t capacity_pixels) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   uint64_t need = 0;
   uh_tile_pack_count(c, c->tp_rank, &need);
   if (!device_out || capacity_pixels < need) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_pack_tiles: buffer too small");
   HIP_TRY(c, hipSetDevice(c->device));
   if (int st = sync_all(c)) return st;
   launch_pack_tiles(cfg(c), c->accumulation.p, (float4*)device_out, c->W, c->H, c->tp_rank, c->tp_world, c->tp_tile);
   HIP_TRY(c, hipStreamSynchronize(c->stream));
   return UH_OK;
}

int uh_compose_tiles(uh_ctx* c, const void* device_all, uint64_t stride_pixels, uint32_t total_samples, uint32_t accumulation_limit) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   uint64_t need = 0;
   for (uint32_t r = 0; r < c->tp_world; r++) {
      uint64_t n = 0;
      uh_tile_pack_count(c, r, &n);
      need = n > need ? n : need;
   }
   if (!device_all || stride_pixels < need) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_compose_tiles: stride smaller than a rank's packed tiles");
   HIP_TRY(c, hipSetDevice(c->device));
   if (int st = sync_all(c)) return st;
   launch_compose_tiles(cfg(c), c->im, (const float4*)device_all, stride_pixels, c->W, c->H, c->tp_rank, c->tp_world, c->tp_tile, total_samples, accumulation_limit);
   HIP_TRY(c, hipStreamSynchronize(c->stream));
   return UH_OK;
}

int uh_unpack_tiles(uh_ctx* c, uint32_t from_rank, const void* device_in, uint64_t num_pixels) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   uint64_t need = 0;
   if (uh_tile_pack_count(c, from_rank, &need) != UH_OK) return UH_ERR_INVALID_ARGUMENT;
   if (!device_in || num_pixels < need) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_unpack_tiles: buffer too small");
   HIP_TRY(c, hipSetDevice(c->device));
   if (int st = sync_all(c)) return st;
   launch_unpack_tiles(cfg(c), c->accumulation.p, (const float4*)device_in, c->W, c->H, from_rank, c->tp_world, c->tp_tile);
   HIP_TRY(c, hipStreamSynchronize(c->stream));
   return UH_OK;
}

int uh_resolve_output(uh_ctx* c, uint32_t total_samples, uint32_t accumulation_limit) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   HIP_TRY(c, hipSetDevice(c->device));
   if (int st = sync_all(c)) return st;
   launch_resolve(cfg(c), c->im, c->W, c->H, total_samples, accumulation_limit);
   HIP_TRY(c, hipGetLastError());
   return UH_OK;
}

int uh_device_pointer(uh_ctx* c, int which, void** out) {
   if (!c || !out) return UH_ERR_INVALID_ARGUMENT;
   if (which == 0)
      *out = c->accumulation.p;
   else if (which == 1)
      *out = c->output.p;
   else
      return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_device_pointer: which must be 0 or 1");
   return UH_OK;
}

int uh_mesh_info(uh_ctx* c, uint32_t mesh_index, uint32_t* num_vertices, uint32_t* num_indices) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (mesh_index >= c->meshes.size()) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_mesh_info: bad mesh index");
   if (num_vertices) *num_vertices = (uint32_t)c->meshes[mesh_index].vertices.size();
   if (num_indices) *num_indices = (uint32_t)c->meshes[mesh_index].indices.size();
   return UH_OK;
}

int uh_read_mesh(uh_ctx* c, uint32_t mesh_index, UhVertex* vertices, uint32_t* indices) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (mesh_index >= c->meshes.size()) return fail(c, UH_ERR_INVALID_ARGUMENT, "uh_read_mesh: bad mesh index");
   const HostMesh& m = c->meshes[mesh_index];
   if (vertices && !m.vertices.empty()) std::memcpy(vertices, m.vertices.data(), m.vertices.size() * sizeof(UhVertex));
   if (indices && !m.indices.empty()) std::memcpy(indices, m.indices.data(), m.indices.size() * sizeof(uint32_t));
   return UH_OK;
}

int uh_stream(uh_ctx* c, void** out) {
   if (!c || !out) return UH_ERR_INVALID_ARGUMENT;
   HIP_TRY(c, hipSetDevice(c->device));  // the caller is about to enqueue on it
   *out = (void*)c->stream;
   return UH_OK;
}

}  // extern "C"
