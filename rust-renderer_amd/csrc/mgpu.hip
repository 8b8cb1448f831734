// mgpu.hip — uh_mgpu_*: one application process driving several GPUs.
// The reference application is a single process with one render thread (prototype/src/main.rs); a
// maintainer who wants N MI355X behind it binds this group instead of a single uh_ctx. It is a thin
// layer over the public uh_* calls: one context per GPU, a full scene replica on each, the
// framebuffer split into tiles t % N == i, no data-path exchange per frame, and one gather of packed
// RGBA32F tiles to GPU 0 (hipMemcpyPeer: a direct xGMI copy, 7 peers land on 7 distinct links) per
// COMPOSED image. (bench.py / the driver's scaling run use the one-process-per-GPU + RCCL form of
// the same partition, distributed.py.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "context_internal.h"
#include "utopian_hip.h"

struct uh_mgpu {
   std::vector<uh_ctx*> ctx;
   std::vector<int> device;
   std::vector<void*> packed;  // on device[i]: this GPU's tiles, packed
   void* staged = nullptr;     // on device[0]: every GPU's packed tiles, GPU i's at i * stride_pixels (uh_compose_tiles' layout)
   uint64_t stride_pixels = 0;
   std::vector<void*> landed;  // hipEvent_t on device[i]: GPU i's tiles of the current composition have been copied into `staged`
   std::vector<uint64_t> pack_pixels;
   uint32_t W = 0, H = 0, tile = 0;
   uint32_t total_samples = 1, accumulation_limit = 999999;  // of the last frame, for the resolve
   bool composed = false;
   // the G-buffer cast and the reservoir passes by bands of rows, one band per GPU, the bands exchanged by peer copies after
   // every spatial pass (uh_set_restir_partition; option "restir_partition", on by default for more than one GPU)
   bool restir_partition = false;
   std::string error;
};

namespace {

thread_local std::string g_create_error;

int fail(uh_mgpu* m, int st, const std::string& what, uh_ctx* from = nullptr) {
   if (m) {
      m->error = what;
      if (from) {
         const char* e = uh_last_error(from);
         if (e && *e) m->error += std::string(": ") + e;
      }
   }
   return st;
}

// run f(i) for every GPU concurrently (host-side work such as the BVH build or blocking copies)
template <typename F>
int for_all(uh_mgpu* m, F f) {
   std::vector<int> st(m->ctx.size(), UH_OK);
   std::vector<std::thread> th;
   for (size_t i = 1; i < m->ctx.size(); i++) th.emplace_back([&, i]() { st[i] = f((int)i); });
   st[0] = f(0);
   for (auto& t : th) t.join();
   for (size_t i = 0; i < st.size(); i++)
      if (st[i] != UH_OK) return fail(m, st[i], "GPU " + std::to_string(m->device[i]), m->ctx[i]);
   return UH_OK;
}

}  // namespace

extern "C" {

int uh_mgpu_create(int ngpus, const int* device_ordinals, uint32_t width, uint32_t height, uint32_t tile_size, uh_mgpu** out) {
   if (!out || ngpus < 1 || ngpus > 64 || tile_size == 0) {
      g_create_error = "uh_mgpu_create: bad argument";
      return UH_ERR_INVALID_ARGUMENT;
   }
   *out = nullptr;
   uh_mgpu* m = new uh_mgpu();
   m->W = width;
   m->H = height;
   m->tile = tile_size;
   for (int i = 0; i < ngpus; i++) {
      const int dev = device_ordinals ? device_ordinals[i] : i;
      uh_ctx* c = nullptr;
      int st = uh_create(dev, width, height, &c);
      if (st == UH_OK) st = uh_set_tile_partition(c, (uint32_t)i, (uint32_t)ngpus, tile_size);
      if (st == UH_OK && ngpus > 1) st = uh_set_restir_partition(c, (uint32_t)i, (uint32_t)ngpus, nullptr, nullptr);  // the group moves the bands itself
      if (st != UH_OK) {
         const char* e = uh_last_error(c);
         g_create_error = std::string("uh_mgpu_create: GPU ") + std::to_string(dev) + ": " + (e ? e : "");
         if (c) uh_destroy(c);
         for (uh_ctx* o : m->ctx) uh_destroy(o);
         delete m;
         return st;
      }
      m->ctx.push_back(c);
      m->device.push_back(dev);
   }
   m->restir_partition = ngpus > 1;
   m->packed.assign(ngpus, nullptr);
   m->landed.assign(ngpus, nullptr);
   m->pack_pixels.assign(ngpus, 0);
   for (int i = 0; i < ngpus; i++) {
      uh_tile_pack_count(m->ctx[i], (uint32_t)i, &m->pack_pixels[i]);
      m->stride_pixels = std::max(m->stride_pixels, m->pack_pixels[i]);
   }
   if (ngpus > 1 && m->stride_pixels) {
      hipError_t e = hipSetDevice(m->device[0]);
      if (e == hipSuccess) e = hipMalloc(&m->staged, (size_t)m->stride_pixels * 16 * ngpus);
      if (e != hipSuccess) {
         g_create_error = std::string("uh_mgpu_create: ") + hipGetErrorString(e);
         uh_mgpu_destroy(m);
         return e == hipErrorOutOfMemory ? UH_ERR_OUT_OF_MEMORY : UH_ERR_HIP;
      }
   }
   for (int i = 0; i < ngpus; i++) {
      if (i == 0 || m->pack_pixels[i] == 0) continue;
      const size_t bytes = m->pack_pixels[i] * 16;
      hipError_t e = hipSetDevice(m->device[i]);
      if (e == hipSuccess) e = hipMalloc(&m->packed[i], bytes);
      if (e == hipSuccess) e = hipEventCreateWithFlags((hipEvent_t*)&m->landed[i], hipEventDisableTiming);
      if (e == hipSuccess && m->device[i] != m->device[0]) {
         // direct xGMI copies; "already enabled" / "not supported" leave hipMemcpyPeer to stage by itself
         (void)hipDeviceEnablePeerAccess(m->device[i], 0);
         (void)hipGetLastError();
      }
      if (e != hipSuccess) {
         g_create_error = std::string("uh_mgpu_create: ") + hipGetErrorString(e);
         *out = m;
         uh_mgpu_destroy(m);
         *out = nullptr;
         return e == hipErrorOutOfMemory ? UH_ERR_OUT_OF_MEMORY : UH_ERR_HIP;
      }
   }
   *out = m;
   return UH_OK;
}

void uh_mgpu_destroy(uh_mgpu* m) {
   if (!m) return;
   for (uh_ctx* c : m->ctx) (void)uh_synchronize(c);  // the copies and the composition run on the contexts' streams
   for (size_t i = 0; i < m->ctx.size(); i++) {
      if (i < m->packed.size() && (m->packed[i] || m->landed[i])) {
         (void)hipSetDevice(m->device[i]);
         if (m->packed[i]) (void)hipFree(m->packed[i]);
         if (m->landed[i]) (void)hipEventDestroy((hipEvent_t)m->landed[i]);
      }
   }
   if (m->staged) {
      (void)hipSetDevice(m->device[0]);
      (void)hipFree(m->staged);
   }
   for (uh_ctx* c : m->ctx) uh_destroy(c);
   delete m;
}

const char* uh_mgpu_last_error(uh_mgpu* m) { return m ? m->error.c_str() : g_create_error.c_str(); }
int uh_mgpu_num_devices(uh_mgpu* m) { return m ? (int)m->ctx.size() : 0; }
uh_ctx* uh_mgpu_context(uh_mgpu* m, int index) { return (m && index >= 0 && index < (int)m->ctx.size()) ? m->ctx[index] : nullptr; }

// ---- scene verbs: replicated on every GPU (full scene replica per GPU) ------------
int uh_mgpu_add_texture_rgba8(uh_mgpu* m, const uint8_t* pixels, uint32_t w, uint32_t h, uint32_t* out_index) {
   if (!m) return UH_ERR_INVALID_ARGUMENT;
   for (uh_ctx* c : m->ctx)
      if (int st = uh_add_texture_rgba8(c, pixels, w, h, out_index)) return fail(m, st, "uh_mgpu_add_texture_rgba8", c);
   return UH_OK;
}
int uh_mgpu_add_mesh(uh_mgpu* m, const UhVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, const UhGpuMaterial* mat, const float world3x4[12],
                     uint32_t* out_mesh_index) {
   if (!m) return UH_ERR_INVALID_ARGUMENT;
   for (uh_ctx* c : m->ctx)
      if (int st = uh_add_mesh(c, v, nv, idx, ni, mat, world3x4, out_mesh_index)) return fail(m, st, "uh_mgpu_add_mesh", c);
   return UH_OK;
}
int uh_mgpu_add_light(uh_mgpu* m, const UhGpuLight* light, uint32_t* out_index) {
   if (!m) return UH_ERR_INVALID_ARGUMENT;
   for (uh_ctx* c : m->ctx)
      if (int st = uh_add_light(c, light, out_index)) return fail(m, st, "uh_mgpu_add_light", c);
   return UH_OK;
}
int uh_mgpu_get_num_lights(uh_mgpu* m, uint32_t* out) { return m ? uh_get_num_lights(m->ctx[0], out) : UH_ERR_INVALID_ARGUMENT; }
int uh_mgpu_set_instance_transform(uh_mgpu* m, uint32_t mesh_index, const float world3x4[12]) {
   if (!m) return UH_ERR_INVALID_ARGUMENT;
   for (uh_ctx* c : m->ctx)
      if (int st = uh_set_instance_transform(c, mesh_index, world3x4)) return fail(m, st, "uh_mgpu_set_instance_transform", c);
   return UH_OK;
}
int uh_mgpu_build_acceleration(uh_mgpu* m) {
   if (!m) return UH_ERR_INVALID_ARGUMENT;
   return for_all(m, [&](int i) { return uh_build_acceleration(m->ctx[i]); });
}
int uh_mgpu_refit_acceleration(uh_mgpu* m) {
   if (!m) return UH_ERR_INVALID_ARGUMENT;
   return for_all(m, [&](int i) { return uh_refit_acceleration(m->ctx[i]); });
}
int uh_mgpu_set_option(uh_mgpu* m, const char* name, int value) {
   if (!m) return UH_ERR_INVALID_ARGUMENT;
   if (name && std::string(name) == "restir_partition") {
      // 1 (default for more than one GPU): reservoir passes by bands of rows; 0: every GPU runs them for the whole frame
      const uint32_t n = (uint32_t)m->ctx.size();
      const bool on = value != 0 && n > 1;
      for (uint32_t i = 0; i < n; i++)
         if (int st = uh_set_restir_partition(m->ctx[i], on ? i : 0, on ? n : 1, nullptr, nullptr)) return fail(m, st, "uh_mgpu_set_option", m->ctx[i]);
      m->restir_partition = on;
      return UH_OK;
   }
   for (uh_ctx* c : m->ctx)
      if (int st = uh_set_option(c, name, value)) return fail(m, st, "uh_mgpu_set_option", c);
   return UH_OK;
}

// ---- frames: every GPU enqueues its tiles of the frame; the calls return without waiting ----
namespace {
// after every GPU's spatial pass of one frame: GPU i pulls band j from GPU j, on its own reservoir stream, behind the event
// GPU j recorded after writing it. Nothing waits on the host; frame f + 1's temporal pass of GPU i follows on the same stream.
int exchange_bands(uh_mgpu* m) {
   const size_t n = m->ctx.size();
   std::vector<void*> base(n), ev(n), stream(n);
   std::vector<UhRestirRows> rows(n);
   uint64_t band_bytes = 0;
   for (size_t i = 0; i < n; i++) {
      if (int st = uhi_exchange_endpoints(m->ctx[i], &base[i], &ev[i], &stream[i], &band_bytes)) return fail(m, st, "exchange endpoints", m->ctx[i]);
      uh_get_restir_rows(m->ctx[i], &rows[i]);
   }
   for (size_t i = 0; i < n; i++) {
      hipError_t e = hipSetDevice(m->device[i]);
      for (size_t j = 0; j < n && e == hipSuccess; j++) {
         const size_t bytes = (size_t)rows[j].band_rows * m->W * sizeof(UhReservoir);
         if (j == i || bytes == 0) continue;
         e = hipStreamWaitEvent((hipStream_t)stream[i], (hipEvent_t)ev[j], 0);
         if (e == hipSuccess)
            e = hipMemcpyPeerAsync((char*)base[i] + j * band_bytes, m->device[i], (const char*)base[j] + j * band_bytes, m->device[j], bytes, (hipStream_t)stream[i]);
      }
      if (e != hipSuccess) return fail(m, UH_ERR_HIP, std::string("band exchange: ") + hipGetErrorString(e));
   }
   return UH_OK;
}

int group_frames(uh_mgpu* m, const UhViewUniformData* view, uint32_t pass_mask, uint32_t count, bool batched) {
   const size_t n = m->ctx.size();
   const bool interleave = m->restir_partition && n > 1 && (pass_mask & UH_PASS_RESTIR);
   if (!interleave) {
      for (uh_ctx* c : m->ctx)
         if (int st = batched ? uh_render_frames(c, view, pass_mask, count) : uh_render_frame(c, view, pass_mask)) return fail(m, st, "uh_mgpu_render_frame(s)", c);
      return UH_OK;
   }
   UhViewUniformData v = *view;
   uint32_t batch = 1;
   if (batched)
      for (size_t i = 0; i < n; i++) {
         uint32_t b = 1;
         if (int st = uhi_plan_batch(m->ctx[i], pass_mask, &b)) return fail(m, st, "uh_mgpu_render_frames", m->ctx[i]);
         batch = i == 0 ? b : std::min(batch, b);
      }
   std::vector<uh_batch*> bs(n, nullptr);
   auto abandon = [&]() {
      for (uh_batch*& b : bs) {
         if (b) uhi_batch_abandon(b);
         b = nullptr;
      }
   };
   for (uint32_t done = 0; done < count;) {
      const uint32_t b = std::min(batch, count - done);
      for (size_t i = 0; i < n; i++)
         if (int st = uhi_batch_begin(m->ctx[i], &v, pass_mask, b, &bs[i])) return abandon(), fail(m, st, "uh_mgpu_render_frames", m->ctx[i]);
      for (uint32_t f = 0; f < b; f++) {
         for (size_t i = 0; i < n; i++)
            if (int st = uhi_batch_restir_frame(m->ctx[i], bs[i], f)) return abandon(), fail(m, st, "uh_mgpu_render_frames", m->ctx[i]);
         if (pass_mask & UH_PASS_SPATIAL_REUSE)
            if (int st = exchange_bands(m)) return abandon(), st;
      }
      for (size_t i = 0; i < n; i++) {
         uh_batch* mine = bs[i];
         bs[i] = nullptr;  // uhi_batch_end frees it
         if (int st = uhi_batch_end(m->ctx[i], mine)) return abandon(), fail(m, st, "uh_mgpu_render_frames", m->ctx[i]);
      }
      done += b;
      v.total_samples += b * v.samples_per_frame;
   }
   return UH_OK;
}
}  // namespace

int uh_mgpu_render_frame(uh_mgpu* m, const UhViewUniformData* view, uint32_t pass_mask) {
   if (!m || !view) return UH_ERR_INVALID_ARGUMENT;
   if (int st = group_frames(m, view, pass_mask, 1, false)) return st;
   m->total_samples = view->total_samples;
   m->accumulation_limit = view->accumulation_limit;
   m->composed = false;
   return UH_OK;
}
int uh_mgpu_render_frames(uh_mgpu* m, const UhViewUniformData* view, uint32_t pass_mask, uint32_t count) {
   if (!m || !view || count == 0) return UH_ERR_INVALID_ARGUMENT;
   if (int st = group_frames(m, view, pass_mask, count, true)) return st;
   m->total_samples = view->total_samples + (count - 1) * view->samples_per_frame;
   m->accumulation_limit = view->accumulation_limit;
   m->composed = false;
   return UH_OK;
}
int uh_mgpu_reset_accumulation(uh_mgpu* m) {
   if (!m) return UH_ERR_INVALID_ARGUMENT;
   for (uh_ctx* c : m->ctx)
      if (int st = uh_reset_accumulation(c)) return fail(m, st, "uh_mgpu_reset_accumulation", c);
   m->composed = false;
   return UH_OK;
}
int uh_mgpu_synchronize(uh_mgpu* m) {
   if (!m) return UH_ERR_INVALID_ARGUMENT;
   for (uh_ctx* c : m->ctx)
      if (int st = uh_synchronize(c)) return fail(m, st, "uh_mgpu_synchronize", c);
   return UH_OK;
}

// Gather every GPU's tiles into GPU 0's accumulation image and recompute pt_output_image there - ENQUEUED, every step ordered by
// an event, nothing waits on the host: GPU i packs its tiles on its context's stream (behind its frames in flight), the peer copy
// into GPU 0's staging buffer follows ON THAT STREAM (behind the pack; and behind GPU 0's previous composition, which may still
// read the buffer), an event recorded behind the copy says "landed", and GPU 0's stream waits for every such event before the one
// launch that scatters the tiles and resolves the output. (Round 4's form copied with hipMemcpyPeer on the null stream, which the
// contexts' non-blocking streams do not wait for - stale tiles in 2 % of the scenes when the GPU was shared - and was patched
// with host waits on both devices' null streams: ordering by guesswork.) The read-backs wait for GPU 0's streams as ever.
int uh_mgpu_compose(uh_mgpu* m) {
   if (!m) return UH_ERR_INVALID_ARGUMENT;
   if (m->composed) return UH_OK;
   const size_t n = m->ctx.size();
   if (n == 1) {
      m->composed = true;  // one GPU holds the whole frame: its own tail wrote both images
      return UH_OK;
   }
   void* const reader_done = uhi_composed_event(m->ctx[0]);  // GPU 0's previous composition (or null)
   for (size_t i = 1; i < n; i++) {
      if (!m->pack_pixels[i]) continue;
      void* stream = nullptr;
      if (int s = uhi_enqueue_pack_tiles(m->ctx[i], m->packed[i], &stream)) return fail(m, s, "uh_mgpu_compose: pack", m->ctx[i]);
      hipError_t e = hipSuccess;  // (this thread's device is GPU i: uhi_enqueue_pack_tiles set it)
      if (reader_done) e = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)reader_done, 0);
      if (e == hipSuccess)
         e = hipMemcpyPeerAsync((char*)m->staged + (size_t)i * m->stride_pixels * 16, m->device[0], m->packed[i], m->device[i], m->pack_pixels[i] * 16, (hipStream_t)stream);
      if (e == hipSuccess) e = hipEventRecord((hipEvent_t)m->landed[i], (hipStream_t)stream);
      if (e != hipSuccess) return fail(m, UH_ERR_HIP, std::string("uh_mgpu_compose: peer copy: ") + hipGetErrorString(e));
   }
   if (int s = uhi_enqueue_compose_tiles(m->ctx[0], m->staged, m->stride_pixels, m->total_samples, m->accumulation_limit, m->landed.data(), (int)n))
      return fail(m, s, "uh_mgpu_compose: compose", m->ctx[0]);
   m->composed = true;
   return UH_OK;
}
int uh_mgpu_read_accumulation(uh_mgpu* m, float* rgba32f) {
   if (!m) return UH_ERR_INVALID_ARGUMENT;
   if (int st = uh_mgpu_compose(m)) return st;
   if (int st = uh_read_accumulation(m->ctx[0], rgba32f)) return fail(m, st, "uh_mgpu_read_accumulation", m->ctx[0]);
   return UH_OK;
}
int uh_mgpu_read_output_bgra8(uh_mgpu* m, uint8_t* bgra) {
   if (!m) return UH_ERR_INVALID_ARGUMENT;
   if (int st = uh_mgpu_compose(m)) return st;
   if (int st = uh_read_output_bgra8(m->ctx[0], bgra)) return fail(m, st, "uh_mgpu_read_output_bgra8", m->ctx[0]);
   return UH_OK;
}

// counters summed over the GPUs (each traces only its tiles); times are the slowest GPU's
// reservoir buffers of the whole frame: spatial_reuse_reservoirs (2) is complete on every GPU after the exchange; the
// initial (0) and temporal (1) buffers are assembled from the rows each GPU's band covers
int uh_mgpu_read_reservoirs(uh_mgpu* m, int which, UhReservoir* out) {
   if (!m || !out || which < 0 || which > 2) return UH_ERR_INVALID_ARGUMENT;
   if (!m->restir_partition || which == 2) {
      if (int st = uh_read_reservoirs(m->ctx[0], which, out)) return fail(m, st, "uh_mgpu_read_reservoirs", m->ctx[0]);
      return UH_OK;
   }
   std::vector<UhReservoir> tmp((size_t)m->W * m->H);
   for (uh_ctx* c : m->ctx) {
      UhRestirRows r;
      uh_get_restir_rows(c, &r);
      if (r.band_rows == 0) continue;
      if (int st = uh_read_reservoirs(c, which, tmp.data())) return fail(m, st, "uh_mgpu_read_reservoirs", c);
      memcpy(out + (size_t)r.band_row0 * m->W, tmp.data() + (size_t)r.band_row0 * m->W, (size_t)r.band_rows * m->W * sizeof(UhReservoir));
   }
   return UH_OK;
}
int uh_mgpu_get_stats(uh_mgpu* m, UhStats* out) {
   if (!m || !out) return UH_ERR_INVALID_ARGUMENT;
   std::memset(out, 0, sizeof(*out));
   for (size_t i = 0; i < m->ctx.size(); i++) {
      UhStats s;
      if (int st = uh_get_stats(m->ctx[i], &s)) return fail(m, st, "uh_mgpu_get_stats", m->ctx[i]);
      for (int k = 0; k < UH_RAY_KINDS; k++) out->rays[k] += (k == UH_RAY_GBUFFER && i > 0 && !m->restir_partition) ? 0 : s.rays[k];  // without the row partition the G-buffer cast is replicated: counted once
      out->nodes_visited += s.nodes_visited;
      out->tris_tested += s.tris_tested;
      out->shadow_nodes_visited += s.shadow_nodes_visited;
      out->shadow_tris_tested += s.shadow_tris_tested;
      out->closest_hits += s.closest_hits;
      out->misses += s.misses;
      out->sun_tree_rays += s.sun_tree_rays;
      out->camera_tree_rays += s.camera_tree_rays;
      out->camera_grid_tris_tested += s.camera_grid_tris_tested;
      out->sun_covered_rays += s.sun_covered_rays;
      out->frames = s.frames;
      out->bvh_nodes = s.bvh_nodes;
      out->bvh_triangles = s.bvh_triangles;
      if (s.build_ms > out->build_ms) out->build_ms = s.build_ms;
      if (s.last_frame_ms > out->last_frame_ms) out->last_frame_ms = s.last_frame_ms;
      if (s.trace_closest_ms > out->trace_closest_ms) out->trace_closest_ms = s.trace_closest_ms;
      if (s.trace_shadow_ms > out->trace_shadow_ms) out->trace_shadow_ms = s.trace_shadow_ms;
      if (s.shade_ms > out->shade_ms) out->shade_ms = s.shade_ms;
      if (s.trace_light_ms > out->trace_light_ms) out->trace_light_ms = s.trace_light_ms;
      out->trace_light_launches += s.trace_light_launches;
      out->light_nodes_visited += s.light_nodes_visited;
      out->light_tris_tested += s.light_tris_tested;
      out->trace_closest_launches += s.trace_closest_launches;
      if (i == 0) {  // every GPU builds the same grid
         out->sun_grid_cells = s.sun_grid_cells;
         out->sun_grid_entries = s.sun_grid_entries;
         out->sun_grid_mean_list = s.sun_grid_mean_list;
      }
      if (s.sun_grid_build_ms > out->sun_grid_build_ms) out->sun_grid_build_ms = s.sun_grid_build_ms;
   }
   return UH_OK;
}
int uh_mgpu_reset_stats(uh_mgpu* m) {
   if (!m) return UH_ERR_INVALID_ARGUMENT;
   for (uh_ctx* c : m->ctx)
      if (int st = uh_reset_stats(c)) return fail(m, st, "uh_mgpu_reset_stats", c);
   return UH_OK;
}

}  // extern "C"
