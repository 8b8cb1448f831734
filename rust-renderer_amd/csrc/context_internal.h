// context_internal.h — the phases of one batch of frames, for the in-process group (mgpu.hip). Not part of the C ABI
// (include/utopian_hip.h): uh_render_frame(s) runs these back to back on one context; a group interleaves them over its
// contexts because, with the reservoir passes partitioned by rows (uh_set_restir_partition), frame f + 1's temporal pass on
// one GPU reads the bands every other GPU wrote in frame f.
#pragma once
#include <cstdint>

#include "utopian_hip.h"

struct uh_batch;
extern "C" {
int uhi_plan_batch(uh_ctx*, uint32_t pass_mask, uint32_t* out_batch);  // frames per wavefront for this pass mask; creates the slots
int uhi_batch_begin(uh_ctx*, const UhViewUniformData*, uint32_t pass_mask, uint32_t batch, uh_batch** out);
int uhi_batch_restir_frame(uh_ctx*, uh_batch*, uint32_t f);  // G-buffer cast + reservoir chain of frame f (this context's rows)
int uhi_batch_end(uh_ctx*, uh_batch*);                       // the path-tracing wavefront; frees the batch
void uhi_batch_abandon(uh_batch*);
// what a peer pulls a band from / into after uhi_batch_restir_frame: the buffer the spatial pass wrote, the event recorded
// behind it (hipEvent_t), the stream the passes run on (hipStream_t), bytes per band
int uhi_exchange_endpoints(uh_ctx*, void** spatial_base, void** band_event, void** stream, uint64_t* band_bytes);
int uhi_iso_reference_triangulation(uh_ctx*);  // option "iso_reference_triangulation" (isosurface.hip)
// composition of a tile-partitioned frame without a host wait (uh_rccl_gather_tiles, uh_mgpu_compose):
// pack this context's tiles into device_out on the context's stream (returned as hipStream_t), behind its frames in flight
int uhi_enqueue_pack_tiles(uh_ctx*, void* device_out, void** out_stream);
// the root: k_compose_tiles over device_all (world buffers of stride_pixels) on the context's stream, behind its frames in flight
// and behind the n_waits events (hipEvent_t) that say the other ranks' tiles have landed
int uhi_enqueue_compose_tiles(uh_ctx*, const void* device_all, uint64_t stride_pixels, uint32_t total_samples, uint32_t accumulation_limit, void* const* wait_events, int n_waits);
void* uhi_composed_event(uh_ctx*);  // hipEvent_t behind the context's last pack / composition (null before the first)
}
