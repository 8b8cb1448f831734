// lbvh.hip — on-device build of the flattened BVH4 ("on-device LBVH build + refit").
// Option "device_build" makes uh_build_acceleration take this path instead of the host SAH builder
// (bvh_build.cpp). It replaces the driver's vkCmdBuildAccelerationStructuresKHR on the path
// utopian/src/raytracing.rs:113-217 (BLAS) / :279-398 (TLAS) for scenes whose geometry changes too
// often for a host rebuild.
//
//   1. bake: object-space corners -> world centroid -> 30-bit Morton code in the scene bounds; the sort key is
//      (morton << 32 | triangle) so keys are unique and the payload rides in the low word
//   2. radix sort of the 64-bit keys (hipcub / rocPRIM)
//   3. binary tree over the sorted triangles, one of (LbvhArgs::kind):
//      PLOC (default; Meister & Bittner 2018, "Parallel Locally-Ordered Clustering"): clusters (initially the triangles, in
//      Morton order) look R places (option "ploc_radius") left and right for the neighbour whose union box has the smallest area; mutual
//      nearest neighbours merge into a node; the survivors are compacted in order (device scan) and the round repeats
//      until LbvhArgs::sah_top clusters are left (about 12 rounds for 8 K of 262 K); a binned-SAH tree over those clusters,
//      built on the host in a millisecond or two, is the top of the tree (bvh_build.cpp build_sah_top).
//      Radix tree (Karras 2012: every internal node finds its range and split from the common-prefix lengths, all nodes
//      in parallel): one launch, a Morton-order tree, about a quarter more traversal work per ray.
//   4. collapse to 4-wide nodes, breadth-first, one launch per level: every leaf is one triangle; a node takes the
//      node slots of its node children from the next level's counter and the packet slots of its triangle children
//      from a packet counter, one atomic each, so both groups are contiguous (bvh.h: implicit child addresses)
//   5. boxes + quantisation: the refit kernels (refit.hip) - a device build is "topology here, boxes by refit"
// Both build in milliseconds instead of the host builder's tens to hundreds (DESIGN.md "On-device build"). Hits do not
// depend on the tree (bvh.h), so a device-built scene renders bit for bit what a host-built one does (tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include <cstring>

#include "bvh.h"
#include "device_scan.h"
#include "device_types.h"

namespace uh {

namespace {

constexpr uint32_t kBlock = 256;

__device__ __forceinline__ uint32_t expand10(uint32_t v) {  // 10 bits -> every third bit
   v = (v * 0x00010001u) & 0xFF0000FFu;
   v = (v * 0x00000101u) & 0x0F00F00Fu;
   v = (v * 0x00000011u) & 0xC30C30C3u;
   v = (v * 0x00000005u) & 0x49249249u;
   return v;
}

__global__ __launch_bounds__(kBlock) void k_lbvh_keys(const float* __restrict__ src_corners, const uint32_t* __restrict__ src_keys, const RefitMesh* __restrict__ meshes,
                                                      float3 lo, float3 inv_ext, unsigned long long* __restrict__ keys, uint32_t n) {
   uint32_t i = blockIdx.x * kBlock + threadIdx.x;
   if (i >= n) return;
   const RefitMesh m = meshes[src_keys[i] >> kPrimBits];
   const float* oc = src_corners + 9 * (size_t)i;
   float c[3] = {0.0f, 0.0f, 0.0f};
   for (int k = 0; k < 3; k++) {
      const float x = oc[3 * k], y = oc[3 * k + 1], z = oc[3 * k + 2];
      if (m.identity) {
         c[0] += x;
         c[1] += y;
         c[2] += z;
      } else {
         c[0] += ((m.o2w[0] * x + m.o2w[1] * y) + m.o2w[2] * z) + m.o2w[3];
         c[1] += ((m.o2w[4] * x + m.o2w[5] * y) + m.o2w[6] * z) + m.o2w[7];
         c[2] += ((m.o2w[8] * x + m.o2w[9] * y) + m.o2w[10] * z) + m.o2w[11];
      }
   }
   auto q = [](float v, float l, float s) {
      float t = (v * (1.0f / 3.0f) - l) * s;
      t = fminf(fmaxf(t, 0.0f), 1.0f);  // NaN -> 0
      return (uint32_t)fminf(t * 1024.0f, 1023.0f);
   };
   const uint32_t morton = (expand10(q(c[0], lo.x, inv_ext.x)) << 2) | (expand10(q(c[1], lo.y, inv_ext.y)) << 1) | expand10(q(c[2], lo.z, inv_ext.z));
   keys[i] = ((unsigned long long)morton << 32) | i;
}

// common-prefix length of sorted keys i and j; -1 outside the array
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ k, int n, int i, int j) {
   if (j < 0 || j >= n) return -1;
   return __clzll((long long)(k[i] ^ k[j]));
}

// internal node i of the binary radix tree over n >= 2 sorted unique keys (Karras 2012, section 3)
__global__ __launch_bounds__(kBlock) void k_lbvh_tree(const unsigned long long* __restrict__ keys, uint4* __restrict__ node2, uint32_t n) {
   const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
   const int N = (int)n;
   if (i >= N - 1) return;
   const int d = (delta(keys, N, i, i + 1) - delta(keys, N, i, i - 1)) >= 0 ? 1 : -1;
   const int dmin = delta(keys, N, i, i - d);
   int lmax = 2;
   while (delta(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
   int l = 0;
   for (int t = lmax / 2; t >= 1; t /= 2)
      if (delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
   const int j = i + l * d;
   const int dnode = delta(keys, N, i, j);
   int s = 0, t = l;
   do {
      t = (t + 1) / 2;
      if (delta(keys, N, i, i + (s + t) * d) > dnode) s += t;
   } while (t > 1);
   const int gamma = i + s * d + (d < 0 ? d : 0);
   const int first = i < j ? i : j, last = i < j ? j : i;
   const uint32_t left = (first == gamma) ? (kLeafBit | (uint32_t)gamma) : (uint32_t)gamma;
   const uint32_t right = (last == gamma + 1) ? (kLeafBit | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
   node2[i] = make_uint4(left, right, (uint32_t)first, (uint32_t)(last - first + 1));
}

// ---- PLOC ------------------------------------------------------------------------------------------------------------
constexpr int kPlocMaxRadius = 64;

struct Box6 {
   float lo[3], hi[3];
};
__device__ __forceinline__ float half_area_union(const Box6& a, const Box6& b) {
   const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]), dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]), dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
   return dx * dy + dy * dz + dz * dx;
}

// world-space box of the triangle at sorted position i (same bake as k_lbvh_keys / refit.hip); cluster i = leaf i
__global__ __launch_bounds__(kBlock) void k_ploc_leaves(const unsigned long long* __restrict__ keys, const float* __restrict__ src_corners, const uint32_t* __restrict__ src_keys,
                                                        const RefitMesh* __restrict__ meshes, Box6* __restrict__ box, uint32_t* __restrict__ cid, uint32_t n) {
   uint32_t i = blockIdx.x * kBlock + threadIdx.x;
   if (i >= n) return;
   const uint32_t s = (uint32_t)(keys[i] & 0xffffffffull);
   const RefitMesh m = meshes[src_keys[s] >> kPrimBits];
   const float* oc = src_corners + 9 * (size_t)s;
   Box6 b;
   for (int a = 0; a < 3; a++) {
      b.lo[a] = INFINITY;
      b.hi[a] = -INFINITY;
   }
   for (int k = 0; k < 3; k++) {
      const float x = oc[3 * k], y = oc[3 * k + 1], z = oc[3 * k + 2];
      float w[3] = {x, y, z};
      if (!m.identity) {
         w[0] = ((m.o2w[0] * x + m.o2w[1] * y) + m.o2w[2] * z) + m.o2w[3];
         w[1] = ((m.o2w[4] * x + m.o2w[5] * y) + m.o2w[6] * z) + m.o2w[7];
         w[2] = ((m.o2w[8] * x + m.o2w[9] * y) + m.o2w[10] * z) + m.o2w[11];
      }
      for (int a = 0; a < 3; a++) {
         // a non-finite corner (never hit) must not poison the area comparisons: it contributes nothing to the box
         if (!(fabsf(w[a]) < 1e30f)) w[a] = 0.0f;
         b.lo[a] = fminf(b.lo[a], w[a]);
         b.hi[a] = fmaxf(b.hi[a], w[a]);
      }
   }
   box[i] = b;
   cid[i] = kLeafBit | i;
}

// nearest neighbour of every cluster within kPlocRadius places (ties: the smaller index)
__global__ __launch_bounds__(kBlock) void k_ploc_nearest(const Box6* __restrict__ box, uint32_t* __restrict__ nn, uint32_t m, int kPlocRadius) {
   __shared__ Box6 tile[kBlock + 2 * kPlocMaxRadius];
   const int base = (int)(blockIdx.x * kBlock) - kPlocRadius;
   for (int k = threadIdx.x; k < (int)kBlock + 2 * kPlocRadius; k += kBlock) {
      const int g = base + k;
      if (g >= 0 && g < (int)m) tile[k] = box[g];
   }
   __syncthreads();
   const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
   if (i >= m) return;
   const Box6 me = tile[threadIdx.x + kPlocRadius];
   float best = INFINITY;
   uint32_t bj = i;  // a lone cluster (m == 1) points at itself
   for (int d = -kPlocRadius; d <= kPlocRadius; d++) {
      const int j = (int)i + d;
      if (d == 0 || j < 0 || j >= (int)m) continue;
      const float a = half_area_union(me, tile[threadIdx.x + kPlocRadius + d]);
      if (a < best) {
         best = a;
         bj = (uint32_t)j;
      }
   }
   nn[i] = bj;
}

// mutual nearest neighbours merge: the lower index keeps the place (new node), the higher one leaves
__global__ __launch_bounds__(kBlock) void k_ploc_merge(const uint32_t* __restrict__ nn, Box6* __restrict__ box, uint32_t* __restrict__ cid, uint32_t* __restrict__ keep,
                                                       uint4* __restrict__ node2, uint32_t* __restrict__ node_count, uint32_t m) {
   const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
   if (i >= m) return;
   const uint32_t j = nn[i];
   uint32_t k = 1;
   if (j != i && nn[j] == i) {
      if (i < j) {
         const Box6 a = box[i], b = box[j];
         Box6 u;
         for (int x = 0; x < 3; x++) {
            u.lo[x] = fminf(a.lo[x], b.lo[x]);
            u.hi[x] = fmaxf(a.hi[x], b.hi[x]);
         }
         const uint32_t id = atomicAdd(node_count, 1u);
         const uint32_t l = cid[i], r = cid[j];
         const uint32_t cnt = ((l & kLeafBit) ? 1u : node2[l].w) + ((r & kLeafBit) ? 1u : node2[r].w);
         node2[id] = make_uint4(l, r, __float_as_uint(half_area_union(a, b)), cnt);
         box[i] = u;  // read by nobody else in this launch: j's thread only tests nn[] and leaves
         cid[i] = id;
      } else {
         k = 0;
      }
   }
   keep[i] = k;
}

__global__ __launch_bounds__(kBlock) void k_ploc_compact(const uint32_t* __restrict__ keep, const uint32_t* __restrict__ offset, const Box6* __restrict__ box_in,
                                                         const uint32_t* __restrict__ cid_in, Box6* __restrict__ box_out, uint32_t* __restrict__ cid_out, uint32_t m) {
   const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
   if (i >= m || !keep[i]) return;
   box_out[offset[i]] = box_in[i];
   cid_out[offset[i]] = cid_in[i];
}

// an unfitted node: boxes and step exponents come from the refit kernels; counts and bases are final
__device__ __forceinline__ void write_topology(uint4* nd, uint32_t n_tri, uint32_t n_child, uint32_t child_base, uint32_t tri_base) {
   const uint32_t meta = (127u | (127u << 8) | (127u << 16)) | (n_tri << kMetaTriShift) | (n_child << kMetaChildShift);
   nd[0] = make_uint4(0u, 0u, 0u, meta);
   nd[1] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0u);
   nd[2] = make_uint4(0u, 0u, (child_base & kChildBaseMask) | (n_tri << kChildBaseBits), tri_base);  // (bvh.h: the traversal reads n_tri here)
}

// one BFS level of the 4-wide tree: level node `idx` (global index level_first + idx) collapses the
// binary subtree rooted at internal node src[idx]
__global__ __launch_bounds__(kBlock) void k_lbvh_collapse(const uint4* __restrict__ node2, const uint32_t* __restrict__ src, uint32_t level_first, uint32_t level_count,
                                                          uint32_t next_first, uint32_t* __restrict__ next_src, uint32_t* __restrict__ next_count, uint32_t* __restrict__ tri_count,
                                                          uint32_t* __restrict__ order, uint4* __restrict__ nodes, bool by_area) {
   const uint32_t idx = blockIdx.x * kBlock + threadIdx.x;
   if (idx >= level_count) return;
   // open the node child with the largest box (PLOC: node2.z = half area as float bits) or the most triangles (radix tree: .z = 0)
   auto weight_of = [&](uint32_t ref) { return (ref & kLeafBit) ? 0.0f : (by_area ? __uint_as_float(node2[ref].z) : (float)node2[ref].w); };
   const uint4 root = node2[src[idx]];
   uint32_t ch[4] = {root.x, root.y, kEmptyRef, kEmptyRef};
   int nc = 2;
   while (nc < 4) {
      int pick = -1;
      float best = -1.0f;
      for (int k = 0; k < nc; k++) {
         if (ch[k] & kLeafBit) continue;
         const float c = weight_of(ch[k]);
         if (c > best) {
            best = c;
            pick = k;
         }
      }
      if (pick < 0) break;
      const uint4 e = node2[ch[pick]];
      ch[pick] = e.x;
      ch[nc++] = e.y;
   }
   // node children in ascending weight (area, or triangle count for the radix tree): a visibility walk takes them from the
   // highest slot down, biggest subtree first (kernels.hip node_compute, bvh_build.cpp)
   uint32_t nd[4];
   uint32_t n_node = 0, n_tri = 0;
   for (int k = 0; k < nc; k++) {
      if (ch[k] & kLeafBit) {
         n_tri++;
         continue;
      }
      uint32_t j = n_node++;
      for (; j > 0 && weight_of(nd[j - 1]) > weight_of(ch[k]); j--) nd[j] = nd[j - 1];
      nd[j] = ch[k];
   }
   const uint32_t tri_base = n_tri ? atomicAdd(tri_count, n_tri) : 0u;
   const uint32_t slot = n_node ? atomicAdd(next_count, n_node) : 0u;
   uint32_t t = 0;
   for (int k = 0; k < nc; k++)
      if (ch[k] & kLeafBit) order[tri_base + t++] = ch[k] & ~kLeafBit;  // position in the sorted order of the packet's triangle
   for (uint32_t m = 0; m < n_node; m++) next_src[slot + m] = nd[m];
   write_topology(nodes + kNodeStride16 * (size_t)(level_first + idx), n_tri, (uint32_t)nc, next_first + slot, tri_base);
}

// a tree without internal binary nodes (n <= 1): one root whose only child, if any, is the triangle
__global__ void k_lbvh_tiny(uint4* __restrict__ nodes, uint32_t* __restrict__ order, uint32_t n) {
   write_topology(nodes, n, n, 0u, 0u);
   if (n) order[0] = 0u;
}

// packet p takes the triangle at position order[p] of the sorted order: object-space corners (refit input), key, shading packet
__global__ __launch_bounds__(kBlock) void k_lbvh_gather(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ order, const float* __restrict__ src_corners,
                                                        const uint32_t* __restrict__ src_keys, const float4* __restrict__ src_shade, float* __restrict__ obj_corners,
                                                        float4* __restrict__ tris, float4* __restrict__ shade, uint32_t n) {
   uint32_t i = blockIdx.x * kBlock + threadIdx.x;
   if (i >= n) return;
   const uint32_t s = (uint32_t)(keys[order[i]] & 0xffffffffull);
   for (int k = 0; k < 9; k++) obj_corners[9 * (size_t)i + k] = src_corners[9 * (size_t)s + k];
   for (int k = 0; k < 4; k++) shade[4 * (size_t)i + k] = src_shade[4 * (size_t)s + k];
   tris[kTriStride16 * (size_t)i + 2] = make_float4(0.0f, __uint_as_float(src_keys[s]), 0.0f, 0.0f);  // the key; refit writes the rest
}

}  // namespace

// Builds topology + packets on `stream`. `level_start` (host) receives the BFS levels. Returns the node count.
// Scratch (keys x2, sort temp, binary nodes, level lists, packet order, counters) is allocated and freed here.
hipError_t lbvh_build(const LbvhArgs& a, hipStream_t stream, std::vector<uint32_t>& level_start, uint32_t* out_nodes) {
   const uint32_t n = a.num_tris;
   level_start.assign({0u, 1u});
   *out_nodes = 1;
   unsigned long long *keys_in = nullptr, *keys_out = nullptr;
   uint4* node2 = nullptr;
   uint32_t *list_a = nullptr, *list_b = nullptr, *counter = nullptr, *order = nullptr;
   Box6 *box_a = nullptr, *box_b = nullptr;
   uint32_t *cid_a = nullptr, *cid_b = nullptr, *nn = nullptr, *keep = nullptr, *offsets = nullptr, *scan_chunks = nullptr;
   unsigned long long* scan_total = nullptr;
   void* temp = nullptr;
   size_t temp_bytes = 0;
   hipError_t e = hipSuccess;
   auto cleanup = [&]() {
      for (void* p : {(void*)keys_in, (void*)keys_out, (void*)node2, (void*)list_a, (void*)list_b, (void*)counter, (void*)order, temp, (void*)box_a, (void*)box_b, (void*)cid_a,
                      (void*)cid_b, (void*)nn, (void*)keep, (void*)offsets, (void*)scan_chunks, (void*)scan_total})
         if (p) (void)hipFree(p);
   };
#define LB_TRY(expr)            \
   if ((e = (expr)) != hipSuccess) { \
      cleanup();                \
      return e;                 \
   }
   const size_t n1 = n ? n : 1;
   LB_TRY(hipMalloc(&order, n1 * sizeof(uint32_t)));
   LB_TRY(hipMalloc(&keys_out, n1 * sizeof(unsigned long long)));
   if (n <= 1) {
      k_lbvh_tiny<<<1, 1, 0, stream>>>(a.nodes, order, n);
      if (n) {
         const unsigned long long zero_key = 0ull;  // the one triangle, source index 0
         LB_TRY(hipMemcpyAsync(keys_out, &zero_key, sizeof(zero_key), hipMemcpyHostToDevice, stream));
         k_lbvh_gather<<<1, kBlock, 0, stream>>>(keys_out, order, a.src_corners, a.src_keys, a.src_shade, a.obj_corners, a.tris, a.shade, n);
      }
      e = hipStreamSynchronize(stream);
      cleanup();
      return e;
   }
   LB_TRY(hipMalloc(&keys_in, n * sizeof(unsigned long long)));
   LB_TRY(hipMalloc(&node2, (size_t)n * sizeof(uint4)));
   LB_TRY(hipMalloc(&list_a, (size_t)n * sizeof(uint32_t)));
   LB_TRY(hipMalloc(&list_b, (size_t)n * sizeof(uint32_t)));
   LB_TRY(hipMalloc(&counter, 2 * sizeof(uint32_t)));  // [0] node slots of the next level, [1] packet slots (whole tree)
   const dim3 grid((n + kBlock - 1) / kBlock);
   const float3 lo = make_float3(a.bounds_lo[0], a.bounds_lo[1], a.bounds_lo[2]);
   auto inv = [](float l, float h) { return h > l ? 1.0f / (h - l) : 0.0f; };
   const float3 inv_ext = make_float3(inv(a.bounds_lo[0], a.bounds_hi[0]), inv(a.bounds_lo[1], a.bounds_hi[1]), inv(a.bounds_lo[2], a.bounds_hi[2]));
   k_lbvh_keys<<<grid, kBlock, 0, stream>>>(a.src_corners, a.src_keys, a.meshes, lo, inv_ext, keys_in, n);
   LB_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, keys_in, keys_out, (int)n, 0, 62, stream));
   LB_TRY(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
   LB_TRY(hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, keys_in, keys_out, (int)n, 0, 62, stream));
   uint32_t binary_root = 0;
   const bool by_area = a.kind != 2;
   if (by_area) {
      // PLOC rounds: nearest neighbour, merge, compact (cluster arrays ping-pong), until one cluster is left
      LB_TRY(hipMalloc(&box_a, (size_t)n * sizeof(Box6)));
      LB_TRY(hipMalloc(&box_b, (size_t)n * sizeof(Box6)));
      LB_TRY(hipMalloc(&cid_a, (size_t)n * sizeof(uint32_t)));
      LB_TRY(hipMalloc(&cid_b, (size_t)n * sizeof(uint32_t)));
      LB_TRY(hipMalloc(&nn, (size_t)n * sizeof(uint32_t)));
      LB_TRY(hipMalloc(&keep, (size_t)n * sizeof(uint32_t)));
      LB_TRY(hipMalloc(&offsets, (size_t)n * sizeof(uint32_t)));
      LB_TRY(hipMalloc(&scan_chunks, (size_t)scan_chunk_count(n) * sizeof(uint32_t)));
      LB_TRY(hipMalloc(&scan_total, sizeof(unsigned long long)));
      LB_TRY(hipMemsetAsync(counter, 0, 2 * sizeof(uint32_t), stream));
      k_ploc_leaves<<<grid, kBlock, 0, stream>>>(keys_out, a.src_corners, a.src_keys, a.meshes, box_a, cid_a, n);
      uint32_t m = n;
      const int radius = a.ploc_radius < 1 ? 1 : (a.ploc_radius > (uint32_t)kPlocMaxRadius ? kPlocMaxRadius : (int)a.ploc_radius);
      Box6 *bi = box_a, *bo = box_b;
      uint32_t *ci = cid_a, *co = cid_b;
      const uint32_t stop_at = a.sah_top > 1 ? a.sah_top : 1;
      for (int round = 0; m > stop_at; round++) {
         if (round > 4096) {
            cleanup();
            return hipErrorUnknown;  // every round merges at least one pair (the globally closest pair is mutual): unreachable
         }
         const dim3 g((m + kBlock - 1) / kBlock);
         k_ploc_nearest<<<g, kBlock, 0, stream>>>(bi, nn, m, radius);
         k_ploc_merge<<<g, kBlock, 0, stream>>>(nn, bi, ci, keep, node2, counter, m);
         LB_TRY(hipMemcpyAsync(offsets, keep, (size_t)m * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
         device_exclusive_scan_u32(offsets, m, scan_chunks, scan_total, stream);
         k_ploc_compact<<<g, kBlock, 0, stream>>>(keep, offsets, bi, ci, bo, co, m);
         unsigned long long kept = 0;
         LB_TRY(hipStreamSynchronize(stream));
         LB_TRY(hipMemcpy(&kept, scan_total, sizeof(kept), hipMemcpyDeviceToHost));  // blocking, behind the wait: an asynchronous copy into a local can land after the wait (context.hip read_back)
         if (kept == 0 || kept >= m) {
            cleanup();
            return hipErrorUnknown;
         }
         m = (uint32_t)kept;
         std::swap(bi, bo);
         std::swap(ci, co);
      }
      if (m > 1) {
         // SAH top: the surviving clusters' boxes come to the host, a binned-SAH binary tree over them (thousands of boxes:
         // a millisecond or two) goes back as binary nodes behind the ones the rounds made. Agglomeration is good at the
         // bottom of the tree and greedy at the top, where a cluster only sees its 2R neighbours in Morton order.
         std::vector<Box6> hb(m);
         std::vector<uint32_t> hc(m);
         uint32_t made = 0;
         LB_TRY(hipStreamSynchronize(stream));
         LB_TRY(hipMemcpy(hb.data(), bi, (size_t)m * sizeof(Box6), hipMemcpyDeviceToHost));
         LB_TRY(hipMemcpy(hc.data(), ci, (size_t)m * sizeof(uint32_t), hipMemcpyDeviceToHost));
         LB_TRY(hipMemcpy(&made, counter, sizeof(uint32_t), hipMemcpyDeviceToHost));
         std::vector<TopNode> top;
         build_sah_top(&hb[0].lo[0], m, top);
         if (top.size() != (size_t)m - 1 || (size_t)made + top.size() > (size_t)n) {
            cleanup();
            return hipErrorUnknown;
         }
         std::vector<uint4> up(top.size());
         auto ref = [&](uint32_t r) { return (r & kLeafBit) ? hc[r & ~kLeafBit] : made + r; };
         for (size_t k = 0; k < top.size(); k++) {
            uint32_t area_bits;
            std::memcpy(&area_bits, &top[k].half_area, sizeof(area_bits));
            up[k] = make_uint4(ref(top[k].left), ref(top[k].right), area_bits, 0u);
         }
         LB_TRY(hipMemcpyAsync(node2 + made, up.data(), up.size() * sizeof(uint4), hipMemcpyHostToDevice, stream));
         LB_TRY(hipStreamSynchronize(stream));  // `up` leaves scope
         binary_root = made;                    // TopNode 0 is the root
      } else {
         LB_TRY(hipStreamSynchronize(stream));
         LB_TRY(hipMemcpy(&binary_root, ci, sizeof(uint32_t), hipMemcpyDeviceToHost));
      }
   } else {
      k_lbvh_tree<<<grid, kBlock, 0, stream>>>(keys_out, node2, n);
   }
   // breadth-first collapse, one launch per level; the level sizes come back through one counter
   level_start.assign({0u});
   uint32_t level_first = 0, level_count = 1;
   LB_TRY(hipMemcpyAsync(list_a, &binary_root, sizeof(uint32_t), hipMemcpyHostToDevice, stream));  // level 0 = the binary root
   LB_TRY(hipMemsetAsync(counter, 0, 2 * sizeof(uint32_t), stream));
   uint32_t *cur = list_a, *nxt = list_b;
   while (level_count) {
      const uint32_t next_first = level_first + level_count;
      if ((size_t)next_first > (size_t)a.node_capacity) {
         cleanup();
         return hipErrorInvalidValue;
      }
      LB_TRY(hipMemsetAsync(counter, 0, sizeof(uint32_t), stream));
      k_lbvh_collapse<<<dim3((level_count + kBlock - 1) / kBlock), kBlock, 0, stream>>>(node2, cur, level_first, level_count, next_first, nxt, counter, counter + 1, order, a.nodes, by_area);
      uint32_t produced = 0;
      LB_TRY(hipStreamSynchronize(stream));
      LB_TRY(hipMemcpy(&produced, counter, sizeof(uint32_t), hipMemcpyDeviceToHost));
      level_start.push_back(next_first);
      level_first = next_first;
      level_count = produced;
      std::swap(cur, nxt);
   }
   *out_nodes = level_first;
   k_lbvh_gather<<<grid, kBlock, 0, stream>>>(keys_out, order, a.src_corners, a.src_keys, a.src_shade, a.obj_corners, a.tris, a.shade, n);
   LB_TRY(hipStreamSynchronize(stream));
   LB_TRY(hipGetLastError());
#undef LB_TRY
   cleanup();
   return hipSuccess;
}

}  // namespace uh
