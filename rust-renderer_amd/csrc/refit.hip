// refit.hip — on-device refit of the flattened BVH4 after instance transforms changed.
// Reference: the per-frame TLAS rebuild of utopian/src/raytracing.rs:400-459, requested by
// view.rebuild_tlas (prototype/src/main.rs:392,526). This build has one flattened world-space BVH
// instead of TLAS + BLAS, so "instances moved" means: re-bake the moved triangles to world space and
// recompute every box bottom-up over the unchanged topology. The closest-hit result does not depend
// on the boxes (bvh.h: conservative padding, ties broken by key), so a refitted tree returns exactly
// what a rebuilt one does; only the traversal cost degrades while instances drift apart, and
// uh_build_acceleration restores it.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "node_quant.h"

namespace uh {

namespace {

constexpr uint32_t kBlock = 256;

// world = ((m0*x + m1*y) + m2*z) + m3 per row, identity copied verbatim: the host bake of
// uh_build_acceleration (context.hip), operation for operation
__device__ __forceinline__ void bake(const float* __restrict__ w, bool ident, float x, float y, float z, float* o) {
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

// one thread per triangle packet (leaf order): object-space corners -> world corners + packet
__global__ __launch_bounds__(kBlock) void k_refit_triangles(const float* __restrict__ obj_corners, const RefitMesh* __restrict__ meshes, float4* __restrict__ tris,
                                                            float* __restrict__ world_corners, uint32_t count) {
   uint32_t i = blockIdx.x * kBlock + threadIdx.x;
   if (i >= count) return;
   float4* pk = tris + kTriStride16 * (size_t)i;
   const uint32_t key = __float_as_uint(pk[2].y);
   const RefitMesh m = meshes[key >> 22];
   const float* oc = obj_corners + 9 * (size_t)i;
   float c[9];
   for (int k = 0; k < 3; k++) bake(m.o2w, m.identity != 0, oc[3 * k], oc[3 * k + 1], oc[3 * k + 2], c + 3 * k);
   float* wc = world_corners + 9 * (size_t)i;
   for (int k = 0; k < 9; k++) wc[k] = c[k];
   // TriPacket: v0 | e1x ; e1y e1z e2x e2y ; e2z key pad pad
   pk[0] = make_float4(c[0], c[1], c[2], c[3] - c[0]);
   pk[1] = make_float4(c[4] - c[1], c[5] - c[2], c[6] - c[0], c[7] - c[1]);
   pk[2] = make_float4(c[8] - c[2], __uint_as_float(key), 0.0f, 0.0f);
}

// what a node's children span: slot k's tight box (node children: from node_box, triangle children: from the baked corners), padded
// as bvh_build.cpp padded() does - the slab test must never cull what the triangle test accepts
__device__ __forceinline__ void child_boxes(const float* __restrict__ node_box, const float* __restrict__ world_corners, uint32_t n_tri, uint32_t n_child, uint32_t child_base,
                                            uint32_t tri_base, float lo[4][3], float hi[4][3], float tlo[3], float thi[3]) {
   for (int a = 0; a < 3; a++) {
      tlo[a] = INFINITY;
      thi[a] = -INFINITY;
   }
   for (uint32_t k = 0; k < 4; k++) {
      float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
      if (k >= n_child) {
         for (int a = 0; a < 3; a++) lo[k][a] = hi[k][a] = 0.0f;
         continue;
      }
      if (k < n_tri) {
         const float* wc = world_corners + 9 * (size_t)(tri_base + k);
         for (int v = 0; v < 3; v++)
            for (int a = 0; a < 3; a++) {
               blo[a] = fminf(blo[a], wc[3 * v + a]);
               bhi[a] = fmaxf(bhi[a], wc[3 * v + a]);
            }
      } else {
         const float* b = node_box + 6 * (size_t)(child_base + k - n_tri);
         for (int a = 0; a < 3; a++) {
            blo[a] = b[a];
            bhi[a] = b[3 + a];
         }
      }
      for (int a = 0; a < 3; a++) {
         tlo[a] = fminf(tlo[a], blo[a]);
         thi[a] = fmaxf(thi[a], bhi[a]);
         float pad = 1e-4f + 1e-5f * fmaxf(fabsf(blo[a]), fabsf(bhi[a]));
         lo[k][a] = blo[a] - pad;
         hi[k][a] = bhi[a] + pad;
      }
   }
}

// pass 1, one thread per node of one BFS level, deepest level first: the node's own tight box for its parent
// (bvh.h Node4C; child slots are implicit: triangles tri_base + k below n_tri, nodes child_base + k - n_tri up to n_child, empty slots above)
__global__ __launch_bounds__(kBlock) void k_refit_boxes(const uint4* __restrict__ nodes, float* __restrict__ node_box, const float* __restrict__ world_corners, uint32_t first,
                                                        uint32_t count) {
   uint32_t j = blockIdx.x * kBlock + threadIdx.x;
   if (j >= count) return;
   const uint32_t ni = first + j;
   const uint4* nd = nodes + kNodeStride16 * (size_t)ni;
   const uint32_t meta = nd[0].w;
   const uint4 w2 = nd[2];
   const uint32_t n_tri = (meta >> kMetaTriShift) & 7u, n_child = (meta >> kMetaChildShift) & 7u;
   float lo[4][3], hi[4][3], tlo[3], thi[3];
   child_boxes(node_box, world_corners, n_tri, n_child, w2.z & kChildBaseMask, w2.w, lo, hi, tlo, thi);
   float* nb = node_box + 6 * (size_t)ni;
   for (int a = 0; a < 3; a++) {
      nb[a] = tlo[a];
      nb[3 + a] = thi[a];
   }
}

// pass 2, one thread per node of one BFS level, root level first: the node takes its own frame (UH_INHERIT_FRAME = 1: the frame is in
// its record already - written by its parent's thread one level up; only the root takes its own), its children's padded boxes are quantised in it exactly as bvh_build.cpp
// quantise_tree does (node_quant.h: the same functions), and the frame every node child inherits goes into that child's record
__global__ __launch_bounds__(kBlock) void k_refit_quantise(uint4* __restrict__ nodes, const float* __restrict__ node_box, const float* __restrict__ world_corners, uint32_t first,
                                                           uint32_t count) {
   uint32_t j = blockIdx.x * kBlock + threadIdx.x;
   if (j >= count) return;
   const uint32_t ni = first + j;
   uint4* nd = nodes + kNodeStride16 * (size_t)ni;
   const uint4 w0 = nd[0], w2 = nd[2];
   const uint32_t meta = w0.w;
   const uint32_t n_tri = (meta >> kMetaTriShift) & 7u, n_child = (meta >> kMetaChildShift) & 7u;
   const uint32_t child_base = w2.z & kChildBaseMask, tri_base = w2.w;
   float lo[4][3], hi[4][3], tlo[3], thi[3];
   child_boxes(node_box, world_corners, n_tri, n_child, child_base, tri_base, lo, hi, tlo, thi);
   float origin[3] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z)};
   uint32_t exps = meta & 0xffffffu;
   if (ni == 0 || !UH_INHERIT_FRAME) qn_own_frame(lo, hi, n_child, origin, exps);
   uint32_t qlo[3], qhi[3], child_exps[4] = {0, 0, 0, 0};
   float child_origin[4][3];
   qn_quantise(origin, exps, lo, hi, n_tri, n_child, qlo, qhi, child_origin, child_exps, UH_INHERIT_FRAME != 0);
   nd[0] = make_uint4(__float_as_uint(origin[0]), __float_as_uint(origin[1]), __float_as_uint(origin[2]), (meta & 0xff000000u) | exps);
   nd[1] = make_uint4(qlo[0], qlo[1], qlo[2], qhi[0]);
   nd[2] = make_uint4(qhi[1], qhi[2], child_base | (n_tri << kChildBaseBits), tri_base);
   for (uint32_t k = n_tri; UH_INHERIT_FRAME && k < n_child && k < 4; k++) {
      uint4* cd = nodes + kNodeStride16 * (size_t)(child_base + k - n_tri);
      const uint32_t cmeta = cd[0].w;  // the child's own counts stay
      cd[0] = make_uint4(__float_as_uint(child_origin[k][0]), __float_as_uint(child_origin[k][1]), __float_as_uint(child_origin[k][2]), (cmeta & 0xff000000u) | child_exps[k]);
   }
}

}  // namespace

void launch_refit(const LaunchCfg& c, const RefitArgs& a) {
   if (a.num_tris == 0) return;
   k_refit_triangles<<<dim3((a.num_tris + kBlock - 1) / kBlock), kBlock, 0, c.stream>>>(a.obj_corners, a.meshes, a.tris, a.world_corners, a.num_tris);
   for (uint32_t l = a.num_levels; l-- > 0;) {  // boxes: leaves to root
      const uint32_t first = a.level_start[l], count = a.level_start[l + 1] - first;
      if (count) k_refit_boxes<<<dim3((count + kBlock - 1) / kBlock), kBlock, 0, c.stream>>>(a.nodes, a.node_box, a.world_corners, first, count);
   }
   for (uint32_t l = 0; l < a.num_levels; l++) {  // frames and planes: root to leaves (a node's frame is inherited from its parent's)
      const uint32_t first = a.level_start[l], count = a.level_start[l + 1] - first;
      if (count) k_refit_quantise<<<dim3((count + kBlock - 1) / kBlock), kBlock, 0, c.stream>>>(a.nodes, a.node_box, a.world_corners, first, count);
   }
}

}  // namespace uh
