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

// one thread per node of one BFS level (deepest level first): tight box of every child, the node's
// own tight box for its parent, and the re-quantised 48-byte node (bvh.h Node4C; child slots are implicit:
// triangles tri_base + k below n_tri, nodes child_base + k - n_tri up to n_child, empty slots above)
__global__ __launch_bounds__(kBlock) void k_refit_level(uint4* __restrict__ nodes, float* __restrict__ node_box, const float* __restrict__ world_corners, uint32_t first,
                                                        uint32_t count) {
   uint32_t j = blockIdx.x * kBlock + threadIdx.x;
   if (j >= count) return;
   const uint32_t ni = first + j;
   uint4* nd = nodes + kNodeStride16 * (size_t)ni;
   const uint32_t meta = nd[0].w;
   const uint4 w2 = nd[2];
   const uint32_t n_tri = (meta >> kMetaTriShift) & 7u, n_child = (meta >> kMetaChildShift) & 7u;
   const uint32_t child_base = w2.z, tri_base = w2.w;
   float lo[4][3], hi[4][3];
   float tlo[3] = {INFINITY, INFINITY, INFINITY}, thi[3] = {-INFINITY, -INFINITY, -INFINITY};
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
         // bvh_build.cpp padded(): the slab test must never cull what the triangle test accepts
         float pad = 1e-4f + 1e-5f * fmaxf(fabsf(blo[a]), fabsf(bhi[a]));
         lo[k][a] = blo[a] - pad;
         hi[k][a] = bhi[a] + pad;
      }
   }
   float* nb = node_box + 6 * (size_t)ni;
   for (int a = 0; a < 3; a++) {
      nb[a] = tlo[a];
      nb[3 + a] = thi[a];
   }
   // quantise exactly as bvh_build.cpp quantise_node does: origin = min lower plane, step = smallest power of two
   // whose 255 steps cover the extent, lower planes round down and upper planes up (in double)
   float origin[3];
   uint32_t qlo[3], qhi[3], exps = 0;
   for (int a = 0; a < 3; a++) {
      double mn = INFINITY, mx = -INFINITY;
      for (uint32_t k = 0; k < n_child; k++) {
         mn = fmin(mn, (double)lo[k][a]);
         mx = fmax(mx, (double)hi[k][a]);
      }
      if (!(mn <= mx)) mn = mx = 0.0;
      const float org = (float)mn;
      const double ext = mx - (double)org;
      int e = -100;
      if (!(ext < 1e38)) {
         e = 120;  // non-finite or overflowing extent (only from non-finite world-space geometry): no search
      } else if (ext > 0) {
         int x;
         double mant = frexp(ext / 255.0, &x);  // ext/255 = mant * 2^x, mant in [0.5, 1)
         e = (mant == 0.5) ? x - 1 : x;
         while (ldexp(255.0, e) < ext) e++;
         if (e < -100) e = -100;
      }
      const double s = ldexp(1.0, e);
      uint32_t wlo = 0, whi = 0;
      for (uint32_t k = 0; k < 4; k++) {
         if (k >= n_child) {
            wlo |= 0xffu << (8 * k);
            continue;
         }
         double a0 = floor(((double)lo[k][a] - (double)org) / s);
         double a1 = ceil(((double)hi[k][a] - (double)org) / s);
         if (a0 < 0) a0 = 0;
         if (a1 > 255) a1 = 255;
         if (a0 > 255) a0 = 255;
         wlo |= (uint32_t)a0 << (8 * k);
         whi |= (uint32_t)a1 << (8 * k);
      }
      origin[a] = org;
      exps |= (uint32_t)(e + 127) << (8 * a);
      qlo[a] = wlo;
      qhi[a] = whi;
   }
   nd[0] = make_uint4(__float_as_uint(origin[0]), __float_as_uint(origin[1]), __float_as_uint(origin[2]), (meta & 0xff000000u) | exps);
   nd[1] = make_uint4(qlo[0], qlo[1], qlo[2], qhi[0]);
   nd[2] = make_uint4(qhi[1], qhi[2], child_base, tri_base);
}

}  // namespace

void launch_refit(const LaunchCfg& c, const RefitArgs& a) {
   if (a.num_tris == 0) return;
   k_refit_triangles<<<dim3((a.num_tris + kBlock - 1) / kBlock), kBlock, 0, c.stream>>>(a.obj_corners, a.meshes, a.tris, a.world_corners, a.num_tris);
   for (uint32_t l = a.num_levels; l-- > 0;) {
      const uint32_t first = a.level_start[l], count = a.level_start[l + 1] - first;
      if (count) k_refit_level<<<dim3((count + kBlock - 1) / kBlock), kBlock, 0, c.stream>>>(a.nodes, a.node_box, a.world_corners, first, count);
   }
}

}  // namespace uh
