// node_quant.h - the quantisation of a BVH4 node's child boxes (bvh.h Node4C), as host + device functions shared by the host
// builder (bvh_build.cpp), the on-device refit (refit.hip, which the device builders end in) and - for the frame a child inherits -
// restated instruction for instruction by the traversal kernels (kernels.hip node_compute).
//
// Round 5 EXPERIMENT (bvh.h UH_INHERIT_FRAME = 1; measured slower on MI355X, off by default - then every node takes qn_own_frame and
// qn_quantise is called with inherit = false): A NODE'S FRAME IS INHERITED. A node's quantisation frame (origin + three power-of-two steps) used to be chosen per node
// from its own children (origin = their lowest planes, steps = the smallest that cover them). Now only the root's is; every other
// node's frame is a fixed function of its parent's frame and of its own quantised box there:
//     origin'[a] = fma(qlo[a], step[a], origin[a])              (float, one rounding: the kernels' v_fma_f32)
//     step'[a]   = step[a] * 2^(bitlength(qhi[a] - qlo[a]) - 8)  (the child's box is qhi - qlo <= 2^k - 1 parent steps wide, so 255 of
//                                                                these steps cover it), exponent byte clamped at 1
// The frame is STILL STORED in the node (Node4C::origin, ::meta - what the builder derived, bit for bit), so the record keeps its 48
// bytes and a reader that knows nothing of this finds what it always found. But a traversal that DESCENDS from a node into its nearest
// child has the parent's frame and the child's planes in registers: it derives the child's frame there and skips the first of the
// child's three 16-byte loads - the vector-memory pipeline of a CU retires about one lane load per clock whatever its size or
// locality, and the traversal kernels sit on it. Only a node reached by a POP loads all three. Three of four
// node visits are descents: about a fifth of the kernel's lane loads go.
// The quantiser keeps the planes conservative as before (lower planes round down, upper planes up, in double) and, for node children,
// widens the quantised box until the frame the child inherits covers the child's own padded box: the float rounding of the fma can
// put origin' half an ulp above the exact plane.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define UH_HD __host__ __device__
#else
#ifndef UH_HD
#define UH_HD
#endif
#endif

namespace uh {

// biased exponent byte of a power-of-two step -> the step
UH_HD inline double qn_step(uint32_t exps, int axis) { return ldexp(1.0, (int)((exps >> (8 * axis)) & 0xffu) - 127); }
UH_HD inline uint32_t qn_bitlength(uint32_t x) {  // 0 for 0, else floor(log2 x) + 1 (x <= 255)
   uint32_t k = 0;
   while (x) {
      k++;
      x >>= 1;
   }
   return k;
}

// the frame of a node taken on its own (the root): origin = the lowest lower plane (rounded down), step = the smallest power of two
// whose 255 steps cover the extent. lo / hi: the children's padded boxes, [child][axis].
UH_HD inline void qn_own_frame(const float lo[4][3], const float hi[4][3], uint32_t n_child, float origin[3], uint32_t& exps) {
   exps = 0;
   for (int a = 0; a < 3; a++) {
      double mn = INFINITY, mx = -INFINITY;
      for (uint32_t k = 0; k < n_child; k++) {
         mn = fmin(mn, (double)lo[k][a]);
         mx = fmax(mx, (double)hi[k][a]);
      }
      if (!(mn <= mx)) mn = mx = 0.0;
      float org = (float)mn;
      if ((double)org > mn) org = nextafterf(org, -INFINITY);
      const double ext = mx - (double)org;
      int e = -100;
      if (!(ext < 1e38)) {
         e = 120;  // non-finite or overflowing extent (only from non-finite world-space geometry): no search
      } else if (ext > 0) {
         int x;
         const double mant = frexp(ext / 255.0, &x);  // ext / 255 = mant * 2^x, mant in [0.5, 1)
         e = (mant == 0.5) ? x - 1 : x;
         while (ldexp(255.0, e) < ext) e++;
         if (e < -100) e = -100;
      }
      origin[a] = org;
      exps |= (uint32_t)(e + 127) << (8 * a);
   }
}

// the frame a node child inherits: from its parent's frame and its quantised box there (one byte per axis each). kernels.hip
// node_compute restates this with v_cvt_f32_ubyte / v_fma_f32 / v_ffbh_u32.
UH_HD inline void qn_inherit(const float origin[3], uint32_t exps, const uint32_t qlo[3], const uint32_t qhi[3], float origin_c[3], uint32_t& exps_c) {
   exps_c = 0;
   for (int a = 0; a < 3; a++) {
      const uint32_t e = (exps >> (8 * a)) & 0xffu;
      const float step = (float)ldexp(1.0, (int)e - 127);
      origin_c[a] = fmaf((float)qlo[a], step, origin[a]);
      const int ec = (int)e + (int)qn_bitlength(qhi[a] >= qlo[a] ? qhi[a] - qlo[a] : 0u) - 8;
      exps_c |= (uint32_t)(ec < 1 ? 1 : ec) << (8 * a);
   }
}

// One node: its children's padded boxes (slots [0, n_tri) triangles, [n_tri, n_child) nodes, the rest empty) against its frame.
// Out: the six plane words (child slot k in byte k; an empty slot is the inverted box 255 / 0) and, for node children, the frame each
// inherits (child_origin[k], child_exps[k]).
UH_HD inline void qn_quantise(const float origin[3], uint32_t exps, const float lo[4][3], const float hi[4][3], uint32_t n_tri, uint32_t n_child, uint32_t qlo_w[3], uint32_t qhi_w[3],
                              float child_origin[4][3], uint32_t child_exps[4], bool inherit) {
   uint32_t ql[4][3], qh[4][3];
   for (uint32_t k = 0; k < 4; k++)
      for (int a = 0; a < 3; a++) {
         if (k >= n_child) {
            ql[k][a] = 255u;
            qh[k][a] = 0u;
            continue;
         }
         const double s = qn_step(exps, a);
         double a0 = floor(((double)lo[k][a] - (double)origin[a]) / s);
         double a1 = ceil(((double)hi[k][a] - (double)origin[a]) / s);
         if (!(a0 >= 0)) a0 = 0;  // (also NaN)
         if (!(a1 <= 255)) a1 = 255;
         if (a0 > 255) a0 = 255;
         if (!(a1 >= 0)) a1 = 0;
         ql[k][a] = (uint32_t)a0;
         qh[k][a] = (uint32_t)a1;
      }
   for (uint32_t k = n_tri; inherit && k < n_child && k < 4; k++) {
      // widen until the inherited frame covers the child's own padded box (every plane of ITS children lies inside it). Ends: each round
      // widens the quantised box by a step, and the box 0..255 inherits this node's own frame shifted by nothing - which covers.
      for (int round = 0; round < 1024; round++) {
         qn_inherit(origin, exps, ql[k], qh[k], child_origin[k], child_exps[k]);
         bool ok = true;
         for (int a = 0; a < 3; a++) {
            const double top = (double)child_origin[k][a] + 255.0 * qn_step(child_exps[k], a);
            if ((double)child_origin[k][a] > (double)lo[k][a] && ql[k][a] > 0) {
               ql[k][a]--;
               ok = false;
            } else if (top < (double)hi[k][a]) {
               if (qh[k][a] < 255)
                  qh[k][a]++;
               else if (ql[k][a] > 0)
                  ql[k][a]--;
               else
                  continue;  // the whole range of this node's frame: nothing wider exists (non-finite geometry)
               ok = false;
            }
         }
         if (ok) break;
      }
   }
   for (int a = 0; a < 3; a++) {
      qlo_w[a] = qhi_w[a] = 0;
      for (uint32_t k = 0; k < 4; k++) {
         qlo_w[a] |= ql[k][a] << (8 * k);
         qhi_w[a] |= qh[k][a] << (8 * k);
      }
   }
}

}  // namespace uh
