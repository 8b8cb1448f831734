// bvh_visits.cpp — CPU study: node visits / triangle tests / loop iterations per ray for tree width 4 and 8 and for different
// child-ordering policies, on the product's host builder (csrc/bvh_build.cpp) and full-precision boxes.
//   input: a binary file written by tools/bvh_visits.py: u32 n_tris, u32 n_rays, 9*n_tris floats (corners), 8*n_rays floats (o, tmin, d, tmax)
// Policies: 0 = full sort by entry distance (push far to near); 1 = nearest child first, the other hits pushed in slot order
// (what kernels.hip does); 2 = as 1 but any-hit (first hit ends the ray).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bvh.h"

using namespace uh;

struct Ray {
   float o[3], tmin, d[3], tmax;
};

static bool tri_hit(const float* c, const Ray& r, float& t_out) {
   const float e1[3] = {c[3] - c[0], c[4] - c[1], c[5] - c[2]}, e2[3] = {c[6] - c[0], c[7] - c[1], c[8] - c[2]};
   const float p[3] = {r.d[1] * e2[2] - r.d[2] * e2[1], r.d[2] * e2[0] - r.d[0] * e2[2], r.d[0] * e2[1] - r.d[1] * e2[0]};
   const float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
   if (det == 0.0f) return false;
   const float inv = 1.0f / det;
   const float tv[3] = {r.o[0] - c[0], r.o[1] - c[1], r.o[2] - c[2]};
   const float u = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) * inv;
   if (!(u >= 0 && u <= 1)) return false;
   const float q[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
   const float v = (r.d[0] * q[0] + r.d[1] * q[1] + r.d[2] * q[2]) * inv;
   if (!(v >= 0 && u + v <= 1)) return false;
   const float t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
   if (!(t > r.tmin)) return false;
   t_out = t;
   return true;
}

int main(int argc, char** argv) {
   if (argc < 2) return 1;
   FILE* f = std::fopen(argv[1], "rb");
   uint32_t nt, nr;
   if (!f || std::fread(&nt, 4, 1, f) != 1 || std::fread(&nr, 4, 1, f) != 1) return 1;
   std::vector<float> corners(9 * (size_t)nt);
   std::vector<Ray> rays(nr);
   if (std::fread(corners.data(), 4, corners.size(), f) != corners.size() || std::fread(rays.data(), sizeof(Ray), nr, f) != nr) return 1;
   std::fclose(f);
   std::vector<uint32_t> keys(nt);
   BuildInput in{corners.data(), keys.data(), nt};
   for (uint32_t width : {4u}) {
      BuildOutput bo;
      build_bvh4(in, bo, 8, false, width);
      double fill = 0;
      for (const NodeW& n : bo.nodes) fill += n.count;
      std::printf("width %u: %zu nodes, %zu levels, %.2f children per node\n", width, bo.nodes.size(), bo.level_start.size() - 1, fill / bo.nodes.size());
      std::vector<float> hint_t(nr, INFINITY);
      for (int policy = 0; policy < 4; policy++) {
         // policy 3 = policy 1 with every ray's search interval cut to its own hit distance beforehand (UH_HINT_SLACK: relative slack,
         // a neighbouring sample's hit instead of the ray's own): what testing a cached triangle first buys a primary ray
         const bool hint_pass = policy == 3;
         const int pol = policy == 3 ? 1 : policy;
         double nodes = 0, tris = 0, pushes = 0, maxsp = 0, node_pushes = 0, pops = 0;
         // the stack by depth: how often a pop (policy 1) finds it this deep, counting all entries and counting node entries only (a stack that
         // carries a node's quantisation frame beside its reference needs the room for the node entries only)
         std::vector<double> depth_all(64, 0.0), depth_nodes(64, 0.0);
         std::vector<uint32_t> iters(nr);
         for (uint32_t ri = 0; ri < nr; ri++) {
            const Ray& r = rays[ri];
            const float idir[3] = {1.0f / r.d[0], 1.0f / r.d[1], 1.0f / r.d[2]};
            float best = r.tmax;
            if (hint_pass && hint_t[ri] < r.tmax) best = hint_t[ri];  // the closest hit of an earlier frame's ray through the same pixel, tested first
            std::vector<uint32_t> st;
            uint32_t cur = 0, it = 0;
            bool done = false;
            while (cur != kEmptyRef && !done) {
               it++;
               if (cur & kLeafBit) {
                  tris++;
                  float t;
                  if (tri_hit(&corners[9 * (size_t)bo.tri_order[cur & ~kLeafBit]], r, t) && t < best) {
                     best = t;
                     if (pol == 2) done = true;
                  }
                  if (st.empty()) break;
                  {
                     size_t nn = 0;
                     for (uint32_t e : st) nn += !(e & kLeafBit);
                     depth_all[std::min<size_t>(st.size(), 63)]++;
                     depth_nodes[std::min<size_t>(nn, 63)]++;
                     pops++;
                  }
                  cur = st.back();
                  st.pop_back();
                  continue;
               }
               nodes++;
               const NodeW& n = bo.nodes[cur];
               float tn[kMaxWidth];
               uint32_t cr[kMaxWidth];
               int nh = 0;
               for (uint32_t k = 0; k < n.count; k++) {
                  float t0 = r.tmin, t1 = best;
                  for (int a = 0; a < 3; a++) {
                     float ta = (n.lo[a][k] - r.o[a]) * idir[a], tb = (n.hi[a][k] - r.o[a]) * idir[a];
                     if (ta > tb) std::swap(ta, tb);
                     t0 = std::fmax(t0, ta);
                     t1 = std::fmin(t1, tb);
                  }
                  if (t0 <= t1) {
                     tn[nh] = t0;
                     cr[nh++] = n.child[k];
                  }
               }
               if (nh == 0) {
                  if (st.empty()) break;
                  {
                     size_t nn = 0;
                     for (uint32_t e : st) nn += !(e & kLeafBit);
                     depth_all[std::min<size_t>(st.size(), 63)]++;
                     depth_nodes[std::min<size_t>(nn, 63)]++;
                     pops++;
                  }
                  cur = st.back();
                  st.pop_back();
                  continue;
               }
               if (pol == 0) {
                  // full sort: nearest next, the others pushed far to near
                  int idx[kMaxWidth];
                  for (int k = 0; k < nh; k++) idx[k] = k;
                  std::sort(idx, idx + nh, [&](int a, int b) { return tn[a] < tn[b]; });
                  for (int k = nh - 1; k >= 1; k--) st.push_back(cr[idx[k]]);
                  cur = cr[idx[0]];
               } else {
                  int near = 0;
                  for (int k = 1; k < nh; k++)
                     if (tn[k] < tn[near]) near = k;
                  for (int k = nh - 1; k >= 0; k--)
                     if (k != near) {
                        st.push_back(cr[k]);
                        node_pushes += !(cr[k] & kLeafBit);
                     }
                  cur = cr[near];
               }
               pushes += nh - 1;
               maxsp = std::fmax(maxsp, (double)st.size());
            }
            iters[ri] = it;
            if (policy == 1) hint_t[ri] = best < r.tmax ? best * (1.0f + (std::getenv("UH_HINT_SLACK") ? (float)std::atof(std::getenv("UH_HINT_SLACK")) : 0.0f)) + 1e-6f : INFINITY;
         }
         double wave_iters = 0, lane_iters = 0;
         for (uint32_t w = 0; w + 64 <= nr; w += 64) {
            uint32_t m = 0;
            for (uint32_t k = 0; k < 64; k++) {
               m = std::max(m, iters[w + k]);
               lane_iters += iters[w + k];
            }
            wave_iters += m;
         }
         std::printf("  policy %d: nodes/ray %.2f tris/ray %.2f iterations/ray %.2f pushes/ray %.2f max stack %.0f | batch-of-64 utilisation %.3f\n", policy, nodes / nr, tris / nr,
                     (nodes + tris) / nr, pushes / nr, maxsp, lane_iters / (64.0 * wave_iters));
         if (pol == 1 && pops > 0) {
            std::printf("    node pushes/ray %.2f, pops/ray %.2f; share of pops that find the stack deeper than d (all entries | node entries):", node_pushes / nr, pops / nr);
            for (int d : {2, 3, 4, 5, 6, 8, 10, 12, 16}) {
               double a = 0, b = 0;
               for (int k = d + 1; k < 64; k++) a += depth_all[k], b += depth_nodes[k];
               std::printf(" d>%d: %.3f|%.3f", d, a / pops, b / pops);
            }
            std::printf("\n");
         }
      }
   }
   return 0;
}
