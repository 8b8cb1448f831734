// bvh_check.cpp — host-side invariants of the BVH4 builder (csrc/bvh_build.cpp), built with
// -fsanitize=address,undefined by tests/test_bvh_builder.py. Triangle soups come from a seeded LCG,
// including degenerate cases (zero-area, duplicated, coincident centroids, huge coordinates).
//   every input triangle is the child of exactly one node (a leaf is one triangle); child refs are in range and
//   form a tree (each interior node referenced once); the slots of a node hold its triangles first, then its nodes,
//   then empty slots; the device node's implicit addressing (tri_base + slot, child_base + slot - n_tri) reproduces
//   the explicit refs; every full-precision child box contains its subtree's triangles; every quantised box contains
//   the full-precision box; empty slots are inverted boxes; level_start describes the breadth-first levels.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "bvh.h"
#include "node_quant.h"

using namespace uh;

static uint32_t g_state = 1;
static float rnd() {
   g_state = g_state * 747796405u + 1u;
   uint32_t w = ((g_state >> ((g_state >> 28) + 4u)) ^ g_state) * 277803737u;
   w = (w >> 22) ^ w;
   return (float)w / 4294967296.0f;
}

static int check(const std::vector<float>& corners, int threads, const char* name, bool geometry = true) {
   const uint32_t n = (uint32_t)(corners.size() / 9);
   std::vector<uint32_t> keys(n);
   for (uint32_t i = 0; i < n; i++) keys[i] = i;
   BuildInput in{corners.data(), keys.data(), n};
   BuildOutput out;
   build_bvh4(in, out, threads);
   int errors = 0;
   auto fail = [&](const char* what, uint32_t a, uint32_t b) {
      if (errors++ < 5) std::printf("FAIL[%s]: %s (%u, %u)\n", name, what, a, b);
   };
   if (out.nodes.empty() || out.nodes.size() != out.cnodes.size()) fail("node arrays", (uint32_t)out.nodes.size(), (uint32_t)out.cnodes.size());
   if (out.level_start.size() < 2 || out.level_start.front() != 0 || out.level_start.back() != out.nodes.size()) fail("level_start", (uint32_t)out.level_start.size(), 0);
   if (out.max_depth + 2 != out.level_start.size()) fail("max_depth vs levels", out.max_depth, (uint32_t)out.level_start.size());
   if (out.tri_order.size() != n) fail("tri_order size", (uint32_t)out.tri_order.size(), n);
   std::vector<uint32_t> seen(n, 0), node_refs(out.nodes.size(), 0);
   std::vector<uint32_t> packet_use(n, 0);
   for (uint32_t i = 0; i < out.tri_order.size(); i++) {
      if (out.tri_order[i] >= n) fail("tri_order entry out of range", i, out.tri_order[i]);
      else seen[out.tri_order[i]]++;
   }
   for (uint32_t i = 0; i < n; i++)
      if (seen[i] != 1) fail("triangle not exactly once in tri_order", i, seen[i]);
   // recursive subtree bounds
   struct Frame {
      uint32_t node;
   };
   std::vector<Frame> st{{0}};
   while (!st.empty()) {
      uint32_t ni = st.back().node;
      st.pop_back();
      const NodeW& nd = out.nodes[ni];
      const Node4C& q = out.cnodes[ni];
      float scale[3];
      for (int a = 0; a < 3; a++) scale[a] = std::ldexp(1.0f, (int)((q.meta >> (8 * a)) & 0xff) - 127);
      const uint32_t n_tri = (q.meta >> kMetaTriShift) & 7, n_child = (q.meta >> kMetaChildShift) & 7;
      if (n_tri > n_child || n_child > 4) fail("child counts", n_tri, n_child);
      if ((q.child_base >> kChildBaseBits) != n_tri) fail("n_tri is not in the top bits of child_base (what the traversal reads)", ni, q.child_base >> kChildBaseBits);
      for (int a = 0; a < 3; a++)
         if (((q.meta >> (8 * a)) & 0xff) < 1 || ((q.meta >> (8 * a)) & 0xff) > 254) fail("step exponent byte out of range", ni, (uint32_t)a);
      for (int k = 0; k < 4; k++) {
         uint32_t c = nd.child[k];
         const uint32_t implicit = (uint32_t)k < n_tri ? (kLeafBit | (q.tri_base + (uint32_t)k)) : ((uint32_t)k < n_child ? (q.child_base & kChildBaseMask) + ((uint32_t)k - n_tri) : kEmptyRef);
         if (implicit != c) fail("implicit child address differs from the explicit ref", ni, (uint32_t)k);
         if (c == kEmptyRef) {
            for (int a = 0; a < 3; a++)
               if (((q.qlo[a] >> (8 * k)) & 0xff) != 0xff || ((q.qhi[a] >> (8 * k)) & 0xff) != 0) fail("empty slot is not an inverted box", ni, (uint32_t)k);
            continue;
         }
         const float lo[3] = {nd.lo[0][k], nd.lo[1][k], nd.lo[2][k]}, hi[3] = {nd.hi[0][k], nd.hi[1][k], nd.hi[2][k]};
         for (int a = 0; a < 3; a++) {
            float qlo = q.origin[a] + scale[a] * (float)((q.qlo[a] >> (8 * k)) & 0xff);
            float qhi = q.origin[a] + scale[a] * (float)((q.qhi[a] >> (8 * k)) & 0xff);
            // the kernel evaluates origin + scale*q in t-space; in world space allow one ulp of the sum
            float slack = 4e-7f * std::fmax(std::fabs(lo[a]), std::fabs(hi[a])) + 1e-30f;
            if (geometry && (!(qlo <= lo[a] + slack) || !(qhi >= hi[a] - slack))) fail("quantised box does not contain the full box", ni, (uint32_t)(k * 3 + a));
         }
         if (c & kLeafBit) {
            const uint32_t p = c & ~kLeafBit;
            if (p >= n) {
               fail("packet index out of range", ni, p);
               continue;
            }
            {
               packet_use[p]++;
               const float* t = &corners[9 * (size_t)out.tri_order[p]];
               for (int v = 0; v < 3; v++)
                  for (int a = 0; a < 3; a++)
                     if (geometry && (!(t[3 * v + a] >= lo[a]) || !(t[3 * v + a] <= hi[a]))) fail("triangle outside its leaf box", ni, p);
            }
         } else {
            if (c >= out.nodes.size()) {
               fail("child index out of range", ni, c);
               continue;
            }
            node_refs[c]++;
            {  // the child's stored frame IS the one a traversal derives from this node's frame and the child's quantised box here
               // (node_quant.h qn_inherit - kernels.hip inherit_frame restates it), and 255 of its steps cover the child's padded box
               uint32_t ql[3], qh[3], ce = 0;
               float co[3];
               for (int a = 0; a < 3; a++) {
                  ql[a] = (q.qlo[a] >> (8 * k)) & 0xff;
                  qh[a] = (q.qhi[a] >> (8 * k)) & 0xff;
               }
               qn_inherit(q.origin, q.meta & 0xffffffu, ql, qh, co, ce);
               const Node4C& cq = out.cnodes[c];
               if (UH_INHERIT_FRAME && (std::memcmp(co, cq.origin, sizeof(co)) != 0 || ce != (cq.meta & 0xffffffu))) fail("a child's stored frame is not the inherited one", ni, c);
               for (int a = 0; a < 3; a++) {
                  const double top = (double)co[a] + 255.0 * std::ldexp(1.0, (int)((ce >> (8 * a)) & 0xff) - 127);
                  if (UH_INHERIT_FRAME && geometry && (!((double)co[a] <= (double)lo[a]) || !(top >= (double)hi[a]))) fail("the inherited frame does not cover the child's padded box", ni, c);
               }
            }
            // child's own children must lie inside this slot's box (boxes are padded outwards at every level)
            const NodeW& ch = out.nodes[c];
            for (int j = 0; j < 4; j++)
               if (ch.child[j] != kEmptyRef) {
                  const float clo[3] = {ch.lo[0][j], ch.lo[1][j], ch.lo[2][j]}, chi[3] = {ch.hi[0][j], ch.hi[1][j], ch.hi[2][j]};
                  for (int a = 0; a < 3; a++) {
                     float pad = 2e-4f + 2e-5f * std::fmax(std::fabs(clo[a]), std::fabs(chi[a]));
                     if (geometry && (!(clo[a] >= lo[a] - pad) || !(chi[a] <= hi[a] + pad))) fail("grandchild box escapes its parent slot", ni, c);
                  }
               }
            st.push_back({c});
         }
      }
   }
   for (uint32_t p = 0; p < n; p++)
      if (packet_use[p] != 1) fail("packet not referenced by exactly one leaf", p, packet_use[p]);
   for (size_t i = 1; i < node_refs.size(); i++)
      if (node_refs[i] != 1) fail("interior node not referenced exactly once", (uint32_t)i, node_refs[i]);
   std::printf("%s: %u tris -> %zu nodes, depth %u, %s\n", name, n, out.nodes.size(), out.max_depth, errors ? "FAILED" : "ok");
   return errors;
}

int main() {
   int errors = 0;
   for (uint32_t seed : {1u, 4u}) {
      for (int threads : {1, 4}) {
         std::vector<float> soup;
         g_state = 12345 + seed;
         for (int i = 0; i < 200000; i++) {
            float c[3] = {rnd() * 40 - 20, rnd() * 15, rnd() * 16 - 8};
            for (int v = 0; v < 3; v++)
               for (int a = 0; a < 3; a++) soup.push_back(c[a] + (rnd() - 0.5f) * 0.4f);
         }
         errors += check(soup, threads, "random soup");
      }
   }
   {
      std::vector<float> empty;
      errors += check(empty, 1, "empty");
      std::vector<float> one = {0, 0, 0, 1, 0, 0, 0, 1, 0};
      errors += check(one, 1, "single triangle");
      std::vector<float> dup;
      for (int i = 0; i < 1000; i++) dup.insert(dup.end(), one.begin(), one.end());
      errors += check(dup, 4, "1000 identical triangles");
      std::vector<float> degenerate;
      g_state = 7;
      for (int i = 0; i < 5000; i++) {
         float p[3] = {rnd(), rnd(), rnd()};
         for (int v = 0; v < 3; v++)
            for (int a = 0; a < 3; a++) degenerate.push_back(p[a]);  // zero-area (point) triangles
      }
      errors += check(degenerate, 2, "point triangles");
      std::vector<float> huge;
      for (int i = 0; i < 3000; i++)
         for (int k = 0; k < 9; k++) huge.push_back((rnd() - 0.5f) * 2e6f);
      errors += check(huge, 2, "huge coordinates");
      std::vector<float> planar;
      for (int i = 0; i < 20000; i++) {
         float x = rnd() * 10, z = rnd() * 10;
         float t[9] = {x, 0, z, x + 0.1f, 0, z, x, 0, z + 0.1f};
         planar.insert(planar.end(), t, t + 9);
      }
      errors += check(planar, 4, "coplanar sheet");
      // non-finite vertices must not break the structure (every packet in exactly one leaf, no crash): such
      // triangles are never hit (NaN fails every comparison in the slab and triangle tests)
      std::vector<float> poisoned = planar;
      for (size_t i = 0; i < poisoned.size(); i += 997) poisoned[i] = (i & 1) ? NAN : ((i & 2) ? INFINITY : -INFINITY);
      errors += check(poisoned, 4, "NaN / inf vertices", false);
   }
   {
      // clusters on a geometric series (each a factor 0.5 closer to the origin than the one before): SAH peels them off one
      // by one and the tree becomes a chain. The traversal stack holds kTraversalStackEntries entries = kMaxTreeLevels
      // levels; the context rebuilds such a scene balanced (csrc/context.hip), which must fit for any triangle count.
      std::vector<float> chain;
      g_state = 99;
      for (int k = 0; k < 120; k++) {
         const float s = std::ldexp(1.0f, -k);
         for (int i = 0; i < 3; i++) {
            float c[3] = {s * (1.0f + 0.1f * rnd()), s * 0.1f * rnd(), s * 0.1f * rnd()};
            for (int v = 0; v < 3; v++)
               for (int a = 0; a < 3; a++) chain.push_back(c[a] + s * 0.01f * rnd());
         }
      }
      errors += check(chain, 2, "geometric-series clusters (SAH)");
      std::vector<uint32_t> keys(chain.size() / 9);
      BuildInput in{chain.data(), keys.data(), (uint32_t)keys.size()};
      BuildOutput sah, bal;
      build_bvh4(in, sah, 1);
      build_bvh4(in, bal, 1, true);
      std::printf("geometric-series clusters: SAH tree %zu levels, balanced tree %zu levels (the stack holds %u)\n", sah.level_start.size() - 1, bal.level_start.size() - 1, kMaxTreeLevels);
      if (bal.level_start.size() - 1 > kMaxTreeLevels) {
         std::printf("FAIL: balanced tree deeper than the traversal stack\n");
         errors++;
      }
      if (sah.level_start.size() - 1 <= 12) {
         std::printf("FAIL: the chain scene no longer produces a deep SAH tree (test lost its point)\n");
         errors++;
      }
   }
   {
      // the same series with >= 65536 triangles and several threads: the serial top of the threaded build must honour
      // `balanced` too (it used to split by SAH whatever the flag said, and a deep clustered scene stayed too deep)
      std::vector<float> chain;
      g_state = 4242;
      for (int k = 0; k < 100; k++) {
         const float s = std::ldexp(1.0f, -k);
         for (int i = 0; i < 700; i++) {
            float c[3] = {s * (1.0f + 0.1f * rnd()), s * 0.1f * rnd(), s * 0.1f * rnd()};
            for (int v = 0; v < 3; v++)
               for (int a = 0; a < 3; a++) chain.push_back(c[a] + s * 0.01f * rnd());
         }
      }
      std::vector<uint32_t> keys(chain.size() / 9);
      BuildInput in{chain.data(), keys.data(), (uint32_t)keys.size()};
      BuildOutput sah, bal;
      build_bvh4(in, sah, 4);
      build_bvh4(in, bal, 4, true);
      std::printf("%zu clustered triangles, 4 threads: SAH tree %zu levels, balanced tree %zu levels (the stack holds %u)\n", keys.size(), sah.level_start.size() - 1,
                  bal.level_start.size() - 1, kMaxTreeLevels);
      if (keys.size() < 65536 || bal.level_start.size() - 1 > kMaxTreeLevels) {
         std::printf("FAIL: threaded balanced tree deeper than the traversal stack\n");
         errors++;
      }
   }
   {
      // build_sah_top (the host half of the device builder): a binary tree over boxes - count - 1 nodes, node 0 the root, every
      // box in exactly one leaf, every other node referenced exactly once, a node's area that of its leaves' union
      for (uint32_t count : {2u, 3u, 17u, 1000u, 5000u}) {
         std::vector<float> boxes;
         g_state = 4242 + count;
         for (uint32_t i = 0; i < count; i++) {
            float c[3] = {rnd() * 30 - 15, rnd() * 10, rnd() * 12 - 6}, e[3] = {rnd(), rnd() * 0.5f, rnd()};
            for (int a = 0; a < 3; a++) boxes.push_back(c[a] - e[a]);
            for (int a = 0; a < 3; a++) boxes.push_back(c[a] + e[a]);
         }
         if (count == 17) boxes.assign(boxes.size(), 0.25f);  // identical degenerate boxes: the split must still terminate
         std::vector<TopNode> top;
         build_sah_top(boxes.data(), count, top);
         int bad = top.size() != count - 1;
         std::vector<int> leaf_use(count, 0), node_use(top.size(), 0);
         for (size_t k = 0; k < top.size() && !bad; k++)
            for (uint32_t r : {top[k].left, top[k].right}) {
               if (r & kLeafBit) {
                  if ((r & ~kLeafBit) >= count) bad = 1; else leaf_use[r & ~kLeafBit]++;
               } else {
                  if (r >= top.size() || r == 0) bad = 1; else node_use[r]++;
               }
            }
         for (uint32_t i = 0; i < count; i++) bad |= leaf_use[i] != 1;
         for (size_t k = 1; k < top.size(); k++) bad |= node_use[k] != 1;
         if (!bad) {
            // area of the root = area of the union of all boxes
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = 0; i < count; i++)
               for (int a = 0; a < 3; a++) lo[a] = std::fmin(lo[a], boxes[6 * i + a]), hi[a] = std::fmax(hi[a], boxes[6 * i + 3 + a]);
            const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            bad |= top[0].half_area != dx * dy + dy * dz + dz * dx;
         }
         std::printf("SAH top over %u boxes: %zu nodes, %s\n", count, top.size(), bad ? "FAILED" : "ok");
         errors += bad;
      }
   }
   std::printf(errors ? "BVH CHECK FAILED (%d)\n" : "BVH CHECK OK\n", errors);
   return errors ? 1 : 0;
}
