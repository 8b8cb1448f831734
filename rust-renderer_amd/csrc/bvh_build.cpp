// bvh_build.cpp — host builder of the flattened BVH4 (see bvh.h). Replaces
// Raytracing::create_bottom_level_acceleration_structure / create_top_level_acceleration_structure
// (reference: utopian/src/raytracing.rs:113-217, :279-398), whose build runs inside the Vulkan
// driver. Binned SAH (16 bins, 3 axes) -> BVH2 down to single triangles -> greedy collapse to 4-wide nodes by surface
// area, emitted breadth-first with every node's triangle children and node children in contiguous slots (bvh.h).
#include "bvh.h"
#include "node_quant.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace uh {
namespace {

struct Box {
   float lo[3], hi[3];
   void reset() {
      for (int a = 0; a < 3; a++) {
         lo[a] = INFINITY;
         hi[a] = -INFINITY;
      }
   }
   void grow(const Box& b) {
      for (int a = 0; a < 3; a++) {
         lo[a] = std::fmin(lo[a], b.lo[a]);
         hi[a] = std::fmax(hi[a], b.hi[a]);
      }
   }
   void grow_pt(const float* p) {
      for (int a = 0; a < 3; a++) {
         lo[a] = std::fmin(lo[a], p[a]);
         hi[a] = std::fmax(hi[a], p[a]);
      }
   }
   float half_area() const {
      float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
      if (!(dx >= 0) || !(dy >= 0) || !(dz >= 0)) return 0.0f;
      return dx * dy + dy * dz + dz * dx;
   }
};

// build knobs (what the studies of rounds 3-4 varied, tools/bvh_visits.py; the values below are what they settled on - the library
// reads no environment variable)
struct Knobs {
   int collapse = 1;        // 0 = greedy by area (rounds 1-3), 1 = SAH-optimal collapse by dynamic programming (Ylitie et al. 2017)
   int sweep_below = 0;     // ranges of at most this many triangles are split by an exact sweep over the sorted centroids instead of 16 bins
   int bins = 16;
#ifndef UH_BVH_CTRI
#define UH_BVH_CTRI 1.0f
#endif
   float c_tri = UH_BVH_CTRI;  // cost of a triangle slot relative to a node visit, per unit of area (collapse = 1); build-time switch for tools/bvh_visits.py
};
static Knobs knobs_from_env() { return Knobs(); }

struct Node2 {
   Box box;
   int32_t left, right;    // children (interior) or -1
   uint32_t first, count;  // leaf range in the permuted index array
};

struct Builder {
   const std::vector<Box>& tb;
   const std::vector<float>& cen;  // 3 per triangle
   std::vector<uint32_t>& idx;
   std::vector<Node2> nodes;
   uint32_t max_depth = 0;
   bool balanced = false;  // median splits only: depth ceil(log2 n), for geometry whose SAH tree would be too deep for the traversal stack
   Knobs kn;

   Builder(const std::vector<Box>& tb_, const std::vector<float>& cen_, std::vector<uint32_t>& idx_) : tb(tb_), cen(cen_), idx(idx_) {}

   // builds the subtree over idx[first, first+count) ; returns node index
   int32_t build(uint32_t first, uint32_t count, uint32_t depth) {
      int32_t me = (int32_t)nodes.size();
      nodes.push_back(Node2());
      Box box, cb;
      box.reset();
      cb.reset();
      for (uint32_t k = first; k < first + count; k++) {
         box.grow(tb[idx[k]]);
         cb.grow_pt(&cen[3 * (size_t)idx[k]]);
      }
      nodes[me].box = box;
      max_depth = std::max(max_depth, depth);
      auto make_leaf = [&]() {
         nodes[me].left = nodes[me].right = -1;
         nodes[me].first = first;
         nodes[me].count = count;
         return me;
      };
      if (count == 1) return make_leaf();
      if (count == 2) {  // nothing to choose
         int32_t l = build(first, 1, depth + 1), r = build(first + 1, 1, depth + 1);
         nodes[me].left = l;
         nodes[me].right = r;
         nodes[me].first = first;
         nodes[me].count = count;
         return me;
      }
      if ((int)count <= kn.sweep_below && !balanced && depth <= 48) {
         // exact sweep: every split position of the range sorted by centroid, on each axis
         float best = INFINITY;
         int axis = -1;
         uint32_t at = 0;
         std::vector<uint32_t> order(idx.begin() + first, idx.begin() + first + count), best_order;
         std::vector<float> right(count);
         for (int a = 0; a < 3; a++) {
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
               const float cx = cen[3 * (size_t)x + a], cy = cen[3 * (size_t)y + a];
               return cx < cy || (cx == cy && x < y);
            });
            Box acc;
            acc.reset();
            for (uint32_t k = count - 1; k > 0; k--) {
               acc.grow(tb[order[k]]);
               right[k] = acc.half_area();
            }
            acc.reset();
            for (uint32_t k = 0; k + 1 < count; k++) {
               acc.grow(tb[order[k]]);
               const float cost = acc.half_area() * (float)(k + 1) + right[k + 1] * (float)(count - k - 1);
               if (cost < best) {
                  best = cost;
                  axis = a;
                  at = k + 1;
               }
            }
            if (axis == a) best_order = order;
         }
         if (axis >= 0) {
            std::copy(best_order.begin(), best_order.end(), idx.begin() + first);
            int32_t l = build(first, at, depth + 1);
            int32_t r = build(first + at, count - at, depth + 1);
            nodes[me].left = l;
            nodes[me].right = r;
            nodes[me].first = first;
            nodes[me].count = count;
            return me;
         }
      }
      constexpr int NBMAX = 64;
      const int NB = kn.bins;
      float best_cost = INFINITY;
      int best_axis = -1, best_split = -1;
      for (int a = 0; a < 3; a++) {
         float lo = cb.lo[a], ext = cb.hi[a] - cb.lo[a];
         if (!(ext > 0) || !std::isfinite(ext)) continue;
         Box bb[NBMAX];
         uint32_t bc[NBMAX];
         for (int b = 0; b < NB; b++) {
            bb[b].reset();
            bc[b] = 0;
         }
         float scale = (float)NB / ext;
         for (uint32_t k = first; k < first + count; k++) {
            uint32_t t = idx[k];
            int b = (int)((cen[3 * (size_t)t + a] - lo) * scale);
            b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
            bb[b].grow(tb[t]);
            bc[b]++;
         }
         float right_area[NBMAX];
         uint32_t right_cnt[NBMAX];
         Box acc;
         acc.reset();
         uint32_t c = 0;
         for (int b = NB - 1; b > 0; b--) {
            acc.grow(bb[b]);
            c += bc[b];
            right_area[b] = acc.half_area();
            right_cnt[b] = c;
         }
         acc.reset();
         c = 0;
         for (int b = 0; b < NB - 1; b++) {
            acc.grow(bb[b]);
            c += bc[b];
            if (c == 0 || right_cnt[b + 1] == 0) continue;
            float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
            if (cost < best_cost) {
               best_cost = cost;
               best_axis = a;
               best_split = b;
            }
         }
      }
      // no SAH termination: every leaf is one triangle (a 4-wide node holds up to four of them as children, each with
      // its own box, and only the ones a ray's slab test passes cost a triangle fetch)
      uint32_t mid;
      if (best_axis < 0 || depth > 48 || balanced) {
         // degenerate centroids (or runaway depth): split the range in half along the widest axis
         int a = 0;
         for (int k = 1; k < 3; k++)
            if (cb.hi[k] - cb.lo[k] > cb.hi[a] - cb.lo[a]) a = k;
         mid = first + count / 2;
         std::nth_element(idx.begin() + first, idx.begin() + mid, idx.begin() + first + count,
                          [&](uint32_t x, uint32_t y) { return cen[3 * (size_t)x + a] < cen[3 * (size_t)y + a]; });
      } else {
         float lo = cb.lo[best_axis], scale = (float)NB / (cb.hi[best_axis] - cb.lo[best_axis]);
         auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [&](uint32_t t) {
            int b = (int)((cen[3 * (size_t)t + best_axis] - lo) * scale);
            b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
            return b <= best_split;
         });
         mid = (uint32_t)(it - idx.begin());
         if (mid == first || mid == first + count) mid = first + count / 2;
      }
      int32_t l = build(first, mid - first, depth + 1);
      int32_t r = build(mid, first + count - mid, depth + 1);
      nodes[me].left = l;
      nodes[me].right = r;
      nodes[me].first = first;
      nodes[me].count = count;
      return me;
   }
};

// Conservative padding: the kernels' slab test must never cull a triangle that the
// ray/triangle test would accept (DESIGN.md "Arithmetic contract", order-independence).
inline void padded(const Box& b, float* lo, float* hi) {
   for (int a = 0; a < 3; a++) {
      float pad = 1e-4f + 1e-5f * std::fmax(std::fabs(b.lo[a]), std::fabs(b.hi[a]));
      lo[a] = b.lo[a] - pad;
      hi[a] = b.hi[a] + pad;
   }
}

}  // namespace

void quantise_tree(const std::vector<NodeW>& nodes, std::vector<Node4C>& out) {
   out.assign(nodes.size(), Node4C{});
   if (nodes.empty()) return;
   // BFS order: a node's children lie behind it, so one pass in index order meets every node after its parent has set its frame
   std::vector<uint8_t> framed(nodes.size(), 0);
   for (size_t i = 0; i < nodes.size(); i++) {
      const NodeW& nd = nodes[i];
      Node4C& q = out[i];
      uint32_t n_tri = 0, n_child = 0;
      float lo[4][3], hi[4][3];
      for (int k = 0; k < 4; k++) {
         for (int a = 0; a < 3; a++) {
            lo[k][a] = nd.lo[a][k];
            hi[k][a] = nd.hi[a][k];
         }
         if (nd.child[k] == kEmptyRef) continue;
         n_child++;
         if (nd.child[k] & kLeafBit) n_tri++;
      }
      if (!framed[i] || !UH_INHERIT_FRAME) {  // the root takes its own frame (and so does every node of a build without inherited frames)
         uint32_t exps = 0;
         qn_own_frame(lo, hi, n_child, q.origin, exps);
         q.meta = exps;
      }
      float child_origin[4][3];
      uint32_t child_exps[4] = {0, 0, 0, 0};
      qn_quantise(q.origin, q.meta & 0xffffffu, lo, hi, n_tri, n_child, q.qlo, q.qhi, child_origin, child_exps, UH_INHERIT_FRAME != 0);
      q.meta = (q.meta & 0xffffffu) | (n_tri << kMetaTriShift) | (n_child << kMetaChildShift);
      uint32_t child_base = 0;
      q.tri_base = 0;
      for (int k = 0; k < 4; k++) {
         if (nd.child[k] == kEmptyRef) continue;
         if ((nd.child[k] & kLeafBit) && k == 0) q.tri_base = nd.child[k] & ~kLeafBit;
         if (!(nd.child[k] & kLeafBit)) {
            if ((uint32_t)k == n_tri) child_base = nd.child[k];
            if (!UH_INHERIT_FRAME) continue;
            Node4C& c = out[nd.child[k]];
            for (int a = 0; a < 3; a++) c.origin[a] = child_origin[k][a];
            c.meta = child_exps[k];
            framed[nd.child[k]] = 1;
         }
      }
      q.child_base = (child_base & kChildBaseMask) | (n_tri << kChildBaseBits);
   }
}

void build_sah_top(const float* boxes6, uint32_t count, std::vector<TopNode>& out) {
   out.clear();
   if (count < 2) return;
   std::vector<Box> tb(count);
   std::vector<float> cen(3 * (size_t)count);
   std::vector<uint32_t> idx(count);
   for (uint32_t i = 0; i < count; i++) {
      for (int a = 0; a < 3; a++) {
         tb[i].lo[a] = boxes6[6 * (size_t)i + a];
         tb[i].hi[a] = boxes6[6 * (size_t)i + 3 + a];
         cen[3 * (size_t)i + a] = 0.5f * (tb[i].lo[a] + tb[i].hi[a]);
      }
      idx[i] = i;
   }
   Builder b(tb, cen, idx);
   b.build(0, count, 0);
   // Builder numbers a node before its children (node 0 = root) and gives leaves nodes of their own: renumber the interior ones
   std::vector<uint32_t> interior(b.nodes.size(), 0);
   uint32_t n_int = 0;
   for (size_t i = 0; i < b.nodes.size(); i++)
      if (b.nodes[i].left >= 0) interior[i] = n_int++;
   out.resize(n_int);
   auto ref = [&](int32_t c) { return b.nodes[c].left >= 0 ? interior[c] : (kLeafBit | idx[b.nodes[c].first]); };
   for (size_t i = 0; i < b.nodes.size(); i++)
      if (b.nodes[i].left >= 0) out[interior[i]] = TopNode{ref(b.nodes[i].left), ref(b.nodes[i].right), b.nodes[i].box.half_area()};
}

void build_bvh4(const BuildInput& in, BuildOutput& out, int num_threads, bool balanced, uint32_t width) {
   if (width < 2) width = 2;
   if (width > (uint32_t)kMaxWidth) width = kMaxWidth;
   out.width = width;
   out.nodes.clear();
   out.cnodes.clear();
   out.tri_order.clear();
   out.level_start.assign({0u, 1u});
   out.max_depth = 0;
   const uint32_t n = in.count;
   std::vector<Box> tb(n);
   std::vector<float> cen(3 * (size_t)n);
   for (uint32_t i = 0; i < n; i++) {
      Box b;
      b.reset();
      bool finite = true;
      const float* c = in.corners + 9 * (size_t)i;
      b.grow_pt(c);
      b.grow_pt(c + 3);
      b.grow_pt(c + 6);
      for (int k = 0; k < 9; k++) finite = finite && std::isfinite(c[k]);
      // a triangle with a non-finite corner can never be hit (NaN fails every comparison of the slab and
      // triangle tests, an infinite edge turns the barycentrics into NaN): it gets a point box at the origin so
      // that the split search below only ever sees finite numbers (NaN breaks the ordering std::nth_element needs)
      for (int a = 0; a < 3; a++) finite = finite && std::isfinite(0.5f * (b.lo[a] + b.hi[a]));
      if (!finite)
         for (int a = 0; a < 3; a++) b.lo[a] = b.hi[a] = 0.0f;
      tb[i] = b;
      for (int a = 0; a < 3; a++) cen[3 * (size_t)i + a] = 0.5f * (b.lo[a] + b.hi[a]);
   }
   out.tri_order.resize(n);
   for (uint32_t i = 0; i < n; i++) out.tri_order[i] = i;

   // ---- BVH2. The top of the tree is built serially down to `num_threads`-ish subtrees, which
   // are then built concurrently on disjoint ranges of the index array.
   std::vector<Node2> n2;
   if (n == 0) {
      NodeW root;
      std::memset(&root, 0, sizeof(root));
      for (int k = 0; k < kMaxWidth; k++) root.child[k] = kEmptyRef;
      out.nodes.push_back(root);
      if (width == 4) {
         quantise_tree(std::vector<NodeW>{root}, out.cnodes);  // empty scene: a root whose four slots are empty (every ray misses)
      }
      return;
   }
   {
      Builder top(tb, cen, out.tri_order);
      top.balanced = balanced;
      top.kn = knobs_from_env();
      if (num_threads <= 1 || n < 65536) {
         top.build(0, n, 0);
         n2.swap(top.nodes);
         out.max_depth = top.max_depth;
      } else {
         // serial top: split until ranges are small enough, recording open jobs
         struct Job {
            uint32_t first, count, depth;
            int32_t parent;
            bool is_left;
         };
         // Simple approach: build whole tree serially for the top levels by limiting leaf size
         // through a recursive lambda that stops at `grain` triangles.
         const uint32_t grain = std::max<uint32_t>(16384, n / (uint32_t)(num_threads * 4));
         std::vector<Job> jobs;
         std::vector<Node2>& nodes = top.nodes;
         struct Frame {
            uint32_t first, count, depth;
            int32_t parent;
            bool is_left;
         };
         std::vector<Frame> st;
         st.push_back(Frame{0, n, 0, -1, false});
         while (!st.empty()) {
            Frame f = st.back();
            st.pop_back();
            if (f.count <= grain) {
               jobs.push_back(Job{f.first, f.count, f.depth, f.parent, f.is_left});
               continue;
            }
            // one SAH split step, reusing Builder::build's logic on a throw-away builder would
            // recurse fully; instead do a median split on the widest centroid axis for the few top
            // levels (they matter little for SAH quality: < log2(threads*4) levels).
            Box box, cb;
            box.reset();
            cb.reset();
            for (uint32_t k = f.first; k < f.first + f.count; k++) {
               box.grow(tb[out.tri_order[k]]);
               cb.grow_pt(&cen[3 * (size_t)out.tri_order[k]]);
            }
            int32_t me = (int32_t)nodes.size();
            nodes.push_back(Node2());
            nodes[me].box = box;
            nodes[me].left = nodes[me].right = -1;
            nodes[me].first = f.first;
            nodes[me].count = f.count;
            if (f.parent >= 0) (f.is_left ? nodes[f.parent].left : nodes[f.parent].right) = me;
            // binned SAH on this range (same as Builder::build)
            constexpr int NB = 16;
            float best_cost = INFINITY;
            int best_axis = -1, best_split = -1;
            for (int a = 0; a < 3; a++) {
               float lo = cb.lo[a], ext = cb.hi[a] - cb.lo[a];
               if (!(ext > 0) || !std::isfinite(ext)) continue;
               Box bb[NB];
               uint32_t bc[NB];
               for (int b = 0; b < NB; b++) {
                  bb[b].reset();
                  bc[b] = 0;
               }
               float scale = (float)NB / ext;
               for (uint32_t k = f.first; k < f.first + f.count; k++) {
                  uint32_t t = out.tri_order[k];
                  int b = (int)((cen[3 * (size_t)t + a] - lo) * scale);
                  b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                  bb[b].grow(tb[t]);
                  bc[b]++;
               }
               float ra[NB];
               uint32_t rc[NB];
               Box acc;
               acc.reset();
               uint32_t c = 0;
               for (int b = NB - 1; b > 0; b--) {
                  acc.grow(bb[b]);
                  c += bc[b];
                  ra[b] = acc.half_area();
                  rc[b] = c;
               }
               acc.reset();
               c = 0;
               for (int b = 0; b < NB - 1; b++) {
                  acc.grow(bb[b]);
                  c += bc[b];
                  if (c == 0 || rc[b + 1] == 0) continue;
                  float cost = acc.half_area() * (float)c + ra[b + 1] * (float)rc[b + 1];
                  if (cost < best_cost) {
                     best_cost = cost;
                     best_axis = a;
                     best_split = b;
                  }
               }
            }
            uint32_t mid;
            if (best_axis < 0 || balanced || f.depth > 48) {
               // median split along the widest centroid axis: what Builder::build does in the same cases, so that the balanced
               // fallback bounds the depth of the serial top as well (a SAH top over clustered geometry can be a long chain)
               int a = 0;
               for (int k = 1; k < 3; k++)
                  if (cb.hi[k] - cb.lo[k] > cb.hi[a] - cb.lo[a]) a = k;
               mid = f.first + f.count / 2;
               std::nth_element(out.tri_order.begin() + f.first, out.tri_order.begin() + mid, out.tri_order.begin() + f.first + f.count,
                                [&](uint32_t x, uint32_t y) { return cen[3 * (size_t)x + a] < cen[3 * (size_t)y + a]; });
            } else {
               float lo = cb.lo[best_axis], scale = (float)NB / (cb.hi[best_axis] - cb.lo[best_axis]);
               auto it = std::partition(out.tri_order.begin() + f.first, out.tri_order.begin() + f.first + f.count, [&](uint32_t t) {
                  int b = (int)((cen[3 * (size_t)t + best_axis] - lo) * scale);
                  b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                  return b <= best_split;
               });
               mid = (uint32_t)(it - out.tri_order.begin());
               if (mid == f.first || mid == f.first + f.count) mid = f.first + f.count / 2;
            }
            st.push_back(Frame{mid, f.first + f.count - mid, f.depth + 1, me, false});
            st.push_back(Frame{f.first, mid - f.first, f.depth + 1, me, true});
         }
         // parallel subtrees
         std::vector<std::vector<Node2>> sub(jobs.size());
         std::vector<uint32_t> sub_depth(jobs.size(), 0);
         std::atomic<size_t> next(0);
         auto worker = [&]() {
            for (;;) {
               size_t j = next.fetch_add(1);
               if (j >= jobs.size()) break;
               Builder b(tb, cen, out.tri_order);
               b.balanced = balanced;
               b.kn = top.kn;
               b.build(jobs[j].first, jobs[j].count, jobs[j].depth);
               sub[j].swap(b.nodes);
               sub_depth[j] = b.max_depth;
            }
         };
         std::vector<std::thread> th;
         for (int t = 0; t < num_threads; t++) th.emplace_back(worker);
         for (auto& t : th) t.join();
         // stitch
         for (size_t j = 0; j < jobs.size(); j++) {
            int32_t base = (int32_t)nodes.size();
            for (Node2 nd : sub[j]) {
               if (nd.left >= 0) {
                  nd.left += base;
                  nd.right += base;
               }
               nodes.push_back(nd);
            }
            if (jobs[j].parent >= 0) (jobs[j].is_left ? nodes[jobs[j].parent].left : nodes[jobs[j].parent].right) = base;
            out.max_depth = std::max(out.max_depth, sub_depth[j]);
         }
         n2.swap(nodes);
      }
   }

   // ---- collapse to BVH4, breadth-first emission. Slot order inside a node: its triangle children first (their
   // packets are consecutive: the packet order is DEFINED here, node by node), then its node children (consecutive
   // node indices), then empty slots.
   // SAH-optimal collapse (Ylitie, Karras, Laine 2017, section 3.1, for single-triangle leaves): T[n][i] = the least cost of
   // representing BVH2 subtree n by at most i + 1 slots of a wide node - a slot being one triangle (cost c_tri x its area) or one
   // wide node (its area, plus the best distribution of its two children over W slots). Replaces the greedy "open the child
   // with the largest area" of rounds 1-3, which fills nodes to 3.0 of 4 children on the config-1 scene.
   const Knobs kn = knobs_from_env();
   const int W = (int)width;
   std::vector<float> T;      // [n * W + i]
   std::vector<uint8_t> split;  // [n * W + i]: how many of the slots the left child gets (0 = n itself is one slot) ...
   std::vector<uint8_t> used;   // ... and how many slots the two children take together (<= i + 1: a smaller budget's plan may be the best)
   if (kn.collapse == 1) {
      T.assign(n2.size() * (size_t)W, 0.0f);
      split.assign(n2.size() * (size_t)W, 0);
      used.assign(n2.size() * (size_t)W, 0);
      // every child before its parent: the reverse of a pre-order walk from the root (the index order of the builders has that
      // property too, but not once the optimiser has moved nodes around)
      std::vector<int32_t> pre;
      pre.reserve(n2.size());
      {
         std::vector<int32_t> stack{0};
         while (!stack.empty()) {
            const int32_t k = stack.back();
            stack.pop_back();
            pre.push_back(k);
            if (n2[k].left >= 0) {
               stack.push_back(n2[k].left);
               stack.push_back(n2[k].right);
            }
         }
      }
      for (size_t pi = pre.size(); pi-- > 0;) {
         const size_t k = (size_t)pre[pi];
         const Node2& nd = n2[k];
         const float area = nd.box.half_area();
         float* t = &T[k * (size_t)W];
         uint8_t* sp = &split[k * (size_t)W];
         uint8_t* us = &used[k * (size_t)W];
         if (nd.left < 0) {
            for (int i = 0; i < W; i++) t[i] = kn.c_tri * area;
            continue;
         }
         const float* tl = &T[(size_t)nd.left * W];
         const float* tr = &T[(size_t)nd.right * W];
         // D[j]: the two children over j + 1 slots (j >= 1)
         float D[kMaxWidth];
         uint8_t Dk[kMaxWidth];
         D[0] = INFINITY;
         Dk[0] = 0;
         for (int j = 1; j < W; j++) {
            D[j] = INFINITY;
            Dk[j] = 1;
            for (int a = 1; a <= j; a++) {  // left gets a slots, right gets j + 1 - a
               const float c = tl[a - 1] + tr[j - a];
               if (c < D[j]) {
                  D[j] = c;
                  Dk[j] = (uint8_t)a;
               }
            }
         }
         t[0] = area + D[W - 1];
         sp[0] = 0;
         us[0] = 1;
         for (int i = 1; i < W; i++) {
            if (D[i] < t[0] && D[i] <= t[i - 1]) {
               t[i] = D[i];
               sp[i] = Dk[i];           // left: Dk[i] in [1, i] slots, right: i + 1 - Dk[i] >= 1
               us[i] = (uint8_t)(i + 1);
            } else if (t[i - 1] < t[0]) {
               t[i] = t[i - 1];        // the smaller budget's plan, as it was made
               sp[i] = sp[i - 1];
               us[i] = us[i - 1];
            } else {
               t[i] = t[0];
               sp[i] = 0;
               us[i] = 1;
            }
         }
      }
   }
   // slots of the wide node made from BVH2 node `root`: its two children over W slots by the table
   auto dp_children = [&](int32_t root, int32_t* ch) {
      int nc = 0;
      struct It {
         int32_t n;
         int slots;
      };
      It st[2 * kMaxWidth];
      int sp_ = 0;
      const Node2& r = n2[root];
      // the root's own distribution: best a for D[W - 1]
      {
         const float* tl = &T[(size_t)r.left * W];
         const float* tr = &T[(size_t)r.right * W];
         int best_a = 1;
         float best = INFINITY;
         for (int a = 1; a <= W - 1; a++) {
            const float c = tl[a - 1] + tr[W - 1 - a];
            if (c < best) {
               best = c;
               best_a = a;
            }
         }
         st[sp_++] = It{r.right, W - best_a};
         st[sp_++] = It{r.left, best_a};
      }
      while (sp_) {
         const It it = st[--sp_];
         const Node2& nd = n2[it.n];
         const uint8_t a = (nd.left < 0 || it.slots < 2) ? 0 : split[(size_t)it.n * W + (it.slots - 1)];
         const int use = a ? used[(size_t)it.n * W + (it.slots - 1)] : 1;  // slots the stored plan takes: a for the left child, use - a >= 1 for the right
         if (a == 0 || use - (int)a < 1 || nc >= W || sp_ + 2 > 2 * kMaxWidth) {
            if (nc < W) ch[nc++] = it.n;
            continue;
         }
         st[sp_++] = It{nd.right, use - a};
         st[sp_++] = It{nd.left, a};
      }
      return nc;
   };
   std::vector<int32_t> queue;  // BVH2 node index of each emitted BVH4 node
   std::vector<uint32_t> packet_order;  // packet p holds input triangle packet_order[p]
   packet_order.reserve(n);
   queue.push_back(0);
   out.nodes.reserve(n2.size() / 2 + 1);
   out.nodes.push_back(NodeW());
   for (size_t qi = 0; qi < queue.size(); qi++) {
      const Node2& src = n2[queue[qi]];
      int32_t ch[kMaxWidth];
      int nc = 0;
      if (src.left < 0) {
         ch[nc++] = queue[qi];  // a single triangle at the root: wrap it
      } else if (kn.collapse == 1) {
         nc = dp_children(queue[qi], ch);
      } else {
         ch[nc++] = src.left;
         ch[nc++] = src.right;
         for (;;) {
            if (nc == W) break;
            int pick = -1;
            float pa = -1.0f;
            for (int k = 0; k < nc; k++)
               if (n2[ch[k]].left >= 0) {
                  float a = n2[ch[k]].box.half_area();
                  if (a > pa) {
                     pa = a;
                     pick = k;
                  }
               }
            if (pick < 0) break;
            int32_t c = ch[pick];
            ch[pick] = n2[c].left;
            ch[nc++] = n2[c].right;
         }
      }
      std::stable_partition(ch, ch + nc, [&](int32_t c) { return n2[c].left < 0; });  // triangles first
      {
         // node children by ASCENDING surface area. A visibility walk (k_trace_shadow) takes a node's children from the
         // highest slot down: the biggest subtree - the likeliest to hold an occluder - first, the triangle children last.
         // Measured on MI355X (tools/any_order_ab.sh): sun shadow rays 13.1 -> 10.6 node visits and 2.6 -> 2.0 triangle
         // tests per ray, +8 % frame rate. Closest-hit rays order the children by distance and are not affected.
         int nt = 0;
         while (nt < nc && n2[ch[nt]].left < 0) nt++;
         std::stable_sort(ch + nt, ch + nc, [&](int32_t a, int32_t b) { return n2[a].box.half_area() < n2[b].box.half_area(); });
      }
      NodeW nd;
      std::memset(&nd, 0, sizeof(nd));
      for (int k = 0; k < kMaxWidth; k++) nd.child[k] = kEmptyRef;
      for (int k = 0; k < nc; k++) {
         const Node2& c = n2[ch[k]];
         float lo[3], hi[3];
         padded(c.box, lo, hi);
         for (int a = 0; a < 3; a++) {
            nd.lo[a][k] = lo[a];
            nd.hi[a][k] = hi[a];
         }
         if (c.left < 0) {
            nd.child[k] = kLeafBit | (uint32_t)packet_order.size();
            packet_order.push_back(out.tri_order[c.first]);
         } else {
            nd.child[k] = (uint32_t)out.nodes.size();
            out.nodes.push_back(NodeW());
            queue.push_back(ch[k]);
         }
      }
      nd.count = (uint32_t)nc;
      out.nodes[qi] = nd;
   }
   out.tri_order.swap(packet_order);

   // ---- BFS levels (for the on-device refit, which walks them deepest first)
   {
      std::vector<uint32_t> depth(out.nodes.size(), 0);
      for (size_t i = 0; i < out.nodes.size(); i++)
         for (int k = 0; k < kMaxWidth; k++) {
            uint32_t ch = out.nodes[i].child[k];
            if (ch != kEmptyRef && !(ch & kLeafBit)) depth[ch] = depth[i] + 1;
         }
      out.level_start.clear();
      for (size_t i = 0; i < out.nodes.size(); i++)
         if (i == 0 || depth[i] != depth[i - 1]) out.level_start.push_back((uint32_t)i);
      out.level_start.push_back((uint32_t)out.nodes.size());
      out.max_depth = depth.empty() ? 0 : depth.back();
   }

   // ---- quantise
   if (width == 4) {
      out.cnodes.resize(out.nodes.size());
      quantise_tree(out.nodes, out.cnodes);
   }
}

}  // namespace uh
