// ploc_study.cpp — CPU mirror of the device PLOC builder (csrc/lbvh.hip) for tree-quality experiments: Morton order, nearest
// neighbour within a radius by union-box area, mutual pairs merge, repeat; optionally the rounds stop at K clusters and a
// binned-SAH top-down build over the cluster boxes finishes the tree (what a "PLOC bottom + SAH top" hybrid would give).
// The binary tree is collapsed to 4-wide nodes by the product's rule (open the child with the largest area) and rays are
// walked through it with the kernels' child order (nearest first). Input: the file tools/bvh_visits.py writes.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <numeric>
#include <vector>

struct Box {
   float lo[3], hi[3];
   void reset() { for (int a = 0; a < 3; a++) lo[a] = INFINITY, hi[a] = -INFINITY; }
   void grow(const Box& b) { for (int a = 0; a < 3; a++) lo[a] = std::fmin(lo[a], b.lo[a]), hi[a] = std::fmax(hi[a], b.hi[a]); }
   float area() const { float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return dx * dy + dy * dz + dz * dx; }
};
static float union_area(const Box& a, const Box& b) { Box u = a; u.grow(b); return u.area(); }
struct Ray { float o[3], tmin, d[3], tmax; };
struct N2 { Box box; int l, r; };  // l < 0: leaf, triangle ~l

static uint32_t expand10(uint32_t v) {
   v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu; v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u;
   return v;
}

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

// binned SAH over a set of boxes (ids index `items`), appended to `nodes`; returns the node index
static int sah_build(std::vector<N2>& nodes, const std::vector<Box>& item_box, const std::vector<int>& item_ref, std::vector<int>& idx, int first, int count) {
   Box box, cb;
   box.reset(); cb.reset();
   for (int k = first; k < first + count; k++) {
      const Box& b = item_box[idx[k]];
      box.grow(b);
      float c[3] = {0.5f * (b.lo[0] + b.hi[0]), 0.5f * (b.lo[1] + b.hi[1]), 0.5f * (b.lo[2] + b.hi[2])};
      for (int a = 0; a < 3; a++) cb.lo[a] = std::fmin(cb.lo[a], c[a]), cb.hi[a] = std::fmax(cb.hi[a], c[a]);
   }
   if (count == 1) return item_ref[idx[first]];
   constexpr int NB = 16;
   float best = INFINITY; int ba = -1, bs = -1;
   for (int a = 0; a < 3; a++) {
      float ext = cb.hi[a] - cb.lo[a];
      if (!(ext > 0)) continue;
      Box bb[NB]; int bc[NB];
      for (int b = 0; b < NB; b++) bb[b].reset(), bc[b] = 0;
      float scale = NB / ext;
      for (int k = first; k < first + count; k++) {
         const Box& x = item_box[idx[k]];
         int b = (int)((0.5f * (x.lo[a] + x.hi[a]) - cb.lo[a]) * scale); b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
         bb[b].grow(x); bc[b]++;
      }
      float ra[NB]; int rc[NB]; Box acc; acc.reset(); int c = 0;
      for (int b = NB - 1; b > 0; b--) { acc.grow(bb[b]); c += bc[b]; ra[b] = acc.area(); rc[b] = c; }
      acc.reset(); c = 0;
      for (int b = 0; b < NB - 1; b++) {
         acc.grow(bb[b]); c += bc[b];
         if (!c || !rc[b + 1]) continue;
         float cost = acc.area() * c + ra[b + 1] * rc[b + 1];
         if (cost < best) best = cost, ba = a, bs = b;
      }
   }
   int mid;
   if (ba < 0) mid = first + count / 2;
   else {
      float lo = cb.lo[ba], scale = NB / (cb.hi[ba] - cb.lo[ba]);
      auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [&](int t) {
         const Box& x = item_box[t];
         int b = (int)((0.5f * (x.lo[ba] + x.hi[ba]) - lo) * scale); b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
         return b <= bs;
      });
      mid = (int)(it - idx.begin());
      if (mid == first || mid == first + count) mid = first + count / 2;
   }
   int me = (int)nodes.size();
   nodes.push_back(N2());
   int l = sah_build(nodes, item_box, item_ref, idx, first, mid - first);
   int r = sah_build(nodes, item_box, item_ref, idx, mid, first + count - mid);
   nodes[me].box = box; nodes[me].l = l; nodes[me].r = r;
   return me;
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
   const int radius = argc > 2 ? std::atoi(argv[2]) : 8;
   const int stop_at = argc > 3 ? std::atoi(argv[3]) : 1;  // clusters left when the SAH top takes over (1 = pure PLOC; nt = pure SAH)
   std::vector<Box> tb(nt);
   Box scene; scene.reset();
   for (uint32_t i = 0; i < nt; i++) {
      Box b; b.reset();
      for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) b.lo[a] = std::fmin(b.lo[a], corners[9 * (size_t)i + 3 * k + a]), b.hi[a] = std::fmax(b.hi[a], corners[9 * (size_t)i + 3 * k + a]);
      tb[i] = b; scene.grow(b);
   }
   std::vector<uint64_t> keys(nt);
   for (uint32_t i = 0; i < nt; i++) {
      uint32_t q[3];
      for (int a = 0; a < 3; a++) {
         float c = (corners[9 * (size_t)i + a] + corners[9 * (size_t)i + 3 + a] + corners[9 * (size_t)i + 6 + a]) * (1.0f / 3.0f);
         float t = (c - scene.lo[a]) / (scene.hi[a] - scene.lo[a]); t = std::fmin(std::fmax(t, 0.0f), 1.0f);
         q[a] = (uint32_t)std::fmin(t * 1024.0f, 1023.0f);
      }
      keys[i] = ((uint64_t)((expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2])) << 32) | i;
   }
   std::sort(keys.begin(), keys.end());
   std::vector<N2> nodes;  // binary nodes; refs: >= 0 node index, < 0 leaf ~triangle
   std::vector<Box> cb(nt); std::vector<int> cid(nt);
   for (uint32_t i = 0; i < nt; i++) { uint32_t t = (uint32_t)(keys[i] & 0xffffffffu); cb[i] = tb[t]; cid[i] = ~(int)t; }
   size_t m = nt; int rounds = 0;
   std::vector<int> nn;
   while ((int)m > stop_at) {
      nn.assign(m, 0);
      for (size_t i = 0; i < m; i++) {
         float best = INFINITY; int bj = (int)i;
         for (int d = -radius; d <= radius; d++) {
            long j = (long)i + d;
            if (!d || j < 0 || j >= (long)m) continue;
            float a = union_area(cb[i], cb[j]);
            if (a < best) best = a, bj = (int)j;
         }
         nn[i] = bj;
      }
      size_t w = 0;
      std::vector<Box> nb; std::vector<int> nc; nb.reserve(m); nc.reserve(m);
      for (size_t i = 0; i < m; i++) {
         int j = nn[i];
         if (j != (int)i && nn[j] == (int)i) {
            if ((int)i < j) {
               N2 n; n.box = cb[i]; n.box.grow(cb[j]); n.l = cid[i]; n.r = cid[j];
               nodes.push_back(n);
               nb.push_back(n.box); nc.push_back((int)nodes.size() - 1);
            }
         } else { nb.push_back(cb[i]); nc.push_back(cid[i]); }
      }
      (void)w;
      cb.swap(nb); cid.swap(nc); m = cb.size(); rounds++;
   }
   int root;
   if (m == 1) root = cid[0];
   else {
      std::vector<int> idx(m); std::iota(idx.begin(), idx.end(), 0);
      root = sah_build(nodes, cb, cid, idx, 0, (int)m);
   }
   // optional: tree rotations (Kensler 2008) over the finished binary tree, `rot` passes bottom-up
   const int rot = argc > 4 ? std::atoi(argv[4]) : 0;
   if (rot > 0) {
      auto box_of = [&](int r) -> Box { return r < 0 ? tb[~r] : nodes[r].box; };
      // post-order of internal nodes
      std::vector<int> order;
      {
         std::vector<int> st{root};
         while (!st.empty()) { int v = st.back(); st.pop_back(); if (v < 0) continue; order.push_back(v); st.push_back(nodes[v].l); st.push_back(nodes[v].r); }
      }
      for (int pass = 0; pass < rot; pass++) {
         size_t applied = 0;
         for (size_t oi = order.size(); oi-- > 0;) {
            N2& n = nodes[order[oi]];
            // refit first (children may have changed)
            n.box = box_of(n.l); n.box.grow(box_of(n.r));
            float best = 0; int which = -1;
            // swap n.l with a child of n.r: n.r's box becomes union(n.l, other child)
            for (int side = 0; side < 2; side++) {
               int a = side ? n.r : n.l, b = side ? n.l : n.r;  // a stays a subtree that moves down into b
               if (b < 0) continue;
               for (int c = 0; c < 2; c++) {
                  int keep = c ? nodes[b].l : nodes[b].r;  // stays under b
                  Box nb = box_of(a); nb.grow(box_of(keep));
                  float gain = nodes[b].box.area() - nb.area();
                  if (gain > best) { best = gain; which = side * 2 + c; }
               }
            }
            if (which >= 0) {
               int side = which / 2, c = which % 2;
               int& a = side ? n.r : n.l; int b = side ? n.l : n.r;
               int& moved_up = c ? nodes[b].r : nodes[b].l;  // the child of b that swaps with a
               std::swap(a, moved_up);
               nodes[b].box = box_of(nodes[b].l); nodes[b].box.grow(box_of(nodes[b].r));
               n.box = box_of(n.l); n.box.grow(box_of(n.r));
               applied++;
            }
         }
         std::printf("rotation pass %d: %zu rotations\n", pass, applied);
      }
   }
   double sah = 0; for (auto& n : nodes) sah += n.box.area();
   std::printf("radius %d, SAH top over %zu clusters after %d rounds: %zu binary nodes, summed area %.4g\n", radius, m, rounds, nodes.size(), sah);
   // collapse to 4-wide + walk
   struct N4 { Box b[4]; int c[4]; int n; };
   std::vector<N4> n4; std::vector<int> src{root};
   n4.push_back(N4());
   for (size_t qi = 0; qi < src.size(); qi++) {
      int ch[4]; int nc = 0;
      if (src[qi] < 0) ch[nc++] = src[qi];
      else {
         ch[nc++] = nodes[src[qi]].l; ch[nc++] = nodes[src[qi]].r;
         while (nc < 4) {
            int pick = -1; float pa = -1;
            for (int k = 0; k < nc; k++) if (ch[k] >= 0 && nodes[ch[k]].box.area() > pa) pa = nodes[ch[k]].box.area(), pick = k;
            if (pick < 0) break;
            int c = ch[pick]; ch[pick] = nodes[c].l; ch[nc++] = nodes[c].r;
         }
      }
      N4 o; o.n = nc;
      for (int k = 0; k < nc; k++) {
         if (ch[k] < 0) { o.b[k] = tb[~ch[k]]; o.c[k] = ch[k]; }
         else { o.b[k] = nodes[ch[k]].box; o.c[k] = (int)n4.size(); n4.push_back(N4()); src.push_back(ch[k]); }
      }
      n4[qi] = o;
   }
   for (int any = 0; any < 2; any++) {
      double nv = 0, tv = 0;
      for (uint32_t ri = 0; ri < nr; ri++) {
         const Ray& r = rays[ri];
         const float idir[3] = {1.0f / r.d[0], 1.0f / r.d[1], 1.0f / r.d[2]};
         float best = r.tmax;
         std::vector<int> st; int cur = 0; bool leaf = false, done = false;
         for (;;) {
            if (leaf) {
               tv++;
               float t;
               if (tri_hit(&corners[9 * (size_t)(~cur)], r, t) && t < best) { best = t; if (any) done = true; }
            } else {
               nv++;
               const N4& n = n4[cur];
               float tn[4]; int cr[4]; int nh = 0;
               for (int k = 0; k < n.n; k++) {
                  float t0 = r.tmin, t1 = best;
                  for (int a = 0; a < 3; a++) {
                     float ta = (n.b[k].lo[a] - r.o[a]) * idir[a], tb2 = (n.b[k].hi[a] - r.o[a]) * idir[a];
                     if (ta > tb2) std::swap(ta, tb2);
                     t0 = std::fmax(t0, ta); t1 = std::fmin(t1, tb2);
                  }
                  if (t0 <= t1) tn[nh] = t0, cr[nh++] = n.c[k];
               }
               int near = 0;
               for (int k = 1; k < nh; k++) if (tn[k] < tn[near]) near = k;
               for (int k = nh - 1; k >= 0; k--) if (k != near) st.push_back(cr[k]);
               if (nh) { cur = cr[near]; leaf = cur < 0; continue; }
            }
            if (done || st.empty()) break;
            cur = st.back(); st.pop_back(); leaf = cur < 0;
         }
      }
      std::printf("   %s: nodes/ray %.2f tris/ray %.2f\n", any ? "any-hit" : "closest", nv / nr, tv / nr);
   }
   return 0;
}
