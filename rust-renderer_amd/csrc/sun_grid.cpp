// sun_grid.cpp — host builder of the sun-direction visibility grid (sun_grid.h). Multi-threaded, one pass to count, one to
// fill, one to sort the cells' lists by far depth.
//
// Margins (why the grid never hides a triangle the any-hit tree walk would have found). A sun ray is accepted by a packet
// iff tri_compute<ANY> says so, and tri_compute is Moeller-Trumbore in float: against the exact triangle its edge tests are
// off by the rounding of tv = o - v0, p = d x e2, det and the two dot products. For the edge e (unit direction ê) the
// accepted rays lie within
//     rho_e = c * 2^-24 * (8 |tv| + 7 |e_other|) / sin(angle(d, ê))
// of the exact edge line, measured in the plane perpendicular to d (derivation: DESIGN.md "Sun grid"); with |tv| <= scene
// diameter S this is 2e-6 (S + L) / sin at c = 4. A packet is therefore binned into every cell that meets its projection
// DILATED, per edge, by
//     pad_e = 2e-4 + 2e-5 * max|coordinate| + 2e-6 (S + L) / max(sin, 1e-4)
// (the first two terms are twice the padding the tree's boxes carry - bvh_build.cpp padded() - and cover the float
// evaluation of the ray's own (u, v) and cell index on the device), the dilation being the intersection of the three
// offset half-planes with the padded bounding box: a superset of the true dilated triangle. Packets whose float determinant
// is exactly zero for this direction are left out: tri_compute rejects every ray for them. The far depth wmax carries the
// same base padding, so "wmax < w(origin)" proves that every accepted t would be negative.
#include "sun_grid.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>

namespace uh {
namespace {

struct Proj {
   double px[3], py[3];  // projected corners
   double pad[3];        // per edge k: corner k -> corner (k+1)%3
   double padmax;
   double x0, x1, y0, y1;  // padded bounding box
   float wmax;
   bool use;
   // the packet's plane over the (u, v) plane, w = pa * u + pb * v + pc, when it faces the sun steeply enough to serve as a
   // cover (|normal . W| >= 0.1: its depth is well conditioned), else can_cover = false
   double pa, pb, pc;
   double cover_drop;  // what the fill pass takes off the cover depth: the rays' tmin and the base margins
   bool can_cover;
};

inline float dot_fma_h(const float* a, const float* b) { return std::fmaf(a[2], b[2], std::fmaf(a[1], b[1], a[0] * b[0])); }
inline void cross_fma_h(const float* a, const float* b, float* r) {
   r[0] = std::fmaf(a[1], b[2], -(a[2] * b[1]));
   r[1] = std::fmaf(a[2], b[0], -(a[0] * b[2]));
   r[2] = std::fmaf(a[0], b[1], -(a[1] * b[0]));
}

template <typename F>
void parallel_for(size_t n, int threads, F f) {
   if (threads <= 1 || n < 4096) {
      f(0, n);
      return;
   }
   std::vector<std::thread> th;
   const size_t chunk = (n + threads - 1) / threads;
   for (int t = 0; t < threads; t++) {
      const size_t a = t * chunk, b = std::min(n, a + chunk);
      if (a < b) th.emplace_back([=]() { f(a, b); });
   }
   for (auto& t : th) t.join();
}

// cells of [ix0, ix1] x [iy0, iy1] that the dilated projection meets; border cells stand for everything beyond the grid
// (the device clamps a ray's cell the same way), so they take every packet whose box reaches them
template <typename Emit>
inline void for_cells(const Proj& p, double u0, double v0, double inv, uint32_t nx, uint32_t ny, Emit emit) {
   auto cell = [](double x, double o, double inv_, uint32_t n) {
      double f = std::floor((x - o) * inv_);
      if (!(f >= 0)) f = 0;
      if (f > (double)(n - 1)) f = (double)(n - 1);
      return (uint32_t)f;
   };
   const uint32_t ix0 = cell(p.x0, u0, inv, nx), ix1 = cell(p.x1, u0, inv, nx), iy0 = cell(p.y0, v0, inv, ny), iy1 = cell(p.y1, v0, inv, ny);
   // outward normals of the projected edges (orientation from the signed area); a degenerate projection keeps only its box
   double nxe[3], nye[3], off[3], epad[3];
   int ne = 0;
   const double area2 = (p.px[1] - p.px[0]) * (p.py[2] - p.py[0]) - (p.py[1] - p.py[0]) * (p.px[2] - p.px[0]);
   if (std::fabs(area2) > 1e-300) {
      const double s = area2 > 0 ? 1.0 : -1.0;
      for (int k = 0; k < 3; k++) {
         const int j = (k + 1) % 3;
         const double dx = p.px[j] - p.px[k], dy = p.py[j] - p.py[k], len = std::sqrt(dx * dx + dy * dy);
         if (!(len > 1e-150)) continue;
         nxe[ne] = s * dy / len;
         nye[ne] = -s * dx / len;
         off[ne] = nxe[ne] * p.px[k] + nye[ne] * p.py[k] + p.pad[k];
         epad[ne] = p.pad[k];
         ne++;
      }
   }
   const double cs = 1.0 / inv;
   for (uint32_t iy = iy0; iy <= iy1; iy++) {
      const bool by = iy == 0 || iy == ny - 1;
      const double cy0 = v0 + iy * cs, cy1 = cy0 + cs;
      for (uint32_t ix = ix0; ix <= ix1; ix++) {
         const bool border = by || ix == 0 || ix == nx - 1;
         bool in = true;
         double cover = -INFINITY;
         if (!border) {
            const double cx0 = u0 + ix * cs, cx1 = cx0 + cs;
            bool inside = ne == 3 && p.can_cover;  // the whole cell inside the projection ERODED by twice the margins
            for (int e = 0; e < ne && in; e++) {
               // the cell's corner deepest inside the half-plane n . x <= off
               const double m = std::min(nxe[e] * cx0, nxe[e] * cx1) + std::min(nye[e] * cy0, nye[e] * cy1);
               in = m <= off[e];
               const double M = std::max(nxe[e] * cx0, nxe[e] * cx1) + std::max(nye[e] * cy0, nye[e] * cy1);  // the corner farthest out
               inside = inside && M <= off[e] - 3.0 * epad[e];
            }
            // tmax (sun_grid.h kSunCoverSlack): the packet's depth over the cell may not leave the cover depth by more than the slack
            inside = inside && (std::fabs(p.pa) + std::fabs(p.pb)) * (cs + p.padmax) + p.cover_drop <= kSunCoverSlack;
            if (in && inside) {
               // the packet's nearest depth over the cell (its plane at the four corners, pushed out by the margin)
               const double wa = std::min(p.pa * cx0, p.pa * cx1), wb = std::min(p.pb * cy0, p.pb * cy1);
               cover = wa + wb + p.pc - (std::fabs(p.pa) + std::fabs(p.pb)) * p.padmax;
            }
         }
         if (in) emit(iy * (size_t)nx + ix, cover);
      }
   }
}

}  // namespace

void sun_grid_frame(const float sun_dir[3], SunGridParams& out) {
   int a = 0;
   for (int k = 1; k < 3; k++)
      if (std::fabs(sun_dir[k]) < std::fabs(sun_dir[a])) a = k;
   double ax[3] = {0, 0, 0}, W[3] = {sun_dir[0], sun_dir[1], sun_dir[2]}, U[3], V[3];
   ax[a] = 1.0;
   U[0] = ax[1] * W[2] - ax[2] * W[1];
   U[1] = ax[2] * W[0] - ax[0] * W[2];
   U[2] = ax[0] * W[1] - ax[1] * W[0];
   const double ul = std::sqrt(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]);
   for (int k = 0; k < 3; k++) U[k] /= ul;
   V[0] = W[1] * U[2] - W[2] * U[1];
   V[1] = W[2] * U[0] - W[0] * U[2];
   V[2] = W[0] * U[1] - W[1] * U[0];
   const double vl = std::sqrt(V[0] * V[0] + V[1] * V[1] + V[2] * V[2]);
   for (int k = 0; k < 3; k++) {
      out.U[k] = (float)U[k];
      out.V[k] = (float)(V[k] / vl);
      out.W[k] = sun_dir[k];
   }
}

bool build_sun_grid(const float* packets12, uint32_t n, const float sun_dir[3], const SunGridLimits& lim, int num_threads, SunGridHost& out, const SunGridParams* forced) {
   const auto t_start = std::chrono::steady_clock::now();
   out = SunGridHost();
   auto refuse = [&](const std::string& why) {
      out.why_not = why;
      out.cell_start.clear();
      out.cell_cover.clear();
      out.entries.clear();
      return false;
   };
   if (n == 0) return refuse("no triangles");
   const double wl = std::sqrt((double)sun_dir[0] * sun_dir[0] + (double)sun_dir[1] * sun_dir[1] + (double)sun_dir[2] * sun_dir[2]);
   if (!std::isfinite(wl) || !(wl > 0.99 && wl < 1.01)) return refuse("sun direction is not a finite unit vector");
   if (num_threads < 1) num_threads = 1;

   // ---- frame: W = the direction as given; U, V complete it (rounded to float: the device uses these very numbers)
   {
      SunGridParams fr;
      sun_grid_frame(sun_dir, fr);
      std::memcpy(out.U, fr.U, sizeof(out.U));
      std::memcpy(out.V, fr.V, sizeof(out.V));
      std::memcpy(out.W, fr.W, sizeof(out.W));
   }
   const double U[3] = {out.U[0], out.U[1], out.U[2]}, V[3] = {out.V[0], out.V[1], out.V[2]}, W[3] = {out.W[0], out.W[1], out.W[2]};

   // ---- scene scale: largest finite |coordinate| and the diameter of the finite corners' box
   double maxabs = 0.0, lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
   for (uint32_t i = 0; i < n; i++) {
      const float* q = packets12 + 12 * (size_t)i;
      const double c[3][3] = {{q[0], q[1], q[2]}, {(double)q[0] + q[3], (double)q[1] + q[4], (double)q[2] + q[5]}, {(double)q[0] + q[6], (double)q[1] + q[7], (double)q[2] + q[8]}};
      for (int k = 0; k < 3; k++)
         for (int a = 0; a < 3; a++)
            if (std::isfinite(c[k][a])) {
               maxabs = std::max(maxabs, std::fabs(c[k][a]));
               lo[a] = std::min(lo[a], c[k][a]);
               hi[a] = std::max(hi[a], c[k][a]);
            }
   }
   double S = 0.0;
   for (int a = 0; a < 3; a++)
      if (hi[a] >= lo[a]) S += (hi[a] - lo[a]) * (hi[a] - lo[a]);
   S = std::sqrt(S);
   const double base = 2e-4 + 2e-5 * maxabs + 2e-6 * S;

   // ---- project every packet, with its margins
   std::vector<Proj> pr(n);
   parallel_for(n, num_threads, [&](size_t a, size_t b) {
      for (size_t i = a; i < b; i++) {
         const float* q = packets12 + 12 * i;
         Proj& p = pr[i];
         p.use = false;
         const float e1[3] = {q[3], q[4], q[5]}, e2[3] = {q[6], q[7], q[8]};
         bool finite = true;
         for (int k = 0; k < 9; k++) finite = finite && std::isfinite(q[k]);
         if (!finite) continue;  // NaN / inf corners fail every comparison of the triangle test
         float pf[3];
         cross_fma_h(sun_dir, e2, pf);
         if (dot_fma_h(e1, pf) == 0.0f) continue;  // tri_compute: det == 0 -> false, whatever the origin
         const double c[3][3] = {{q[0], q[1], q[2]}, {(double)q[0] + q[3], (double)q[1] + q[4], (double)q[2] + q[5]}, {(double)q[0] + q[6], (double)q[1] + q[7], (double)q[2] + q[8]}};
         double w = -INFINITY, len3[3], L = 0.0;
         for (int k = 0; k < 3; k++) {
            p.px[k] = U[0] * c[k][0] + U[1] * c[k][1] + U[2] * c[k][2];
            p.py[k] = V[0] * c[k][0] + V[1] * c[k][1] + V[2] * c[k][2];
            w = std::max(w, W[0] * c[k][0] + W[1] * c[k][1] + W[2] * c[k][2]);
         }
         for (int k = 0; k < 3; k++) {
            const int j = (k + 1) % 3;
            len3[k] = std::sqrt((c[j][0] - c[k][0]) * (c[j][0] - c[k][0]) + (c[j][1] - c[k][1]) * (c[j][1] - c[k][1]) + (c[j][2] - c[k][2]) * (c[j][2] - c[k][2]));
            L = std::max(L, len3[k]);
         }
         if (!std::isfinite(L) || !std::isfinite(w)) continue;
         p.padmax = 0.0;
         for (int k = 0; k < 3; k++) {
            const int j = (k + 1) % 3;
            const double l2 = std::sqrt((p.px[j] - p.px[k]) * (p.px[j] - p.px[k]) + (p.py[j] - p.py[k]) * (p.py[j] - p.py[k]));
            const double sn = len3[k] > 0 ? std::min(1.0, l2 / len3[k]) : 1.0;
            p.pad[k] = base + 2e-6 * (S + L) / std::max(sn, 1e-4);
            p.padmax = std::max(p.padmax, p.pad[k]);
         }
         p.x0 = std::min(p.px[0], std::min(p.px[1], p.px[2])) - p.padmax;
         p.x1 = std::max(p.px[0], std::max(p.px[1], p.px[2])) + p.padmax;
         p.y0 = std::min(p.py[0], std::min(p.py[1], p.py[2])) - p.padmax;
         p.y1 = std::max(p.py[0], std::max(p.py[1], p.py[2])) + p.padmax;
         double wm = w + base;
         float wf = (float)wm;
         if ((double)wf < wm) wf = std::nextafterf(wf, INFINITY);
         p.wmax = wf;
         p.use = std::isfinite(p.x0) && std::isfinite(p.x1) && std::isfinite(p.y0) && std::isfinite(p.y1);
         // cover candidate: the plane w(u, v) through the three projected corners
         p.can_cover = false;
         p.cover_drop = 1.01e-3 + 4.0 * base;
         {
            double pw[3];
            for (int k = 0; k < 3; k++) pw[k] = W[0] * c[k][0] + W[1] * c[k][1] + W[2] * c[k][2];
            const double ax = p.px[1] - p.px[0], ay = p.py[1] - p.py[0], bx = p.px[2] - p.px[0], by = p.py[2] - p.py[0];
            const double det2 = ax * by - ay * bx;
            // |normal . W| = projected area / true area
            const double nx3 = (double)q[4] * q[8] - (double)q[5] * q[7], ny3 = (double)q[5] * q[6] - (double)q[3] * q[8], nz3 = (double)q[3] * q[7] - (double)q[4] * q[6];
            const double area3 = std::sqrt(nx3 * nx3 + ny3 * ny3 + nz3 * nz3);
            if (p.use && area3 > 0 && std::fabs(det2) >= 0.1 * area3) {
               const double dw1 = pw[1] - pw[0], dw2 = pw[2] - pw[0];
               p.pa = (dw1 * by - dw2 * ay) / det2;
               p.pb = (dw2 * ax - dw1 * bx) / det2;
               p.pc = pw[0] - p.pa * p.px[0] - p.pb * p.py[0];
               p.can_cover = std::isfinite(p.pa) && std::isfinite(p.pb) && std::isfinite(p.pc);
            }
         }
      }
   });

   // ---- extent: the 0.5 % .. 99.5 % range of the boxes' centres per axis, widened; what lies outside lands in the border cells
   std::vector<double> cx, cy;
   cx.reserve(n);
   cy.reserve(n);
   for (uint32_t i = 0; i < n; i++)
      if (pr[i].use) {
         cx.push_back(0.5 * (pr[i].x0 + pr[i].x1));
         cy.push_back(0.5 * (pr[i].y0 + pr[i].y1));
      }
   if (cx.empty()) return refuse("no triangle can occlude a ray of this direction");
   auto quantile = [](std::vector<double>& v, double q) {
      size_t k = (size_t)(q * (v.size() - 1));
      std::nth_element(v.begin(), v.begin() + k, v.end());
      return v[k];
   };
   double ex0 = quantile(cx, 0.005), ex1 = quantile(cx, 0.995), ey0 = quantile(cy, 0.005), ey1 = quantile(cy, 0.995);
   {
      const double mx = 0.05 * (ex1 - ex0) + 4 * base, my = 0.05 * (ey1 - ey0) + 4 * base;
      ex0 -= mx;
      ex1 += mx;
      ey0 -= my;
      ey1 += my;
   }
   const double ext_x = ex1 - ex0, ext_y = ey1 - ey0;
   if (!(ext_x > 0) || !(ext_y > 0) || !std::isfinite(ext_x) || !std::isfinite(ext_y)) return refuse("degenerate projected extent");

   // ---- cell size: the finest grid whose estimated entry count meets the target (about 24 entries per triangle; a finer grid
   // shortens every list, a coarser one saves memory), within the cell budget
   const size_t used = cx.size();
   const double target = std::min((double)lim.max_entries * 0.5, std::max(2.0e6, lim.entries_per_triangle * (double)used));
   const size_t stride = used > 400000 ? used / 200000 : 1;
   auto estimate = [&](double s) {
      double e = 0.0;
      for (uint32_t i = 0; i < n; i += (uint32_t)stride) {
         const Proj& p = pr[i];
         if (!p.use) continue;
         const double bw = std::min(p.x1, ex1) - std::max(p.x0, ex0), bh = std::min(p.y1, ey1) - std::max(p.y0, ey0);
         if (bw < 0 || bh < 0) {
            e += 1.0;
            continue;
         }
         e += 0.5 * (bw / s) * (bh / s) + (bw + bh) / s + 1.0;
      }
      return e * (double)stride;
   };
   double s_lo = std::sqrt(ext_x * ext_y / (double)lim.max_cells) * 1.001, s_hi = std::max(ext_x, ext_y);
   if (estimate(s_lo) > target) {
      for (int it = 0; it < 48; it++) {
         const double mid = std::sqrt(s_lo * s_hi);
         (estimate(mid) > target ? s_lo : s_hi) = mid;
      }
   } else {
      s_hi = s_lo;
   }
   double cell = s_hi;

   std::vector<uint32_t> counts;
   uint64_t total = 0;
   for (int attempt = 0;; attempt++) {
      out.nx = (uint32_t)std::ceil(ext_x / cell) + 2;  // + the two border columns
      out.ny = (uint32_t)std::ceil(ext_y / cell) + 2;
      out.inv_cell = (float)(1.0 / cell);
      double inv = out.inv_cell;  // the device's number
      out.u0 = (float)(ex0 - 1.0 / inv);
      out.v0 = (float)(ey0 - 1.0 / inv);
      if (forced) {
         out.nx = forced->nx;
         out.ny = forced->ny;
         out.inv_cell = forced->inv_cell;
         out.u0 = forced->u0;
         out.v0 = forced->v0;
         inv = out.inv_cell;
      }
      const size_t ncell = (size_t)out.nx * out.ny;
      counts.assign(ncell + 1, 0u);
      parallel_for(n, num_threads, [&](size_t a, size_t b) {
         for (size_t i = a; i < b; i++)
            if (pr[i].use) for_cells(pr[i], out.u0, out.v0, inv, out.nx, out.ny, [&](size_t c, double) { __atomic_fetch_add(&counts[c], 1u, __ATOMIC_RELAXED); });
      });
      total = 0;
      for (size_t c = 0; c < ncell; c++) total += counts[c];
      if (total <= lim.max_entries) break;
      if (attempt >= 10 || forced) return refuse("entry budget exceeded");
      cell *= 1.3;
   }

   {  // lists too long to beat the tree walk: known after the count pass - no fill, no sort
      uint64_t occupied = 0;
      for (size_t c = 0; c < (size_t)out.nx * out.ny; c++) occupied += counts[c] ? 1 : 0;
      out.mean_list = occupied ? (double)total / (double)occupied : 0.0;
      if (out.mean_list > lim.max_mean_list) {
         out.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
         return refuse("lists too long for this direction (mean " + std::to_string(out.mean_list) + " entries per occupied cell)");
      }
   }
   // ---- offsets, fill, sort
   const size_t ncell = (size_t)out.nx * out.ny;
   out.cell_start.resize(ncell + 1);
   {
      uint64_t run = 0;
      for (size_t c = 0; c < ncell; c++) {
         out.cell_start[c] = (uint32_t)run;
         run += counts[c];
      }
      out.cell_start[ncell] = (uint32_t)run;
      if (run > 0xfffffff0ull) return refuse("entry budget exceeded");
   }
   out.entries.resize(total);
   std::vector<uint32_t> cursor(out.cell_start.begin(), out.cell_start.end() - 1);
   std::vector<uint32_t> cover_key(ncell, 0u);  // key 0 = below every float: no cover
   {
      const double inv = out.inv_cell;
      parallel_for(n, num_threads, [&](size_t a, size_t b) {
         for (size_t i = a; i < b; i++)
            if (pr[i].use)
               for_cells(pr[i], out.u0, out.v0, inv, out.nx, out.ny, [&](size_t c, double cover) {
                  const uint32_t at = __atomic_fetch_add(&cursor[c], 1u, __ATOMIC_RELAXED);
                  out.entries[at] = SunGridEntry{(uint32_t)i, pr[i].wmax};
                  if (cover > -INFINITY) {
                     // a ray of this cell that starts below `cover` (less the ray's tmin and the margins) is occluded by this packet
                     // whatever else the cell lists: keep the highest such depth (atomic max on order-preserving keys)
                     double cw = cover - pr[i].cover_drop;
                     float cf = (float)cw;
                     if ((double)cf > cw) cf = std::nextafterf(cf, -INFINITY);
                     uint32_t bits;
                     std::memcpy(&bits, &cf, 4);
                     const uint32_t key = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
                     uint32_t seen = __atomic_load_n(&cover_key[c], __ATOMIC_RELAXED);
                     while (key > seen && !__atomic_compare_exchange_n(&cover_key[c], &seen, key, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                     }
                  }
               });
      });
   }
   uint64_t nonempty = 0;
   uint32_t longest = 0;
   for (size_t c = 0; c < ncell; c++) {
      const uint32_t len = out.cell_start[c + 1] - out.cell_start[c];
      nonempty += len ? 1 : 0;
      longest = std::max(longest, len);
   }
   parallel_for(ncell, num_threads, [&](size_t a, size_t b) {
      for (size_t c = a; c < b; c++) {
         SunGridEntry* first = out.entries.data() + out.cell_start[c];
         SunGridEntry* last = out.entries.data() + out.cell_start[c + 1];
         if (last - first > 1)
            std::sort(first, last, [](const SunGridEntry& x, const SunGridEntry& y) { return x.wmax > y.wmax || (x.wmax == y.wmax && x.packet < y.packet); });
      }
   });
   out.mean_list = nonempty ? (double)total / (double)nonempty : 0.0;
   out.cell_cover.resize(ncell);
   uint64_t covered = 0;
   for (size_t c = 0; c < ncell; c++) {
      const uint32_t key = cover_key[c];
      float f = -INFINITY;
      if (key) {
         const uint32_t bits = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
         std::memcpy(&f, &bits, 4);
         covered++;
      }
      out.cell_cover[c] = f;
   }
   out.covered_cells = covered;
   // Where do sun rays start? On surfaces - so weigh every triangle by its area and ask whether the cell under its centre would
   // serve a ray well (an interior cell with a short list) or hand it to the tree (a border cell: everything beyond the dense
   // part of the scene, e.g. a ground plane around a detailed object; or a long list: walls seen edge-on). A grid that sends a
   // large share of the surface to the tree only adds its look-up to the tree walk: refused as a whole.
   {
      double area_all = 0.0, area_bad = 0.0;
      const double inv = out.inv_cell;
      for (uint32_t i = 0; i < n; i++) {
         const Proj& p = pr[i];
         if (!p.use) continue;
         const float* q = packets12 + 12 * (size_t)i;
         const double e1[3] = {q[3], q[4], q[5]}, e2[3] = {q[6], q[7], q[8]};
         const double cx3 = e1[1] * e2[2] - e1[2] * e2[1], cy3 = e1[2] * e2[0] - e1[0] * e2[2], cz3 = e1[0] * e2[1] - e1[1] * e2[0];
         const double area = 0.5 * std::sqrt(cx3 * cx3 + cy3 * cy3 + cz3 * cz3);
         if (!std::isfinite(area)) continue;
         double fx = std::floor(((p.px[0] + p.px[1] + p.px[2]) / 3.0 - out.u0) * inv), fy = std::floor(((p.py[0] + p.py[1] + p.py[2]) / 3.0 - out.v0) * inv);
         fx = !(fx >= 0) ? 0 : (fx > out.nx - 1 ? out.nx - 1 : fx);
         fy = !(fy >= 0) ? 0 : (fy > out.ny - 1 ? out.ny - 1 : fy);
         const uint32_t ix = (uint32_t)fx, iy = (uint32_t)fy;
         const size_t c = (size_t)iy * out.nx + ix;
         const bool border = ix == 0 || iy == 0 || ix == out.nx - 1 || iy == out.ny - 1;
         area_all += area;
         if (border || out.cell_start[c + 1] - out.cell_start[c] > lim.max_walk) area_bad += area;
      }
      out.fallback_area = area_all > 0 ? area_bad / area_all : 0.0;
   }
   out.max_list = longest;
   out.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
   if (out.fallback_area > lim.max_fallback_area)
      return refuse("too much of the scene's surface (" + std::to_string(out.fallback_area) + ") lies beyond the dense extent or in cells with long lists: its rays would walk the tree anyway");
   if (out.mean_list > lim.max_mean_list) return refuse("lists too long for this direction (mean " + std::to_string(out.mean_list) + " entries per occupied cell)");
   return true;
}

}  // namespace uh
