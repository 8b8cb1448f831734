// sun_grid_check.cpp - CPU test of the sun-direction visibility grid (product code: csrc/sun_grid.cpp), run under ASan + UBSan.
// For random and adversarial triangle soups and sun directions it replays, on the host and in the kernel's float arithmetic,
// what k_trace_sun_grid does for a ray (project, clamp to a cell, walk the cell's list until the far-depth break) and compares
// the verdict with a brute-force any-hit over ALL triangles through the same Moeller-Trumbore arithmetic (tri_compute<ANY> of
// kernels.hip, restated with std::fmaf). Rays are the hard ones: origins on surfaces, rays through shared edges and vertices
// (exactly and a few ulps off), edge-on triangles, slivers, huge / tiny / non-finite triangles, axis-aligned directions.
// Build: g++ -std=c++17 -O1 -ffp-contract=off -mfma -fsanitize=address,undefined ... sun_grid_check.cpp ../../rust-renderer_amd/csrc/sun_grid.cpp
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sun_grid.h"

using namespace uh;

static uint32_t rng_state = 0x1234567u;
static float rnd() {
   rng_state = rng_state * 747796405u + 2891336453u;
   uint32_t w = ((rng_state >> ((rng_state >> 28) + 4u)) ^ rng_state) * 277803737u;
   w = (w >> 22) ^ w;
   return (float)w * 2.3283064365386963e-10f;
}

struct F3 {
   float x, y, z;
};
static float dot_fma(F3 a, F3 b) { return std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)); }
static F3 cross_fma(F3 a, F3 b) { return F3{std::fmaf(a.y, b.z, -(a.z * b.y)), std::fmaf(a.z, b.x, -(a.x * b.z)), std::fmaf(a.x, b.y, -(a.y * b.x))}; }

// tri_compute<ANY = true> of kernels.hip with tmin 0.001, tmax 10000, tlimit inf
// solid = true additionally ignores packets whose plane contains the direction to within 1e-5 rad: for a ray that lies in such a
// plane the test divides rounding noise by rounding noise, and which of those packets gets asked (hence the verdict) differs
// between any two conservative culling schemes - tree or grid. Shadow rays leave surfaces displaced along the normal
// (view.glsl offsetRay), so they do not lie in planes; the check keeps such rays, but scores them on the solid packets only.
static bool accepts(const float* q, F3 o, F3 d, bool solid = false) {
   F3 v0{q[0], q[1], q[2]}, e1{q[3], q[4], q[5]}, e2{q[6], q[7], q[8]};
   F3 p = cross_fma(d, e2);
   float det = dot_fma(e1, p);
   if (det == 0.0f) return false;
   if (solid && std::fabs(det) <= 1e-5f * std::sqrt(dot_fma(e1, e1) * dot_fma(e2, e2))) return false;
   float inv = 1.0f / det;
   F3 tv{o.x - v0.x, o.y - v0.y, o.z - v0.z};
   float u = dot_fma(tv, p) * inv;
   if (!(u >= 0.0f && u <= 1.0f)) return false;
   F3 qq = cross_fma(tv, e1);
   float v = dot_fma(d, qq) * inv;
   if (!(v >= 0.0f && u + v <= 1.0f)) return false;
   float t = dot_fma(e2, qq) * inv;
   if (!(t > 0.001f)) return false;
   return t < 10000.0f && t <= INFINITY;
}

// What a padded-box tree asks before it tests a packet: does the ray meet the packet's own padded box within (tmin, tmax)?
// (padding and slab arithmetic of oracle.cpp build_bvh / slab.) A ray that lies IN the plane of a triangle whose plane contains
// the direction makes Moeller-Trumbore divide rounding noise by rounding noise; such a packet can "accept" a ray it lies wholly
// behind. No tree ever asks it (its box is behind the origin), so the reference verdict is "accepted by a packet whose box the
// ray meets" - the predicate the tree walk of k_trace_shadow and the oracle's own tree both compute.
static bool meets_box(const float* q, F3 o, F3 d) {
   float lo[3], hi[3];
   for (int a = 0; a < 3; a++) {
      const float c0 = q[a], c1 = q[a] + q[3 + a], c2 = q[a] + q[6 + a];
      float l = std::fmin(c0, std::fmin(c1, c2)), h = std::fmax(c0, std::fmax(c1, c2));
      const float pad = 1e-4f + 1e-5f * std::fmax(std::fabs(l), std::fabs(h));
      lo[a] = l - pad;
      hi[a] = h + pad;
   }
   const float id[3] = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z}, oo[3] = {o.x, o.y, o.z};
   float tn = 0.001f, tf = 10000.0f;
   for (int a = 0; a < 3; a++) {
      const float t0 = (lo[a] - oo[a]) * id[a], t1 = (hi[a] - oo[a]) * id[a];
      tn = std::fmax(tn, std::fmin(t0, t1));
      tf = std::fmin(tf, std::fmax(t0, t1));
   }
   return tn <= tf * 1.0000005f + 1e-30f;
}

// build_sun_coarse_cover (sun_grid_build.hip k_sg_coarse) on the host: the lowest cover depth of each block of cells, none where a
// cell of the block has none or where the block's cover depths lie more than kSunCoarseSpread apart
static std::vector<float> coarse_cover;
static uint32_t coarse_shift = 2, coarse_nx = 0;
static uint64_t coarse_answers = 0;
static void build_coarse(const SunGridHost& g) {
   const uint32_t b = 1u << coarse_shift, cnx = (g.nx + b - 1) / b, cny = (g.ny + b - 1) / b;
   coarse_nx = cnx;
   coarse_cover.assign((size_t)cnx * cny, -INFINITY);
   for (uint32_t by = 0; by < cny; by++)
      for (uint32_t bx = 0; bx < cnx; bx++) {
         float lo = INFINITY, hi = -INFINITY;
         for (uint32_t y = by << coarse_shift; y < ((by + 1) << coarse_shift) && y < g.ny; y++)
            for (uint32_t x = bx << coarse_shift; x < ((bx + 1) << coarse_shift) && x < g.nx; x++) {
               const float c = g.cell_cover[(size_t)y * g.nx + x];
               lo = c < lo ? c : lo;
               hi = c > hi ? c : hi;
            }
         coarse_cover[(size_t)by * cnx + bx] = (lo > -INFINITY && hi - lo <= uh::kSunCoarseSpread) ? lo : -INFINITY;
      }
}

// k_trace_sun_grid for one ray, on the host
static bool grid_occluded(const SunGridHost& g, const std::vector<float>& pk, F3 o, F3 d, uint32_t* tests, bool solid = false) {
   const float pu = dot_fma(F3{g.U[0], g.U[1], g.U[2]}, o), pv = dot_fma(F3{g.V[0], g.V[1], g.V[2]}, o), pw = dot_fma(F3{g.W[0], g.W[1], g.W[2]}, o);
   float fx = (pu - g.u0) * g.inv_cell, fy = (pv - g.v0) * g.inv_cell;
   const float max_x = (float)(g.nx - 1), max_y = (float)(g.ny - 1);
   fx = !(fx >= 0.0f) ? 0.0f : fx;
   fy = !(fy >= 0.0f) ? 0.0f : fy;
   fx = fx > max_x ? max_x : fx;
   fy = fy > max_y ? max_y : fy;
   const uint32_t cell = (uint32_t)fy * g.nx + (uint32_t)fx;
   {  // the coarse cover first, as the kernel asks it (sun_grid.h kSunCoarseReach)
      const float c = coarse_cover[(size_t)((uint32_t)fy >> coarse_shift) * coarse_nx + ((uint32_t)fx >> coarse_shift)];
      if (pw < c && c - pw < uh::kSunCoarseReach) {
         coarse_answers++;
         return true;
      }
   }
   if (pw < g.cell_cover[cell] && g.cell_cover[cell] - pw < uh::kSunCoverReach) return true;  // the cell's cover: no packet is asked (k_trace_sun_grid does the same)
   for (uint32_t e = g.cell_start[cell]; e < g.cell_start[cell + 1]; e++) {
      if (g.entries[e].wmax < pw) break;
      (*tests)++;
      if (accepts(&pk[12 * (size_t)g.entries[e].packet], o, d, solid)) return true;
   }
   return false;
}

static void add_tri(std::vector<float>& pk, F3 a, F3 b, F3 c) {
   const float q[12] = {a.x, a.y, a.z, b.x - a.x, b.y - a.y, b.z - a.z, c.x - a.x, c.y - a.y, c.z - a.z, 0, 0, 0};
   pk.insert(pk.end(), q, q + 12);
}

static F3 normalised(F3 v) {
   // the host's normalize(view.sun_dir): v * (1 / sqrt(dot)) in float (context.hip make_params)
   float d = (v.x * v.x + v.y * v.y) + v.z * v.z;
   float inv = 1.0f / std::sqrt(d);
   return F3{v.x * inv, v.y * inv, v.z * inv};
}

static float nudge(float x, int ulps) {
   for (int i = 0; i < (ulps < 0 ? -ulps : ulps); i++) x = std::nextafterf(x, ulps < 0 ? -INFINITY : INFINITY);
   return x;
}

static int failures = 0;
static uint32_t tri_stride = 1;  // argv[1]: surface origins from every k-th triangle (the check is O(triangles x rays))
static int num_suns = 9;           // argv[2]
static uint64_t rays_checked = 0, occluded_rays = 0, grid_tests = 0, noise_accepts = 0, in_plane_noise = 0;

static void check_scene(const char* name, const std::vector<float>& pk, F3 sun, const std::vector<F3>& extra_origins, int random_origins, float lo, float hi, bool expect_grid) {
   const uint32_t n = (uint32_t)(pk.size() / 12);
   const F3 d = normalised(sun);
   const float dir[3] = {d.x, d.y, d.z};
   SunGridHost g;
   SunGridLimits lim;
   lim.max_mean_list = 1e9;
   lim.max_fallback_area = 1.0;  // this check is about what the walk finds, not about when the grid is worth building (checked at the end)
   const bool ok = build_sun_grid(pk.data(), n, dir, lim, 3, g);
   if (!ok) {
      std::printf("%-34s grid refused: %s%s\n", name, g.why_not.c_str(), expect_grid ? "   <-- UNEXPECTED" : "");
      if (expect_grid) failures++;
      return;
   }
   if (g.cell_start.size() != (size_t)g.nx * g.ny + 1 || g.cell_start.back() != g.entries.size()) {
      std::printf("%s: malformed offsets\n", name);
      failures++;
      return;
   }
   build_coarse(g);
   for (size_t c = 0; c + 1 < g.cell_start.size(); c++)
      for (uint32_t e = g.cell_start[c]; e + 1 < g.cell_start[c + 1]; e++)
         if (g.entries[e].wmax < g.entries[e + 1].wmax || g.entries[e].packet >= n) {
            std::printf("%s: list of cell %zu is not sorted by descending far depth\n", name, c);
            failures++;
            return;
         }
   std::vector<F3> origins = extra_origins;
   for (int i = 0; i < random_origins; i++) origins.push_back(F3{lo + rnd() * (hi - lo), lo + rnd() * (hi - lo), lo + rnd() * (hi - lo)});
   // origins ON the surfaces (what a shadow ray's origin is), pushed back along the ray so that the ray meets the surface again,
   // and rays aimed exactly at edges and corners of triangles, and a few ulps beside them
   for (uint32_t i = 0; i < n; i += tri_stride) {
      const float* q = &pk[12 * (size_t)i];
      const F3 v0{q[0], q[1], q[2]}, e1{q[3], q[4], q[5]}, e2{q[6], q[7], q[8]};
      const float bary[7][2] = {{0, 0}, {1, 0}, {0, 1}, {0.5f, 0}, {0, 0.5f}, {0.5f, 0.5f}, {rnd() * 0.5f, rnd() * 0.5f}};
      for (auto& b : bary) {
         F3 p{v0.x + b[0] * e1.x + b[1] * e2.x, v0.y + b[0] * e1.y + b[1] * e2.y, v0.z + b[0] * e1.z + b[1] * e2.z};
         const float back = 0.01f + rnd() * 3.0f;
         F3 o{p.x - back * d.x, p.y - back * d.y, p.z - back * d.z};
         origins.push_back(o);
         origins.push_back(p);
         {  // a shadow ray's real origin: the hit point moved off the surface (view.glsl offsetRay: ~1e-4 .. 1e-5 along the normal)
            F3 nrm = cross_fma(e1, e2);
            const float nl = std::sqrt(dot_fma(nrm, nrm));
            if (nl > 0) {
               const float off = (rnd() < 0.5f ? 1.0f : -1.0f) * (1.0f / 65536.0f + rnd() * 1e-4f) / nl;
               origins.push_back(F3{p.x + off * nrm.x, p.y + off * nrm.y, p.z + off * nrm.z});
            }
         }
         origins.push_back(F3{nudge(o.x, (int)(rnd() * 9) - 4), nudge(o.y, (int)(rnd() * 9) - 4), nudge(o.z, (int)(rnd() * 9) - 4)});
      }
   }
   uint32_t tests = 0, bad = 0;
   for (const F3& o : origins) {
      bool brute = false, raw = false;
      for (uint32_t i = 0; i < n && !brute; i++) {
         const bool acc = accepts(&pk[12 * (size_t)i], o, d);
         raw = raw || acc;
         brute = acc && meets_box(&pk[12 * (size_t)i], o, d);
      }
      const bool viagrid = grid_occluded(g, pk, o, d, &tests);
      rays_checked++;
      occluded_rays += brute;
      noise_accepts += (raw && !brute);
      if (brute != viagrid) {
         // in-plane noise, or a real difference? score the ray again on the solid packets only
         bool b2 = false;
         uint32_t unused = 0;
         for (uint32_t i = 0; i < n && !b2; i++) b2 = accepts(&pk[12 * (size_t)i], o, d, true) && meets_box(&pk[12 * (size_t)i], o, d);
         const bool g2 = grid_occluded(g, pk, o, d, &unused, true);
         if (b2 == g2) {
            in_plane_noise++;
         } else if (bad++ < 5) {
            std::printf("%s: MISMATCH at origin (%.9g, %.9g, %.9g): reference %d, grid %d\n", name, o.x, o.y, o.z, (int)b2, (int)g2);
         }
      }
   }
   grid_tests += tests;
   failures += bad;
   std::printf("%-34s %7u tris  grid %4u x %-4u  %8zu entries  mean list %6.2f  max %5u  %7zu rays  %5.2f tests/ray  build %6.1f ms  %s\n", name, n, g.nx, g.ny,
               g.entries.size(), g.mean_list, g.max_list, origins.size(), (double)tests / (double)origins.size(), g.build_ms, bad ? "FAILED" : "ok");
}

// a tessellated quad patch with shared edges (rays through the seams must not leak)
static void add_patch(std::vector<float>& pk, F3 o, F3 eu, F3 ev, int nu, int nv, float bump) {
   std::vector<F3> P((size_t)(nu + 1) * (nv + 1));
   for (int i = 0; i <= nu; i++)
      for (int j = 0; j <= nv; j++) {
         const float s = (float)i / nu, t = (float)j / nv, h = bump * std::sin(7.0f * s) * std::cos(5.0f * t);
         F3 nrm = cross_fma(eu, ev);
         const float nl = std::sqrt(dot_fma(nrm, nrm));
         P[(size_t)i * (nv + 1) + j] = F3{o.x + s * eu.x + t * ev.x + h * nrm.x / nl, o.y + s * eu.y + t * ev.y + h * nrm.y / nl, o.z + s * eu.z + t * ev.z + h * nrm.z / nl};
      }
   for (int i = 0; i < nu; i++)
      for (int j = 0; j < nv; j++) {
         const F3 a = P[(size_t)i * (nv + 1) + j], b = P[(size_t)(i + 1) * (nv + 1) + j], c = P[(size_t)(i + 1) * (nv + 1) + j + 1], dd = P[(size_t)i * (nv + 1) + j + 1];
         add_tri(pk, a, b, c);
         add_tri(pk, a, c, dd);
      }
}

int main(int argc, char** argv) {
   if (argc > 1) tri_stride = (uint32_t)std::atoi(argv[1]) > 0 ? (uint32_t)std::atoi(argv[1]) : 1u;
   if (argc > 2) num_suns = std::atoi(argv[2]);
   const F3 all_suns[] = {{0.0f, 0.9f, 0.15f}, {0, 1, 0}, {1, 0, 0}, {0, 0, -1}, {0.3f, 0.5f, -0.8f}, {-0.6f, 0.2f, 0.1f}, {1e-4f, 1.0f, 0.0f}, {0.70710678f, 0.70710678f, 0.0f}, {-0.2f, -0.9f, 0.4f}};
   std::vector<F3> suns(all_suns, all_suns + (num_suns < 1 ? 1 : (num_suns > 9 ? 9 : num_suns)));
   // 1. an atrium-like set of patches: floor, walls (edge-on for a vertical sun), a roof with a gap, a column ring
   {
      std::vector<float> pk;
      add_patch(pk, F3{-4, 0, -3}, F3{8, 0, 0}, F3{0, 0, 6}, 24, 18, 0.01f);
      add_patch(pk, F3{-4, 0, -3}, F3{8, 0, 0}, F3{0, 5, 0}, 24, 14, 0.02f);
      add_patch(pk, F3{-4, 0, 3}, F3{0, 0, -6}, F3{0, 5, 0}, 18, 14, 0.0f);
      add_patch(pk, F3{-4, 5, -3}, F3{3, 0, 0}, F3{0, 0, 6}, 10, 18, 0.03f);
      add_patch(pk, F3{1, 5, -3}, F3{3, 0, 0}, F3{0, 0, 6}, 10, 18, 0.0f);
      for (int k = 0; k < 12; k++) {
         const float a0 = 6.2831853f * k / 12, a1 = 6.2831853f * (k + 1) / 12;
         add_patch(pk, F3{0.5f * std::cos(a0), 0, 0.5f * std::sin(a0)}, F3{0.5f * (std::cos(a1) - std::cos(a0)), 0, 0.5f * (std::sin(a1) - std::sin(a0))}, F3{0, 4, 0}, 1, 16, 0.0f);
      }
      for (const F3& s : suns) check_scene("atrium patches", pk, s, {}, 2000, -5.0f, 6.0f, true);
   }
   // 2. random soup of mixed sizes incl. slivers
   {
      std::vector<float> pk;
      for (int i = 0; i < 3000; i++) {
         const F3 c{rnd() * 20 - 10, rnd() * 20 - 10, rnd() * 20 - 10};
         const float sz = i % 7 == 0 ? 4.0f : (i % 5 == 0 ? 0.01f : 0.4f);
         F3 a{c.x + (rnd() - 0.5f) * sz, c.y + (rnd() - 0.5f) * sz, c.z + (rnd() - 0.5f) * sz};
         F3 b{c.x + (rnd() - 0.5f) * sz, c.y + (rnd() - 0.5f) * sz, c.z + (rnd() - 0.5f) * sz};
         F3 cc{c.x + (rnd() - 0.5f) * sz, c.y + (rnd() - 0.5f) * sz, c.z + (rnd() - 0.5f) * sz};
         if (i % 11 == 0) cc = F3{a.x + (b.x - a.x) * 0.5f + 1e-5f * rnd(), a.y + (b.y - a.y) * 0.5f, a.z + (b.z - a.z) * 0.5f + 1e-5f * rnd()};  // sliver
         add_tri(pk, a, b, cc);
      }
      for (const F3& s : suns) check_scene("random soup with slivers", pk, s, {}, 2000, -11.0f, 11.0f, true);
   }
   // 3. torture: coincident quads, zero-area, huge, tiny, fan, non-finite corner, triangles whose plane contains the direction
   {
      std::vector<float> pk;
      const F3 q0{-1, -1, 0}, q1{1, -1, 0}, q2{1, 1, 0}, q3{-1, 1, 0};
      for (int rep = 0; rep < 3; rep++) {
         add_tri(pk, q0, q1, q2);
         add_tri(pk, q0, q2, q3);
      }
      add_tri(pk, F3{0, 0, 1}, F3{0, 0, 1}, F3{0, 0, 1});
      add_tri(pk, F3{0, 0, 1}, F3{1, 0, 1}, F3{2, 0, 1});
      add_tri(pk, F3{-1e5f, -1e5f, -3}, F3{1e5f, -1e5f, -3}, F3{0, 1e5f, -3});
      add_tri(pk, F3{0.25f, 0.25f, 0.5f}, F3{0.250001f, 0.25f, 0.5f}, F3{0.25f, 0.250001f, 0.5f});
      for (int k = 0; k < 12; k++) add_tri(pk, F3{0, 0, 2}, F3{std::cos(0.5236f * k), std::sin(0.5236f * k), 2}, F3{std::cos(0.5236f * (k + 1)), std::sin(0.5236f * (k + 1)), 2});
      add_tri(pk, F3{NAN, 0, 0}, F3{1, 0, 0}, F3{0, 1, 0});
      add_tri(pk, F3{0, 0, 0}, F3{INFINITY, 0, 0}, F3{0, 1, 0});
      add_tri(pk, F3{3, 0, 0}, F3{3, 4, 0}, F3{3, 4, 1});  // contains the y axis
      std::vector<F3> extra = {{NAN, 0, 0}, {INFINITY, 0, 0}, {0, -INFINITY, 0}, {1e30f, 1e30f, 1e30f}, {0, 0, -1e4f}, {0, 0, -2.999f}, {0, 0, -3.0f}};
      for (const F3& s : suns) check_scene("torture set", pk, s, extra, 3000, -4.0f, 4.0f, false);
   }
   // 4. one big wall exactly edge-on to the sun (every triangle's plane contains the direction) + a floor
   {
      std::vector<float> pk;
      add_patch(pk, F3{0, 0, -5}, F3{0, 10, 0}, F3{0, 0, 10}, 40, 40, 0.0f);
      add_patch(pk, F3{-5, 0, -5}, F3{10, 0, 0}, F3{0, 0, 10}, 20, 20, 0.0f);
      check_scene("edge-on wall, vertical sun", pk, F3{0, 1, 0}, {}, 3000, -6.0f, 11.0f, false);
      check_scene("edge-on wall, grazing sun", pk, F3{1e-3f, 1, 0}, {}, 3000, -6.0f, 11.0f, false);
   }
   // 4b. deeper than the rays' tmax along the sun: a floor, a roof 20,000 units above it, a mid-air slab; origins nearer
   // and farther than tmax = 10000 below each covering patch - the cover shortcut must not call a ray occluded whose occluder lies
   // beyond tmax (the tree walk and the per-packet test reject t >= 10000)
   {
      std::vector<float> pk;
      add_patch(pk, F3{-4, 0, -4}, F3{8, 0, 0}, F3{0, 0, 8}, 16, 16, 0.0f);
      add_patch(pk, F3{-40, 20000, -40}, F3{80, 0, 0}, F3{0, 0, 80}, 2, 2, 0.0f);  // big triangles: the margins at |y| = 20000 are ~0.5, and a cover must contain its cell eroded by three of them
      add_patch(pk, F3{-20, 12000, -20}, F3{40, 0, 0}, F3{0, 0, 40}, 1, 1, 0.0f);
      std::vector<F3> extra;
      for (int i = 0; i < 400; i++) {
         const float x = -3.9f + 7.8f * rnd(), z = -3.9f + 7.8f * rnd();
         for (float y : {1.0f, 1999.0f, 2001.0f, 9990.0f, 9999.5f, 10000.5f, 10010.0f, 11000.0f, 11999.0f, 19990.0f, 19999.9f}) extra.push_back(F3{x, y, z});
      }
      check_scene("deeper than tmax, vertical sun", pk, F3{0, 1, 0}, extra, 500, -4.0f, 4.0f, false);
      check_scene("deeper than tmax, tilted sun", pk, F3{1e-4f, 1, 2e-4f}, extra, 500, -4.0f, 4.0f, false);
   }
   // 5. refusals: empty scene, degenerate directions
   {
      SunGridHost g;
      SunGridLimits lim;
      const float z[3] = {0, 0, 0}, nanv[3] = {NAN, 0, 1}, ok[3] = {0, 1, 0};
      std::vector<float> one;
      add_tri(one, F3{0, 0, 0}, F3{1, 0, 0}, F3{0, 0, 1});
      if (build_sun_grid(one.data(), 1, z, lim, 1, g) || build_sun_grid(one.data(), 1, nanv, lim, 1, g) || build_sun_grid(nullptr, 0, ok, lim, 1, g)) {
         std::printf("a degenerate request was not refused\n");
         failures++;
      }
      lim.max_fallback_area = 1.0;
      if (!build_sun_grid(one.data(), 1, ok, lim, 1, g)) {
         std::printf("single triangle refused: %s\n", g.why_not.c_str());
         failures++;
      }
      lim.max_entries = 4;  // budget far too small for a soup
      std::vector<float> soup;
      for (int i = 0; i < 500; i++) add_tri(soup, F3{rnd(), rnd(), rnd()}, F3{rnd(), rnd(), rnd()}, F3{rnd(), rnd(), rnd()});
      if (build_sun_grid(soup.data(), 500, ok, lim, 2, g) && g.entries.size() > 4) {
         std::printf("entry budget ignored\n");
         failures++;
      }
   }
   {
      // when is the grid refused as a whole? A detailed object on a ground plane far larger than it: most of the surface (by area)
      // lies beyond the dense extent, its rays would all be handed to the tree. A uniformly tessellated room: accepted.
      std::vector<float> object, room;
      add_patch(object, F3{-200, 0, -200}, F3{400, 0, 0}, F3{0, 0, 400}, 8, 8, 0.0f);
      for (int k = 0; k < 6; k++) add_patch(object, F3{-1.0f + 0.3f * k, 0.2f, -1}, F3{0.25f, 0, 0}, F3{0, 1.5f, 2}, 24, 24, 0.02f);
      add_patch(room, F3{-4, 0, -3}, F3{8, 0, 0}, F3{0, 0, 6}, 40, 30, 0.01f);
      add_patch(room, F3{-4, 3, -3}, F3{8, 0, 0}, F3{0, 0, 6}, 40, 30, 0.01f);
      add_patch(room, F3{-4, 0, -3}, F3{8, 0, 0}, F3{0, 3, 0}, 40, 15, 0.01f);
      SunGridHost g;
      SunGridLimits lim;
      const F3 d = normalised(F3{0.0f, 0.9f, 0.15f});
      const float dir[3] = {d.x, d.y, d.z};
      const bool small_on_huge = build_sun_grid(object.data(), (uint32_t)(object.size() / 12), dir, lim, 2, g);
      std::printf("object on a huge ground plane: %s (surface beyond the grid's reach: %.2f)\n", small_on_huge ? "grid built" : g.why_not.c_str(), g.fallback_area);
      const bool uniform = build_sun_grid(room.data(), (uint32_t)(room.size() / 12), dir, lim, 2, g);
      std::printf("uniform room: %s (surface beyond the grid's reach: %.2f)\n", uniform ? "grid built" : g.why_not.c_str(), g.fallback_area);
      if (small_on_huge || !uniform) {
         std::printf("the refusal heuristic misjudged a scene\n");
         failures++;
      }
   }
   std::printf("%llu rays checked (%llu occluded; %llu more 'accepted' only by packets whose box they never meet; %llu rays lying in the plane of an edge-on packet scored on the "
               "solid packets), %.2f grid tests per ray, %llu rays answered by the coarse cover\n", (unsigned long long)rays_checked, (unsigned long long)occluded_rays, (unsigned long long)noise_accepts,
               (unsigned long long)in_plane_noise, (double)grid_tests / (double)(rays_checked ? rays_checked : 1), (unsigned long long)coarse_answers);
   if (failures) {
      std::printf("SUN GRID CHECK FAILED (%d)\n", failures);
      return 1;
   }
   std::printf("SUN GRID CHECK OK\n");
   return 0;
}
