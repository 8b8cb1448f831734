// sun_grid_build.hip — the sun-direction grid of sun_grid.h built ON THE DEVICE, from the triangle packets where they lie
// (round 4). The host builder (sun_grid.cpp: 130-550 ms per sun direction, more than the 64 frames of the headline image
// take) stays as the reference implementation: its margins are held against brute force by tests/cpp/sun_grid_check.cpp, and
// this file restates its per-packet arithmetic - same double-precision expressions in the same order - so that for the same
// grid parameters the two builders bin every packet into the same cells and find the same cover depths
// (tests/test_gpu_sun_grid.py::test_device_built_grid_equals_the_host_built_one reads the device grid back and compares).
//
//   bounds    one pass over the packets: largest |coordinate|, box of the finite corners         -> margins (host, 7 numbers)
//   project   per packet: the (u, v) projection, per-edge pads, padded box, far depth, cover plane (SgProj, device-resident)
//   sample    every k-th packet's box to the host -> extent (0.5 % .. 99.5 % of the centres) and cell size (the host builder's
//             bisection on its estimate, over the sample)
//   count     one WAVE per packet: its lanes walk the cells of the packet's box, test each against the three dilated edges,
//             one atomicAdd per (packet, cell) pair
//   scan      device_scan.h, in place -> cell_start; total and occupied cells to the host (refusals: budget, mean list)
//   fill      the same walk: entry = (packet, far depth) at an atomic cursor per cell; the cell's cover depth by atomicMax on
//             order-preserving keys
//   sort      a thread per cell: insertion sort by far depth, descending (ties: packet index) - only the cells a ray may walk
//             (interior, at most max_walk entries: k_trace_sun_grid hands the rays of every other cell to the tree)
//   area      share of the surface whose cell would hand its rays to the tree (refusal), cell records (offset | cover)
// Nothing but a few dozen numbers crosses PCIe; the entries never leave the device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bvh.h"
#include "camera_grid.h"
#include "device_scan.h"
#include "sun_grid.h"

namespace uh {
namespace {

constexpr int kBlock = 256;

struct SgFrame {
   double U[3], V[3], W[3];
   float sun[3];
   double base, S;
};

__device__ __forceinline__ unsigned long long key_of(double x) {
   const unsigned long long u = (unsigned long long)__double_as_longlong(x);
   return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
inline double double_of(unsigned long long k) {
   const unsigned long long u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
   double d;
   std::memcpy(&d, &u, 8);
   return d;
}

__device__ __forceinline__ double wave_max(double x) {
   for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o));
   return x;
}

// keys[0] = max |coordinate|, keys[1..3] = max of -lo, keys[4..6] = max of hi over the finite corners (0 = none seen)
__global__ __launch_bounds__(kBlock) void k_sg_bounds(const float4* __restrict__ tris, uint32_t n, unsigned long long* __restrict__ keys) {
   double m[7];
   for (int k = 0; k < 7; k++) m[k] = -INFINITY;
   for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
      const float4 a = tris[kTriStride16 * (size_t)i], b = tris[kTriStride16 * (size_t)i + 1], c4 = tris[kTriStride16 * (size_t)i + 2];
      const float q[9] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c4.x};
      const double c[3][3] = {{q[0], q[1], q[2]}, {(double)q[0] + q[3], (double)q[1] + q[4], (double)q[2] + q[5]}, {(double)q[0] + q[6], (double)q[1] + q[7], (double)q[2] + q[8]}};
      for (int k = 0; k < 3; k++)
         for (int ax = 0; ax < 3; ax++)
            if (isfinite(c[k][ax])) {
               m[0] = fmax(m[0], fabs(c[k][ax]));
               m[1 + ax] = fmax(m[1 + ax], -c[k][ax]);
               m[4 + ax] = fmax(m[4 + ax], c[k][ax]);
            }
   }
   // one atomic per block and value (thousands of waves on seven words serialise in the L2)
   __shared__ double s_m[kBlock / 64][7];
   for (int k = 0; k < 7; k++) {
      const double w = wave_max(m[k]);
      if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6][k] = w;
   }
   __syncthreads();
   if (threadIdx.x < 7) {
      double w = s_m[0][threadIdx.x];
      for (int v = 1; v < kBlock / 64; v++) w = fmax(w, s_m[v][threadIdx.x]);
      if (w > -INFINITY) atomicMax(&keys[threadIdx.x], key_of(w));
   }
}

__device__ __forceinline__ float dot_fma_d(const float* a, const float* b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }

// sun_grid.cpp "project every packet, with its margins", expression by expression
__global__ __launch_bounds__(kBlock) void k_sg_project(const float4* __restrict__ tris, uint32_t n, SgFrame fr, SgProj* __restrict__ out) {
   const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
   if (i >= n) return;
   const float4 a = tris[kTriStride16 * (size_t)i], b = tris[kTriStride16 * (size_t)i + 1], c4 = tris[kTriStride16 * (size_t)i + 2];
   const float q[9] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c4.x};
   SgProj p;
   memset(&p, 0, sizeof(p));
   p.flags = 0;
   const float e1[3] = {q[3], q[4], q[5]}, e2[3] = {q[6], q[7], q[8]};
   bool finite = true;
   for (int k = 0; k < 9; k++) finite = finite && isfinite(q[k]);
   bool live = finite;
   if (live) {
      float pf[3];
      pf[0] = fmaf(fr.sun[1], e2[2], -(fr.sun[2] * e2[1]));
      pf[1] = fmaf(fr.sun[2], e2[0], -(fr.sun[0] * e2[2]));
      pf[2] = fmaf(fr.sun[0], e2[1], -(fr.sun[1] * e2[0]));
      if (dot_fma_d(e1, pf) == 0.0f) live = false;  // tri_compute: det == 0 -> false, whatever the origin
   }
   if (live) {
      const double c[3][3] = {{q[0], q[1], q[2]}, {(double)q[0] + q[3], (double)q[1] + q[4], (double)q[2] + q[5]}, {(double)q[0] + q[6], (double)q[1] + q[7], (double)q[2] + q[8]}};
      double w = -INFINITY, len3[3], L = 0.0;
      for (int k = 0; k < 3; k++) {
         p.px[k] = fr.U[0] * c[k][0] + fr.U[1] * c[k][1] + fr.U[2] * c[k][2];
         p.py[k] = fr.V[0] * c[k][0] + fr.V[1] * c[k][1] + fr.V[2] * c[k][2];
         w = fmax(w, fr.W[0] * c[k][0] + fr.W[1] * c[k][1] + fr.W[2] * c[k][2]);
      }
      for (int k = 0; k < 3; k++) {
         const int j = (k + 1) % 3;
         len3[k] = sqrt((c[j][0] - c[k][0]) * (c[j][0] - c[k][0]) + (c[j][1] - c[k][1]) * (c[j][1] - c[k][1]) + (c[j][2] - c[k][2]) * (c[j][2] - c[k][2]));
         L = fmax(L, len3[k]);
      }
      if (isfinite(L) && isfinite(w)) {
         p.padmax = 0.0;
         for (int k = 0; k < 3; k++) {
            const int j = (k + 1) % 3;
            const double l2 = sqrt((p.px[j] - p.px[k]) * (p.px[j] - p.px[k]) + (p.py[j] - p.py[k]) * (p.py[j] - p.py[k]));
            const double sn = len3[k] > 0 ? fmin(1.0, l2 / len3[k]) : 1.0;
            p.pad[k] = fr.base + 2e-6 * (fr.S + L) / fmax(sn, 1e-4);
            p.padmax = fmax(p.padmax, p.pad[k]);
         }
         p.x0 = fmin(p.px[0], fmin(p.px[1], p.px[2])) - p.padmax;
         p.x1 = fmax(p.px[0], fmax(p.px[1], p.px[2])) + p.padmax;
         p.y0 = fmin(p.py[0], fmin(p.py[1], p.py[2])) - p.padmax;
         p.y1 = fmax(p.py[0], fmax(p.py[1], p.py[2])) + p.padmax;
         const double wm = w + fr.base;
         float wf = (float)wm;
         if ((double)wf < wm) wf = nextafterf(wf, INFINITY);
         p.wmax = wf;
         const bool use = isfinite(p.x0) && isfinite(p.x1) && isfinite(p.y0) && isfinite(p.y1);
         p.cover_drop = 1.01e-3 + 4.0 * fr.base;
         bool can_cover = false;
         {
            double pw[3];
            for (int k = 0; k < 3; k++) pw[k] = fr.W[0] * c[k][0] + fr.W[1] * c[k][1] + fr.W[2] * c[k][2];
            const double ax = p.px[1] - p.px[0], ay = p.py[1] - p.py[0], bx = p.px[2] - p.px[0], by = p.py[2] - p.py[0];
            const double det2 = ax * by - ay * bx;
            const double nx3 = (double)q[4] * q[8] - (double)q[5] * q[7], ny3 = (double)q[5] * q[6] - (double)q[3] * q[8], nz3 = (double)q[3] * q[7] - (double)q[4] * q[6];
            const double area3 = sqrt(nx3 * nx3 + ny3 * ny3 + nz3 * nz3);
            if (use && area3 > 0 && fabs(det2) >= 0.1 * area3) {
               const double dw1 = pw[1] - pw[0], dw2 = pw[2] - pw[0];
               p.pa = (dw1 * by - dw2 * ay) / det2;
               p.pb = (dw2 * ax - dw1 * bx) / det2;
               p.pc = pw[0] - p.pa * p.px[0] - p.pb * p.py[0];
               can_cover = isfinite(p.pa) && isfinite(p.pb) && isfinite(p.pc);
            }
         }
         p.flags = (use ? 1u : 0u) | (can_cover ? 2u : 0u);
      }
   }
   out[i] = p;
}

// the camera grid's projection: camera_grid.h pg_project_packet, one thread per packet
__global__ __launch_bounds__(kBlock) void k_pg_project(const float4* __restrict__ tris, uint32_t n, PgCam cam, SgProj* __restrict__ out) {
   const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
   if (i >= n) return;
   const float4 a4 = tris[kTriStride16 * (size_t)i], b4 = tris[kTriStride16 * (size_t)i + 1], c4 = tris[kTriStride16 * (size_t)i + 2];
   const float q[9] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w, c4.x};
   SgProj p;
   pg_project_packet(q, cam, p);
   out[i] = p;
}

// every stride-th packet's padded box (NaN x0: not in use), for the host's choice of extent and cell size
__global__ __launch_bounds__(kBlock) void k_sg_sample(const SgProj* __restrict__ pr, uint32_t n, uint32_t stride, uint32_t n_samples, double* __restrict__ out) {
   const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
   if (k >= n_samples) return;
   const size_t i = (size_t)k * stride;
   const SgProj& p = pr[i < n ? i : n - 1];
   const bool use = i < n && (p.flags & 1u);
   out[4 * (size_t)k + 0] = use ? p.x0 : NAN;
   out[4 * (size_t)k + 1] = p.x1;
   out[4 * (size_t)k + 2] = p.y0;
   out[4 * (size_t)k + 3] = p.y1;
}

// sun_grid.cpp for_cells, one wave per packet: the lanes share the packet's edge set-up and walk the cells of its box
template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_sg_bin(const SgProj* __restrict__ pr, uint32_t n, SgGrid g, uint32_t* __restrict__ counts_or_cursor, SunGridEntry* __restrict__ entries,
                                                   uint32_t* __restrict__ cover_key) {
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t waves = gridDim.x * (kBlock / 64);
   for (uint32_t i = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); i < n; i += waves) {
      const SgProj& p = pr[i];
      if (!(p.flags & 1u)) continue;
      const uint32_t ix0 = sg_cell_of(p.x0, g.u0, g.inv, g.nx), ix1 = sg_cell_of(p.x1, g.u0, g.inv, g.nx), iy0 = sg_cell_of(p.y0, g.v0, g.inv, g.ny), iy1 = sg_cell_of(p.y1, g.v0, g.inv, g.ny);
      double nxe[3], nye[3], off[3], epad[3];
      const int ne = sg_packet_edges(p, nxe, nye, off, epad);
      const double cs = 1.0 / g.inv;
      const uint32_t bw = ix1 - ix0 + 1, bh = iy1 - iy0 + 1;
      const uint64_t cells = (uint64_t)bw * bh;
      const bool can_cover = (p.flags & 2u) != 0;
      for (uint64_t j = lane; j < cells; j += 64) {
         const uint32_t ix = ix0 + (uint32_t)(j % bw), iy = iy0 + (uint32_t)(j / bw);
         const bool border = iy == 0 || iy == g.ny - 1 || ix == 0 || ix == g.nx - 1;
         bool in = true;
         double cover = -INFINITY;
         if (!border) {
            const double cy0 = g.v0 + iy * cs, cy1 = cy0 + cs;
            const double cx0 = g.u0 + ix * cs, cx1 = cx0 + cs;
            bool inside = ne == 3 && can_cover;
            in = sg_cell_touches(ne, nxe, nye, off, epad, cx0, cx1, cy0, cy1, inside);
            inside = inside && (fabs(p.pa) + fabs(p.pb)) * (cs + p.padmax) + p.cover_drop <= kSunCoverSlack;
            if (in && inside) {
               const double wa = fmin(p.pa * cx0, p.pa * cx1), wb = fmin(p.pb * cy0, p.pb * cy1);
               cover = wa + wb + p.pc - (fabs(p.pa) + fabs(p.pb)) * p.padmax;
            }
         }
         if (!in) continue;
         const size_t c = iy * (size_t)g.nx + ix;
         if (!FILL) {
            atomicAdd(&counts_or_cursor[c], 1u);
         } else {
            const uint32_t at = atomicAdd(&counts_or_cursor[c], 1u);
            entries[at] = SunGridEntry{i, p.wmax};
            if (cover > -INFINITY) {
               const double cw = cover - p.cover_drop;
               float cf = (float)cw;
               if ((double)cf > cw) cf = nextafterf(cf, -INFINITY);
               const uint32_t bits = __float_as_uint(cf);
               const uint32_t key = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
               atomicMax(&cover_key[c], key);
            }
         }
      }
   }
}

// occupied cells and the longest list, from the counts (before the scan turns them into offsets)
// longest[0]: over all cells; longest[1]: over the interior cells (the ones a ray can be in: the border ring takes what projects
// beyond the raster), nx x ny being the raster
__global__ __launch_bounds__(kBlock) void k_sg_occupancy(const uint32_t* __restrict__ counts, uint32_t ncell, uint32_t nx, uint32_t ny, unsigned long long* __restrict__ occupied,
                                                         uint32_t* __restrict__ longest) {
   uint32_t occ = 0, mx = 0, mi = 0;
   for (uint32_t c = blockIdx.x * kBlock + threadIdx.x; c < ncell; c += gridDim.x * kBlock) {
      const uint32_t v = counts[c];
      occ += v ? 1u : 0u;
      mx = v > mx ? v : mx;
      const uint32_t ix = c % nx, iy = c / nx;
      if (ix != 0 && iy != 0 && ix + 1 != nx && iy + 1 != ny) mi = v > mi ? v : mi;
   }
   for (int o = 32; o > 0; o >>= 1) {
      occ += __shfl_xor(occ, o);
      const uint32_t other = __shfl_xor(mx, o), other_i = __shfl_xor(mi, o);
      mx = other > mx ? other : mx;
      mi = other_i > mi ? other_i : mi;
   }
   __shared__ uint32_t s_occ[kBlock / 64], s_mx[kBlock / 64], s_mi[kBlock / 64];
   if ((threadIdx.x & 63) == 0) {
      s_occ[threadIdx.x >> 6] = occ;
      s_mx[threadIdx.x >> 6] = mx;
      s_mi[threadIdx.x >> 6] = mi;
   }
   __syncthreads();
   if (threadIdx.x == 0) {  // one atomic per block
      for (int v = 1; v < kBlock / 64; v++) {
         occ += s_occ[v];
         mx = s_mx[v] > mx ? s_mx[v] : mx;
         mi = s_mi[v] > mi ? s_mi[v] : mi;
      }
      if (occ) atomicAdd(occupied, (unsigned long long)occ);
      if (mx) atomicMax(longest, mx);
      if (mi) atomicMax(longest + 1, mi);
   }
}

// the lists a ray may walk (interior cells of at most max_walk entries), by far depth descending, ties by packet index
__global__ __launch_bounds__(kBlock) void k_sg_sort(const uint32_t* __restrict__ start, SunGridEntry* __restrict__ entries, uint32_t nx, uint32_t ny, uint32_t max_walk) {
   const uint32_t ncell = nx * ny;
   for (uint32_t c = blockIdx.x * kBlock + threadIdx.x; c < ncell; c += gridDim.x * kBlock) {
      const uint32_t ix = c % nx, iy = c / nx;
      if (ix == 0 || iy == 0 || ix == nx - 1 || iy == ny - 1) continue;
      const uint32_t a = start[c], len = start[c + 1] - a;
      if (len < 2 || len > max_walk) continue;
      SunGridEntry* e = entries + a;
      for (uint32_t k = 1; k < len; k++) {
         const SunGridEntry x = e[k];
         uint32_t j = k;
         while (j > 0 && (e[j - 1].wmax < x.wmax || (e[j - 1].wmax == x.wmax && e[j - 1].packet > x.packet))) {
            e[j] = e[j - 1];
            j--;
         }
         e[j] = x;
      }
   }
}

// the records k_trace_sun_grid reads: offset | cover depth (float bits), ncell + 1 of them
// (round 5: four words - the list's FIRST entry rides in the cell record (packet index, far depth; kEmptyRef for an empty list), so that
// a walk over the plain lists asks for its first packet straight from the cell record: one dependent round trip less, as with the
// 64-byte records, for 8 more bytes per cell)
__global__ __launch_bounds__(kBlock) void k_sg_cells(const uint32_t* __restrict__ start, const uint32_t* __restrict__ cover_key, const SunGridEntry* __restrict__ entries, uint32_t ncell,
                                                     uint32_t* __restrict__ cells) {
   for (uint32_t c = blockIdx.x * kBlock + threadIdx.x; c <= ncell; c += gridDim.x * kBlock) {
      uint32_t bits = 0xff800000u;  // -inf: no cover
      if (c < ncell && cover_key[c]) {
         const uint32_t key = cover_key[c];
         bits = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
      }
      SunGridEntry first{kEmptyRef, 0.0f};
      if (c < ncell && start[c + 1] > start[c]) first = entries[start[c]];
      cells[kSunCellWords * (size_t)c] = start[c];
      cells[kSunCellWords * (size_t)c + 1] = bits;
      cells[kSunCellWords * (size_t)c + 2] = first.packet;
      cells[kSunCellWords * (size_t)c + 3] = __float_as_uint(first.wmax);
   }
}

// sun_grid.cpp's last check: the share of the surface (by area) whose cell hands its rays to the tree. out[0] = all, out[1] = bad
__global__ __launch_bounds__(kBlock) void k_sg_area(const float4* __restrict__ tris, const SgProj* __restrict__ pr, uint32_t n, SgGrid g, const uint32_t* __restrict__ start, uint32_t max_walk,
                                                    double* __restrict__ out) {
   double all = 0.0, bad = 0.0;
   for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
      const SgProj& p = pr[i];
      if (!(p.flags & 1u)) continue;
      const float4 a = tris[kTriStride16 * (size_t)i], b = tris[kTriStride16 * (size_t)i + 1], c4 = tris[kTriStride16 * (size_t)i + 2];
      const double e1[3] = {a.w, b.x, b.y}, e2[3] = {b.z, b.w, c4.x};
      const double cx3 = e1[1] * e2[2] - e1[2] * e2[1], cy3 = e1[2] * e2[0] - e1[0] * e2[2], cz3 = e1[0] * e2[1] - e1[1] * e2[0];
      const double area = 0.5 * sqrt(cx3 * cx3 + cy3 * cy3 + cz3 * cz3);
      if (!isfinite(area)) continue;
      double fx = floor(((p.px[0] + p.px[1] + p.px[2]) / 3.0 - g.u0) * g.inv), fy = floor(((p.py[0] + p.py[1] + p.py[2]) / 3.0 - g.v0) * g.inv);
      fx = !(fx >= 0) ? 0 : (fx > g.nx - 1 ? g.nx - 1 : fx);
      fy = !(fy >= 0) ? 0 : (fy > g.ny - 1 ? g.ny - 1 : fy);
      const uint32_t ix = (uint32_t)fx, iy = (uint32_t)fy;
      const size_t c = (size_t)iy * g.nx + ix;
      const bool border = ix == 0 || iy == 0 || ix == g.nx - 1 || iy == g.ny - 1;
      all += area;
      if (border || (start && start[c + 1] - start[c] > max_walk)) bad += area;  // start == null: the border's share alone (known before the count pass)
   }
   for (int o = 32; o > 0; o >>= 1) {
      all += __shfl_xor(all, o);
      bad += __shfl_xor(bad, o);
   }
   if ((threadIdx.x & 63) == 0) {
      atomicAdd(&out[0], all);
      atomicAdd(&out[1], bad);
   }
}

// SunGridDev::recs: every entry with its packet in one 64-byte record
__global__ __launch_bounds__(kBlock) void k_sg_inline(const float4* __restrict__ tris, const SunGridEntry* __restrict__ entries, uint64_t n, float4* __restrict__ out) {
   for (uint64_t e = (uint64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (uint64_t)gridDim.x * kBlock) {
      const SunGridEntry en = entries[e];
      const float next = e + 1 < n ? entries[e + 1].wmax : -INFINITY;  // of the next cell's first entry at a list's end: the walk knows where its list ends
      const float4 a = tris[kTriStride16 * (size_t)en.packet], b = tris[kTriStride16 * (size_t)en.packet + 1], c = tris[kTriStride16 * (size_t)en.packet + 2];
      out[4 * e] = a;
      out[4 * e + 1] = b;
      out[4 * e + 2] = make_float4(c.x, c.y, en.wmax, next);
   }
}

// SunGridDev::coarse: the lowest cover depth of each block of cells (sun_grid.h kSunCoarseSpread)
__global__ __launch_bounds__(kBlock) void k_sg_coarse(const uint32_t* __restrict__ cells, uint32_t nx, uint32_t ny, uint32_t shift, uint32_t cnx, uint32_t cny, float* __restrict__ out) {
   const uint32_t b = 1u << shift;
   for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < cnx * cny; k += gridDim.x * kBlock) {
      const uint32_t bx = (k % cnx) << shift, by = (k / cnx) << shift;
      float lo = INFINITY, hi = -INFINITY;
      for (uint32_t y = by; y < by + b && y < ny; y++)
         for (uint32_t x = bx; x < bx + b && x < nx; x++) {
            const float c = __uint_as_float(cells[kSunCellWords * ((size_t)y * nx + x) + 1]);
            lo = c < lo ? c : lo;  // a NaN never gets in: the builders write depths or -inf
            hi = c > hi ? c : hi;
         }
      out[k] = (lo > -INFINITY && hi - lo <= kSunCoarseSpread) ? lo : -INFINITY;
   }
}

struct Scratch {
   void* p = nullptr;
   hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
   ~Scratch() {
      if (p) (void)hipFree(p);
   }
};

}  // namespace

int build_sun_inline_records(void* stream_v, const void* d_packets, const SunGridEntry* d_entries, uint64_t num_entries, void* out) {
   if (num_entries == 0) return (int)hipSuccess;
   const uint32_t blocks = (uint32_t)std::min<uint64_t>((num_entries + kBlock - 1) / kBlock, 1u << 16);
   k_sg_inline<<<blocks, kBlock, 0, (hipStream_t)stream_v>>>((const float4*)d_packets, d_entries, num_entries, (float4*)out);
   return (int)hipGetLastError();
}

int build_sun_coarse_cover(void* stream_v, const uint32_t* d_cells, uint32_t nx, uint32_t ny, uint32_t shift, float* out) {
   const uint32_t b = 1u << shift, cnx = (nx + b - 1) / b, cny = (ny + b - 1) / b;
   if (cnx * cny == 0) return (int)hipSuccess;
   k_sg_coarse<<<std::min<uint32_t>((cnx * cny + kBlock - 1) / kBlock, 4096u), kBlock, 0, (hipStream_t)stream_v>>>(d_cells, nx, ny, shift, cnx, cny, out);
   return (int)hipGetLastError();
}

void SunGridDevice::release() {
   if (cells) (void)hipFree(cells);
   if (entries) (void)hipFree(entries);
   cells = nullptr;
   entries = nullptr;
}

// sun grid (cam == null: frame from sun_dir, raster chosen from a sample of the boxes unless forced) or camera grid (cam given:
// k_pg_project, the raster is the frame itself)
static bool build_grid_impl(void* stream_v, const void* d_packets, uint32_t n, const float* sun_dir, const PgCam* cam, const SunGridLimits& lim, const SunGridParams* forced, SunGridDevice& out) {
   hipStream_t stream = (hipStream_t)stream_v;
   const float4* d_tris = (const float4*)d_packets;
   const auto t_start = std::chrono::steady_clock::now();
   out.release();
   out = SunGridDevice();
   auto refuse = [&](const std::string& why) {
      out.release();
      out.why_not = why;
      out.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
      return false;
   };
#define SG_TRY(expr)                                                                                                      \
   do {                                                                                                                   \
      hipError_t e_ = (expr);                                                                                             \
      if (e_ == hipErrorOutOfMemory) {  /* no room for the grid is a refusal - the tree walk serves - not a failed frame */  \
         (void)hipGetLastError();                                                                                          \
         return refuse(std::string("no device memory for the grid (") + #expr + ")");                                      \
      }                                                                                                                    \
      if (e_ != hipSuccess) return refuse(std::string("device build: ") + #expr + ": " + hipGetErrorString(e_));          \
   } while (0)
   if (n == 0) return refuse("no triangles");
   SunGridParams prm;
   SgFrame fr{};
   const uint32_t blocks_n = (n + kBlock - 1) / kBlock;
   Scratch d_proj;
   SG_TRY(d_proj.alloc((size_t)n * sizeof(SgProj)));
   SgProj* pr = (SgProj*)d_proj.p;
   double base = 0.0;
   SunGridParams cam_raster;
   if (cam) {
      k_pg_project<<<blocks_n, kBlock, 0, stream>>>(d_tris, n, *cam, pr);
      cam_raster.u0 = -1.0f;
      cam_raster.v0 = -1.0f;
      cam_raster.inv_cell = 1.0f;
      cam_raster.nx = (uint32_t)cam->W + 2;
      cam_raster.ny = (uint32_t)cam->H + 2;
      forced = &cam_raster;
   } else {
      const double wl = std::sqrt((double)sun_dir[0] * sun_dir[0] + (double)sun_dir[1] * sun_dir[1] + (double)sun_dir[2] * sun_dir[2]);
      if (!std::isfinite(wl) || !(wl > 0.99 && wl < 1.01)) return refuse("sun direction is not a finite unit vector");
      sun_grid_frame(sun_dir, prm);  // U, V, W: the host builder's frame (sun_grid.cpp)
      for (int k = 0; k < 3; k++) {
         fr.U[k] = prm.U[k];
         fr.V[k] = prm.V[k];
         fr.W[k] = prm.W[k];
         fr.sun[k] = sun_dir[k];
      }

      // ---- scene scale
      Scratch d_keys;
      SG_TRY(d_keys.alloc(7 * sizeof(unsigned long long)));
      SG_TRY(hipMemsetAsync(d_keys.p, 0, 7 * sizeof(unsigned long long), stream));
      k_sg_bounds<<<std::min<uint32_t>(blocks_n, 512), kBlock, 0, stream>>>(d_tris, n, (unsigned long long*)d_keys.p);
      unsigned long long keys[7];
      SG_TRY(hipStreamSynchronize(stream));
      SG_TRY(hipMemcpy(keys, d_keys.p, sizeof(keys), hipMemcpyDeviceToHost));  // (blocking: the destination is on this frame's stack / in a local vector)
      const double maxabs = keys[0] ? double_of(keys[0]) : 0.0;
      double S = 0.0;
      for (int a = 0; a < 3; a++)
         if (keys[1 + a] && keys[4 + a]) {
            const double lo = -double_of(keys[1 + a]), hi = double_of(keys[4 + a]);
            if (hi >= lo) S += (hi - lo) * (hi - lo);
         }
      S = std::sqrt(S);
      fr.S = S;
      fr.base = 2e-4 + 2e-5 * maxabs + 2e-6 * S;
      base = fr.base;

      // ---- projection
      k_sg_project<<<blocks_n, kBlock, 0, stream>>>(d_tris, n, fr, pr);
   }

   // ---- extent and cell size from a sample of the boxes (at most 8 Ki of them)
   double ex0, ex1, ey0, ey1, cell;
   if (forced) {
      const SunGridParams frame = prm;
      prm = *forced;
      if (!cam) {  // a forced sun raster keeps the direction's frame (the camera grid has none)
         std::memcpy(prm.U, frame.U, sizeof(prm.U));
         std::memcpy(prm.V, frame.V, sizeof(prm.V));
         std::memcpy(prm.W, frame.W, sizeof(prm.W));
      }
   } else {
      // (8 Ki boxes, 32 bisection steps: the host's share of the build stays under a millisecond; 32 Ki x 48 cost more than the kernels)
      const uint32_t stride = n > 8192 ? (n + 8191) / 8192 : 1, n_samples = (n + stride - 1) / stride;
      Scratch d_sample;
      SG_TRY(d_sample.alloc((size_t)n_samples * 4 * sizeof(double)));
      k_sg_sample<<<(n_samples + kBlock - 1) / kBlock, kBlock, 0, stream>>>(pr, n, stride, n_samples, (double*)d_sample.p);
      std::vector<double> box(4 * (size_t)n_samples);
      SG_TRY(hipStreamSynchronize(stream));
      SG_TRY(hipMemcpy(box.data(), d_sample.p, box.size() * sizeof(double), hipMemcpyDeviceToHost));  // (blocking: the destination is on this frame's stack / in a local vector)
      std::vector<double> cx, cy;
      for (uint32_t k = 0; k < n_samples; k++)
         if (box[4 * (size_t)k] == box[4 * (size_t)k]) {
            cx.push_back(0.5 * (box[4 * (size_t)k] + box[4 * (size_t)k + 1]));
            cy.push_back(0.5 * (box[4 * (size_t)k + 2] + box[4 * (size_t)k + 3]));
         }
      if (cx.empty()) return refuse("no triangle can occlude a ray of this direction");
      auto quantile = [](std::vector<double>& v, double q) {
         size_t k = (size_t)(q * (v.size() - 1));
         std::nth_element(v.begin(), v.begin() + k, v.end());
         return v[k];
      };
      double qq = 0.005;
      ex0 = quantile(cx, qq), ex1 = quantile(cx, 1.0 - qq), ey0 = quantile(cy, qq), ey1 = quantile(cy, 1.0 - qq);
      const double mx = 0.05 * (ex1 - ex0) + 4 * base, my = 0.05 * (ey1 - ey0) + 4 * base;
      ex0 -= mx;
      ex1 += mx;
      ey0 -= my;
      ey1 += my;
      const double ext_x = ex1 - ex0, ext_y = ey1 - ey0;
      if (!(ext_x > 0) || !(ext_y > 0) || !std::isfinite(ext_x) || !std::isfinite(ext_y)) return refuse("degenerate projected extent");
      // the host builder's bisection on its estimate of the entry count, over the sample (each sampled box stands for `stride`)
      const double used = (double)cx.size() * stride;
      const double target = std::min((double)lim.max_entries * 0.5, std::max(2.0e6, lim.entries_per_triangle * used));
      auto estimate = [&](double s) {
         double e = 0.0;
         for (uint32_t k = 0; k < n_samples; k++) {
            const double* b = &box[4 * (size_t)k];
            if (!(b[0] == b[0])) continue;
            const double bw = std::min(b[1], ex1) - std::max(b[0], ex0), bh = std::min(b[3], ey1) - std::max(b[2], ey0);
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
         for (int it = 0; it < 32; it++) {
            const double mid = std::sqrt(s_lo * s_hi);
            (estimate(mid) > target ? s_lo : s_hi) = mid;
         }
      } else {
         s_hi = s_lo;
      }
      cell = s_hi;
   }

   Scratch d_counts, d_chunks, d_tot, d_cover, d_area;
   SG_TRY(d_tot.alloc(2 * sizeof(unsigned long long) + sizeof(uint32_t) * 2));
   unsigned long long* d_total = (unsigned long long*)d_tot.p;
   unsigned long long* d_occupied = d_total + 1;
   uint32_t* d_longest = (uint32_t*)(d_total + 2);
   uint64_t total = 0, occupied = 0;
   uint32_t longest = 0;
   SgGrid g{};
   size_t ncell = 0;
   for (int attempt = 0;; attempt++) {
      if (!forced) {
         const double ext_x = ex1 - ex0, ext_y = ey1 - ey0;
         prm.nx = (uint32_t)std::ceil(ext_x / cell) + 2;  // + the two border columns
         prm.ny = (uint32_t)std::ceil(ext_y / cell) + 2;
         prm.inv_cell = (float)(1.0 / cell);
         const double inv = prm.inv_cell;  // the device's number
         prm.u0 = (float)(ex0 - 1.0 / inv);
         prm.v0 = (float)(ey0 - 1.0 / inv);
      }
      g.u0 = prm.u0;
      g.v0 = prm.v0;
      g.inv = prm.inv_cell;
      g.nx = prm.nx;
      g.ny = prm.ny;
      ncell = (size_t)prm.nx * prm.ny;
      if (ncell + 1 > 0xfffffff0ull || prm.nx < 3 || prm.ny < 3) return refuse("cell budget exceeded");
      if (d_counts.p) {
         (void)hipFree(d_counts.p);
         d_counts.p = nullptr;
      }
      if (d_chunks.p) {
         (void)hipFree(d_chunks.p);
         d_chunks.p = nullptr;
      }
      SG_TRY(d_counts.alloc((ncell + 1) * sizeof(uint32_t)));
      SG_TRY(d_chunks.alloc((size_t)scan_chunk_count((uint32_t)ncell + 1) * sizeof(uint32_t)));
      uint32_t* counts = (uint32_t*)d_counts.p;
      if (!cam && attempt == 0) {
         // the share of the surface that lies in the border ring is known from the raster alone: a scene whose ground plane reaches far
         // beyond the dense extent (configs 0 and 4) is refused here, before the count pass
         if (!d_area.p) SG_TRY(d_area.alloc(2 * sizeof(double)));
         SG_TRY(hipMemsetAsync(d_area.p, 0, 2 * sizeof(double), stream));
         k_sg_area<<<std::min<uint32_t>(blocks_n, 2048), kBlock, 0, stream>>>(d_tris, pr, n, g, nullptr, lim.max_walk, (double*)d_area.p);
         double area[2] = {0, 0};
         SG_TRY(hipStreamSynchronize(stream));
         SG_TRY(hipMemcpy(area, d_area.p, sizeof(area), hipMemcpyDeviceToHost));  // (blocking: the destination is on this frame's stack / in a local vector)
         out.fallback_area = area[0] > 0 ? area[1] / area[0] : 0.0;
         if (out.fallback_area > lim.max_fallback_area)
            return refuse("too much of the scene's surface (" + std::to_string(out.fallback_area) + ") lies beyond the dense extent: its rays would walk the tree anyway");
      }
      SG_TRY(hipMemsetAsync(counts, 0, (ncell + 1) * sizeof(uint32_t), stream));
      SG_TRY(hipMemsetAsync(d_tot.p, 0, 2 * sizeof(unsigned long long) + 2 * sizeof(uint32_t), stream));
      const uint32_t bin_blocks = std::min<uint32_t>((n + (kBlock / 64) - 1) / (kBlock / 64), 1u << 16);
      k_sg_bin<false><<<bin_blocks, kBlock, 0, stream>>>(pr, n, g, counts, nullptr, nullptr);
      k_sg_occupancy<<<std::min<uint32_t>((uint32_t)((ncell + kBlock - 1) / kBlock), 1024), kBlock, 0, stream>>>(counts, (uint32_t)ncell, prm.nx, prm.ny, d_occupied, d_longest);
      device_exclusive_scan_u32(counts, (uint32_t)ncell + 1, (uint32_t*)d_chunks.p, d_total, stream);
      struct {
         unsigned long long total, occupied;
         uint32_t longest, longest_interior;
      } h;
      SG_TRY(hipStreamSynchronize(stream));
      SG_TRY(hipMemcpy(&h, d_tot.p, sizeof(h), hipMemcpyDeviceToHost));  // (blocking: the destination is on this frame's stack / in a local vector)
      SG_TRY(hipGetLastError());
      total = h.total;
      occupied = h.occupied;
      longest = h.longest;
      out.max_list_interior = h.longest_interior;
      if (total <= lim.max_entries && total < 0xfffffff0ull) break;
      if (attempt >= 10 || forced) return refuse("entry budget exceeded");
      cell *= 1.3;
   }
   out.params = prm;
   out.mean_list = occupied ? (double)total / (double)occupied : 0.0;
   out.max_list = longest;
   if (out.mean_list > lim.max_mean_list) return refuse("lists too long for this direction (mean " + std::to_string(out.mean_list) + " entries per occupied cell)");

   uint32_t* start = (uint32_t*)d_counts.p;  // the scan turned the counts into offsets, start[ncell] = total
   // ---- the last reason to refuse needs the offsets only: before the fill, so that a refused grid costs the count pass and no more
   // (the 512^3 iso-surface of config 4 - a ground plane beyond the dense extent - cost 23-38 ms to refuse after the fill)
   if (!cam) {
      if (!d_area.p) SG_TRY(d_area.alloc(2 * sizeof(double)));
      SG_TRY(hipMemsetAsync(d_area.p, 0, 2 * sizeof(double), stream));
      k_sg_area<<<std::min<uint32_t>(blocks_n, 2048), kBlock, 0, stream>>>(d_tris, pr, n, g, start, lim.max_walk, (double*)d_area.p);
      double area[2] = {0, 0};
      SG_TRY(hipStreamSynchronize(stream));
      SG_TRY(hipMemcpy(area, d_area.p, sizeof(area), hipMemcpyDeviceToHost));  // (blocking: the destination is on this frame's stack / in a local vector)
      SG_TRY(hipGetLastError());
      out.fallback_area = area[0] > 0 ? area[1] / area[0] : 0.0;
      if (out.fallback_area > lim.max_fallback_area)
         return refuse("too much of the scene's surface (" + std::to_string(out.fallback_area) + ") lies beyond the dense extent or in cells with long lists: its rays would walk the tree anyway");
   }

   // ---- fill, sort, cell records
   Scratch d_cursor;
   SG_TRY(d_cursor.alloc(ncell * sizeof(uint32_t)));
   SG_TRY(d_cover.alloc(ncell * sizeof(uint32_t)));
   SG_TRY(hipMemcpyAsync(d_cursor.p, start, ncell * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
   SG_TRY(hipMemsetAsync(d_cover.p, 0, ncell * sizeof(uint32_t), stream));
   SG_TRY(hipMalloc((void**)&out.entries, (total ? total : 1) * sizeof(SunGridEntry)));
   if (!cam) SG_TRY(hipMalloc((void**)&out.cells, kSunCellWords * (ncell + 1) * sizeof(uint32_t)));
   {
      const uint32_t bin_blocks = std::min<uint32_t>((n + (kBlock / 64) - 1) / (kBlock / 64), 1u << 16);
      k_sg_bin<true><<<bin_blocks, kBlock, 0, stream>>>(pr, n, g, (uint32_t*)d_cursor.p, out.entries, (uint32_t*)d_cover.p);
   }
   const uint32_t cell_blocks = std::min<uint32_t>((uint32_t)((ncell + kBlock) / kBlock), 1u << 15);
   k_sg_sort<<<cell_blocks, kBlock, 0, stream>>>(start, out.entries, prm.nx, prm.ny, lim.max_walk);
   if (!cam) k_sg_cells<<<cell_blocks, kBlock, 0, stream>>>(start, (const uint32_t*)d_cover.p, out.entries, (uint32_t)ncell, out.cells);
   SG_TRY(hipStreamSynchronize(stream));
   SG_TRY(hipGetLastError());
   out.num_entries = total;
   if (cam) {  // the camera grid has no cover depths: the scanned counts ARE its cell records (ncell + 1 offsets) - they change owner
      out.cells = (uint32_t*)d_counts.p;
      d_counts.p = nullptr;
   }
   out.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
   return true;
#undef SG_TRY
}

bool build_sun_grid_device(void* stream_v, const void* d_packets, uint32_t n, const float sun_dir[3], const SunGridLimits& lim, const SunGridParams* forced, SunGridDevice& out) {
   return build_grid_impl(stream_v, d_packets, n, sun_dir, nullptr, lim, forced, out);
}

bool build_camera_grid_device(void* stream_v, const void* d_packets, uint32_t n, const float inverse_view[16], const float inverse_projection[16], uint32_t W, uint32_t H,
                              const SunGridLimits& lim, SunGridDevice& out) {
   PgCam cam{};
   const char* why = "";
   if (!pg_make_cam(inverse_view, inverse_projection, W, H, cam, &why)) {  // camera_grid.h
      out.release();
      out = SunGridDevice();
      out.why_not = why;
      return false;
   }
   return build_grid_impl(stream_v, d_packets, n, nullptr, &cam, lim, nullptr, out);
}

}  // namespace uh
