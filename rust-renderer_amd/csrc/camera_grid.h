// camera_grid.h - the arithmetic of the camera grid's builder (and the cell test it shares with the sun grid's), as host + device
// functions: csrc/sun_grid_build.hip runs them in its kernels, tests/cpp/camera_grid_check.cpp runs THE SAME EXPRESSIONS on the
// host and holds the grid they make against brute force over all packets (rays through pixel corners, edges and centres of a set
// of cameras, including one in the plane of a floor and packets behind and across the camera plane) - what tests/cpp/sun_grid_check.cpp
// does for the sun grid's host builder. Nothing here touches the GPU or the HIP runtime.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <cmath>

#if defined(__HIPCC__)
#define UH_HD __host__ __device__
#else
#define UH_HD
#endif

namespace uh {

struct SgProj {
   double px[3], py[3], pad[3], padmax;
   double x0, x1, y0, y1;
   double pa, pb, pc, cover_drop;
   float wmax;      // the entries' sort key: a cell's list is sorted by it, descending (sun grid: far depth; camera grid: -near distance)
   uint32_t flags;  // bit 0: use, bit 1: can_cover, bit 2: the dilated edges are given below (camera grid) instead of derived from px / py / pad
   // bit 2: up to three half-planes nxe x + nye y <= off in grid coordinates, each carrying a margin of epad; bits 8-9: how many
   double nxe[3], nye[3], off[3], epad[3];
};


struct SgGrid {
   double u0, v0, inv;  // the device kernel's numbers (float u0, v0, inv_cell), widened
   uint32_t nx, ny;
};

// ------------------------------------------------------------------------------------------
// The camera grid: the same binning for the rays that leave ONE POINT - the primary rays of reference.rgen:31-47 and the G-buffer
// cast. The grid is the frame itself, one cell per pixel (plus the border ring the binning kernel expects), in pixel coordinates; a
// packet is listed in every pixel from which some ray through the pixel's square can be accepted by the float triangle test; the
// lists are sorted by a lower bound of the distance at which the packet can be hit.
// ------------------------------------------------------------------------------------------
struct PgCam {
   double O[3];                  // the rays' common origin (inverse_view * (0,0,0,1) as the kernels compute it)
   double D0[3], Dx[3], Dy[3];   // direction through NDC (dx, dy), up to its length: D0 + dx Dx + dy Dy (world space)
   double Kinv[9];               // row-major inverse of [Dx Dy D0]: Kinv r = lambda (dx, dy, 1) for a point O + r in front of the camera
   double Dmax;                  // largest |D0 + dx Dx + dy Dy| over the frame
   double smax;                  // largest stretch of a unit vector by the upper 3x3 of inverse_view (1 for a rigid camera): t >= distance / smax
   double lam_clip;              // points of the frame with lambda below this are nearer than tmin / 2 (0: unknown - a packet that crosses the camera plane takes the whole frame as its box)
   double W, H;
};


UH_HD inline double dot3d(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
UH_HD inline void cross3d(const double* a, const double* b, double* r) {
   r[0] = a[1] * b[2] - a[2] * b[1];
   r[1] = a[2] * b[0] - a[0] * b[2];
   r[2] = a[0] * b[1] - a[1] * b[0];
}

// distance from the origin to the triangle (r0, r1, r2) (Ericson, Real-Time Collision Detection 5.1.5, for the point 0)
UH_HD inline double origin_triangle_distance(const double* a, const double* b, const double* c) {
   double ab[3], ac[3], ap[3], bp[3], cp[3];
   for (int k = 0; k < 3; k++) {
      ab[k] = b[k] - a[k];
      ac[k] = c[k] - a[k];
      ap[k] = -a[k];
      bp[k] = -b[k];
      cp[k] = -c[k];
   }
   const double d1 = dot3d(ab, ap), d2 = dot3d(ac, ap);
   if (d1 <= 0 && d2 <= 0) return sqrt(dot3d(a, a));
   const double d3 = dot3d(ab, bp), d4 = dot3d(ac, bp);
   if (d3 >= 0 && d4 <= d3) return sqrt(dot3d(b, b));
   const double vc = d1 * d4 - d3 * d2;
   double q[3];
   if (vc <= 0 && d1 >= 0 && d3 <= 0) {
      const double v = d1 / (d1 - d3);
      for (int k = 0; k < 3; k++) q[k] = a[k] + v * ab[k];
      return sqrt(dot3d(q, q));
   }
   const double d5 = dot3d(ab, cp), d6 = dot3d(ac, cp);
   if (d6 >= 0 && d5 <= d6) return sqrt(dot3d(c, c));
   const double vb = d5 * d2 - d1 * d6;
   if (vb <= 0 && d2 >= 0 && d6 <= 0) {
      const double w = d2 / (d2 - d6);
      for (int k = 0; k < 3; k++) q[k] = a[k] + w * ac[k];
      return sqrt(dot3d(q, q));
   }
   const double va = d3 * d6 - d5 * d4;
   if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
      const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
      for (int k = 0; k < 3; k++) q[k] = b[k] + w * (c[k] - b[k]);
      return sqrt(dot3d(q, q));
   }
   const double denom = 1.0 / (va + vb + vc), v = vb * denom, w = vc * denom;
   for (int k = 0; k < 3; k++) q[k] = a[k] + ab[k] * v + ac[k] * w;
   return sqrt(dot3d(q, q));
}

// Margins (DESIGN.md "Camera grid"). With tv = O - v0 the three barycentric tests of tri_compute are signs of linear forms in the
// direction d: u det = d . (e2 x tv), v det = d . (tv x e1), (1 - u - v) det = d . (normal of the plane through O and the third
// edge) - each the plane through O and one edge. In float each form is off by at most delta = 8 eps |tv| |e| |d| (the third: the
// sum of the other two plus 4 eps |det|): a ray can pass the test of edge k only within the ANGLE rho_k = Delta / |r_a x r_b| of
// that plane (r_a, r_b the edge's corners seen from O; Delta = 4 eps (16 |tv| Lmax + 4 |e1| |e2|), safety factor 4 included),
// widened by the rounding of the ray's own direction (rho_dir). In the plane of NDC coordinates the plane through O and an edge is
// a line, and "within rho of it" a band of width rho Dmax / |gradient|: the dilated half-planes the binning kernel intersects
// with the packet's padded box. A packet seen edge-on (O in its plane to within the margins) keeps only its box.
UH_HD inline void pg_project_packet(const float q[9], const PgCam& cam, SgProj& p) {
   p = SgProj{};
   bool finite = true;
   for (int k = 0; k < 9; k++) finite = finite && isfinite(q[k]);
   if (finite) {
      const double c[3][3] = {{q[0], q[1], q[2]}, {(double)q[0] + q[3], (double)q[1] + q[4], (double)q[2] + q[5]}, {(double)q[0] + q[6], (double)q[1] + q[7], (double)q[2] + q[8]}};
      double r[3][3], lam[3], sx[3], sy[3], dist[3];
      bool front = true, behind = true;
      for (int k = 0; k < 3; k++) {
         for (int a = 0; a < 3; a++) r[k][a] = c[k][a] - cam.O[a];
         const double hx = cam.Kinv[0] * r[k][0] + cam.Kinv[1] * r[k][1] + cam.Kinv[2] * r[k][2];
         const double hy = cam.Kinv[3] * r[k][0] + cam.Kinv[4] * r[k][1] + cam.Kinv[5] * r[k][2];
         lam[k] = cam.Kinv[6] * r[k][0] + cam.Kinv[7] * r[k][1] + cam.Kinv[8] * r[k][2];
         dist[k] = sqrt(dot3d(r[k], r[k]));
         const double lmin = 1e-9 * dist[k];  // "in front" with room to spare: the projection below divides by it
         front = front && lam[k] > lmin;
         behind = behind && lam[k] <= lmin;
         const double dx = hx / lam[k], dy = hy / lam[k];
         sx[k] = (dx + 1.0) * 0.5 * cam.W;
         sy[k] = (1.0 - dy) * 0.5 * cam.H;
      }
      // a packet wholly behind the camera plane cannot be met by a ray of the frame (every ray has lambda > 0)
      if (!behind) {
         const double e1[3] = {q[3], q[4], q[5]}, e2[3] = {q[6], q[7], q[8]};
         const double l1 = sqrt(dot3d(e1, e1)), l2 = sqrt(dot3d(e2, e2));
         const double e3[3] = {e2[0] - e1[0], e2[1] - e1[1], e2[2] - e1[2]};
         const double Lmax = fmax(fmax(l1, l2), sqrt(dot3d(e3, e3)));
         const double eps = 5.9604644775390625e-8;  // 2^-24
         const double Delta = 4.0 * eps * (16.0 * dist[0] * Lmax + 4.0 * l1 * l2);
         const double rho_dir = 2e-6;
         // orientation: the forward cone is where the three edge forms have the sign of the volume r0 . (r1 x r2)
         double n01[3];
         cross3d(r[0], r[1], n01);
         const double vol = dot3d(n01, r[2]);
         const double dmaxv = fmax(dist[0], fmax(dist[1], dist[2]));
         int ne = 0;
         double padmax = 2e-3;
         const bool edge_on = !(fabs(vol) > 1e-12 * dmaxv * dmaxv * dmaxv);
         if (!edge_on) {
            const double s = vol > 0 ? 1.0 : -1.0;
            for (int k = 0; k < 3; k++) {
               const int j = (k + 1) % 3;
               double nk[3];
               cross3d(r[k], r[j], nk);
               const double nl = sqrt(dot3d(nk, nk));
               if (!(nl > 1e-300)) continue;  // the edge points at the camera (or has no length): no constraint from it
               const double nh[3] = {nk[0] / nl, nk[1] / nl, nk[2] / nl};
               const double fa = dot3d(nh, cam.Dx), fb = dot3d(nh, cam.Dy), fc = dot3d(nh, cam.D0);
               // inside: s (fa dx + fb dy + fc) >= -rho Dmax; with dx = 2 u / W - 1, dy = 1 - 2 v / H:
               const double gx = -s * 2.0 * fa / cam.W, gy = s * 2.0 * fb / cam.H, hh = s * (fc - fa + fb);
               const double gl = sqrt(gx * gx + gy * gy);
               if (!(gl > 1e-12)) continue;  // the plane through O and this edge does not cross the frame's plane at a usable angle
               const double rho = Delta / nl + rho_dir;
               const double pad = rho * cam.Dmax / gl + 2e-3;  // pixels; 2e-3: the float evaluation of the ray's own screen position
               p.nxe[ne] = gx / gl;
               p.nye[ne] = gy / gl;
               p.off[ne] = hh / gl + pad;
               p.epad[ne] = pad;
               padmax = fmax(padmax, pad);
               ne++;
            }
         }
         if (front) {
            p.x0 = fmin(sx[0], fmin(sx[1], sx[2])) - padmax;
            p.x1 = fmax(sx[0], fmax(sx[1], sx[2])) + padmax;
            p.y0 = fmin(sy[0], fmin(sy[1], sy[2])) - padmax;
            p.y1 = fmax(sy[0], fmax(sy[1], sy[2])) + padmax;
            for (int k = 0; k < 3; k++) {
               p.px[k] = sx[k];
               p.py[k] = sy[k];
            }
         } else if (cam.lam_clip > 0) {
            // crosses the camera plane: the box of the part with lambda >= lam_clip (the rest is behind the camera or nearer than the
            // rays' tmin: no ray can be accepted there) - Sutherland-Hodgman against that one plane, then the projection
            double bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY;
            int kept = 0;
            for (int k = 0; k < 3; k++) {
               const int j = (k + 1) % 3;
               const bool ik = lam[k] >= cam.lam_clip, ij = lam[j] >= cam.lam_clip;
               double pts[2][3];
               int np = 0;
               if (ik) {
                  for (int a = 0; a < 3; a++) pts[np][a] = r[k][a];
                  np++;
               }
               if (ik != ij) {
                  const double t = (cam.lam_clip - lam[k]) / (lam[j] - lam[k]);
                  for (int a = 0; a < 3; a++) pts[np][a] = r[k][a] + t * (r[j][a] - r[k][a]);
                  np++;
               }
               for (int m = 0; m < np; m++) {
                  const double hx = cam.Kinv[0] * pts[m][0] + cam.Kinv[1] * pts[m][1] + cam.Kinv[2] * pts[m][2];
                  const double hy = cam.Kinv[3] * pts[m][0] + cam.Kinv[4] * pts[m][1] + cam.Kinv[5] * pts[m][2];
                  double lm = cam.Kinv[6] * pts[m][0] + cam.Kinv[7] * pts[m][1] + cam.Kinv[8] * pts[m][2];
                  lm = fmax(lm, 0.5 * cam.lam_clip);  // (an intersection point: lambda = lam_clip up to rounding)
                  const double u = (hx / lm + 1.0) * 0.5 * cam.W, v = (1.0 - hy / lm) * 0.5 * cam.H;
                  // a coordinate beyond the frame by more than this is as good as infinite (and keeps the arithmetic finite)
                  const double big = 16.0 * (cam.W + cam.H);
                  bx0 = fmin(bx0, fmax(u, -big));
                  bx1 = fmax(bx1, fmin(u, big));
                  by0 = fmin(by0, fmax(v, -big));
                  by1 = fmax(by1, fmin(v, big));
                  kept++;
               }
            }
            if (kept == 0) {
               front = false;  // nothing of it at lambda >= lam_clip: marked unused below
               p.x0 = NAN;
            } else {
               const double slack = 1.0 + padmax;  // the clipped outline is a chord of the true one: a pixel of room
               p.x0 = bx0 - slack;
               p.x1 = bx1 + slack;
               p.y0 = by0 - slack;
               p.y1 = by1 + slack;
            }
            if (edge_on) ne = 0;
         } else {
            // crosses the camera plane and the camera is not rigid: the whole frame is its box, the edge planes (which need no
            // projection) cut it down to the pixels it can be seen from
            p.x0 = -1.0;
            p.x1 = cam.W + 1.0;
            p.y0 = -1.0;
            p.y1 = cam.H + 1.0;
            if (edge_on) ne = 0;
         }
         // the list's sort key: minus a lower bound of the t at which a ray of the frame can hit this packet in float arithmetic.
         // t_exact >= distance(O, triangle) / smax; the float t = (e2 . q) / det is off by about 16 eps / |cos phi| relatively
         // (phi: ray against the packet's normal), |cos phi| >= h / dmax over the packet (h: distance of O from its plane)
         double tnear = 0.0;
         {
            double nrm[3];
            const double ee1[3] = {r[1][0] - r[0][0], r[1][1] - r[0][1], r[1][2] - r[0][2]}, ee2[3] = {r[2][0] - r[0][0], r[2][1] - r[0][1], r[2][2] - r[0][2]};
            cross3d(ee1, ee2, nrm);
            const double nl = sqrt(dot3d(nrm, nrm));
            if (nl > 0 && dmaxv > 0) {
               const double h = fabs(dot3d(nrm, r[0])) / nl, cosmin = h / dmaxv;
               if (cosmin > 1e-3) {
                  const double dmin = origin_triangle_distance(r[0], r[1], r[2]);
                  tnear = dmin / cam.smax * (1.0 - 4e-6 / cosmin) - 1e-6;
                  if (!(tnear > 0)) tnear = 0.0;
               }
            }
         }
         float kf = (float)(-tnear);
         if ((double)kf < -tnear) kf = nextafterf(kf, INFINITY);  // -key <= tnear: rounded towards the camera
         p.wmax = kf;
         p.padmax = padmax;
         const bool use = isfinite(p.x0) && isfinite(p.x1) && isfinite(p.y0) && isfinite(p.y1);
         p.flags = (use ? 1u : 0u) | 4u | ((uint32_t)ne << 8);
      }
   }
}

inline bool pg_invert3(const double* m, double* inv) {
   const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
   const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
   const double det = a * A + b * B + c * C;
   const double scale = std::fabs(a) + std::fabs(b) + std::fabs(c) + std::fabs(d) + std::fabs(e) + std::fabs(f) + std::fabs(g) + std::fabs(h) + std::fabs(i);
   if (!std::isfinite(det) || !(std::fabs(det) > 1e-12 * scale * scale * scale)) return false;
   const double id = 1.0 / det;
   inv[0] = A * id;
   inv[1] = -(b * i - c * h) * id;
   inv[2] = (b * f - c * e) * id;
   inv[3] = B * id;
   inv[4] = (a * i - c * g) * id;
   inv[5] = -(a * f - c * d) * id;
   inv[6] = C * id;
   inv[7] = -(a * h - b * g) * id;
   inv[8] = (a * e - b * d) * id;
   return true;
}


// The bundle of primary rays of (inverse_view, inverse_projection) on a W x H frame, as primary_ray (device_math.h) makes them:
// origin = inverse_view * (0,0,0,1) = its fourth column; target = inverse_projection * (dx, dy, 1, 1); direction = (upper 3x3 of
// inverse_view) * normalize(target.xyz). false (and why): the matrices are not finite or no perspective bundle this grid can raster.
inline bool pg_make_cam(const float inverse_view[16], const float inverse_projection[16], uint32_t W, uint32_t H, PgCam& cam, const char** why) {
   cam = PgCam{};
   const float* iv = inverse_view;
   const float* ip = inverse_projection;
   for (int k = 0; k < 16; k++)
      if (!std::isfinite(iv[k]) || !std::isfinite(ip[k])) {
         if (why) *why = "the camera matrices are not finite";
         return false;
      }
   for (int r = 0; r < 3; r++) cam.O[r] = iv[12 + r];
   double M[9], Px[3], Py[3], P0[3];
   for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) M[3 * r + c] = iv[4 * c + r];
      Px[r] = ip[0 + r];
      Py[r] = ip[4 + r];
      P0[r] = (double)ip[8 + r] + ip[12 + r];
   }
   auto mul = [&](const double* v, double* o) {
      for (int r = 0; r < 3; r++) o[r] = M[3 * r] * v[0] + M[3 * r + 1] * v[1] + M[3 * r + 2] * v[2];
   };
   mul(Px, cam.Dx);
   mul(Py, cam.Dy);
   mul(P0, cam.D0);
   const double K[9] = {cam.Dx[0], cam.Dy[0], cam.D0[0], cam.Dx[1], cam.Dy[1], cam.D0[1], cam.Dx[2], cam.Dy[2], cam.D0[2]};
   if (!pg_invert3(K, cam.Kinv) || W == 0 || H == 0 || (uint64_t)(W + 2) * (H + 2) > (64ull << 20)) {
      if (why) *why = "the camera matrices do not describe a perspective bundle this grid can raster";
      return false;
   }
   cam.Dmax = 0.0;
   for (int sx = -1; sx <= 1; sx += 2)
      for (int sy = -1; sy <= 1; sy += 2) {
         double d[3];
         for (int r = 0; r < 3; r++) d[r] = cam.D0[r] + sx * cam.Dx[r] + sy * cam.Dy[r];
         cam.Dmax = std::max(cam.Dmax, std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]));
      }
   // the largest stretch of a unit vector by M: 1 for a rigid camera, else bounded by the Frobenius norm
   {
      double fro = 0.0, dev = 0.0;
      for (int a = 0; a < 3; a++)
         for (int b = 0; b < 3; b++) {
            double g = 0.0;
            for (int r = 0; r < 3; r++) g += M[3 * r + a] * M[3 * r + b];
            dev = std::max(dev, std::fabs(g - (a == b ? 1.0 : 0.0)));
            if (a == b) fro += g;
         }
      cam.smax = dev < 1e-5 ? 1.0 + 1e-5 : std::sqrt(fro);
      // a point of the frame at lambda is O + lambda (D0 + dx Dx + dy Dy): at most lambda Dmax away, i.e. at t <= lambda Dmax / smin. With
      // a rigid camera (smin = 1) everything below lam_clip is nearer than half the rays' tmin = 0.001 (rgen:44)
      cam.lam_clip = dev < 1e-5 && cam.Dmax > 0 ? 2.5e-4 / cam.Dmax : 0.0;
   }
   cam.W = (double)W;
   cam.H = (double)H;
   return true;
}

// ---- the cell walk both grids' binning shares (k_sg_bin): a packet's cell range, its dilated edges, and "may a ray of this cell be
// accepted by the packet"
UH_HD inline uint32_t sg_cell_of(double x, double o, double inv, uint32_t n) {
   double f = floor((x - o) * inv);
   if (!(f >= 0)) f = 0;
   if (f > (double)(n - 1)) f = (double)(n - 1);
   return (uint32_t)f;
}
// up to three half-planes nxe x + nye y <= off (each carrying a margin epad): given by the projection (camera grid, flags bit 2) or
// derived from the projected corners and their pads (sun grid); returns how many
UH_HD inline int sg_packet_edges(const SgProj& p, double nxe[3], double nye[3], double off[3], double epad[3]) {
   int ne = 0;
   const double area2 = (p.px[1] - p.px[0]) * (p.py[2] - p.py[0]) - (p.py[1] - p.py[0]) * (p.px[2] - p.px[0]);
   if (p.flags & 4u) {
      ne = (int)((p.flags >> 8) & 3u);
      for (int k = 0; k < ne; k++) {
         nxe[k] = p.nxe[k];
         nye[k] = p.nye[k];
         off[k] = p.off[k];
         epad[k] = p.epad[k];
      }
   } else if (fabs(area2) > 1e-300) {
      const double s = area2 > 0 ? 1.0 : -1.0;
      for (int k = 0; k < 3; k++) {
         const int j = (k + 1) % 3;
         const double dx = p.px[j] - p.px[k], dy = p.py[j] - p.py[k], len = sqrt(dx * dx + dy * dy);
         if (!(len > 1e-150)) continue;
         nxe[ne] = s * dy / len;
         nye[ne] = -s * dx / len;
         off[ne] = nxe[ne] * p.px[k] + nye[ne] * p.py[k] + p.pad[k];
         epad[ne] = p.pad[k];
         ne++;
      }
   }
   return ne;
}
// the cell [cx0, cx1] x [cy0, cy1] meets every dilated half-plane (`in`); `inside`: it lies within all of them with three margins to
// spare (the sun grid's cover test starts from it)
UH_HD inline bool sg_cell_touches(int ne, const double nxe[3], const double nye[3], const double off[3], const double epad[3], double cx0, double cx1, double cy0, double cy1,
                                 bool& inside) {
   bool in = true;
   for (int e = 0; e < ne && in; e++) {
      const double m = fmin(nxe[e] * cx0, nxe[e] * cx1) + fmin(nye[e] * cy0, nye[e] * cy1);
      in = m <= off[e];
      const double M = fmax(nxe[e] * cx0, nxe[e] * cx1) + fmax(nye[e] * cy0, nye[e] * cy1);
      inside = inside && M <= off[e] - 3.0 * epad[e];
   }
   return in;
}

}  // namespace uh
