// isosurface.hip — uh_add_isosurface_mesh: on-device extraction of the reference's marching-cubes density field
// into a triangle mesh (BASELINE.json configs[4]).
//
// Reference: utopian/shaders/marching_cubes/marching_cubes.comp:83-119 (density = max(-1, -sdTorus, -sdBox,
// -sdSphere(8 |sin(0.3 t)|)), positive inside) and :179-254 (per-voxel extraction with a wave-level append),
// driven by utopian/src/renderers/marching_cubes.rs:17-83. Differences, stated once:
//   * the cube is split into the 6 tetrahedra around its 0-6 diagonal and each tetrahedron is cut directly
//     (marching tetrahedra) instead of looking the cube up in the 256-case triangle table of tables.glsl - the
//     table is data of the reference that this repository does not copy; the surface is the same iso-surface,
//     triangulated differently (about 2x the triangles);
//   * the shapes sit in a [lo, hi]^3 domain (32 units for config 5) instead of around the camera block origin;
//   * in the reference the mesh only feeds a raster pass; here it becomes an ordinary mesh of the scene
//     (uh_add_mesh semantics: replicated host copy, any builder), which is what path tracing it needs.
// Two passes over the cells (count per 256-cell block, then emit at the scanned offsets), normals from central
// differences of the density like the host generator (rust-renderer_amd/scenes.py::isosurface_scene).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "utopian_hip.h"

namespace {

constexpr int kBlock = 256;

struct IsoParams {
   uint32_t res;
   float lo, h;         // domain origin and cell size
   float sphere_r;      // 8 |sin(0.3 t)|
   float inv_domain;    // uv = pos.xz * inv_domain
};

__device__ __forceinline__ float len2(float a, float b) { return sqrtf(a * a + b * b); }
__device__ __forceinline__ float len3(float a, float b, float c) { return sqrtf(a * a + b * b + c * c); }

// marching_cubes.comp:83-103 with the shapes at (16,20,16) torus, (16,10,16) box, (16,26,16) sphere
__device__ __forceinline__ float density(float x, float y, float z, float sphere_r) {
   const float tx = x - 16.0f, ty = y - 20.0f, tz = z - 16.0f;
   const float torus = len2(len2(tx, tz) - 5.0f, ty) - 3.0f;                                  // sdTorus(p, (5, 3))
   const float dx = fabsf(x - 16.0f) - 5.0f, dy = fabsf(y - 10.0f) - 5.0f, dz = fabsf(z - 16.0f) - 5.0f;
   const float box = fminf(fmaxf(dx, fmaxf(dy, dz)), 0.0f) + len3(fmaxf(dx, 0.0f), fmaxf(dy, 0.0f), fmaxf(dz, 0.0f));  // sdBox(p, 5)
   const float sphere = len3(x - 16.0f, y - 26.0f, z - 16.0f) - sphere_r;                      // sdSphere
   float d = fmaxf(-torus, -1.0f);
   d = fmaxf(-box, d);
   if (sphere_r > 0.0f) d = fmaxf(-sphere, d);
   return d;
}

__constant__ int c_corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};  // marching_cubes.rs:23-32
__constant__ int c_tet[6][4] = {{0, 5, 1, 6}, {0, 1, 2, 6}, {0, 2, 3, 6}, {0, 3, 7, 6}, {0, 7, 4, 6}, {0, 4, 5, 6}};

struct Cell {
   float p[8][3], v[8];
};

__device__ __forceinline__ bool load_cell(const IsoParams& q, uint64_t cell, Cell& c) {
   const uint64_t r = q.res;
   if (cell >= r * r * r) return false;
   const uint32_t ix = (uint32_t)(cell % r), iy = (uint32_t)((cell / r) % r), iz = (uint32_t)(cell / (r * r));
   int inside = 0;
   for (int k = 0; k < 8; k++) {
      c.p[k][0] = q.lo + q.h * (float)(ix + c_corner[k][0]);
      c.p[k][1] = q.lo + q.h * (float)(iy + c_corner[k][1]);
      c.p[k][2] = q.lo + q.h * (float)(iz + c_corner[k][2]);
      c.v[k] = density(c.p[k][0], c.p[k][1], c.p[k][2], q.sphere_r);
      inside += c.v[k] > 0.0f ? 1 : 0;
   }
   return inside > 0 && inside < 8;
}

__device__ __forceinline__ uint32_t cell_triangles(const Cell& c) {
   uint32_t n = 0;
   for (int t = 0; t < 6; t++) {
      int k = 0;
      for (int j = 0; j < 4; j++) k += c.v[c_tet[t][j]] > 0.0f ? 1 : 0;
      n += (k == 1 || k == 3) ? 1u : (k == 2 ? 2u : 0u);
   }
   return n;
}

__device__ __forceinline__ uint32_t block_sum(uint32_t x, uint32_t* scratch) {
   // plain shared-memory tree; 256 threads
   scratch[threadIdx.x] = x;
   __syncthreads();
   for (int s = kBlock / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) scratch[threadIdx.x] += scratch[threadIdx.x + s];
      __syncthreads();
   }
   const uint32_t total = scratch[0];
   __syncthreads();
   return total;
}

__global__ __launch_bounds__(kBlock) void k_iso_count(IsoParams q, uint32_t* __restrict__ block_counts) {
   __shared__ uint32_t scratch[kBlock];
   const uint64_t cell = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
   Cell c;
   const uint32_t n = load_cell(q, cell, c) ? cell_triangles(c) : 0u;
   const uint32_t total = block_sum(n, scratch);
   if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

__device__ __forceinline__ void emit_vertex(const IsoParams& q, UhVertex* out, const float* a, float va, const float* b, float vb) {
   const float t = va / (va - vb);
   const float p[3] = {a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1]), a[2] + t * (b[2] - a[2])};
   const float eps = 1e-3f;
   float g[3] = {density(p[0] + eps, p[1], p[2], q.sphere_r) - density(p[0] - eps, p[1], p[2], q.sphere_r),
                 density(p[0], p[1] + eps, p[2], q.sphere_r) - density(p[0], p[1] - eps, p[2], q.sphere_r),
                 density(p[0], p[1], p[2] + eps, q.sphere_r) - density(p[0], p[1], p[2] - eps, q.sphere_r)};
   const float gl = fmaxf(len3(g[0], g[1], g[2]), 1e-20f);
   UhVertex v;
   memset(&v, 0, sizeof(v));
   v.pos[0] = p[0];
   v.pos[1] = p[1];
   v.pos[2] = p[2];
   v.pos[3] = 1.0f;
   v.normal[0] = -g[0] / gl;  // density grows inwards
   v.normal[1] = -g[1] / gl;
   v.normal[2] = -g[2] / gl;
   v.uv[0] = p[0] * q.inv_domain;
   v.uv[1] = p[2] * q.inv_domain;
   v.color[0] = v.color[1] = v.color[2] = v.color[3] = 1.0f;
   *out = v;
}

__global__ __launch_bounds__(kBlock) void k_iso_emit(IsoParams q, const uint32_t* __restrict__ block_offsets, UhVertex* __restrict__ verts) {
   __shared__ uint32_t scan[kBlock];
   const uint64_t cell = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
   Cell c;
   const bool mixed = load_cell(q, cell, c);
   const uint32_t n = mixed ? cell_triangles(c) : 0u;
   // exclusive scan of n inside the block (Hillis-Steele on 256 values)
   scan[threadIdx.x] = n;
   __syncthreads();
   for (int s = 1; s < kBlock; s <<= 1) {
      const uint32_t add = (int)threadIdx.x >= s ? scan[threadIdx.x - s] : 0u;
      __syncthreads();
      scan[threadIdx.x] += add;
      __syncthreads();
   }
   if (!n) return;
   uint64_t tri = (uint64_t)block_offsets[blockIdx.x] + (scan[threadIdx.x] - n);
   for (int t = 0; t < 6; t++) {
      // inside vertices first, stable (the host generator's argsort(~inside, kind="stable"))
      int ord[4], k = 0, m = 0;
      for (int j = 0; j < 4; j++)
         if (c.v[c_tet[t][j]] > 0.0f) ord[k++] = c_tet[t][j];
      m = k;
      for (int j = 0; j < 4; j++)
         if (!(c.v[c_tet[t][j]] > 0.0f)) ord[m++] = c_tet[t][j];
      auto cut = [&](UhVertex* o, int a, int b) { emit_vertex(q, o, c.p[ord[a]], c.v[ord[a]], c.p[ord[b]], c.v[ord[b]]); };
      UhVertex* o = verts + 3 * tri;
      if (k == 1) {
         cut(o + 0, 0, 1);
         cut(o + 1, 0, 2);
         cut(o + 2, 0, 3);
         tri += 1;
      } else if (k == 3) {
         cut(o + 0, 0, 3);
         cut(o + 1, 1, 3);
         cut(o + 2, 2, 3);
         tri += 1;
      } else if (k == 2) {
         cut(o + 0, 0, 2);  // a
         cut(o + 1, 0, 3);  // b
         cut(o + 2, 1, 3);  // c
         cut(o + 3, 0, 2);  // a
         cut(o + 4, 1, 3);  // c
         cut(o + 5, 1, 2);  // d
         tri += 2;
      }
   }
}

}  // namespace

extern "C" int uh_add_isosurface_mesh(uh_ctx* ctx, uint32_t resolution, float lo, float hi, float time, const UhGpuMaterial* material, const float world3x4[12],
                                       uint32_t* out_mesh_index, uint32_t* out_triangles) {
   if (!ctx || !material || !world3x4) return UH_ERR_INVALID_ARGUMENT;
   if (resolution < 1 || resolution > 1024 || !(hi > lo)) return UH_ERR_INVALID_ARGUMENT;
   void* stream_v = nullptr;
   if (int st = uh_stream(ctx, &stream_v)) return st;  // also selects the context's device
   hipStream_t stream = (hipStream_t)stream_v;
   IsoParams q;
   q.res = resolution;
   q.lo = lo;
   q.h = (hi - lo) / (float)resolution;
   q.sphere_r = 8.0f * std::fabs(std::sin(time * 0.3f));
   q.inv_domain = 1.0f / (hi - lo);
   const uint64_t cells = (uint64_t)resolution * resolution * resolution;
   const uint32_t blocks = (uint32_t)((cells + kBlock - 1) / kBlock);
   uint32_t* d_counts = nullptr;
   UhVertex* d_verts = nullptr;
   auto fail = [&](int st) {
      if (d_counts) (void)hipFree(d_counts);
      if (d_verts) (void)hipFree(d_verts);
      return st;
   };
   if (hipMalloc(&d_counts, (size_t)blocks * sizeof(uint32_t)) != hipSuccess) return fail(UH_ERR_OUT_OF_MEMORY);
   k_iso_count<<<blocks, kBlock, 0, stream>>>(q, d_counts);
   std::vector<uint32_t> counts(blocks);
   if (hipMemcpyAsync(counts.data(), d_counts, (size_t)blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, stream) != hipSuccess) return fail(UH_ERR_HIP);
   if (hipStreamSynchronize(stream) != hipSuccess) return fail(UH_ERR_HIP);
   uint64_t total = 0;
   for (uint32_t b = 0; b < blocks; b++) {  // exclusive scan of the per-block counts (<= 4 Mi entries) on the host
      const uint32_t n = counts[b];
      counts[b] = (uint32_t)total;
      total += n;
   }
   if (total > (3ull << 22)) return fail(UH_ERR_CAPACITY);  // raw count; a mesh holds at most 4 Mi triangles after the sliver filter (checked by uh_add_mesh)
   std::vector<UhVertex> verts((size_t)total * 3);
   if (total) {
      if (hipMalloc(&d_verts, verts.size() * sizeof(UhVertex)) != hipSuccess) return fail(UH_ERR_OUT_OF_MEMORY);
      if (hipMemcpyAsync(d_counts, counts.data(), (size_t)blocks * sizeof(uint32_t), hipMemcpyHostToDevice, stream) != hipSuccess) return fail(UH_ERR_HIP);
      k_iso_emit<<<blocks, kBlock, 0, stream>>>(q, d_counts, d_verts);
      if (hipMemcpyAsync(verts.data(), d_verts, verts.size() * sizeof(UhVertex), hipMemcpyDeviceToHost, stream) != hipSuccess) return fail(UH_ERR_HIP);
      if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) return fail(UH_ERR_HIP);
   }
   (void)fail(0);
   // zero-area slivers (a cut that lands on a grid corner collapses an edge; about a third of the raw output): they can
   // never be hit (det == 0) and would only cost tree nodes - dropped like the host generator does
   {
      size_t kept = 0;
      for (size_t t = 0; t < (size_t)total; t++) {
         const float* a = verts[3 * t].pos;
         const float* b = verts[3 * t + 1].pos;
         const float* c = verts[3 * t + 2].pos;
         const double e1[3] = {(double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2]}, e2[3] = {(double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2]};
         const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
         if (std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) > 1e-12) {
            if (kept != t)
               for (int k = 0; k < 3; k++) verts[3 * kept + k] = verts[3 * t + k];
            kept++;
         }
      }
      total = kept;
      verts.resize(3 * kept);
   }
   if (out_triangles) *out_triangles = (uint32_t)total;
   if (!total) {
      if (out_mesh_index) *out_mesh_index = 0xffffffffu;
      return UH_OK;  // nothing crosses the iso value: no mesh is added
   }
   std::vector<uint32_t> indices(verts.size());
   for (size_t i = 0; i < indices.size(); i++) indices[i] = (uint32_t)i;
   return uh_add_mesh(ctx, verts.data(), (uint32_t)verts.size(), indices.data(), (uint32_t)indices.size(), material, world3x4, out_mesh_index);
}
