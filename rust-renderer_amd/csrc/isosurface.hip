// isosurface.hip — uh_add_isosurface_mesh: on-device marching-cubes extraction of the reference's density field
// into a triangle mesh (BASELINE.json configs[4]).
//
// Reference: utopian/shaders/marching_cubes/marching_cubes.comp:83-119 (density = max(-1, -sdTorus, -sdBox,
// -sdSphere(8 |sin(0.3 t)|)), positive inside), :179-254 (per-voxel case index, edge vertices by linear interpolation,
// triangles from the 256-case table), driven by utopian/src/renderers/marching_cubes.rs:17-83.
// Default (option "iso_reference_triangulation" = 1): the reference's triangles, cell by cell - the triangle table is the public
// Lorensen-Cline table tables.glsl:38-293 holds (mc_reference_tables.h, numbers as data), every edge vertex is vertexInterp
// (marching_cubes.comp:134-137) in the corner order the shader names (:204-226), and every triangle of the case's list is
// emitted in list order, zero-area ones included (:231-251). What differs, stated once:
//   * output order is deterministic: triangles are counted per cell, the counts are scanned ON THE DEVICE (three small
//     kernels below) and a second pass writes each cell's triangles at its offset, cells in x-fastest order - where the
//     reference appends with one atomicAdd per triangle in arrival order (marching_cubes.comp:236);
// With the option at 0 (round 3's form, kept for the tree it gives):
//   * the case tables are the ones generated in this repository from the cube's geometry (tools/gen_mc_tables.py ->
//     mc_tables.h: same crossed edges, triangle counts and boundary polygons, other interior diagonals in 160 cases);
//   * an edge vertex is interpolated from the endpoint with the smaller grid index to the other one, so the two to
//     four cells that share an edge compute the same bits;
//   * triangles of zero area (a cut that lands on a grid corner collapses an edge) are dropped on the device in both
//     passes: they can never be hit (det == 0) and would only cost tree nodes;
//   * the shapes sit in a [lo, hi]^3 domain (32 units for config 5) instead of around the camera block origin, and the
//     mesh becomes an ordinary mesh of the scene (uh_add_mesh semantics: replicated host copy, any builder) instead of
//     feeding a raster pass.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "device_scan.h"
#include "mc_reference_tables.h"
#include "mc_tables.h"
#include "context_internal.h"
#include "utopian_hip.h"

namespace {

constexpr int kBlock = 256;

struct IsoParams {
   uint32_t res;
   float lo, h;         // domain origin and cell size
   float sphere_r;      // 8 |sin(0.3 t)|
   float inv_domain;    // uv = pos.xz * inv_domain
   uint32_t reference;  // 1: the reference's triangle table, corner order and zero-area triangles (the default)
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
   d = fmaxf(-sphere, d);  // also at radius 0, as the reference: -|p - c| only wins at the centre point itself (density 0 there)
   return d;
}

__constant__ int c_corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};  // marching_cubes.rs:23-32
__constant__ uint16_t c_edge_mask[256];
__constant__ uint8_t c_tri_count[256];
__constant__ uint8_t c_tris[256][3 * kMcMaxTris];
// the edge's endpoints, the one with the smaller grid index first (every cell sharing the edge interpolates alike)
__constant__ uint8_t c_edge_lo_hi[12][2] = {{0, 1}, {1, 2}, {3, 2}, {0, 3}, {4, 5}, {5, 6}, {7, 6}, {4, 7}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
// ... and in the order marching_cubes.comp:204-226 hands the corners to vertexInterp
__constant__ uint8_t c_edge_shader[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
__constant__ int8_t c_ref_tris[256][16];

struct Cell {
   float p[8][3], v[8];
   uint32_t cube_index;  // bit i: corner i outside (density < 0), marching_cubes.comp:186-190
};

__device__ __forceinline__ bool load_cell(const IsoParams& q, uint64_t cell, Cell& c) {
   const uint64_t r = q.res;
   c.cube_index = 0;
   if (cell >= r * r * r) return false;
   const uint32_t ix = (uint32_t)(cell % r), iy = (uint32_t)((cell / r) % r), iz = (uint32_t)(cell / (r * r));
   for (int k = 0; k < 8; k++) {
      c.p[k][0] = q.lo + q.h * (float)(ix + c_corner[k][0]);
      c.p[k][1] = q.lo + q.h * (float)(iy + c_corner[k][1]);
      c.p[k][2] = q.lo + q.h * (float)(iz + c_corner[k][2]);
      c.v[k] = density(c.p[k][0], c.p[k][1], c.p[k][2], q.sphere_r);
      if (c.v[k] < 0.0f) c.cube_index |= 1u << k;
   }
   return c_edge_mask[c.cube_index] != 0;
}

// vertexInterp (marching_cubes.comp:134-137) at iso level 0 on edge e
__device__ __forceinline__ void edge_point(const Cell& c, int e, float* p, bool reference) {
   if (reference) {
      const int a = c_edge_shader[e][0], b = c_edge_shader[e][1];
      const float t = (0.0f - c.v[a]) / (c.v[b] - c.v[a]);  // (isolevel - val1) / (val2 - val1), :136
      for (int k = 0; k < 3; k++) p[k] = c.p[a][k] + t * (c.p[b][k] - c.p[a][k]);  // mix(x, y, a) = x + a (y - x): the oracle's reading
      return;
   }
   const int a = c_edge_lo_hi[e][0], b = c_edge_lo_hi[e][1];
   const float t = c.v[a] / (c.v[a] - c.v[b]);
   for (int k = 0; k < 3; k++) p[k] = c.p[a][k] + t * (c.p[b][k] - c.p[a][k]);
}

__device__ __forceinline__ bool has_area(const float* a, const float* b, const float* c) {
   const double e1[3] = {(double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2]}, e2[3] = {(double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2]};
   const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
   return sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) > 1e-12;
}

__device__ __forceinline__ void write_vertex(const IsoParams& q, UhVertex* out, const float* p) {
   const float eps = 1.0f;  // generateNormal: "float d = 1.0f / 1.0f" (marching_cubes.comp:167), whatever the voxel size
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
   v.normal[0] = -g[0] / gl;  // generateNormal (marching_cubes.comp:160-177): the density grows inwards
   v.normal[1] = -g[1] / gl;
   v.normal[2] = -g[2] / gl;
   v.uv[0] = p[0] * q.inv_domain;
   v.uv[1] = p[2] * q.inv_domain;
   v.color[0] = v.color[1] = v.color[2] = v.color[3] = 1.0f;
   *out = v;
}

// the cell's triangles with area; EMIT writes them from `out` on
template <bool EMIT>
__device__ __forceinline__ uint32_t cell_triangles(const IsoParams& q, const Cell& c, UhVertex* out) {
   if (q.reference) {
      // marching_cubes.comp:231-251: the case's list up to the first -1, nothing dropped, nothing reordered
      uint32_t n = 0;
      for (int i = 0; i < 15 && c_ref_tris[c.cube_index][i] >= 0; i += 3, n++) {
         if (!EMIT) continue;
         for (int k = 0; k < 3; k++) {
            float p[3];
            edge_point(c, c_ref_tris[c.cube_index][i + k], p, true);
            write_vertex(q, out + 3 * n + k, p);
         }
      }
      return n;
   }
   const uint32_t n = c_tri_count[c.cube_index];
   uint32_t kept = 0;
   for (uint32_t t = 0; t < n; t++) {
      float p[3][3];
      for (int k = 0; k < 3; k++) edge_point(c, c_tris[c.cube_index][3 * t + k], p[k], false);
      if (!has_area(p[0], p[1], p[2])) continue;
      if (EMIT)
         for (int k = 0; k < 3; k++) write_vertex(q, out + 3 * kept + k, p[k]);
      kept++;
   }
   return kept;
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

// exclusive scan of 256 values across the block (Hillis-Steele); returns this thread's prefix
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t x, uint32_t* scan) {
   scan[threadIdx.x] = x;
   __syncthreads();
   for (int s = 1; s < kBlock; s <<= 1) {
      const uint32_t add = (int)threadIdx.x >= s ? scan[threadIdx.x - s] : 0u;
      __syncthreads();
      scan[threadIdx.x] += add;
      __syncthreads();
   }
   const uint32_t r = scan[threadIdx.x] - x;
   __syncthreads();
   return r;
}

__global__ __launch_bounds__(kBlock) void k_iso_count(IsoParams q, uint32_t* __restrict__ block_counts) {
   __shared__ uint32_t scratch[kBlock];
   const uint64_t cell = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
   Cell c;
   const uint32_t n = load_cell(q, cell, c) ? cell_triangles<false>(q, c, nullptr) : 0u;
   const uint32_t total = block_sum(n, scratch);
   if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void k_iso_emit(IsoParams q, const uint32_t* __restrict__ block_offsets, UhVertex* __restrict__ verts) {
   __shared__ uint32_t scan[kBlock];
   const uint64_t cell = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
   Cell c;
   const bool mixed = load_cell(q, cell, c);
   const uint32_t n = mixed ? cell_triangles<false>(q, c, nullptr) : 0u;
   const uint32_t before = block_exclusive_scan(n, scan);
   if (n) cell_triangles<true>(q, c, verts + 3 * ((size_t)block_offsets[blockIdx.x] + before));
}

// diagnostics (uh_isosurface_cells): what each cell decided - its case index and the triangles it keeps
__global__ __launch_bounds__(kBlock) void k_iso_cells(IsoParams q, uint8_t* __restrict__ cube_index, uint8_t* __restrict__ tri_count) {
   const uint64_t cell = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
   if (cell >= (uint64_t)q.res * q.res * q.res) return;
   Cell c;
   const bool mixed = load_cell(q, cell, c);
   if (cube_index) cube_index[cell] = (uint8_t)c.cube_index;
   if (tri_count) tri_count[cell] = (uint8_t)(mixed ? cell_triangles<false>(q, c, nullptr) : 0u);
}

static bool load_tables() {
   static bool tables_loaded[64] = {false};
   int dev = 0;
   (void)hipGetDevice(&dev);
   if (dev >= 0 && dev < 64 && !tables_loaded[dev]) {
      if (hipMemcpyToSymbol(HIP_SYMBOL(c_edge_mask), kMcEdgeMask, sizeof(kMcEdgeMask)) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(c_tri_count), kMcTriCount, sizeof(kMcTriCount)) != hipSuccess ||
          hipMemcpyToSymbol(HIP_SYMBOL(c_tris), kMcTris, sizeof(kMcTris)) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(c_ref_tris), kMcRefTris, sizeof(kMcRefTris)) != hipSuccess)
         return false;
      tables_loaded[dev] = true;
   }
   return true;
}

}  // namespace

extern "C" int uh_isosurface_cells(uh_ctx* ctx, uint32_t resolution, float lo, float hi, float time, uint8_t* out_cube_index, uint8_t* out_triangle_count) {
   if (!ctx || resolution < 1 || resolution > 1024 || !(hi > lo)) return UH_ERR_INVALID_ARGUMENT;
   void* stream_v = nullptr;
   if (int st = uh_stream(ctx, &stream_v)) return st;
   hipStream_t stream = (hipStream_t)stream_v;
   if (!load_tables()) return UH_ERR_HIP;
   IsoParams q;
   q.res = resolution;
   q.lo = lo;
   q.h = (hi - lo) / (float)resolution;
   q.sphere_r = 8.0f * std::fabs(std::sin(time * 0.3f));
   q.inv_domain = 1.0f / (hi - lo);
   q.reference = uhi_iso_reference_triangulation(ctx) ? 1u : 0u;
   const uint64_t cells = (uint64_t)resolution * resolution * resolution;
   uint8_t *d_a = nullptr, *d_b = nullptr;
   if (hipMalloc(&d_a, cells) != hipSuccess || hipMalloc(&d_b, cells) != hipSuccess) {
      if (d_a) (void)hipFree(d_a);
      return UH_ERR_OUT_OF_MEMORY;
   }
   k_iso_cells<<<(uint32_t)((cells + kBlock - 1) / kBlock), kBlock, 0, stream>>>(q, d_a, d_b);
   int st = UH_OK;
   if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) st = UH_ERR_HIP;
   // blocking copies behind the wait (context.hip read_back: asynchronous device-to-host copies into pageable memory can land late)
   if (out_cube_index && hipMemcpy(out_cube_index, d_a, cells, hipMemcpyDeviceToHost) != hipSuccess) st = UH_ERR_HIP;
   if (out_triangle_count && hipMemcpy(out_triangle_count, d_b, cells, hipMemcpyDeviceToHost) != hipSuccess) st = UH_ERR_HIP;
   (void)hipFree(d_a);
   (void)hipFree(d_b);
   return st;
}

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
   q.reference = uhi_iso_reference_triangulation(ctx) ? 1u : 0u;
   const uint64_t cells = (uint64_t)resolution * resolution * resolution;
   const uint32_t blocks = (uint32_t)((cells + kBlock - 1) / kBlock);
   uint32_t *d_counts = nullptr, *d_chunks = nullptr;
   unsigned long long* d_total = nullptr;
   UhVertex* d_verts = nullptr;
   auto fail = [&](int st) {
      for (void* p : {(void*)d_counts, (void*)d_chunks, (void*)d_total, (void*)d_verts})
         if (p) (void)hipFree(p);
      return st;
   };
   if (!load_tables()) return UH_ERR_HIP;
   const uint32_t n_chunks = scan_chunk_count(blocks);
   if (hipMalloc(&d_counts, (size_t)blocks * sizeof(uint32_t)) != hipSuccess || hipMalloc(&d_chunks, (size_t)n_chunks * sizeof(uint32_t)) != hipSuccess ||
       hipMalloc(&d_total, sizeof(unsigned long long)) != hipSuccess)
      return fail(UH_ERR_OUT_OF_MEMORY);
   k_iso_count<<<blocks, kBlock, 0, stream>>>(q, d_counts);
   // exclusive scan of the per-block counts, on the device
   device_exclusive_scan_u32(d_counts, blocks, d_chunks, d_total, stream);
   unsigned long long total = 0;
   if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) return fail(UH_ERR_HIP);
   if (hipMemcpy(&total, d_total, sizeof(total), hipMemcpyDeviceToHost) != hipSuccess) return fail(UH_ERR_HIP);  // blocking (a local)
   // a mesh holds at most 4 Mi triangles (key = mesh << 22 | primitive); `total` is the true 64-bit sum (device_scan.h), so a
   // grid whose count would wrap the 32-bit offsets is refused here, before the emit pass sizes anything by it
   if (total > (1ull << 22)) return fail(UH_ERR_CAPACITY);
   std::vector<UhVertex> verts((size_t)total * 3);
   if (total) {
      if (hipMalloc(&d_verts, verts.size() * sizeof(UhVertex)) != hipSuccess) return fail(UH_ERR_OUT_OF_MEMORY);
      k_iso_emit<<<blocks, kBlock, 0, stream>>>(q, d_counts, d_verts);
      if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) return fail(UH_ERR_HIP);
      if (hipMemcpy(verts.data(), d_verts, verts.size() * sizeof(UhVertex), hipMemcpyDeviceToHost) != hipSuccess) return fail(UH_ERR_HIP);  // blocking
   }
   (void)fail(0);
   if (out_triangles) *out_triangles = (uint32_t)total;
   if (!total) {
      if (out_mesh_index) *out_mesh_index = 0xffffffffu;
      return UH_OK;  // nothing crosses the iso value: no mesh is added
   }
   std::vector<uint32_t> indices(verts.size());
   for (size_t i = 0; i < indices.size(); i++) indices[i] = (uint32_t)i;
   return uh_add_mesh(ctx, verts.data(), (uint32_t)verts.size(), indices.data(), (uint32_t)indices.size(), material, world3x4, out_mesh_index);
}
