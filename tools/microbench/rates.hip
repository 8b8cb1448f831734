// rates.hip — two calibration microbenchmarks for the traversal kernels (MI355X, gfx950):
//  (1) VALU issue: wave-instructions per cycle per SIMD for the instruction classes of the BVH node step, at 1..8 waves/SIMD;
//  (2) record gather: random R-byte records (R = 64, 128: a BVH4 / BVH8 node) fetched by every lane from a table of T MiB,
//      independent (throughput) or as a dependent chain (the next index comes out of the loaded record: what a traversal does).
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/rates.hip -o tools/microbench/rates.bin ; run on the GPU box: tools/microbench/rates.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(float* out, int iters, float seed) {
   float a0 = seed + threadIdx.x, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
   uint32_t q = __float_as_uint(a0) * 2654435761u;
   for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
         if (KIND == 0) {  // 8 independent v_fma_f32
            a0 = fmaf(a0, 1.0001f, 0.5f); a1 = fmaf(a1, 1.0001f, 0.5f); a2 = fmaf(a2, 1.0001f, 0.5f); a3 = fmaf(a3, 1.0001f, 0.5f);
            a4 = fmaf(a4, 1.0001f, 0.5f); a5 = fmaf(a5, 1.0001f, 0.5f); a6 = fmaf(a6, 1.0001f, 0.5f); a7 = fmaf(a7, 1.0001f, 0.5f);
         } else if (KIND == 1) {  // v_cvt_f32_ubyteN + v_fma_f32 (the slab plane evaluation): 4 + 4
            a0 = fmaf((float)(q & 0xffu), a4, a0); a1 = fmaf((float)((q >> 8) & 0xffu), a5, a1);
            a2 = fmaf((float)((q >> 16) & 0xffu), a6, a2); a3 = fmaf((float)(q >> 24), a7, a3);
            q += 0x01010101u;
         } else if (KIND == 2) {  // v_max3 / v_min3 (4 + 4)
            a0 = fmaxf(fmaxf(a0, a1), a2); a1 = fminf(fminf(a1, a2), a3); a2 = fmaxf(fmaxf(a2, a3), a4); a3 = fminf(fminf(a3, a4), a5);
            a4 = fmaxf(fmaxf(a4, a5), a6); a5 = fminf(fminf(a5, a6), a7); a6 = fmaxf(fmaxf(a6, a7), a0); a7 = fminf(fminf(a7, a0), a1);
         } else if (KIND == 3) {  // v_cmp + v_cndmask (4 + 4)
            a0 = a0 <= a1 ? a2 : a3; a1 = a1 <= a2 ? a3 : a4; a2 = a2 <= a3 ? a4 : a5; a3 = a3 <= a4 ? a5 : a6;
            a4 = a4 + 1.0f; a5 = a5 + 1.0f; a6 = a6 - 1.0f; a7 = a7 + 2.0f;
         } else {  // v_rcp_f32 x 4 + 4 fma
            a0 = __builtin_amdgcn_rcpf(a0) + 1.0f; a1 = __builtin_amdgcn_rcpf(a1) + 1.0f; a2 = __builtin_amdgcn_rcpf(a2) + 1.0f; a3 = __builtin_amdgcn_rcpf(a3) + 1.0f;
         }
      }
   }
   out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + __uint_as_float(q);
}

// every lane fetches `steps` records of REC bytes (REC/16 dwordx4 loads each) from a table of n_rec records
template <int REC, bool DEP>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ table, uint32_t n_rec, int steps, uint32_t* out) {
   uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
   uint32_t acc = 0;
   for (int s = 0; s < steps; s++) {
      idx = idx * 747796405u + 2891336453u;
      const uint32_t r = (uint32_t)(((uint64_t)(idx >> 4) * n_rec) >> 28);
      const uint4* p = table + (size_t)r * (REC / 16);
      uint32_t v = 0;
#pragma unroll
      for (int k = 0; k < REC / 16; k++) {
         const uint4 w = p[k];
         v ^= w.x + w.y + w.z + w.w;
      }
      acc += v;
      if (DEP) idx ^= v;  // the next record depends on this one's contents
   }
   out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// LPR lanes share one record: lane j of a group loads the 16-byte chunks j, j + LPR, ... of the group's record (REC/16/LPR loads per lane);
// do the lanes of a group that read one line cost the address path one slot or LPR slots?
template <int REC, int LPR>
__global__ __launch_bounds__(256) void k_gather_group(const uint4* __restrict__ table, uint32_t n_rec, int steps, uint32_t* out) {
   const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
   uint32_t idx = (tid / LPR) * 2654435761u;
   const uint32_t j = tid % LPR;
   uint32_t acc = 0;
   for (int s = 0; s < steps; s++) {
      idx = idx * 747796405u + 2891336453u;
      const uint32_t r = (uint32_t)(((uint64_t)(idx >> 4) * n_rec) >> 28);
      const uint4* p = table + (size_t)r * (REC / 16);
      uint32_t v = 0;
#pragma unroll
      for (int k = 0; k < REC / 16 / LPR; k++) {
         const uint4 w = p[j + k * LPR];
         v ^= w.x + w.y + w.z + w.w;
      }
      acc += v;
   }
   out[tid] = acc;
}

int main() {
   hipDeviceProp_t prop;
   CK(hipGetDeviceProperties(&prop, 0));
   const int cus = prop.multiProcessorCount;
   printf("device: %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
   float* out;
   CK(hipMalloc(&out, sizeof(float) * cus * 8 * 256));
   hipEvent_t e0, e1;
   CK(hipEventCreate(&e0));
   CK(hipEventCreate(&e1));
   const char* names[5] = {"8x v_fma_f32", "4x cvt_ubyte + 4x fma", "4x max3 + 4x min3", "4x (cmp+cndmask) + 4x add", "4x (rcp + add)"};
   const int per_unroll[5] = {8, 8, 8, 12, 8};
   for (int kind = 0; kind < 5; kind++) {
      for (int bpc = 1; bpc <= 8; bpc *= 2) {
         const int iters = 4096;
         float ms = 0;
         for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            switch (kind) {
               case 0: k_valu<0><<<cus * bpc, 256>>>(out, iters, 1.0f); break;
               case 1: k_valu<1><<<cus * bpc, 256>>>(out, iters, 1.0f); break;
               case 2: k_valu<2><<<cus * bpc, 256>>>(out, iters, 1.0f); break;
               case 3: k_valu<3><<<cus * bpc, 256>>>(out, iters, 1.0f); break;
               default: k_valu<4><<<cus * bpc, 256>>>(out, iters, 1.0f); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
         }
         const double instr_per_simd = (double)bpc * iters * 8 * per_unroll[kind];  // bpc waves per SIMD (4 waves per block, 4 SIMDs)
         printf("VALU %-28s %d waves/SIMD: %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f clk at 2.4 GHz)\n", names[kind], bpc, ms, ms * 1e6 / instr_per_simd,
                ms * 1e6 / instr_per_simd * 2.4);
      }
   }
   // gather
   uint32_t* gout;
   CK(hipMalloc(&gout, sizeof(uint32_t) * cus * 8 * 256));
   const size_t max_bytes = 512ull << 20;
   uint4* table;
   CK(hipMalloc(&table, max_bytes));
   CK(hipMemset(table, 1, max_bytes));
   const int mibs[4] = {6, 24, 96, 512};
   for (int dep = 0; dep < 2; dep++)
      for (int rec = 64; rec <= 128; rec *= 2)
         for (int t = 0; t < 4; t++)
            for (int bpc = 4; bpc <= 8; bpc *= 2) {
               const uint32_t n_rec = (uint32_t)(((size_t)mibs[t] << 20) / rec);
               const int steps = 256;
               float ms = 0;
               for (int rep = 0; rep < 2; rep++) {
                  CK(hipEventRecord(e0));
                  if (rec == 64) {
                     if (dep) k_gather<64, true><<<cus * bpc, 256>>>(table, n_rec, steps, gout);
                     else k_gather<64, false><<<cus * bpc, 256>>>(table, n_rec, steps, gout);
                  } else {
                     if (dep) k_gather<128, true><<<cus * bpc, 256>>>(table, n_rec, steps, gout);
                     else k_gather<128, false><<<cus * bpc, 256>>>(table, n_rec, steps, gout);
                  }
                  CK(hipEventRecord(e1));
                  CK(hipEventSynchronize(e1));
                  CK(hipEventElapsedTime(&ms, e0, e1));
               }
               const double recs = (double)cus * bpc * 256 * steps;
               printf("GATHER %3d-B records, table %3d MiB, %s, %d blocks/CU: %.3f ms -> %.1f G records/s, %.2f TB/s, %.1f B/clk/CU\n", rec, mibs[t], dep ? "dependent  " : "independent", bpc, ms,
                      recs / ms / 1e6, recs * rec / ms / 1e9, recs * rec / (ms * 1e-3) / cus / 2.4e9);
            }
   for (int t = 0; t < 2; t++) {
      const uint32_t n_rec = (uint32_t)(((size_t)mibs[t] << 20) / 64);
      const int steps = 256, bpc = 8;
      for (int lpr = 1; lpr <= 4; lpr *= 2) {
         float ms = 0;
         for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            if (lpr == 1) k_gather_group<64, 1><<<cus * bpc, 256>>>(table, n_rec, steps, gout);
            else if (lpr == 2) k_gather_group<64, 2><<<cus * bpc, 256>>>(table, n_rec, steps, gout);
            else k_gather_group<64, 4><<<cus * bpc, 256>>>(table, n_rec, steps, gout);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
         }
         const double recs = (double)cus * bpc * 256 / lpr * steps;
         printf("GROUP  64-B records, table %3d MiB, %d lanes per record (%d loads per lane): %.3f ms -> %.1f G records/s, %.2f clk per record per CU\n", mibs[t], lpr, 4 / lpr, ms,
                recs / ms / 1e6, ms * 1e-3 * 2.4e9 * cus / recs);
      }
   }
   return 0;
}
