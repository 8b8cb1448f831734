// valu_classes.hip — issue cost of single VALU instruction classes on gfx950 (wave-instructions per SIMD clock), measured with
// 8 independent dependency chains per lane at 8 waves per SIMD. Which of the node step's instructions run at the FMA's rate?
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/valu_classes.hip -o tools/microbench/valu_classes.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

#define OP8(STR)                                                                                                   \
   asm volatile(STR(0) STR(1) STR(2) STR(3) STR(4) STR(5) STR(6) STR(7)                                              \
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc")

#define S_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
#define S_ADD(k) "v_add_f32 %" #k ", %" #k ", %8\n\t"
#define S_MUL(k) "v_mul_f32 %" #k ", %" #k ", %8\n\t"
#define S_MAX(k) "v_max_f32 %" #k ", %" #k ", %8\n\t"
#define S_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n\t"
#define S_MIN3(k) "v_min3_f32 %" #k ", %" #k ", %8, %9\n\t"
#define S_CVTUB(k) "v_cvt_f32_ubyte1 %" #k ", %" #k "\n\t"
#define S_CVTU32(k) "v_cvt_f32_u32 %" #k ", %" #k "\n\t"
#define S_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n\t"
#define S_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n\t"
#define S_LSHR(k) "v_lshrrev_b32 %" #k ", 8, %" #k "\n\t"
#define S_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 8, 8\n\t"
#define S_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n\t"
#define S_ADDU(k) "v_add_u32 %" #k ", %" #k ", %8\n\t"
#define S_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %8\n\t"
#define S_CMP(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n\t"
#define S_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n\t"
#define S_CMPCND(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n\tv_cndmask_b32 %" #k ", %" #k ", %9, vcc\n\t"
#define S_PKFMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
#define S_MOV(k) "v_mov_b32 %" #k ", %8\n\t"
#define S_FMAMIX(k) "v_fma_mix_f32 %" #k ", %" #k ", %8, %9\n\t"
#define S_SUBREV(k) "v_sub_f32 %" #k ", %8, %" #k "\n\t"

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
   float a0 = threadIdx.x + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
   float b = 1.0001f, c = 0.5f;
   typedef float f2 __attribute__((ext_vector_type(2)));
   f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0, p5 = p1, p6 = p2, p7 = p3, pb = {b, b}, pc = {c, c};
   for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
         if (KIND == 0) OP8(S_FMA);
         else if (KIND == 1) OP8(S_ADD);
         else if (KIND == 2) OP8(S_MUL);
         else if (KIND == 3) OP8(S_MAX);
         else if (KIND == 4) OP8(S_MAX3);
         else if (KIND == 5) OP8(S_MIN3);
         else if (KIND == 6) OP8(S_CVTUB);
         else if (KIND == 7) OP8(S_CVTU32);
         else if (KIND == 8) OP8(S_AND);
         else if (KIND == 9) OP8(S_ANDOR);
         else if (KIND == 10) OP8(S_LSHR);
         else if (KIND == 11) OP8(S_BFE);
         else if (KIND == 12) OP8(S_PERM);
         else if (KIND == 13) OP8(S_ADDU);
         else if (KIND == 14) OP8(S_LSHLADD);
         else if (KIND == 15) OP8(S_CMP);
         else if (KIND == 16) OP8(S_CNDMASK);
         else if (KIND == 17) OP8(S_CMPCND);
         else if (KIND == 18) OP8(S_MOV);
         else if (KIND == 19) OP8(S_SUBREV);
         else {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t"
                         "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9\n\t"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
         }
      }
   }
   out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

int main() {
   int cus = 256;
   hipDeviceProp_t prop;
   CK(hipGetDeviceProperties(&prop, 0));
   cus = prop.multiProcessorCount;
   float* out;
   CK(hipMalloc(&out, sizeof(float) * cus * 8 * 256));
   hipEvent_t e0, e1;
   CK(hipEventCreate(&e0));
   CK(hipEventCreate(&e1));
   const char* names[21] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_max_f32", "v_max3_f32", "v_min3_f32", "v_cvt_f32_ubyte1", "v_cvt_f32_u32", "v_and_b32", "v_and_or_b32",
                            "v_lshrrev_b32", "v_bfe_u32", "v_perm_b32", "v_add_u32", "v_lshl_add_u32", "v_cmp_lt_f32 (vcc)", "v_cndmask_b32", "v_cmp + v_cndmask (pair)", "v_mov_b32",
                            "v_sub_f32", "v_pk_fma_f32"};
   const int per[21] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 16, 8, 8, 8};
   // reference: the clock the device reports, and the fma's rate as the yardstick
   printf("device %s, %d CUs, reported clock %d kHz\n", prop.name, cus, prop.clockRate);
   double fma_ns = 0;
   for (int kind = 0; kind < 21; kind++) {
      const int iters = 4096, bpc = 8;
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
         CK(hipEventRecord(e0));
         switch (kind) {
#define C(K) case K: k<K><<<cus * bpc, 256>>>(out, iters); break;
            C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19)
            default: k<20><<<cus * bpc, 256>>>(out, iters); break;
         }
         CK(hipEventRecord(e1));
         CK(hipEventSynchronize(e1));
         CK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double instr_per_simd = (double)bpc * iters * 8 * per[kind];
      const double ns = ms * 1e6 / instr_per_simd;
      if (kind == 0) fma_ns = ns;
      printf("%-26s %.3f ms -> %.3f ns per wave-instruction per SIMD = %.2f x v_fma_f32\n", names[kind], ms, ns, ns / fma_ns);
   }
   return 0;
}
