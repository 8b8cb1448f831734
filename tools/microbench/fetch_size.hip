// fetch_size.hip - what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of this library's kernels, against
// byte counts known by construction (the guide calibrates the wide streaming read only - FETCH_SIZE reports half of it - and says
// "calibrate on a known byte count in your own access pattern before trusting an absolute"). Every table is 2 GiB (eight times the
// Infinity Cache) and every line is touched once, so what a kernel asks of memory is what HBM delivers.
//   stream_read16    every lane 16 contiguous bytes (the path-state planes, the hit records)
//   stream_write16   the same, stores
//   gather64         every lane a whole 64-byte record, four 16-byte loads, records in a scattered order (shading packets, grid records)
//   gather48         every lane three 16-byte loads of a 64-byte-stride record (tree nodes, triangle packets)
//   gather16         every lane one 16-byte load of a scattered 128-byte line
//   gather4          every lane one 4-byte load of a scattered 64-byte line (texels)
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/fetch_size.hip -o tools/microbench/fetch_size.bin
// run:   rocprofv3 --pmc FETCH_SIZE -d <dir> --output-format csv -- tools/microbench/fetch_size.bin ; the same with WRITE_SIZE ;
//        python tools/microbench/fetch_size_report.py <dir> ...
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x)                                                                                   \
   do {                                                                                         \
      hipError_t e_ = (x);                                                                      \
      if (e_ != hipSuccess) {                                                                   \
         printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);          \
         return 1;                                                                              \
      }                                                                                         \
   } while (0)

constexpr uint64_t kBytes = 2ull << 30;
constexpr uint32_t kOdd = 2654435761u;  // scatter: i -> (i * odd) mod 2^k is a permutation of the records

__global__ __launch_bounds__(256) void stream_read16(const float4* __restrict__ t, uint64_t n, float* out) {
   float acc = 0.0f;
   for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
      typedef float f4_t __attribute__((ext_vector_type(4)));
      const f4_t v = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(t) + i);  // (the library's ld_stream)
      acc += v.x + v.y + v.z + v.w;
   }
   if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void stream_write16(float4* __restrict__ t, uint64_t n) {
   for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
      typedef float f4_t __attribute__((ext_vector_type(4)));
      const f4_t v = {1.0f, 2.0f, 3.0f, (float)i};
      __builtin_nontemporal_store(v, reinterpret_cast<f4_t*>(t) + i);  // (the library's st_stream)
   }
}
// records of `stride16` quads, `loads` of them read per record, `nrec` records (a power of two), each by one lane
template <int LOADS>
__global__ __launch_bounds__(256) void gather_rec(const float4* __restrict__ t, uint32_t nrec, uint32_t stride16, float* out) {
   float acc = 0.0f;
   for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nrec; i += gridDim.x * 256) {
      const uint32_t r = (i * kOdd) & (nrec - 1);
      const float4* p = t + (uint64_t)r * stride16;
#pragma unroll
      for (int k = 0; k < LOADS; k++) {
         const float4 v = p[k];
         acc += v.x + v.w;
      }
   }
   if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void gather4(const float* __restrict__ t, uint32_t nline, float* out) {
   float acc = 0.0f;
   for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nline; i += gridDim.x * 256) {
      const uint32_t r = (i * kOdd) & (nline - 1);
      acc += t[(uint64_t)r * 16 + (i & 15)];
   }
   if (acc == 123.456f) out[0] = acc;
}

int main() {
   float4* t = nullptr;
   float* out = nullptr;
   CK(hipMalloc(&t, kBytes));
   CK(hipMalloc(&out, 64));
   CK(hipMemset(t, 0, kBytes));
   const uint64_t n16 = kBytes / 16;
   const dim3 grid(256 * 16);
   for (int rep = 0; rep < 2; rep++) {
      stream_read16<<<grid, 256>>>(t, n16, out);
      stream_write16<<<grid, 256>>>(t, n16);
      gather_rec<4><<<grid, 256>>>(t, (uint32_t)(kBytes / 64), 4, out);   // gather64
      gather_rec<3><<<grid, 256>>>(t, (uint32_t)(kBytes / 64), 4, out);   // gather48
      gather_rec<1><<<grid, 256>>>(t, (uint32_t)(kBytes / 128), 8, out);  // gather16
      gather4<<<grid, 256>>>((const float*)t, (uint32_t)(kBytes / 64), out);
      CK(hipDeviceSynchronize());
   }
   printf("true bytes: stream_read16 %llu, stream_write16 %llu, gather64 %llu asked = lines, gather48 %llu asked / %llu in 64-byte lines, gather16 %llu asked / %llu in 64-byte lines / %llu in 128-byte lines, gather4 %llu asked / %llu in 64-byte lines\n",
          (unsigned long long)kBytes, (unsigned long long)kBytes, (unsigned long long)kBytes, (unsigned long long)(kBytes / 64 * 48), (unsigned long long)kBytes,
          (unsigned long long)(kBytes / 128 * 16), (unsigned long long)(kBytes / 128 * 64), (unsigned long long)kBytes, (unsigned long long)(kBytes / 64 * 4), (unsigned long long)kBytes);
   return 0;
}
