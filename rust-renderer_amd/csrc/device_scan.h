// device_scan.h — exclusive scan of a uint32 array on the device (in place), hand-written: chunks of 2048 values
// (8 per thread), the chunk totals by one block, then the chunk offsets added back. Used by the marching-cubes
// extraction (isosurface.hip: per-block triangle counts) and the PLOC builder (lbvh.hip: surviving clusters).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace uh_scan {

constexpr uint32_t kBlock = 256, kPer = 8, kChunk = kBlock * kPer;

// exclusive scan of 256 values across the block (Hillis-Steele); returns this thread's prefix
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t x, uint32_t* scan) {
   scan[threadIdx.x] = x;
   __syncthreads();
   for (uint32_t s = 1; s < kBlock; s <<= 1) {
      const uint32_t add = threadIdx.x >= s ? scan[threadIdx.x - s] : 0u;
      __syncthreads();
      scan[threadIdx.x] += add;
      __syncthreads();
   }
   const uint32_t r = scan[threadIdx.x] - x;
   __syncthreads();
   return r;
}

static __global__ __launch_bounds__(kBlock) void k_scan_chunks(uint32_t* __restrict__ data, uint32_t n, uint32_t* __restrict__ chunk_totals) {
   __shared__ uint32_t scan[kBlock];
   const uint32_t base = blockIdx.x * kChunk + threadIdx.x * kPer;
   uint32_t v[kPer], sum = 0;
   for (uint32_t k = 0; k < kPer; k++) {
      v[k] = base + k < n ? data[base + k] : 0u;
      sum += v[k];
   }
   uint32_t prefix = block_exclusive_scan(sum, scan);
   for (uint32_t k = 0; k < kPer; k++) {
      if (base + k < n) data[base + k] = prefix;
      prefix += v[k];
   }
   if (threadIdx.x == kBlock - 1) chunk_totals[blockIdx.x] = prefix;
}
// one block over the chunk totals (serial over tiles of 2048 when there are more), leaves their exclusive scan and the grand total
static __global__ __launch_bounds__(kBlock) void k_scan_totals(uint32_t* __restrict__ chunk_totals, uint32_t n_chunks, unsigned long long* __restrict__ grand_total) {
   __shared__ uint32_t scan[kBlock];
   __shared__ unsigned long long s_total;  // the true sum, in 64 bits: the 32-bit prefixes below wrap silently past 2^32
   if (threadIdx.x == 0) s_total = 0ull;
   __syncthreads();
   unsigned long long carry = 0;
   for (uint32_t tile = 0; tile < n_chunks; tile += kChunk) {
      const uint32_t base = tile + threadIdx.x * kPer;
      uint32_t v[kPer], sum = 0;
      unsigned long long wide = 0ull;
      for (uint32_t k = 0; k < kPer; k++) {
         v[k] = base + k < n_chunks ? chunk_totals[base + k] : 0u;
         sum += v[k];
         wide += v[k];
      }
      if (wide) atomicAdd(&s_total, wide);
      uint32_t prefix = block_exclusive_scan(sum, scan) + (uint32_t)carry;
      for (uint32_t k = 0; k < kPer; k++) {
         if (base + k < n_chunks) chunk_totals[base + k] = prefix;
         prefix += v[k];
      }
      __shared__ unsigned long long s_carry;
      if (threadIdx.x == kBlock - 1) s_carry = (unsigned long long)prefix;
      __syncthreads();
      carry = s_carry;
      __syncthreads();
   }
   __syncthreads();
   if (threadIdx.x == 0) *grand_total = s_total;
}
static __global__ __launch_bounds__(kBlock) void k_scan_add(uint32_t* __restrict__ data, uint32_t n, const uint32_t* __restrict__ chunk_offsets) {
   const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
   if (i < n) data[i] += chunk_offsets[i / kChunk];
}

}  // namespace uh_scan

inline uint32_t scan_chunk_count(uint32_t n) { return (n + uh_scan::kChunk - 1) / uh_scan::kChunk; }

// data[0..n) -> its exclusive scan, *grand_total (device) = the sum, accumulated in 64 bits from the chunk totals: a caller
// must refuse a total >= 2^32 (the 32-bit offsets have wrapped by then); a single chunk of 2,048 values must itself sum
// below 2^32. chunk_scratch: scan_chunk_count(n) values.
inline void device_exclusive_scan_u32(uint32_t* data, uint32_t n, uint32_t* chunk_scratch, unsigned long long* grand_total, hipStream_t stream) {
   const uint32_t chunks = scan_chunk_count(n);
   uh_scan::k_scan_chunks<<<chunks ? chunks : 1, uh_scan::kBlock, 0, stream>>>(data, n, chunk_scratch);
   uh_scan::k_scan_totals<<<1, uh_scan::kBlock, 0, stream>>>(chunk_scratch, chunks, grand_total);
   uh_scan::k_scan_add<<<n / uh_scan::kBlock + 1, uh_scan::kBlock, 0, stream>>>(data, n, chunk_scratch);
}
