// hip_churn_repro.cpp - a LIBRARY-FREE reproducer for the host heap corruption round 4's context-churn soaks met under the HIP
// runtime bundled with the PyTorch wheel (profiles/README.md "The soak crash"). Nothing of libutopian_hip.so is in here: plain HIP
// calls in the pattern a context's life makes - streams, events and device allocations created and destroyed a few hundred times a
// second, work on several non-blocking streams ordered by events, and pageable device-to-host copies (hipMemcpyAsync + a wait for the
// stream, and blocking hipMemcpy) into host buffers that are FREED STRAIGHT AFTER - with two checks the soaks did not have:
//   (1) when the wait returns, every byte of the destination must be there (a copy signalled complete too early shows as stale bytes);
//   (2) the freed block is allocated again at once, filled with a canary, and must still hold it a moment later (a write that lands
//       after the copy "completed" shows as a broken canary - before glibc trips over it).
// Build:  hipcc -O2 --offload-arch=gfx950 -DWITH_KERNEL tools/hip_churn_repro.cpp -o repro_kernel      (device code by hipcc 7.2)
//         g++ -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/hip_churn_repro.cpp -L/opt/rocm/lib -lamdhip64 -o repro_plain
//                                                                     (no device code at all: only the C API's headers are "7.2")
// Run:    ./repro_kernel 100            against /opt/rocm's runtime (RUNPATH / LD_LIBRARY_PATH)
//         LD_PRELOAD=<site-packages>/torch/lib/libamdhip64.so ./repro_kernel 100     against the wheel's copy
// Exit status 0 = clean, 3 = a check failed (message on stderr); a crash is a crash. tools/run_churn_repro.sh runs the matrix.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define CHECK(expr)                                                                                  \
   do {                                                                                              \
      hipError_t e_ = (expr);                                                                        \
      if (e_ != hipSuccess) {                                                                        \
         std::fprintf(stderr, "%s: %s (iteration %llu)\n", #expr, hipGetErrorString(e_), (unsigned long long)g_iter); \
         std::exit(2);                                                                               \
      }                                                                                              \
   } while (0)

static uint64_t g_iter = 0;
static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
   g_rng ^= g_rng << 13;
   g_rng ^= g_rng >> 7;
   g_rng ^= g_rng << 17;
   return (uint32_t)(g_rng >> 32);
}

#ifdef WITH_KERNEL
__global__ void k_fill(uint8_t* p, size_t n, uint8_t v) {
   for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
#endif

static void fill(uint8_t* dev, size_t n, uint8_t v, hipStream_t s) {
#ifdef WITH_KERNEL
   hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, s, dev, n, v);
   CHECK(hipGetLastError());
#else
   CHECK(hipMemsetAsync(dev, v, n, s));
#endif
}

static int fail(const char* what, size_t at, size_t n, unsigned got, unsigned want) {
   std::fprintf(stderr, "CHECK FAILED (%s): byte %zu of %zu is 0x%02x, expected 0x%02x, iteration %llu\n", what, at, n, got, want, (unsigned long long)g_iter);
   return 3;
}

int main(int argc, char** argv) {
   const double seconds = argc > 1 ? std::atof(argv[1]) : 30.0;
   g_rng ^= (uint64_t)(argc > 2 ? std::atoll(argv[2]) : 1) * 0xD1B54A32D192ED03ull;
   int rt = 0;
   CHECK(hipRuntimeGetVersion(&rt));
   std::printf("HIP_VERSION (headers) %d, runtime %d, %s\n", HIP_VERSION, rt,
#ifdef WITH_KERNEL
               "device code by hipcc"
#else
               "no device code"
#endif
   );
   CHECK(hipSetDevice(0));
   const auto t0 = std::chrono::steady_clock::now();
   uint64_t copies = 0, bytes = 0;
   std::map<uint64_t, std::vector<uint8_t>> churn;  // host-side small objects: what reuses a freed destination at once
   while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
      g_iter++;
      // ---- one "context": streams, events, allocations
      constexpr int NS = 4, NE = 10, NB = 12;
      hipStream_t st[NS];
      hipEvent_t ev[NE];
      uint8_t* buf[NB];
      size_t len[NB];
      for (auto& s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      for (int k = 0; k < NE; k++) CHECK(k < 2 ? hipEventCreate(&ev[k]) : hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
      for (int k = 0; k < NB; k++) {
         const uint32_t cls = rnd() % 4;  // 4 KB .. 64 MB
         len[k] = cls == 0 ? 4096 + rnd() % 65536 : cls == 1 ? (1u << 20) + rnd() % (1u << 20) : cls == 2 ? (8u << 20) + rnd() % (8u << 20) : (32u << 20) + rnd() % (32u << 20);
         CHECK(hipMalloc((void**)&buf[k], len[k]));
      }
      // ---- "frames": fills on the streams, ordered by events across them
      const uint8_t pat = (uint8_t)(1 + g_iter % 250);
      for (int k = 0; k < NB; k++) fill(buf[k], len[k], (uint8_t)(pat + k), st[k % NS]);
      for (int k = 0; k < NS; k++) CHECK(hipEventRecord(ev[2 + k], st[k]));
      for (int k = 0; k < NS; k++) CHECK(hipStreamWaitEvent(st[(k + 1) % NS], ev[2 + k], 0));
      // ---- "read-backs": pageable destinations, freed straight after
      for (int k = 0; k < NB; k++) {
         const bool async = (rnd() & 1) != 0;
         const size_t n = len[k] > (16u << 20) ? (16u << 20) : len[k];
         uint8_t* dst = (uint8_t*)std::malloc(n);
         std::memset(dst, 0xEE, n);
         hipStream_t s = st[k % NS];
         if (async) {
            CHECK(hipMemcpyAsync(dst, buf[k], n, hipMemcpyDeviceToHost, s));
            CHECK(hipStreamSynchronize(s));
         } else {
            CHECK(hipStreamSynchronize(s));
            CHECK(hipMemcpy(dst, buf[k], n, hipMemcpyDeviceToHost));
         }
         const uint8_t want = (uint8_t)(pat + k);
         for (size_t i = 0; i < n; i += (i + 4096 < n ? 4096 - (i % 7) : 1))
            if (dst[i] != want) return fail(async ? "hipMemcpyAsync + hipStreamSynchronize returned before the data" : "hipMemcpy returned before the data", i, n, dst[i], want);
         if (dst[n - 1] != want) return fail("last byte", n - 1, n, dst[n - 1], want);
         std::free(dst);
         // the block goes back to work at once: same size (glibc hands the same chunk back), canary, a little host-side churn, canary check
         uint8_t* again = (uint8_t*)std::malloc(n);
         std::memset(again, 0xA5, n);
         for (int j = 0; j < 64; j++) churn[rnd() % 4096] = std::vector<uint8_t>(16 + rnd() % 512, (uint8_t)j);
         for (size_t i = 0; i < n; i += 61)
            if (again[i] != 0xA5) return fail("a write landed in memory freed after the copy had 'completed'", i, n, again[i], 0xA5);
         std::free(again);
         copies++;
         bytes += n;
      }
      // small copies into locals (the grid builders' totals and samples)
      for (int k = 0; k < NB; k++) {
         uint32_t local[16];
         std::memset(local, 0, sizeof(local));
         CHECK(hipMemcpyAsync(local, buf[k], sizeof(local), hipMemcpyDeviceToHost, st[k % NS]));
         CHECK(hipStreamSynchronize(st[k % NS]));
         const uint8_t want = (uint8_t)(pat + k);
         if (((const uint8_t*)local)[63] != want) return fail("copy into a local", 63, 64, ((const uint8_t*)local)[63], want);
      }
      // ---- the context goes: device idle, allocations, events, streams (uh_destroy's order)
      CHECK(hipDeviceSynchronize());
      for (int k = 0; k < NB; k++) CHECK(hipFree(buf[k]));
      for (auto& e : ev) CHECK(hipEventDestroy(e));
      for (auto& s : st) CHECK(hipStreamDestroy(s));
      if (g_iter % 200 == 0) {
         std::printf("  %llu contexts, %llu copies, %.1f GB, %.0f s\n", (unsigned long long)g_iter, (unsigned long long)copies, bytes / 1e9,
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
         std::fflush(stdout);
      }
   }
   std::printf("clean: %llu contexts, %llu read-backs, %.1f GB\n", (unsigned long long)g_iter, (unsigned long long)copies, bytes / 1e9);
   return 0;
}
