// kernels.hip — gfx950 kernels of the wavefront path tracer and the ReSTIR passes.
//
// Replaces (reference, utopian/shaders/): pathtrace_reference/reference.{rgen,rchit,rmiss},
// restir/{reset_reservoirs.comp,initial_ris.rgen,temporal_reuse.rgen,spatial_reuse.rgen} and the
// driver's traceRayEXT traversal. One launch of reference.rgen (vkCmdTraceRaysKHR(W,H,1),
// utopian/src/renderers/mod.rs:357) becomes, per sample and bounce, a short chain of launches over
// compacted queues of path ids:
//     generate -> [ trace_closest -> shade_miss | shade_hit -> trace_shadow(sun) -> trace_shadow(light) ]*
//              -> finish_sample
// Traversal kernels are persistent: a grid that just fills the chip pulls 64-ray batches from a
// device-side cursor until the queue (whose length only the device knows) is drained.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_math.h"
#include "device_types.h"

namespace uh {

constexpr int kBlock = 256;                 // 4 waves
constexpr int kWavesPerBlock = kBlock / 64;
#ifndef UH_SUN_RAYS_PER_LANE
#define UH_SUN_RAYS_PER_LANE 2  // k_trace_sun_grid (inline records): chains of dependent loads a lane keeps in flight
#endif
#ifndef UH_SHADE_HIT_BLOCKS
#define UH_SHADE_HIT_BLOCKS 4  // blocks per CU the register budget of k_shade_hit is sized for
#endif
#ifndef UH_LDS_STACK
#define UH_LDS_STACK 16
#endif
// Build-time experiments of round 5 on the traversal step (profiles/README.md "k_trace_closest, round 5"; tools/ab.sh compares builds):
#ifndef UH_PK_SLAB
#define UH_PK_SLAB 0  // 1: the slab test's 24 fused multiply-adds as 12 v_pk_fma_f32 (near and far plane of an axis in one instruction)
#endif
#ifndef UH_GATE_LEAVES
#define UH_GATE_LEAVES 0  // k > 0: lanes at a leaf wait (no load, no test) until k lanes of the wave stand at one, a lane has waited UH_GATE_WAIT iterations, or no lane is at a node
#endif
#ifndef UH_GATE_WAIT
#define UH_GATE_WAIT 3
#endif
constexpr int kLdsStack = UH_LDS_STACK;      // per-lane traversal stack entries kept in LDS (16 KiB per 256-thread block)
constexpr int kSpillStack = (int)kTraversalStackEntries - kLdsStack;             // overflow entries in private memory (rarely touched); the host refuses trees deeper than the two together hold

// ------------------------------------------------------------------------------------------
// BVH4 traversal (thread per ray). Closest hit: min t over all triangles with tmin < t < tmax,
// ties broken by the smaller key (mesh << 22 | prim) — independent of traversal order, because
// node boxes are padded conservatively by the builder. Any hit: first triangle with
// tmin < t < tmax and t <= tlimit.
// ------------------------------------------------------------------------------------------
struct Hit {
   float t, u, v;
   uint32_t idx;  // packet index, kEmptyRef = miss
   uint32_t key;
};

__device__ __forceinline__ float dot_fma(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 cross_fma(V3 a, V3 b) {
   return v3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

// Moeller-Trumbore on a baked packet (a = v0.xyz e1.x, b = e1.yz e2.xy, c = e2.z key): DESIGN.md "Arithmetic contract"
template <bool ANY>
__device__ __forceinline__ bool tri_compute(float4 a, float4 b, float4 c, uint32_t i, V3 o, V3 d, float tmin, float tlimit, Hit& best) {
   V3 v0 = v3(a.x, a.y, a.z), e1 = v3(a.w, b.x, b.y), e2 = v3(b.z, b.w, c.x);
   uint32_t key = __float_as_uint(c.y);
   V3 p = cross_fma(d, e2);
   float det = dot_fma(e1, p);
   if (det == 0.0f) return false;
   float inv = 1.0f / det;
   V3 tv = o - v0;
   float u = dot_fma(tv, p) * inv;
   if (!(u >= 0.0f && u <= 1.0f)) return false;
   V3 q = cross_fma(tv, e1);
   float v = dot_fma(d, q) * inv;
   if (!(v >= 0.0f && u + v <= 1.0f)) return false;
   float t = dot_fma(e2, q) * inv;
   if (!(t > tmin)) return false;
   if (ANY) {
      return t < best.t && t <= tlimit;
   } else {
      if (t < best.t || (t == best.t && key < best.key)) {
         best.t = t;
         best.u = u;
         best.v = v;
         best.idx = i;
         best.key = key;
         return true;
      }
      return false;
   }
}
template <bool ANY>
__device__ __forceinline__ bool tri_test(const float4* __restrict__ tris, uint32_t i, V3 o, V3 d, float tmin, float tlimit, Hit& best) {
   float4 a = tris[kTriStride16 * (size_t)i + 0], b = tris[kTriStride16 * (size_t)i + 1], c = tris[kTriStride16 * (size_t)i + 2];
   return tri_compute<ANY>(a, b, c, i, o, d, tmin, tlimit, best);
}

__device__ __forceinline__ float safe_rcp_dir(float x) {
   // the slab test only has to be conservative (boxes are padded by >= 1e-5 relative, the 1-ulp hardware
   // reciprocal is 1e-7): no IEEE division here; a zero component becomes +-1e-30 so no inf/NaN appears
   return __builtin_amdgcn_rcpf(fabsf(x) < 1e-30f ? copysignf(1e-30f, x) : x);
}

struct Trav {
   V3 o, d, idir;
   float tmin, tlimit;
   Hit best;
   int sp;
   uint32_t cur;
   // the quantisation frame of `cur` when the lane DESCENDED into it from its parent (node_quant.h qn_inherit): origin, and the three
   // step exponent bytes - 0 = none (cur was popped, is a leaf, or is the root: the visit loads the record's first quad)
#if UH_INHERIT_FRAME
   float fx, fy, fz;
   uint32_t fexp;
#endif
#if UH_GATE_LEAVES
   uint32_t wait;  // iterations this lane has stood at a leaf without testing it
#endif
};

__device__ __forceinline__ void trav_init(Trav& t, float4 ro, float4 rd, float tmin, float tmax, float tlimit) {
   t.o = v3(ro.x, ro.y, ro.z);
   t.d = v3(rd.x, rd.y, rd.z);
   t.idir = v3(safe_rcp_dir(t.d.x), safe_rcp_dir(t.d.y), safe_rcp_dir(t.d.z));
   t.tmin = tmin;
   t.tlimit = tlimit;
   t.best.t = tmax;
   t.best.u = t.best.v = 0.0f;
   t.best.idx = kEmptyRef;
   t.best.key = 0xffffffffu;
   t.sp = 0;
   t.cur = 0;
#if UH_GATE_LEAVES
   t.wait = 0;
#endif
#if UH_INHERIT_FRAME
   t.fx = t.fy = t.fz = 0.0f;
   t.fexp = 0u;
#endif
}

__device__ __forceinline__ void trav_push(Trav& t, uint32_t* lds_col, uint32_t* spill, uint32_t ref) {
   if (t.sp < kLdsStack)
      lds_col[t.sp * 64] = ref;
   else if (t.sp < kLdsStack + kSpillStack)
      spill[t.sp - kLdsStack] = ref;
   else
      return;
   t.sp++;
}
__device__ __forceinline__ uint32_t trav_pop(Trav& t, const uint32_t* lds_col, const uint32_t* spill) {
   if (t.sp == 0) return kEmptyRef;
   t.sp--;
   if (t.sp < kLdsStack) {
      // inline asm: written as `sp < kLdsStack ? lds_col[..] : spill[..]` hipcc selects between the LDS and the scratch
      // pointer and issues ONE flat_load - every pop then takes a slot of the vector memory addresser, the unit the
      // traversal kernels load most (one slot per lane and load, profiles/r02_microbench_rates.txt)
      uint32_t v;
      const uint32_t at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)lds_col + 256u * (uint32_t)t.sp;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(at) : "memory");
      return v;
   }
   return spill[t.sp - kLdsStack];
}

// one interior node (Node4C, 48 B = three loads): slab-test the 4 children, continue with the nearest, push
// the other hits. plane = origin + scale * q  =>  t = q * (scale * idir) + (origin - o) * idir.
//   w0 = origin.xyz, step exponents (the node's FRAME) ; w1 = qlo.xyz, qhi.x ; w2 = qhi.y, qhi.z, child_base | n_tri << 29, tri_base
// Child references are implicit (bvh.h): slot k is triangle packet tri_base + k below n_tri, node child_base + k - n_tri above.
// The frame of the child the lane continues with is derived here (node_quant.h qn_inherit, restated: v_cvt_f32_ubyte, v_fma_f32,
// v_ffbh_u32) and left in t.fx .. t.fexp: the next visit does not load that node's first quad.
// Instruction diet: the near / far plane words are picked once per axis by the sign of idir instead of min/max per
// plane; an empty slot is an inverted box (no child != empty test); only the nearest child is fully ordered
// (3 comparators); pushes are branch-free (write always, advance the stack pointer by the hit bit).
// Returns false when no child was hit: the caller pops (trav_step pops once for its node lanes and its leaf lanes together).
// the frame of node child `next` (= node0 + slot) of the node (w0, w1, w2), or none when `next` is no node child that was hit
__device__ __forceinline__ void inherit_frame(const uint4 w0, const uint4 w1, const uint4 w2, float sx, float sy, float sz, uint32_t node0, uint32_t next, bool to_node, Trav& t) {
#if UH_INHERIT_FRAME
   const uint32_t sh = ((next - node0) & 3u) << 3;
   const uint32_t lx = (w1.x >> sh) & 0xffu, ly = (w1.y >> sh) & 0xffu, lz = (w1.z >> sh) & 0xffu;
   const uint32_t hx = (w1.w >> sh) & 0xffu, hy = (w2.x >> sh) & 0xffu, hz = (w2.y >> sh) & 0xffu;
   t.fx = fmaf((float)lx, sx, __uint_as_float(w0.x));
   t.fy = fmaf((float)ly, sy, __uint_as_float(w0.y));
   t.fz = fmaf((float)lz, sz, __uint_as_float(w0.z));
   // exponent byte + bit length of the box's width in steps - 8, at least 1 (bit length of 0 is 0: __clz(0) = 32)
   const int ex = (int)(w0.w & 0xffu) + (24 - __clz((int)((hx - lx) & 0xffu))), ey = (int)((w0.w >> 8) & 0xffu) + (24 - __clz((int)((hy - ly) & 0xffu))),
             ez = (int)((w0.w >> 16) & 0xffu) + (24 - __clz((int)((hz - lz) & 0xffu)));
   const uint32_t packed = (uint32_t)(ex < 1 ? 1 : ex) | ((uint32_t)(ey < 1 ? 1 : ey) << 8) | ((uint32_t)(ez < 1 ? 1 : ez) << 16);
   t.fexp = to_node ? packed : 0u;
#endif
}

// CAP (closest-hit order, k_path_fused): children beyond t.tlimit are culled as in a visibility walk - the kernel's shadow rays go
// through the closest-hit walk (occluded <=> the closest hit lies within the limit) beside the other lanes' bounce rays
template <bool ANY, bool CAP = false>
__device__ __forceinline__ bool node_compute(const uint4 w0, const uint4 w1, const uint4 w2, Trav& t, uint32_t* lds_col, uint32_t* spill) {
   const uint32_t meta = w0.w;
   // a power-of-two step is its biased exponent moved to bits 23..30
   const float sx = __uint_as_float((meta & 0xffu) << 23), sy = __uint_as_float((meta << 15) & 0x7f800000u), sz = __uint_as_float((meta << 7) & 0x7f800000u);
   const float ax = sx * t.idir.x, ay = sy * t.idir.y, az = sz * t.idir.z;
   const float bx = (__uint_as_float(w0.x) - t.o.x) * t.idir.x, by = (__uint_as_float(w0.y) - t.o.y) * t.idir.y, bz = (__uint_as_float(w0.z) - t.o.z) * t.idir.z;
   const bool nx = t.idir.x < 0.0f, ny = t.idir.y < 0.0f, nz = t.idir.z < 0.0f;
   // qlo = (w1.x, w1.y, w1.z), qhi = (w1.w, w2.x, w2.y)
   const uint32_t qnx = nx ? w1.w : w1.x, qfx = nx ? w1.x : w1.w;
   const uint32_t qny = ny ? w2.x : w1.y, qfy = ny ? w1.y : w2.x;
   const uint32_t qnz = nz ? w2.y : w1.z, qfz = nz ? w1.z : w2.y;
   const float tcap = (ANY || CAP) ? fminf(t.best.t, t.tlimit) : t.best.t;  // closest: tlimit is +inf
   float tn[4];
   const uint32_t n_tri = w2.z >> kChildBaseBits;
   const uint32_t tri0 = kLeafBit | w2.w, node0 = (w2.z & kChildBaseMask) - n_tri;
   uint32_t cr[4];
#pragma unroll
   for (int k = 0; k < 4; k++) cr[k] = ((uint32_t)k < n_tri ? tri0 : node0) + (uint32_t)k;
   bool hit[4];
#if UH_PK_SLAB
   typedef float f2_t __attribute__((ext_vector_type(2)));
   const f2_t ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az}, bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz};
#endif
#pragma unroll
   for (int k = 0; k < 4; k++) {
#if UH_PK_SLAB
      const f2_t qx = {(float)((qnx >> (8 * k)) & 0xffu), (float)((qfx >> (8 * k)) & 0xffu)}, qy = {(float)((qny >> (8 * k)) & 0xffu), (float)((qfy >> (8 * k)) & 0xffu)},
                 qz = {(float)((qnz >> (8 * k)) & 0xffu), (float)((qfz >> (8 * k)) & 0xffu)};
      const f2_t tx = __builtin_elementwise_fma(qx, ax2, bx2), ty = __builtin_elementwise_fma(qy, ay2, by2), tz = __builtin_elementwise_fma(qz, az2, bz2);
      const float t0x = tx.x, t1x = tx.y, t0y = ty.x, t1y = ty.y, t0z = tz.x, t1z = tz.y;
#else
      const float t0x = fmaf((float)((qnx >> (8 * k)) & 0xffu), ax, bx), t1x = fmaf((float)((qfx >> (8 * k)) & 0xffu), ax, bx);
      const float t0y = fmaf((float)((qny >> (8 * k)) & 0xffu), ay, by), t1y = fmaf((float)((qfy >> (8 * k)) & 0xffu), ay, by);
      const float t0z = fmaf((float)((qnz >> (8 * k)) & 0xffu), az, bz), t1z = fmaf((float)((qfz >> (8 * k)) & 0xffu), az, bz);
#endif
      const float tnear = fmaxf(fmaxf(t0x, t0y), fmaxf(t0z, t.tmin));
      const float tfar = fminf(fminf(t1x, t1y), fminf(t1z, tcap));
      hit[k] = tnear <= tfar;
      tn[k] = hit[k] ? tnear : INFINITY;
   }
   if (ANY) {
      // visibility walk: no ordering network. The order of the children does not matter to an unoccluded ray (it visits
      // them all); an occluded one ends sooner if the likelier occluder comes first. The builders store a node's node
      // children in ascending surface area behind its triangle children (bvh_build.cpp, lbvh.hip), and the walk takes the
      // hit children from the HIGHEST slot down: biggest subtree first, triangles last (tools/any_order_ab.sh: 13.1 ->
      // 10.6 node visits per sun shadow ray). The other hits are pushed, lowest slot first, so that they pop in the same order.
      const bool any = hit[0] || hit[1] || hit[2] || hit[3];
      const uint32_t next = hit[3] ? cr[3] : hit[2] ? cr[2] : hit[1] ? cr[1] : cr[0];
      const bool p2 = hit[2] && hit[3], p1 = hit[1] && (hit[3] || hit[2]), p0 = hit[0] && (hit[3] || hit[2] || hit[1]);
      if (t.sp + 3 <= kLdsStack) {
         uint32_t* p = lds_col + t.sp * 64;
         p[0] = cr[0];
         p += (p0 ? 1 : 0) * 64;
         p[0] = cr[1];
         p += (p1 ? 1 : 0) * 64;
         p[0] = cr[2];
         t.sp += (p0 ? 1 : 0) + (p1 ? 1 : 0) + (p2 ? 1 : 0);
      } else {
         if (p0) trav_push(t, lds_col, spill, cr[0]);
         if (p1) trav_push(t, lds_col, spill, cr[1]);
         if (p2) trav_push(t, lds_col, spill, cr[2]);
      }
      t.cur = next;
      inherit_frame(w0, w1, w2, sx, sy, sz, node0, next, any && !(next & kLeafBit), t);
      return any;
   }
   {
      // closest hit: bring the nearest hit to slot 0 (3 comparators); slots 1..3 stay unordered
      auto cswap = [&](int i, int j) {
         bool s = tn[j] < tn[i];
         float ta = s ? tn[j] : tn[i], tb = s ? tn[i] : tn[j];
         uint32_t ca = s ? cr[j] : cr[i], cb = s ? cr[i] : cr[j];
         tn[i] = ta;
         tn[j] = tb;
         cr[i] = ca;
         cr[j] = cb;
      };
      cswap(0, 1);
      cswap(2, 3);
      cswap(0, 2);
   }
   if (t.sp + 3 <= kLdsStack) {
      // branch-free pushes of slots 3, 2, 1
      uint32_t* p = lds_col + t.sp * 64;
      int h3 = tn[3] < INFINITY ? 1 : 0, h2 = tn[2] < INFINITY ? 1 : 0, h1 = tn[1] < INFINITY ? 1 : 0;
      p[0] = cr[3];
      p += h3 * 64;
      p[0] = cr[2];
      p += h2 * 64;
      p[0] = cr[1];
      t.sp += h3 + h2 + h1;
   } else {
      if (tn[3] < INFINITY) trav_push(t, lds_col, spill, cr[3]);
      if (tn[2] < INFINITY) trav_push(t, lds_col, spill, cr[2]);
      if (tn[1] < INFINITY) trav_push(t, lds_col, spill, cr[1]);
   }
   t.cur = cr[0];
   inherit_frame(w0, w1, w2, sx, sy, sz, node0, cr[0], tn[0] < INFINITY && !(cr[0] & kLeafBit), t);
   return tn[0] < INFINITY;
}

template <bool ANY>
__device__ __forceinline__ void node_step(const uint4* __restrict__ nodes, Trav& t, uint32_t* lds_col, uint32_t* spill) {
   const uint4* n = nodes + kNodeStride16 * (size_t)t.cur;
   const uint4 w0 = n[0], w1 = n[1], w2 = n[2];  // (this walk loads every node whole: the stored frame is the inherited one, bit for bit)
   if (!node_compute<ANY>(w0, w1, w2, t, lds_col, spill)) t.cur = trav_pop(t, lds_col, spill);
}


// batch (if-if) traversal of one ray to its end: the batch kernels (variant 0) and the stand-alone any-hit query
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool traverse(const SceneDev& sc, V3 o, V3 d, float tmin, float tmax, float tlimit, Hit& best, uint32_t* lds_col,
                                         uint32_t& n_nodes, uint32_t& n_tris) {
   Trav t;
   trav_init(t, make_float4(o.x, o.y, o.z, tmin), make_float4(d.x, d.y, d.z, tmax), tmin, tmax, ANY ? tlimit : INFINITY);
   uint32_t spill[kSpillStack];
   const uint4* __restrict__ nodes = sc.nodes;
   const float4* __restrict__ tris = sc.tris;
   bool occluded = false;
   // if-if: in a divergent wave both branches are issued every iteration, so a lane that the node step has just
   // sent to a triangle uses this iteration's triangle branch too instead of idling through it
   while (t.cur != kEmptyRef) {
      if (!(t.cur & kLeafBit)) {
         if (COUNT) n_nodes++;
         node_step<ANY>(nodes, t, lds_col, spill);
      }
      if (t.cur != kEmptyRef && (t.cur & kLeafBit)) {
         if (COUNT) n_tris++;
         if (tri_test<ANY>(tris, t.cur & ~kLeafBit, t.o, t.d, t.tmin, t.tlimit, t.best) && ANY) {
            occluded = true;
            break;
         }
         t.cur = trav_pop(t, lds_col, spill);
      }
   }
   best = t.best;
   return ANY ? occluded : (best.idx != kEmptyRef);
}

// persistent-thread batch fetch: lane 0 pulls the next 64-item batch of its shard
__device__ __forceinline__ uint32_t next_batch(uint32_t* cursor) {
   uint32_t base = 0;
   if (lane_id() == 0) base = atomicAdd(cursor, 64u);
   return __builtin_amdgcn_readfirstlane(base);
}

// blocks are bound to queue shards by blockIdx % kShards (device_types.h)
struct ShardCtx {
   uint32_t shard, lb, nb;  // shard id, this block's index within the shard, blocks per shard
};
__device__ __forceinline__ ShardCtx shard_ctx() {
   ShardCtx c;
   c.shard = blockIdx.x % kShards;
   c.lb = blockIdx.x / kShards;
   c.nb = gridDim.x / kShards;
   return c;
}

// ------------------------------------------------------------------------------------------
// Ray replacement ("refill") from a per-wave LDS ray pool.
//
// Thread-per-ray traversal leaves a lane idle from the moment its ray ends until the slowest ray of
// the wave ends (bounce rays: mean 18 steps against a wave maximum of 43, profiles/README.md). Here a
// wave is persistent and every lane whose ray has ended takes the next ray out of a 64-entry pool in
// LDS - a couple of ds_read_b128, no global round trip on the wave's critical path. The pool is kept
// fed by a three-stage pipeline, one stage per "refill event" (= the pool ran empty):
//    stage 1  lane 0 reserves the next 64-ray chunk of the shard's queue (one returning atomic),
//    stage 2  the 64 path ids of the chunk reserved one event earlier are loaded,
//    stage 3  the rays of the ids loaded one event earlier go global -> LDS by LDS-DMA
//             (global_load_lds_dwordx4: per-lane source address, 64 x 16 B contiguous in LDS).
// Every stage consumes what was issued a whole pool (about 20 wave iterations) earlier, so its wait
// finds the data there. The traversal kernels write nothing but their per-path result: no queue is
// built here (the shading kernels classify hits and misses themselves), so the loop has no atomic
// whose value it needs at once.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kPool = 64;

template <int NA>
struct alignas(16) RayPool {
   float4 v[NA][kPool];  // one LDS-DMA instruction fills one of these arrays
   uint32_t id[kPool];
};

__device__ __forceinline__ void dma16(const float4* gsrc, float4* lds_dst) {
   // lds_dst is wave-uniform; lane l's 16 bytes land at lds_dst + l. aux = 2: the nt cache policy
   if (kStreamNt)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 2);
   else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Where a wave's rays come from: a (shard segment of a) queue of path ids, or the identity (ray i = record i)
// when `queue` is null; chunks are handed out by an atomic cursor, or - `cursor` null - statically
// (chunk k of wave w = (k * num_waves + w) * 64).
struct RaySource {
   const uint32_t* queue;
   uint32_t count;
   uint32_t* cursor;
   uint32_t wave_index, num_waves;  // static chunk assignment only
};

template <int NA>
struct Feeder {
   // wave-uniform state (ballot / readfirstlane derived: lives in SGPRs)
   uint32_t pos = 0, n = 0;     // pool entries [pos, n) are unread
   uint32_t load_n = 0;         // entries the LDS-DMA in flight delivers
   uint32_t q_n = 0;            // valid lanes of q_id
   uint32_t static_k = 0;
   uint32_t iterations = 0;
   uint32_t q_base = 0, load_base = 0, pool_base = 0;  // first queue position of the chunk in q_id / in flight / in the pool
   bool loading = false, have_base = false, drained = false;
   // per-lane pipeline registers
   uint32_t r_base = 0;         // lane 0: what the newest cursor atomic returned
   uint32_t q_id = 0;           // path id of lane l's ray in the chunk that enters the pool next

   __device__ __forceinline__ bool empty() const { return pos >= n && !loading && q_n == 0 && !have_base && drained; }

   // the DMA issued one event ago has landed (its wait also covers every older load of the wave)
   __device__ __forceinline__ void land() {
      if (loading) {
         wait_vm0();
         pos = 0;
         n = load_n;
         pool_base = load_base;
         loading = false;
      }
   }

   // one refill event: called when the pool is empty and nothing is in flight
   template <typename SrcFn>
   __device__ __forceinline__ void advance(const RaySource& src, RayPool<NA>& pool, SrcFn&& source_of) {
      const uint32_t lane = lane_id();
      if (q_n) {  // stage 3
         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the pool's last entries have been read
         if (lane < q_n) {
#pragma unroll
            for (int a = 0; a < NA; a++) dma16(source_of(a, q_id), pool.v[a]);
            pool.id[lane] = q_id;
         }
         load_n = q_n;
         load_base = q_base;
         loading = true;
         q_n = 0;
      }
      if (have_base) {  // stage 2
         const uint32_t b = __builtin_amdgcn_readfirstlane(r_base);
         have_base = false;
         if (b < src.count) {
            q_base = b;
            q_n = src.count - b < kPool ? src.count - b : kPool;
            if (lane < q_n) q_id = src.queue ? ld_stream(src.queue + b + lane) : b + lane;
         } else {
            drained = true;
         }
      }
      if (!drained) {  // stage 1
         if (src.cursor) {
            if (lane == 0) r_base = atomicAdd(src.cursor, kPool);
         } else {
            r_base = (static_k * src.num_waves + src.wave_index) * kPool;
            static_k++;
         }
         have_base = true;
      }
   }
};

// what a traversal wave does per iteration for its idle lanes; returns false when the wave is out of work.
// kRefill: idle lanes that make a refill worth its instructions (every lane that ends costs a pool read + trav_init);
// 4 / 8 / 16 measured 3.10 / 3.10 / 3.15 ms per frame (profiles/README.md)
constexpr int kRefill = 8;
template <int NA, typename SrcFn, typename TakeFn>
__device__ __forceinline__ bool refill_lanes(Feeder<NA>& f, const RaySource& src, RayPool<NA>& pool, bool lane_idle, SrcFn&& source_of, TakeFn&& take) {
   // exit condition every wave reaches whatever the data: a traversal step visits a node or a triangle once, so a
   // wave that has run this many iterations is not walking a tree any more (corrupt references) - leave
   if (++f.iterations > (1u << 24)) return false;
   const unsigned long long idle = __ballot(lane_idle);
   if (idle == 0ull) return true;
   const uint32_t n_idle = (uint32_t)__popcll(idle);
   f.land();
   const uint32_t avail = f.n - f.pos;
   if (avail != 0 && (n_idle >= (uint32_t)kRefill || n_idle == 64u)) {
      const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
      if (lane_idle && prefix < avail) take(f.pos + prefix);
      f.pos += n_idle < avail ? n_idle : avail;
   }
   if (f.pos >= f.n && !f.loading) {
      if (f.empty()) return n_idle != 64u || avail != 0;  // nothing left to hand out: done once every lane is idle
      f.advance(src, pool, source_of);
   }
   return true;
}

// one step of a lane's traversal; returns true when the ray has ended. ONE load phase per iteration: a lane at a node loads
// its node, a lane at a leaf its triangle, then both groups compute. A lane that reaches a leaf tests it an iteration later,
// but the wave waits for memory once per iteration (a chained node -> triangle step, two dependent round trips per
// iteration, measured 3.21 against 3.10 ms per frame: profiles/README.md).
template <bool ANY, bool COUNT, bool CAP = false>
__device__ __forceinline__ bool trav_step(const uint4* __restrict__ nodes, const float4* __restrict__ tris, Trav& t, uint32_t* lds_col, uint32_t* spill, bool& occluded,
                                          uint32_t& n_nodes, uint32_t& n_tris) {
   const bool at_node = !(t.cur & kLeafBit);
   const uint32_t packet = t.cur & ~kLeafBit;
#if UH_GATE_LEAVES
   {  // the triangle branch (~70 instructions for the handful of lanes that stand at a leaf) only when enough lanes want it
      const unsigned long long leaves = __ballot(!at_node), at_nodes = __ballot(at_node);
      const bool run = (uint32_t)__popcll(leaves) >= (uint32_t)UH_GATE_LEAVES || at_nodes == 0ull || __ballot(!at_node && t.wait >= (uint32_t)UH_GATE_WAIT) != 0ull;
      if (!at_node) {
         if (!run) {
            t.wait++;
            return false;
         }
         t.wait = 0;
      }
   }
#endif
   // nodes and triangle packets are both three-quad records: ONE address and ONE set of loads for the whole wave.
   // (Written as two branches, each with its own loads, the compiler gave the second branch's address the first
   // branch's destination registers and made it wait for them: the two groups' loads ran one after the other.)
   const uint4* rec = at_node ? nodes + kNodeStride16 * (size_t)packet : (const uint4*)tris + kTriStride16 * (size_t)packet;
   // a lane that descended into this node brought the node's frame along (node_compute): it loads two quads, not three
#if UH_INHERIT_FRAME
   uint4 w0 = make_uint4(__float_as_uint(t.fx), __float_as_uint(t.fy), __float_as_uint(t.fz), t.fexp);
   if (!(at_node && t.fexp != 0u)) w0 = rec[0];
   uint4 w1 = rec[1], w2 = rec[2];
#else
   uint4 w0 = rec[0], w1 = rec[1], w2 = rec[2];
#endif
   // the packet's last two dwords are padding: without this the compiler loads them in the node branch only (a fourth load)
   asm volatile("" : "+v"(w2.z), "+v"(w2.w));
   bool pop;  // one pop for both groups: an LDS read and its wait once per iteration, not once per branch
   if (at_node) {
      if (COUNT) n_nodes++;
      pop = !node_compute<ANY, CAP>(w0, w1, w2, t, lds_col, spill);
   } else {
      if (COUNT) n_tris++;
      const float4 ta = make_float4(__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w));
      const float4 tb = make_float4(__uint_as_float(w1.x), __uint_as_float(w1.y), __uint_as_float(w1.z), __uint_as_float(w1.w));
      const float4 tc = make_float4(__uint_as_float(w2.x), __uint_as_float(w2.y), __uint_as_float(w2.z), __uint_as_float(w2.w));
      pop = true;
      if (tri_compute<ANY>(ta, tb, tc, packet, t.o, t.d, t.tmin, t.tlimit, t.best) && ANY) {
         occluded = true;
         t.cur = kEmptyRef;
         pop = false;
      }
   }
   if (pop) {
      t.cur = trav_pop(t, lds_col, spill);
#if UH_INHERIT_FRAME
      t.fexp = 0u;  // a popped node's frame is in its record
#endif
   }
   return t.cur == kEmptyRef;
}

// ------------------------------------------------------------------------------------------
// trace_closest — reference.rgen:47 traceRayEXT(..., payload 0) minus the shaders it invokes.
// Reads the bounce's ray queue, writes hit[queue position] = (t, u, v, packet) or packet = kEmptyRef: the shading kernel walks
// the same queue and reads the records back as one contiguous stream (a record per path id was a 64-byte sector per 16-byte
// record there).
// ------------------------------------------------------------------------------------------
// Path rays (sharded): ray_o / ray_d are the origin / direction planes of the bounce's state set, hit_out the hit plane - all by
// queue position, so the rays of a 64-entry chunk are two contiguous kilobytes; their w components carry RNG words: the range is
// rgen:45-47's constants. Raw rays (!sharded, RawRays): record i = ray i, the range is in the w components.
// listed != null (sharded only): the rays are the queue positions listed there (count: Control q_count of kind Q_CAM_TREE) - the
// primary rays k_trace_camera_grid handed over -, already counted.
template <bool COUNT>
__global__ __launch_bounds__(kBlock, 6) void k_trace_closest(SceneDev sc, bool sharded, const float4* __restrict__ ray_o,
                                                               const float4* __restrict__ ray_d, float4* __restrict__ hit_out, uint32_t shard_cap, Control* ctl,
                                                               DeviceStats* stats, uint32_t bounce, uint32_t cursor_slot, int ray_kind, uint32_t raw_count,
                                                               const uint32_t* __restrict__ listed) {
   const bool range_in_w = !sharded;
   __shared__ uint32_t s_stack[kWavesPerBlock][kLdsStack][64];
   __shared__ RayPool<2> s_pool[kWavesPerBlock];
   const uint32_t lane = lane_id();
   const uint32_t wave = threadIdx.x >> 6;
   uint32_t* lds_col = &s_stack[wave][0][lane];
   RayPool<2>& pool = s_pool[wave];
   RaySource src;
   uint32_t seg = 0;  // first record of this block's rays and results: its shard's segment (path rays) or 0 (raw rays)
   if (sharded) {  // path tracer: this block's shard of the bounce's ray queue, chunks from the shard's cursor; ray = record at its position
      const ShardCtx sx = shard_ctx();
      seg = sx.shard * shard_cap;
      src.queue = listed ? listed + seg : nullptr;
      src.count = ctl->q_count[qc_index(bounce, listed ? Q_CAM_TREE : Q_RAY, sx.shard)];
      src.cursor = &ctl->cursor[cursor_index(cursor_slot, sx.shard)];
      src.wave_index = src.num_waves = 0;
      if (sx.lb == 0 && threadIdx.x == 0) {
         if (!listed) atomicAdd(&stats->rays[ray_kind], (unsigned long long)src.count);
         else if (src.count) atomicAdd(&stats->cam_tree_rays, (unsigned long long)src.count);  // counted as primary rays by the grid kernel already
      }
   } else {
      // stand-alone query over raw_count rays (uh_trace_closest, the G-buffer cast): ray i = record i
      src.queue = nullptr;
      src.count = raw_count;
      src.cursor = nullptr;
      src.wave_index = blockIdx.x * kWavesPerBlock + wave;
      src.num_waves = gridDim.x * kWavesPerBlock;
   }
   const uint4* __restrict__ nodes = sc.nodes;
   const float4* __restrict__ tris = sc.tris;
   auto source_of = [&](int a, uint32_t pos) { return (a == 0 ? ray_o : ray_d) + seg + pos; };
   Feeder<2> f;
   Trav t;
   t.cur = kEmptyRef;
   t.sp = 0;
   uint32_t where = 0, n_nodes = 0, n_tris = 0;
   uint32_t spill[kSpillStack];
   auto take = [&](uint32_t slot) {
      where = seg + (listed ? pool.id[slot] : f.pool_base + slot);  // the ray's position in the queue (the pool holds one chunk: positions pool_base ..)
      const float4 ro = pool.v[0][slot], rd = pool.v[1][slot];
      trav_init(t, ro, rd, range_in_w ? ro.w : 0.001f, range_in_w ? rd.w : 10000.0f, INFINITY);
   };
   while (refill_lanes<2>(f, src, pool, t.cur == kEmptyRef, source_of, take)) {
      if (t.cur != kEmptyRef) {
         bool occluded = false;
         if (trav_step<false, COUNT>(nodes, tris, t, lds_col, spill, occluded, n_nodes, n_tris))
            st_rec(hit_out + where, make_float4(t.best.t, t.best.u, t.best.v, __uint_as_float(t.best.idx)));
      }
   }
   if (COUNT) {
      atomicAdd(&stats->nodes_visited, (unsigned long long)n_nodes);
      atomicAdd(&stats->tris_tested, (unsigned long long)n_tris);
   }
}

// ------------------------------------------------------------------------------------------
// trace_shadow — reference.rgen:67 (sun) and :115 (light): visibility only. The reference runs
// its closest-hit / miss shaders on these rays too, but the raygen reads nothing except
// colorDistance.w (rgen:69,118-119), so the sky integral and material fetch are dead work here.
// Occluded <=> some triangle has tmin < t < tmax (sun) and additionally t <= distance_to_light.
// An unoccluded ray adds its path's throughput (x light weight) to the path's radiance (rgen:69-78 / :118-122).
// ------------------------------------------------------------------------------------------
struct ShadowRay {
   float4 ro, rd, lit;  // origin / direction (tmin, tmax in w) and the radiance record the path gets if the ray is unoccluded
   float tlimit;
};
template <bool LIGHT>
__device__ __forceinline__ ShadowRay make_shadow_ray(const SceneDev& sc, const FrameParams& fp, float4 ro, float4 thr, float4 rad) {
   ShadowRay s;
   s.ro = make_float4(ro.x, ro.y, ro.z, 0.001f);
   s.tlimit = INFINITY;
   if (LIGHT) {
      const int light_index = (int)__float_as_uint(rad.w);
      V3 lpos = v3(0, 0, 0);
      if (light_index >= 0 && (uint32_t)light_index < sc.num_lights) lpos = xyz(sc.lights[2 * light_index]);
      const V3 o = v3(ro.x, ro.y, ro.z);
      const V3 dir = normalize3(lpos - o);  // rgen:113
      s.tlimit = length3(lpos - o);         // rgen:114
      const float f = thr.w;
      s.lit = make_float4(rad.x + thr.x * f, rad.y + thr.y * f, rad.z + thr.z * f, rad.w);
      s.rd = make_float4(dir.x, dir.y, dir.z, 10000.0f);
   } else {
      s.lit = make_float4(rad.x + thr.x, rad.y + thr.y, rad.z + thr.z, rad.w);
      s.rd = make_float4(fp.sun_dir[0], fp.sun_dir[1], fp.sun_dir[2], 10000.0f);  // rgen:64
   }
   return s;
}

// leftovers (sun rays only): the rays k_trace_sun_grid could not serve from its grid (queue 3); they are already counted
template <bool COUNT, bool LIGHT>
__global__ __launch_bounds__(kBlock, 5) void k_trace_shadow(SceneDev sc, FrameParams fp, PathState ps, Control* ctl, DeviceStats* stats, uint32_t bounce,
                                                              uint32_t cursor_slot, bool leftovers) {
   __shared__ uint32_t s_stack[kWavesPerBlock][kLdsStack][64];
   __shared__ RayPool<3> s_pool[kWavesPerBlock];
   const uint32_t lane = lane_id();
   const uint32_t wave = threadIdx.x >> 6;
   uint32_t* lds_col = &s_stack[wave][0][lane];
   RayPool<3>& pool = s_pool[wave];
   const ShardCtx sx = shard_ctx();
   const uint32_t seg = sx.shard * ps.shard_cap;
   RaySource src;
   // sun rays leave from every scattered path = every position of the next bounce's ray queue (identity); light rays and the sun
   // rays the grid handed over from the positions listed in their queues
   src.queue = LIGHT ? ps.queue[2] + seg : leftovers ? ps.queue[3] + seg : nullptr;
   src.count = LIGHT ? ctl->q_count[qc_index(bounce, Q_LIGHT, sx.shard)] : leftovers ? ctl->q_count[qc_index(bounce, Q_SUN_TREE, sx.shard)] : ctl->q_count[qc_index(bounce + 1, Q_RAY, sx.shard)];
   src.cursor = &ctl->cursor[cursor_index(cursor_slot, sx.shard)];
   src.wave_index = src.num_waves = 0;
   if (sx.lb == 0 && threadIdx.x == 0) {
      if (!leftovers) atomicAdd(&stats->rays[LIGHT ? UH_RAY_LIGHT_SHADOW : UH_RAY_SUN_SHADOW], (unsigned long long)src.count);
      else if (src.count) atomicAdd(&stats->sun_tree_rays, (unsigned long long)src.count);  // counted as sun rays by the grid kernel already
   }
   const uint4* __restrict__ nodes = sc.nodes;
   const float4* __restrict__ tris = sc.tris;
   const PathRecs rec = ps.set[(bounce + 1) & 1];  // the state of the scattered paths, at their positions in the next bounce's queue
   float4* rad = rec_quad(rec, seg, REC_RAD);
   auto source_of = [&](int a, uint32_t pos) { return (const float4*)rec_quad(rec, seg + pos, a == 0 ? REC_ORIGIN : a == 1 ? REC_THR : REC_RAD); };
   Feeder<3> f;
   Trav t;
   t.cur = kEmptyRef;
   t.sp = 0;
   uint32_t id = 0, n_nodes = 0, n_tris = 0;  // id: the path's position in the next bounce's queue
   float4 lit = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
   uint32_t spill[kSpillStack];
   auto take = [&](uint32_t slot) {
      id = pool.id[slot];
      const ShadowRay s = make_shadow_ray<LIGHT>(sc, fp, pool.v[0][slot], pool.v[1][slot], pool.v[2][slot]);
      lit = s.lit;
      trav_init(t, s.ro, s.rd, s.ro.w, s.rd.w, s.tlimit);
   };
   while (refill_lanes<3>(f, src, pool, t.cur == kEmptyRef, source_of, take)) {
      if (t.cur != kEmptyRef) {
         bool occluded = false;
         if (trav_step<true, COUNT>(nodes, tris, t, lds_col, spill, occluded, n_nodes, n_tris) && !occluded) st_stream(rad + id, lit);
      }
   }
   if (COUNT) {
      atomicAdd(LIGHT ? &stats->light_nodes_visited : &stats->shadow_nodes_visited, (unsigned long long)n_nodes);
      atomicAdd(LIGHT ? &stats->light_tris_tested : &stats->shadow_tris_tested, (unsigned long long)n_tris);
   }
}

// ------------------------------------------------------------------------------------------
// trace_sun_grid - the sun shadow rays of reference.rgen:63-79 through the per-direction grid of sun_grid.h instead of the
// tree: one cell look-up, then the cell's packets front (sun side) to back through the same tri_compute<ANY> until one
// occludes the ray or the list falls behind the ray's origin. Same predicate as k_trace_shadow<.., false>: occluded <=> some
// triangle accepts the ray with 0.001 < t < 10000 - bit for bit, the grid only prunes (conservatively) which triangles are asked.
// Batch form: rays do a handful of steps, and their lengths (0..a few tests) differ little inside a wave.
// ------------------------------------------------------------------------------------------
// The lists come in two forms (SunGridDev). Plain: 8-byte entries (packet index, far depth) beside the packet array - two dependent
// reads per test. INLINE: 64-byte records that CARRY their packet (v0 e1 e2 key | this entry's far depth | the next entry's) - one
// sector and one round trip per test; sixty-four bytes per entry where the plain list has eight, so by default only while the records
// stay within four times the packet array (context.hip attach_sun_inline_records, option "sun_grid_inline_max_mb").
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop the compiler cannot leave rolled (arrays indexed by its
// counter stay in registers whatever the body holds)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
   if constexpr (I < N) {
      f(std::integral_constant<int, I>{});
      static_for<N, I + 1>(f);
   }
}

// The ray's cell: (pu, pv) = its origin in the grid's frame; outside the grid (or NaN) it is a border cell.
__device__ __forceinline__ void sun_cell_of(const SunGridDev& g, float pu, float pv, uint32_t& cx, uint32_t& cy) {
   float fx = (pu - g.u0) * g.inv_cell, fy = (pv - g.v0) * g.inv_cell;
   fx = !(fx >= 0.0f) ? 0.0f : fx;
   fy = !(fy >= 0.0f) ? 0.0f : fy;
   const float max_x = (float)(g.nx - 1), max_y = (float)(g.ny - 1);
   fx = fx > max_x ? max_x : fx;
   fy = fy > max_y ? max_y : fy;
   cx = (uint32_t)fx;
   cy = (uint32_t)fy;
}
// The coarse cover (SunGridDev::coarse, sun_grid.h): one depth per block of cells, below every cell's own cover depth - a ray that
// starts below it is below its own cell's cover too. 0.7 MB on the config-1 scene: it stays in an XCD's L2, where the cell
// records (23 MB) are a request to the memory side per ray.
__device__ __forceinline__ bool sun_coarse_covered(float coarse, float pw) { return pw < coarse && coarse - pw < kSunCoarseReach; }

// Every position of the next bounce's ray queue is a ray; the coarse cover - when the grid has one - is asked first. (Round 4 also had
// k_shade_hit ask the coarse cover and list the rays it did not answer - level on the frame - and a one-ray-per-lane loop for the plain
// lists: both removed in round 5.)
template <bool COUNT, bool INLINE>
__global__ __launch_bounds__(kBlock) void k_trace_sun_grid(SceneDev sc, FrameParams fp, PathState ps, Control* ctl, DeviceStats* stats, uint32_t bounce,
                                                            uint32_t cursor_slot, SunGridDev g) {
   const uint32_t lane = lane_id();
   const ShardCtx sx = shard_ctx();
   const uint32_t seg = sx.shard * ps.shard_cap;
   const PathRecs rec = ps.set[(bounce + 1) & 1];  // a sun ray leaves every scattered path: every position of the next bounce's ray queue
   const uint32_t count = ctl->q_count[qc_index(bounce + 1, Q_RAY, sx.shard)];
   uint32_t* cursor = &ctl->cursor[cursor_index(cursor_slot, sx.shard)];
   const V3 d = v3(fp.sun_dir[0], fp.sun_dir[1], fp.sun_dir[2]);  // rgen:64
   uint32_t n_tris = 0, n_covered = 0;
   uint32_t* q_tree = ps.queue[3] + seg;
   uint32_t* n_tree = &ctl->q_count[qc_index(bounce, Q_SUN_TREE, sx.shard)];
   // K rays per lane, each step of the chain (origin -> coarse cover -> cell record -> first list entry [-> its packet] -> throughput /
   // radiance) requested for all K before the first is used: the kernel is a chain of dependent round trips with a few instructions
   // between them, and a wave that keeps K chains in flight hides K times the latency (registers are no constraint: 40 at K = 1)
   constexpr int K = UH_SUN_RAYS_PER_LANE;
   typedef float f4_t __attribute__((ext_vector_type(4)));  // (arrays of HIP's float4 struct assigned under a condition end up in scratch)
   const f4_t* __restrict__ recs = reinterpret_cast<const f4_t*>(g.recs);       // INLINE
   const f4_t* __restrict__ tris = reinterpret_cast<const f4_t*>(sc.tris);      // plain lists: the packets
   const uint2* __restrict__ entries = reinterpret_cast<const uint2*>(g.entries);
   const uint4* __restrict__ cells = reinterpret_cast<const uint4*>(g.cell_start);  // offset | cover depth | the first entry: packet, far depth
   for (;;) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(cursor, 64u * K);
      base = __builtin_amdgcn_readfirstlane(base);
      if (base >= count) break;
      uint32_t id[K];  // the path's position in the next bounce's queue: its state lies there (a wave reads and writes contiguous kilobytes)
      bool valid[K];
      static_for<K>([&](auto kc) {
         constexpr int k = decltype(kc)::value;
         id[k] = base + 64u * k + lane;
         valid[k] = id[k] < count;
      });
      float4 ro[K];
      static_for<K>([&](auto kc) {
         constexpr int k = decltype(kc)::value;
         ro[k] = valid[k] ? ld_rec(rec_quad(rec, seg + id[k], REC_ORIGIN)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      });
      float pw[K];
      uint32_t cx[K], cy[K];
      bool covered[K], ask[K];
      static_for<K>([&](auto kc) {
         constexpr int k = decltype(kc)::value;
         const V3 o = v3(ro[k].x, ro[k].y, ro[k].z);
         const float pu = dot_fma(v3(g.U[0], g.U[1], g.U[2]), o), pv = dot_fma(v3(g.V[0], g.V[1], g.V[2]), o);
         pw[k] = dot_fma(v3(g.W[0], g.W[1], g.W[2]), o);
         sun_cell_of(g, pu, pv, cx[k], cy[k]);
         covered[k] = false;
      });
      if (g.coarse) {
         float cw[K];
         static_for<K>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            cw[k] = valid[k] ? g.coarse[(cy[k] >> g.coarse_shift) * g.coarse_nx + (cx[k] >> g.coarse_shift)] : -INFINITY;
         });
         static_for<K>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            covered[k] = valid[k] && sun_coarse_covered(cw[k], pw[k]);
         });
      }
      uint4 cs[K];  // offset into the entries | cover depth | first entry: packet | far depth
      uint32_t end[K];
      static_for<K>([&](auto kc) {
         constexpr int k = decltype(kc)::value;
         ask[k] = valid[k] && !covered[k];
         cs[k] = make_uint4(0u, 0u, 0u, 0u);
         end[k] = 0u;
         if (ask[k]) {
            const uint32_t cell = cy[k] * g.nx + cx[k];
            cs[k] = cells[cell];
            end[k] = cells[cell + 1].x;
         }
      });
      bool defer[K], walk[K], lit[K];
      f4_t ra[K], rb[K], rc[K];  // the first packet; INLINE: rc.z / rc.w = this entry's far depth / the next one's
      uint2 en[K], nxt[K];       // plain lists: this entry (packet, far depth) and the next
      static_for<K>([&](auto kc) {
         constexpr int k = decltype(kc)::value;
         // the cell's cover: some packet spans the whole cell (with the margins to spare) and every ray that starts below this depth has
         // it in front, further than tmin away and nearer than tmax (sun_grid.h kSunCoverReach) - occluded, as the tree walk would say
         const float cover = __uint_as_float(cs[k].y);
         const bool cov = ask[k] && pw[k] < cover && cover - pw[k] < kSunCoverReach;
         covered[k] = covered[k] || cov;
         // a border cell stands for everything beyond the dense part of the scene, and some interior cells list a great many packets
         // (walls edge-on to the sun): such a ray is cheaper in the tree - k_trace_shadow takes it from queue 3
         defer[k] = ask[k] && !cov && (cx[k] == 0 || cy[k] == 0 || cx[k] + 1 == g.nx || cy[k] + 1 == g.ny || end[k] - cs[k].x > g.max_walk);
         walk[k] = ask[k] && !cov && !defer[k] && cs[k].x < end[k];
         lit[k] = ask[k] && !cov && !defer[k] && !walk[k];  // nothing projects into the cell
         if (walk[k]) {
            if constexpr (INLINE) {
               const f4_t* r = recs + 4 * (size_t)cs[k].x;
               ra[k] = r[0];
               rb[k] = r[1];
               rc[k] = r[2];
            } else {
               // the list's first entry came with the cell record: its packet is asked for at once, the second entry beside it
               en[k] = make_uint2(cs[k].z, cs[k].w);
               // sorted by far depth, descending: a list whose first entry ends behind the origin has nothing in front of the ray
               if (__uint_as_float(en[k].y) < pw[k]) {
                  walk[k] = false;
                  lit[k] = true;
               } else {
                  const f4_t* r = tris + kTriStride16 * (size_t)en[k].x;
                  ra[k] = r[0];
                  rb[k] = r[1];
                  rc[k] = r[2];
                  nxt[k] = cs[k].x + 1 < end[k] ? entries[cs[k].x + 1] : make_uint2(0u, 0u);
               }
            }
         }
         if (COUNT) n_covered += covered[k] ? 1u : 0u;
      });
      static_for<K>([&](auto kc) {
         constexpr int k = decltype(kc)::value;
         if (walk[k]) {
            const V3 o = v3(ro[k].x, ro[k].y, ro[k].z);
            Hit best;
            best.t = 10000.0f;  // tmax (rgen:66)
            best.u = best.v = 0.0f;
            best.idx = kEmptyRef;
            best.key = 0xffffffffu;
            uint32_t e = cs[k].x;
            float4 a = make_float4(ra[k].x, ra[k].y, ra[k].z, ra[k].w), b = make_float4(rb[k].x, rb[k].y, rb[k].z, rb[k].w), c = make_float4(rc[k].x, rc[k].y, rc[k].z, rc[k].w);
            bool occluded = false;
            if constexpr (INLINE) {
               const f4_t* r = recs + 4 * (size_t)e;
               for (;;) {
                  if (c.z < pw[k]) break;  // from here on every packet ends behind the origin (t < 0 for all of them)
                  if (COUNT) n_tris++;
                  if (tri_compute<true>(a, b, c, 0u, o, d, 0.001f, INFINITY, best)) {
                     occluded = true;
                     break;
                  }
                  e++;
                  if (e >= end[k] || c.w < pw[k]) break;  // the next record is only asked for when the ray has to go on
                  r += 4;
                  const f4_t na = r[0], nb = r[1], nc = r[2];
                  a = make_float4(na.x, na.y, na.z, na.w);
                  b = make_float4(nb.x, nb.y, nb.z, nb.w);
                  c = make_float4(nc.x, nc.y, nc.z, nc.w);
               }
            } else {
               uint2 cur = en[k], nx = nxt[k];
               for (;;) {
                  if (COUNT) n_tris++;
                  if (tri_compute<true>(a, b, c, cur.x, o, d, 0.001f, INFINITY, best)) {
                     occluded = true;
                     break;
                  }
                  e++;
                  if (e >= end[k] || __uint_as_float(nx.y) < pw[k]) break;
                  cur = nx;
                  if (e + 1 < end[k]) nx = entries[e + 1];  // in flight with the packet
                  const f4_t* r = tris + kTriStride16 * (size_t)cur.x;
                  const f4_t na = r[0], nb = r[1], nc = r[2];
                  a = make_float4(na.x, na.y, na.z, na.w);
                  b = make_float4(nb.x, nb.y, nb.z, nb.w);
                  c = make_float4(nc.x, nc.y, nc.z, nc.w);
               }
            }
            lit[k] = !occluded;
         }
      });
      float4 thr[K], rad[K];
      static_for<K>([&](auto kc) {
         constexpr int k = decltype(kc)::value;
         if (lit[k]) {  // rgen:69-78: radiance += throughput
            thr[k] = ld_rec(rec_quad(rec, seg + id[k], REC_THR));
            rad[k] = ld_rec(rec_quad(rec, seg + id[k], REC_RAD));
         }
      });
      static_for<K>([&](auto kc) {
         constexpr int k = decltype(kc)::value;
         if (lit[k]) st_rec(rec_quad(rec, seg + id[k], REC_RAD), make_float4(rad[k].x + thr[k].x, rad[k].y + thr[k].y, rad[k].z + thr[k].z, rad[k].w));
      });
      static_for<K>([&](auto kc) {
         constexpr int k = decltype(kc)::value;
         if (__ballot(defer[k]) != 0ull) {  // wave-uniform: every lane of the wave takes part in the append
            const uint32_t slot = wave_append(n_tree, defer[k]);
            if (defer[k]) st_stream(q_tree + slot, id[k]);
         }
      });
   }
   if (sx.lb == 0 && threadIdx.x == 0) {
      atomicAdd(&stats->rays[UH_RAY_SUN_SHADOW], (unsigned long long)count);
      if (COUNT) atomicAdd(&stats->shadow_nodes_visited, (unsigned long long)count);  // every sun ray looked one cell (or its block's coarse cover) up
   }
   if (COUNT) {
      atomicAdd(&stats->shadow_tris_tested, (unsigned long long)n_tris);
      atomicAdd(&stats->sun_covered_rays, (unsigned long long)n_covered);
   }
}

// ------------------------------------------------------------------------------------------
// trace_camera_grid - the primary rays of reference.rgen:31-47 (bounce 0 of the path tracer) through the per-camera grid of
// sun_grid.h "camera grid" instead of the tree: the ray's cell is its own pixel, the cell's packets are tested front to back
// with the closest-hit test of the tree walk (tri_compute<false>: the same t, u, v and the same tie-break by key) until the next
// packet's distance bound exceeds the best hit. Same hit record as k_trace_closest, bit for bit: the grid only prunes
// (conservatively) which triangles are asked. A pixel whose list is longer than max_walk hands its ray to the tree walk
// (queue 3, Q_CAM_TREE: k_trace_closest over the listed positions, right after this kernel).
// The 64 rays of a wave are 64 neighbouring pixels: cell records, entries and packets are shared or adjacent - this kernel runs
// out of the caches where the tree walk of the same rays fetched ~20 records per ray.
// ------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace_camera_grid(SceneDev sc, FrameParams fp, PathState ps, Control* ctl, DeviceStats* stats, uint32_t cursor_slot, SunGridDev g) {
   const uint32_t lane = lane_id();
   const ShardCtx sx = shard_ctx();
   const uint32_t seg = sx.shard * ps.shard_cap;
   const uint32_t* __restrict__ queue = ps.queue[0] + seg;
   const uint32_t count = ctl->q_count[qc_index(0, Q_RAY, sx.shard)];
   uint32_t* cursor = &ctl->cursor[cursor_index(cursor_slot, sx.shard)];
   const PathRecs rec = ps.set[0];
   const float4* __restrict__ tris = sc.tris;
   uint32_t* q_tree = ps.queue[3] + seg;
   uint32_t* n_tree = &ctl->q_count[qc_index(0, Q_CAM_TREE, sx.shard)];
   uint32_t n_tris = 0;
   for (;;) {
      const uint32_t base = next_batch(cursor);
      if (base >= count) break;
      const uint32_t i = base + lane;
      const bool valid = i < count;
      bool defer = false;
      if (valid) {
         const uint32_t id = ld_stream(queue + i);
         const uint32_t k = id % fp.n_owned;
         const uint32_t pix = fp.owned_pixels ? fp.owned_pixels[k] : k;
         const uint32_t px = pix % fp.W, py = pix / fp.W;
         const uint32_t cell = (py + 1) * g.nx + (px + 1);
         uint32_t e = g.cell_start[cell];  // the camera grid's cell records are plain offsets (no cover depths)
         const uint32_t end = g.cell_start[cell + 1];
         const bool sorted = end - e <= g.max_walk;  // the builder sorts the lists a ray may walk with the early exit
         defer = end - e > (g.walk_whole > g.max_walk ? g.walk_whole : g.max_walk);
         if (defer && fp.primary_implicit) {  // the tree walk reads its rays from the planes: this one's is written after all
            float4 ro, rd;
            primary_state(fp, id, ro, rd);
            st_rec(rec_quad(rec, seg + i, REC_ORIGIN), ro);
            st_rec(rec_quad(rec, seg + i, REC_DIR), rd);
         }
         if (!defer) {
            float4 ro, rd;
            if (fp.primary_implicit)
               primary_state(fp, id, ro, rd);
            else {
               ro = ld_rec(rec_quad(rec, seg + i, REC_ORIGIN));
               rd = ld_rec(rec_quad(rec, seg + i, REC_DIR));
            }
            const V3 o = v3(ro.x, ro.y, ro.z), d = v3(rd.x, rd.y, rd.z);
            Hit best;
            best.t = 10000.0f;  // rgen:45: tmax
            best.u = best.v = 0.0f;
            best.idx = kEmptyRef;
            best.key = 0xffffffffu;
            uint2 en = make_uint2(0u, 0u);
            if (e < end) en = reinterpret_cast<const uint2*>(g.entries)[e];
            while (e < end) {
               // sorted by the bound, ascending: from here on no packet can be hit nearer than the best hit (nor tie with it)
               if (sorted && -__uint_as_float(en.y) > best.t) break;
               const uint32_t pk = en.x;
               uint2 nxt = make_uint2(0u, 0u);
               if (e + 1 < end) nxt = reinterpret_cast<const uint2*>(g.entries)[e + 1];  // in flight with the packet
               const float4 a = tris[kTriStride16 * (size_t)pk + 0], b = tris[kTriStride16 * (size_t)pk + 1], c = tris[kTriStride16 * (size_t)pk + 2];
               if (COUNT) n_tris++;
               tri_compute<false>(a, b, c, pk, o, d, 0.001f, INFINITY, best);
               en = nxt;
               e++;
            }
            st_rec(ps.hit + seg + i, make_float4(best.t, best.u, best.v, __uint_as_float(best.idx)));
         }
      }
      if (__ballot(defer) != 0ull) {  // wave-uniform: every lane of the wave takes part in the append
         const uint32_t slot = wave_append(n_tree, defer);
         if (defer) st_stream(q_tree + slot, i);
      }
   }
   if (sx.lb == 0 && threadIdx.x == 0) atomicAdd(&stats->rays[UH_RAY_PRIMARY], (unsigned long long)count);
   if (COUNT) atomicAdd(&stats->cam_tris_tested, (unsigned long long)n_tris);
}

// the G-buffer cast (gbuffer.rs:11-52 as a primary-ray cast: k_gbuffer_generate) through the camera grid: ray j of the cast is the ray
// through the centre of pixel spans.pixel_of(j). A pixel whose list is too long to have been sorted is walked whole (no early exit).
__global__ __launch_bounds__(kBlock) void k_gbuffer_camera_grid(SceneDev sc, RawRays ps, RowSpans spans, uint32_t W, SunGridDev g) {
   const uint32_t n = spans.total();
   const float4* __restrict__ tris = sc.tris;
   for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
      const uint32_t pix = spans.pixel_of(j);
      const uint32_t px = pix % W, py = pix / W;
      const uint32_t cell = (py + 1) * g.nx + (px + 1);
      uint32_t e = g.cell_start[cell];
      const uint32_t end = g.cell_start[cell + 1];
      const bool sorted = end - e <= g.max_walk;
      const float4 ro = ps.ray_o[j], rd = ps.ray_d[j];
      const V3 o = v3(ro.x, ro.y, ro.z), d = v3(rd.x, rd.y, rd.z);
      Hit best;
      best.t = rd.w;
      best.u = best.v = 0.0f;
      best.idx = kEmptyRef;
      best.key = 0xffffffffu;
      for (; e < end; e++) {
         const uint2 en = reinterpret_cast<const uint2*>(g.entries)[e];
         if (sorted && -__uint_as_float(en.y) > best.t) break;
         const uint32_t pk = en.x;
         const float4 a = tris[kTriStride16 * (size_t)pk + 0], b = tris[kTriStride16 * (size_t)pk + 1], c = tris[kTriStride16 * (size_t)pk + 2];
         tri_compute<false>(a, b, c, pk, o, d, ro.w, INFINITY, best);
      }
      ps.hit[j] = make_float4(best.t, best.u, best.v, __uint_as_float(best.idx));
   }
}

// stand-alone any-hit query (uh_trace_any): occluded[i] = 1 when some triangle lies in (tmin, tmax) of ray i
__global__ __launch_bounds__(kBlock) void k_trace_any_raw(SceneDev sc, const float4* __restrict__ ray_o, const float4* __restrict__ ray_d,
                                                          uint32_t* __restrict__ occluded, uint32_t count) {
   __shared__ uint32_t s_stack[kWavesPerBlock][kLdsStack][64];
   uint32_t* lds_col = &s_stack[threadIdx.x >> 6][0][lane_id()];
   uint32_t n_nodes = 0, n_tris = 0;
   for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
      float4 ro = ray_o[i], rd = ray_d[i];
      Hit h;
      occluded[i] = traverse<true, false>(sc, v3(ro.x, ro.y, ro.z), v3(rd.x, rd.y, rd.z), ro.w, rd.w, INFINITY, h, lds_col, n_nodes, n_tris) ? 1u : 0u;
   }
}

// ------------------------------------------------------------------------------------------
// generate — reference.rgen:24-40: RNG init, payload seed copy, jitter, primary ray
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_generate(FrameParams fp, PathState ps, Control* ctl, uint32_t sample) {
   const uint32_t n = fp.n_owned * fp.batch_frames;  // owned pixels x frames of the batch
   const uint32_t lane = lane_id();
   const uint32_t groups = (n + 63) / 64;
   // one wave per 64 owned pixels; a path's shard (shard_of_run of its 64-path run) decides which queue segment receives its id,
   // and its state goes to the position it gets there (set 0: the state of bounce 0's queue) - the 64 paths of a wave are one
   // run, so one append hands them 64 consecutive positions: plain coalesced 1-KiB stores per plane
   for (uint32_t g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); g < groups; g += gridDim.x * kWavesPerBlock) {
      const uint32_t q = g * 64u + lane;
      bool own = q < n;
      uint32_t id = 0;
      float4 s_o = make_float4(0, 0, 0, 0), s_d = make_float4(0, 0, 0, 0);
      if (own) id = q;
      if (own && !fp.primary_implicit) {  // (primary_implicit: the kernels of bounce 0 compute the state from the id)
         // path id = frame of the batch x owned pixels + index into the rank's owned-pixel list: dense, so a rank's wavefront
         // carries as many frames as its share of the frame allows (on one GPU: frame x W x H + pixel)
         const uint32_t f = q / fp.n_owned, k = q - f * fp.n_owned;
         const uint32_t pix = fp.owned_pixels ? fp.owned_pixels[k] : k;
         uint32_t px = pix % fp.W, py = pix / fp.W;
         uint32_t rng = sample == 0 ? init_rng(px, py, fp.W, fp.frame_numbers[f]) : __float_as_uint(ps.radf[id].w);  // rgen:24; later samples: where the last one left it
         uint32_t seed = rng;                                                                   // rgen:30
         float jx = random_float(rng), jy = random_float(rng);                                  // rgen:31
         V3 o, d;
         primary_ray(fp, px, py, jx, jy, o, d);
         s_o = make_float4(o.x, o.y, o.z, __uint_as_float(rng));
         s_d = make_float4(d.x, d.y, d.z, __uint_as_float(seed));
         // radiance = 0 / pixelColor = 0 (rgen:26,40) are not materialised: the bounce-0 shading kernels and the first
         // finish_sample use the constants directly
      }
      // the 64 paths of a wave are one run of ids (hence one shard); the loop below is kept for the general case
      uint32_t shard = own ? shard_of_run(id >> 6) : 0xffffffffu;
      unsigned long long todo = __ballot(own);
      while (todo) {
         const int leader = __ffsll((long long)todo) - 1;
         const uint32_t s = __shfl(shard, leader);
         const bool mine = own && shard == s;
         uint32_t slot = wave_append(&ctl->q_count[qc_index(0, Q_RAY, s)], mine);
         if (mine) {
            const uint32_t pos = s * ps.shard_cap + slot;
            ps.queue[0][pos] = id;
            if (!fp.primary_implicit) {
               st_rec(rec_quad(ps.set[0], pos, REC_ORIGIN), s_o);
               st_rec(rec_quad(ps.set[0], pos, REC_DIR), s_d);
               st_rec(rec_quad(ps.set[0], pos, REC_THR), make_float4(1.0f, 1.0f, 1.0f, 0.0f));  // throughput = 1 (rgen:39)
            }
         }
         todo &= ~__ballot(mine);
      }
   }
}

// ------------------------------------------------------------------------------------------
// shade_miss — reference.rmiss:10-31 + rgen:48-57 for paths whose ray left the scene.
// Walks the bounce's RAY queue and picks out the paths whose hit record says "miss" (the traversal
// kernels build no hit / miss queues). The sky integral is ~3k VALU instructions per path, so the
// misses are first compacted inside the wave: ids collect in a per-wave LDS list and the integral
// runs on 64 of them at a time with every lane live.
// ------------------------------------------------------------------------------------------
// pos: the path's position in the bounce's ray queue (shard segment included), id: its path id. The path ends here: its radiance
// goes to the per-id array, with the raygen's RNG word (the frame's next sample starts from it, rgen:28-31)
__device__ __forceinline__ void shade_miss_path(const FrameParams& fp, const PathState& ps, uint32_t pos, uint32_t id, uint32_t bounce) {
   const PathRecs rec = ps.set[bounce & 1];
   const bool implicit = bounce == 0 && fp.primary_implicit;  // origin and throughput of a primary ray: not stored (FrameParams)
   float4 ro, rd_implicit = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
   if (implicit)
      primary_state(fp, id, ro, rd_implicit);
   else
      ro = ld_rec(rec_quad(rec, pos, REC_ORIGIN));
   V3 sky_color = v3(0.0f, 0.0f, 0.0f);
   if (fp.furnace) {
      sky_color = v3(1.0f, 1.0f, 1.0f);  // rmiss:12 with FURNACE_TEST defined: the #ifndef block (rmiss:14-28) is compiled out
   } else if (fp.sky_enabled == 1) {
      const float4 rd = implicit ? rd_implicit : ld_rec(rec_quad(rec, pos, REC_DIR));
      V3 c = sky::integrate_scattering(v3(ro.x, ro.y, ro.z), v3(rd.x, rd.y, rd.z), 999999999.0f, v3(fp.sun_dir[0], fp.sun_dir[1], fp.sun_dir[2]));
      sky_color = v3(fminf(c.x, 1.0f), fminf(c.y, 1.0f), fminf(c.z, 1.0f));  // rmiss:22
   }
   float4 thr = make_float4(1.0f, 1.0f, 1.0f, 0.0f);  // rgen:39
   if (!implicit) thr = ld_rec(rec_quad(rec, pos, REC_THR));
   float4 rad = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
   if (bounce != 0) rad = ld_rec(rec_quad(rec, pos, REC_RAD));
   V3 t = v3(thr.x, thr.y, thr.z) * sky_color;                               // rgen:48
   st_stream(ps.radf + id, make_float4(rad.x + t.x, rad.y + t.y, rad.z + t.z, ro.w));   // rgen:55
}

// The bounce's misses arrive as a queue of their own (Q_MISS), written by k_shade_hit while it classifies the bounce's RAY
// queue: every lane has a path, no hit record is read here, and nothing but the sky integral (~600 VALU instructions per
// path) and four 16-byte records per path is left in the kernel.
__global__ __launch_bounds__(kBlock) void k_shade_miss(FrameParams fp, PathState ps, Control* ctl, DeviceStats* stats, uint32_t bounce) {
   const ShardCtx sx = shard_ctx();
   const uint32_t seg = sx.shard * ps.shard_cap;
   const uint2* __restrict__ queue = reinterpret_cast<const uint2*>(ps.queue[4]) + seg;  // (position in the bounce's ray queue, path id)
   const uint32_t count = ctl->q_count[qc_index(bounce, Q_MISS, sx.shard)];
   for (uint32_t i = sx.lb * kBlock + threadIdx.x; i < count; i += sx.nb * kBlock) {
      const uint2 e = ld_stream(queue + i);
      shade_miss_path(fp, ps, seg + e.x, e.y, bounce);
   }
   if (sx.lb == 0 && threadIdx.x == 0 && count) atomicAdd(&stats->misses, (unsigned long long)count);
}

// ------------------------------------------------------------------------------------------
// shade_hit — reference.rchit:20-92 + rgen:48-61 and the light-sample selection of rgen:81-110
// ------------------------------------------------------------------------------------------
// Material type 4 - an EXTENSION that no reference scene uses: the Cook-Torrance BRDF of
// include/pbr_lighting.glsl:20-79 / include/brdf.glsl:3-36,82-85 (GGX D, Smith-Schlick G with k = (r+1)^2/8, Schlick F,
// kD = (1-F)(1-metallic), +0.0001 in the denominator) evaluated for the Lambertian-style scatter direction
// L = normalize(n + randomPointInUnitSphere) and returned as BRDF * cos / pdf with pdf = cos/pi, i.e.
// kD * baseColor + specular * pi. Same operations in the same order as oracle.cpp::pbr_weight (bit-identical).
__device__ __forceinline__ V3 pbr_weight(V3 N, V3 V, V3 L, V3 base, float metallic, float roughness) {
   const float PI = 3.14159265359f;
   const V3 H = normalize3(V + L);
   const float a = roughness * roughness, a2 = a * a;
   const float NdotH = fmaxf(dot3(N, H), 0.0f), NdotH2 = NdotH * NdotH;
   float denom = NdotH2 * (a2 - 1.0f) + 1.0f;
   denom = (PI * denom) * denom;
   const float NDF = a2 / denom;
   const float NdotV = fmaxf(dot3(N, V), 0.0f), NdotL = fmaxf(dot3(N, L), 0.0f);
   const float r = roughness + 1.0f, k = (r * r) / 8.0f;
   const float gV = NdotV / (NdotV * (1.0f - k) + k), gL = NdotL / (NdotL * (1.0f - k) + k);
   const float G = gL * gV;
   const float c = fminf(fmaxf(1.0f - fmaxf(dot3(H, V), 0.0f), 0.0f), 1.0f);
   const float c5 = ((c * c) * (c * c)) * c;
   const float om = 1.0f - metallic;
   const V3 F0 = v3(0.04f * om + base.x * metallic, 0.04f * om + base.y * metallic, 0.04f * om + base.z * metallic);
   const V3 F = v3(F0.x + (1.0f - F0.x) * c5, F0.y + (1.0f - F0.y) * c5, F0.z + (1.0f - F0.z) * c5);
   const V3 kD = v3((1.0f - F.x) * om, (1.0f - F.y) * om, (1.0f - F.z) * om);
   const float den = (4.0f * NdotV) * NdotL + 0.0001f;
   const float dg = NDF * G;
   const V3 spec = v3((dg * F.x) / den, (dg * F.y) / den, (dg * F.z) / den);
   return v3(kD.x * base.x + spec.x * PI, kD.y * base.y + spec.y * PI, kD.z * base.z + spec.z * PI);
}

__device__ __forceinline__ float schlick_reflectance(float cosine, float ref_idx) {  // rchit:12-18
   float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
   r0 = r0 * r0;
   float x = 1.0f - cosine;
   float x5 = ((x * x) * (x * x)) * x;  // pow(x, 5.0)
   return r0 + (1.0f - r0) * x5;
}
__device__ __forceinline__ V3 reflect3(V3 I, V3 N) { return I - N * (2.0f * dot3(N, I)); }
__device__ __forceinline__ V3 refract3(V3 I, V3 N, float eta) {
   float dn = dot3(N, I);
   float k = 1.0f - eta * eta * (1.0f - dn * dn);
   if (k < 0.0f) return v3(0, 0, 0);
   return I * eta - N * (eta * dn + sqrtf(k));
}

// ---- the closest-hit shader's arithmetic, shared by k_shade_hit (one bounce of a wavefront) and k_path_fused (a lone frame's later
// bounces inside one persistent kernel): the same expressions in the same order, so both give the path the same words
struct SurfaceHit {
   V3 world_normal, origin;  // rchit:32-37; where the path goes on from (and its shadow rays start): rgen:59-60
   float uu, vv;             // rchit:39
   uint32_t mesh_index;
};
// s0..s3: the hit's shading packet (SceneDev::shade); the mesh record comes second because its index is in the packet
__device__ __forceinline__ void surface_normal_uv(const float4 s0, const float4 s1, const float4 s2, const float4 s3, float bu, float bv, V3& normal, float& uu, float& vv) {
   const V3 n0 = v3(s0.x, s0.y, s0.z), n1 = v3(s0.w, s1.x, s1.y), n2 = v3(s1.z, s1.w, s2.x);
   const float uv0x = s2.y, uv0y = s2.z, uv1x = s2.w, uv1y = s3.x, uv2x = s3.y, uv2y = s3.z;
   const float bx = 1.0f - bu - bv, by = bu, bz = bv;                            // rchit:30
   normal = (n0 * bx + n1 * by) + n2 * bz;                                       // rchit:31
   uu = (uv0x * bx + uv1x * by) + uv2x * bz;                                     // rchit:39
   vv = (uv0y * bx + uv1y * by) + uv2y * bz;
}
__device__ __forceinline__ V3 world_normal_of(const MeshShade& ms, V3 normal, V3 ray_dir) {
   V3 wn = v3((normal.x * ms.w2o[0] + normal.y * ms.w2o[3]) + normal.z * ms.w2o[6],
              (normal.x * ms.w2o[1] + normal.y * ms.w2o[4]) + normal.z * ms.w2o[7],
              (normal.x * ms.w2o[2] + normal.y * ms.w2o[5]) + normal.z * ms.w2o[8]);  // rchit:32
   V3 world_normal = normalize3(wn);
   if (dot3(world_normal, ray_dir) > 0.0f) world_normal = vneg(world_normal);   // rchit:35-37
   return world_normal;
}
// Does the path go on? Decided by the material type and the side the ray came from (rchit:47-89) - nothing the texels bring
__device__ __forceinline__ bool path_scatters(const MeshShade& ms, V3 ray_dir, V3 world_normal) {
   return (ms.type == 0.0f || ms.type == 4.0f) ? dot3(ray_dir, world_normal) < 0.0f : (ms.type == 1.0f || ms.type == 2.0f);
}
// rchit:47-89: the scatter direction; `color` in: texel x base colour (rchit:40-41), out: what the throughput is multiplied by
__device__ __forceinline__ V3 material_scatter(const MeshShade& ms, V3 ray_dir, V3 world_normal, V3& color, uint32_t& seed) {
   V3 scatter = v3(0, 0, 0);
   if (ms.type == 0.0f) {                                                        // rchit:47-50
      scatter = world_normal + random_point_in_unit_sphere(seed);            // scattered = dot(ray, normal) < 0: path_scatters
   } else if (ms.type == 1.0f) {                                                 // rchit:52-59
      scatter = reflect3(normalize3(ray_dir), world_normal);
      scatter = scatter + ms.property * random_point_in_unit_sphere(seed);
      color = v3(1, 1, 1);
   } else if (ms.type == 2.0f) {                                                 // rchit:61-83
      V3 nd = normalize3(ray_dir);
      float dnd = dot3(nd, world_normal);
      V3 outward = dnd > 0 ? vneg(world_normal) : world_normal;
      float ratio = ms.property;
      ratio = dnd > 0 ? ratio : 1.0f / ratio;
      float cos_theta = fminf(dot3(-1.0f * nd, outward), 1.0f);
      float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
      bool cannot_refract = ratio * sin_theta > 1.0f;
      float reflectance = schlick_reflectance(cos_theta, ratio);
      if (cannot_refract || reflectance > random_float(seed))
         scatter = reflect3(nd, outward);
      else
         scatter = refract3(nd, outward, ratio);
      color = v3(1, 1, 1);
   } else if (ms.type == 4.0f) {
      // EXTENSION (never produced by the reference's scenes): Cook-Torrance, see pbr_weight()
      scatter = world_normal + random_point_in_unit_sphere(seed);
      color = pbr_weight(world_normal, -1.0f * normalize3(ray_dir), normalize3(scatter), color, ms.metallic, ms.roughness);
   } else {                                                                      // rchit:85-89: the path ends
      color = v3(1, 1, 1);
   }
   return scatter;
}
// rgen:81-121: which light the scattered path asks, and the weight f its throughput gets if the light is visible from `origin`
__device__ __forceinline__ bool select_light(const FrameParams& fp, const SceneDev& sc, uint32_t id, uint32_t& rng_x, V3 origin, float& f, int& light_index) {
   float light_sample_weight = 0.0f, total_weights = 1.0f;
   const uint32_t k = id % fp.n_owned;
   const uint32_t pix = fp.owned_pixels ? fp.owned_pixels[k] : k;
   uint32_t px = pix % fp.W;
   bool use_reservoir = (px > fp.W / 2 || fp.full_frame_restir) && fp.use_ris == 1;  // rgen:87
   if (use_reservoir) {
      UhReservoir rs = fp.spatial_of[id / fp.n_owned][pix];                // rgen:98 (the path's own frame of the batch)
      light_sample_weight = rs.W_X;
      total_weights = rs.W_sum;
      light_index = rs.Y;
   } else {
      sample_light_uniform(fp.num_lights_used, rng_x, light_index, light_sample_weight);  // rgen:107
      light_sample_weight = 1.0f / light_sample_weight;                    // rgen:108
   }
   if (total_weights != 0.0f) {                                            // rgen:112
      f = target_function(sc.lights, sc.num_lights, light_index, origin) * light_sample_weight;  // rgen:121
      return true;
   }
   return false;
}

__global__ __launch_bounds__(kBlock, UH_SHADE_HIT_BLOCKS) void k_shade_hit(FrameParams fp, SceneDev sc, PathState ps, Control* ctl, DeviceStats* stats, uint32_t bounce) {
   // c / 255.0f table in LDS: the 12 per-fetch table gathers were texture-addresser traffic (the
   // kernel ran 86 % TA-busy at 2 % VALU, profiles/r01c_*); LDS serves them at no TA cost
   __shared__ float s_lut[256];
   s_lut[threadIdx.x] = sc.unorm_lut[threadIdx.x];
   // per-mesh shading records and texture descriptors of the first kLdsMeshes / kLdsTextures entries: every hit
   // gathers 64 + 24 bytes of them per lane, and the kernel is bound by its gather traffic (texture addresser)
   constexpr uint32_t kLdsMeshes = 192, kLdsTextures = 64;
   __shared__ MeshShade s_mesh[kLdsMeshes];
   __shared__ TexInfo s_tex[kLdsTextures];
   const uint32_t n_lds_mesh = sc.num_meshes < kLdsMeshes ? sc.num_meshes : kLdsMeshes;
   const uint32_t n_lds_tex = sc.num_textures < kLdsTextures ? sc.num_textures : kLdsTextures;
   __shared__ uint32_t s_hits;
   __shared__ uint32_t s_list[kWavesPerBlock][6][128];  // per wave: path ids, their queue positions, and the hit record (t, u, v, packet) the classification read with them
   __shared__ uint32_t s_miss[kWavesPerBlock][2][128];  // per wave: (position, id) of the paths that missed, handed to k_shade_miss 64 at a time
   if (threadIdx.x == 0) s_hits = 0;
   if (threadIdx.x < n_lds_mesh) s_mesh[threadIdx.x] = sc.meshes[threadIdx.x];
   if (threadIdx.x < n_lds_tex) s_tex[threadIdx.x] = sc.textures[threadIdx.x];
   __syncthreads();
   const ShardCtx sx = shard_ctx();
   const uint32_t seg = sx.shard * ps.shard_cap;
   // the bounce's RAY queue: paths whose hit record says "miss" belong to k_shade_miss and are skipped here
   const uint32_t count = ctl->q_count[qc_index(bounce, Q_RAY, sx.shard)];
   const uint32_t* __restrict__ queue = ps.queue[bounce & 1] + seg;
   const PathRecs cur = ps.set[bounce & 1], nxt = ps.set[(bounce + 1) & 1];  // state by position in this bounce's queue / in the next one's
   uint32_t* q_next = ps.queue[(bounce + 1) & 1] + seg;
   uint32_t* n_next = &ctl->q_count[qc_index(bounce + 1, Q_RAY, sx.shard)];
   uint32_t* q_light = ps.queue[2] + seg;
   uint32_t* n_light = &ctl->q_count[qc_index(bounce, Q_LIGHT, sx.shard)];
   uint2* q_miss = reinterpret_cast<uint2*>(ps.queue[4]) + seg;
   uint32_t* n_miss = &ctl->q_count[qc_index(bounce, Q_MISS, sx.shard)];
   const uint32_t stride = sx.nb * kBlock;
   const uint32_t rounds = (count + stride - 1) / stride;
   uint32_t n_hits = 0;
   // one path per lane: `valid` lanes shade their hit; every lane of the wave takes part in the queue appends.
   // pos: the path's position in this bounce's queue (without the shard segment)
   auto shade = [&](uint32_t id, uint32_t pos, float4 hr, bool valid) {
      const uint32_t pk = __float_as_uint(hr.w);
      bool scattered = false, want_light = false;
      uint32_t slot = 0;        // the scattered path's position in the next bounce's queue
      float4 n_o = make_float4(0, 0, 0, 0), n_d = make_float4(0, 0, 0, 0), n_t = make_float4(0, 0, 0, 0), n_r = make_float4(0, 0, 0, 0);  // the scattered path's new state
      if (valid) {
         // every record of the path is requested up front, together with the shading packet (whose index the
         // classification below already read): one round trip for all of them, then one for the texels.
         // The path's state: four planes at its queue position (the RNG words ride in the rays' w components); the lanes of a wave
         // hold hits of nearly consecutive positions, so these are nearly contiguous reads
         float4 ro, rd, thr4;
         if (bounce == 0 && fp.primary_implicit) {  // a primary ray's state is not stored (FrameParams::primary_implicit): from its id
            primary_state(fp, id, ro, rd);
            thr4 = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
         } else {
            ro = ld_rec(rec_quad(cur, seg + pos, REC_ORIGIN));
            rd = ld_rec(rec_quad(cur, seg + pos, REC_DIR));
            thr4 = ld_rec(rec_quad(cur, seg + pos, REC_THR));
         }
         uint2 rng = make_uint2(__float_as_uint(ro.w), __float_as_uint(rd.w));
         float4 rad4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // radiance so far: zero before the first bounce (not materialised)
         if (bounce != 0) rad4 = ld_rec(rec_quad(cur, seg + pos, REC_RAD));  // (the compiler waits for it on the spot: measured level with an unconditional read, which costs 16 bytes per hit at bounce 0)
         const V3 ray_dir = v3(rd.x, rd.y, rd.z);
         const float t = hr.x, bu = hr.y, bv = hr.z;
         const float4* sp = sc.shade + 4 * (size_t)pk;  // pk = hr.w, known since the classification: no wait for hr before these
         float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
         const uint32_t mesh_index = __float_as_uint(s3.w);
         // rchit:22-23. The LDS copy is read unconditionally (clamped index, explicit ds_read: device_math.h lds_fetch)
         // and replaced from global memory for the meshes beyond the LDS table
         MeshShade ms = lds_fetch(s_mesh + (mesh_index < kLdsMeshes ? mesh_index : kLdsMeshes - 1));
         if (mesh_index >= n_lds_mesh) ms = sc.meshes[mesh_index];
         V3 normal;
         float uu, vv;
         surface_normal_uv(s0, s1, s2, s3, bu, bv, normal, uu, vv);                    // rchit:30-31, :39
         // keeps the compiler from sinking the early loads to their first use
         asm volatile("" : "+v"(rng.x), "+v"(rng.y), "+v"(thr4.x), "+v"(thr4.y), "+v"(thr4.z), "+v"(rad4.x), "+v"(rad4.y), "+v"(rad4.z));
         const V3 world_normal = world_normal_of(ms, normal, ray_dir);                 // rchit:32-37
         // where the path goes on from (and its sun ray starts): needs nothing the texels bring
         V3 origin = v3(ro.x, ro.y, ro.z) + t * ray_dir;                               // rgen:59
         origin = offset_ray(origin, world_normal);                                    // rgen:60
         // Does the path go on? Decided by the material type and the side the ray came from (rchit:47-89) - nothing the texels
         // bring -, so the scattered paths' places in the next bounce's queue are asked for together with the texels: the returning
         // atomic is in flight with them instead of a round trip of its own after the material evaluation. (Inside the divergent
         // block: the ballot sees the valid lanes, which are the only ones that can scatter.)
         scattered = path_scatters(ms, ray_dir, world_normal);
         const unsigned long long scat_mask = __ballot(scattered);
         const int scat_leader = __ffsll((long long)scat_mask) - 1;
         uint32_t scat_base = 0;
         V3 color = sample_texture_pre(sc, s_lut, ms.diffuse_map, uu, vv, s_tex, n_lds_tex, [&] {  // rchit:40
            // (the counter's address goes through a register the compiler cannot see into: for a wave-uniform address its atomic
            // optimiser rewrites the add as a wave reduction with a readfirstlane of the result - and a wait for it - on the spot)
            typedef __attribute__((address_space(1))) uint32_t* global_u32_t;  // (a pointer of unknown address space would make it a flat atomic, which LDS waits wait for)
            global_u32_t counter = (global_u32_t)n_next;
            asm volatile("" : "+v"(counter));
            if (scattered && (int)lane_id() == scat_leader) scat_base = __hip_atomic_fetch_add(counter, (uint32_t)__popcll(scat_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         });
         color = color * v3(ms.base_color[0], ms.base_color[1], ms.base_color[2]);    // rchit:41

         uint32_t seed = rng.y;
         const V3 scatter = material_scatter(ms, ray_dir, world_normal, color, seed);  // rchit:47-89
         rng.y = seed;                                                                 // rchit:91
         if (scat_leader >= 0) slot = (uint32_t)__builtin_amdgcn_readlane((int)scat_base, scat_leader) + __builtin_amdgcn_mbcnt_hi((uint32_t)(scat_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)scat_mask, 0u));

         V3 thr = v3(thr4.x, thr4.y, thr4.z) * color;                                  // rgen:48
         if (!scattered) {                                                             // rgen:53-57: the path ends here
            // its radiance goes to the per-id array k_finish_sample reads, with the raygen's RNG word: the next sample of the
            // frame starts from it (rgen:28-31)
            st_stream(ps.radf + id, make_float4(rad4.x + thr.x, rad4.y + thr.y, rad4.z + thr.z, __uint_as_float(rng.x)));
         } else {
            float f = 0.0f;
            int light_index = 0;
            if (fp.lights_enabled == 1) want_light = select_light(fp, sc, id, rng.x, origin, f, light_index);  // rgen:81-121
            // the path's state for the next bounce; written below at the position the path gets in the next bounce's queue
            n_o = make_float4(origin.x, origin.y, origin.z, __uint_as_float(rng.x));
            n_d = make_float4(scatter.x, scatter.y, scatter.z, __uint_as_float(rng.y));  // rgen:61
            n_t = make_float4(thr.x, thr.y, thr.z, f);
            n_r = make_float4(rad4.x, rad4.y, rad4.z, __uint_as_float((uint32_t)light_index));  // the radiance travels with the path
         }
      }
      // a wave's scattered paths get consecutive positions: their new state leaves as contiguous stores, and the next bounce's
      // traversal, sun-ray and shading kernels read it back as streams
      // the light queue: one round trip - none at all when no lane has an entry
      const uint32_t lslot = wave_append(n_light, want_light);
      if (scattered) {
         st_stream(q_next + slot, id);
         st_rec(rec_quad(nxt, seg + slot, REC_ORIGIN), n_o);
         if (bounce + 1 < fp.num_bounces) st_rec(rec_quad(nxt, seg + slot, REC_DIR), n_d);  // (after the last bounce no ray is traced: k_flush_survivors reads radiance and the raygen RNG word)
         st_rec(rec_quad(nxt, seg + slot, REC_THR), n_t);
         st_rec(rec_quad(nxt, seg + slot, REC_RAD), n_r);
      }
      if (want_light) st_stream(q_light + lslot, slot);  // the light ray leaves from the path's new position
   };
   // The bounce's RAY queue holds hits and misses (the traversal kernels build no hit / miss queues). Shading a wave of
   // queue entries as they come leaves the lanes of the misses idle through the whole material evaluation, so the hits
   // are first compacted inside the wave: their ids collect in a per-wave LDS list and are shaded 64 at a time.
   uint32_t(*list)[128] = s_list[threadIdx.x >> 6];  // [0] ids, [1] positions, [2..5] the hit record's four words
   uint32_t(*missed)[128] = s_miss[threadIdx.x >> 6];  // [0] positions, [1] ids
   const uint32_t lane = lane_id();
   uint32_t n_list = 0, n_missed = 0;  // wave-uniform
   // `n` entries from the front of the wave's miss list go to the bounce's miss queue (one atomic per call)
   auto flush_misses = [&](uint32_t n) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(n_miss, n);
      base = __shfl(base, 0);
      if (lane < n) st_stream(q_miss + base + lane, make_uint2(missed[0][lane], missed[1][lane]));
   };
   auto entry = [&](uint32_t k) { return make_float4(__uint_as_float(list[2][k]), __uint_as_float(list[3][k]), __uint_as_float(list[4][k]), __uint_as_float(list[5][k])); };
   for (uint32_t r = 0; r < rounds; r++) {
      const uint32_t i = r * stride + sx.lb * kBlock + threadIdx.x;
      uint32_t id = 0;
      float4 hr = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kEmptyRef));
      if (i < count) {
         id = ld_stream(queue + i);
         // the whole record (one 16-byte lane load, as the packet index alone would be): shade() does not read it again. The
         // traversal kernel left it at the ray's queue position: the wave reads 1 KiB in one piece
         hr = ld_rec(ps.hit + seg + i);
      }
      const bool is_hit = __float_as_uint(hr.w) != kEmptyRef;
      const unsigned long long mask = __ballot(is_hit);
      const unsigned long long mmask = __ballot(i < count && !is_hit);
      if (mmask) {
         const uint32_t mp = __builtin_amdgcn_mbcnt_hi((uint32_t)(mmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mmask, 0u));
         if (i < count && !is_hit) {
            missed[0][n_missed + mp] = i;
            missed[1][n_missed + mp] = id;
         }
         n_missed += (uint32_t)__popcll(mmask);
         __builtin_amdgcn_wave_barrier();
         if (n_missed >= 64u) {
            flush_misses(64u);
            const uint32_t rest = n_missed - 64u;  // at most 63: to the front (one wave, LDS operations execute in order)
            uint32_t tmp0 = 0, tmp1 = 0;
            if (lane < rest) {
               tmp0 = missed[0][64u + lane];
               tmp1 = missed[1][64u + lane];
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) {
               missed[0][lane] = tmp0;
               missed[1][lane] = tmp1;
            }
            __builtin_amdgcn_wave_barrier();
            n_missed = rest;
         }
      }
      if (mask == 0ull) continue;
      const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
      if (is_hit) {
         list[0][n_list + prefix] = id;
         list[1][n_list + prefix] = i;
         list[2][n_list + prefix] = __float_as_uint(hr.x);
         list[3][n_list + prefix] = __float_as_uint(hr.y);
         list[4][n_list + prefix] = __float_as_uint(hr.z);
         list[5][n_list + prefix] = __float_as_uint(hr.w);
      }
      n_list += (uint32_t)__popcll(mask);
      __builtin_amdgcn_wave_barrier();
      if (n_list >= 64u) {
         shade(list[0][lane], list[1][lane], entry(lane), true);
         n_hits += 64u;
         const uint32_t rest = n_list - 64u;  // move the tail (at most 63 entries) to the front: one wave, LDS operations execute in order
         uint32_t tmp[6] = {0, 0, 0, 0, 0, 0};
         if (lane < rest)
            for (int k = 0; k < 6; k++) tmp[k] = list[k][64u + lane];
         __builtin_amdgcn_wave_barrier();
         if (lane < rest)
            for (int k = 0; k < 6; k++) list[k][lane] = tmp[k];
         __builtin_amdgcn_wave_barrier();
         n_list = rest;
      }
   }
   if (n_list) {
      shade(lane < n_list ? list[0][lane] : 0u, lane < n_list ? list[1][lane] : 0u, lane < n_list ? entry(lane) : make_float4(0.0f, 0.0f, 0.0f, 0.0f), lane < n_list);
      n_hits += n_list;
   }
   if (n_missed) flush_misses(n_missed);
   // closest_hits: per-block sum, one atomic per block (n_hits is wave-uniform)
   if (lane_id() == 0 && n_hits) atomicAdd(&s_hits, n_hits);
   __syncthreads();
   if (threadIdx.x == 0 && s_hits) atomicAdd(&stats->closest_hits, (unsigned long long)s_hits);
}

// The paths still alive after the last bounce (the ray queue shade_hit(num_bounces - 1) built): their radiance - with what the
// last bounce's shadow rays added - and their raygen RNG word go to the per-id array k_finish_sample and the next sample's
// k_generate read (rgen:127, :28-31). Dense reads by queue position, one 16-byte store per surviving path.
__global__ __launch_bounds__(kBlock) void k_flush_survivors(FrameParams fp, PathState ps, Control* ctl) {
   const ShardCtx sx = shard_ctx();
   const uint32_t seg = sx.shard * ps.shard_cap, b = fp.num_bounces;
   const uint32_t count = ctl->q_count[qc_index(b, Q_RAY, sx.shard)];
   const uint32_t* __restrict__ queue = ps.queue[b & 1] + seg;
   const PathRecs rec = ps.set[b & 1];
   for (uint32_t i = sx.lb * kBlock + threadIdx.x; i < count; i += sx.nb * kBlock) {
      const uint32_t id = ld_stream(queue + i);
      const float4 rad = ld_rec(rec_quad(rec, seg + i, REC_RAD)), ro = ld_rec(rec_quad(rec, seg + i, REC_ORIGIN));
      st_stream(ps.radf + id, make_float4(rad.x, rad.y, rad.z, ro.w));
   }
}

// ------------------------------------------------------------------------------------------
// path_fused - ONE FRAME PER CALL (a moving camera, renderers/mod.rs:357, main.rs:460-471): bounces 1 .. num_bounces - 1 of a lone
// frame inside one persistent kernel. A lone frame's wavefront is some 28 launches of which every traversal launch ends in the tail of
// its longest ray (about 60 dependent steps: 0.30-0.38 ms per bounce for 1.5 M rays against 0.19 ms at the batched rate, DESIGN.md
// section 4) - nothing of the same frame can fill those tails across a launch boundary. Here EVERY BLOCK RUNS ITS OWN WAVEFRONT: a
// block owns a contiguous range of the positions of bounce 1's ray queue (its shard's count / blocks of the shard) and takes those paths
// through all the remaining bounces by itself - trace phase, block barrier, shading phase, block barrier, ... - with no word to any
// other block.
// The paths never move: a path's state stays in its record of set 1 at its position p (shading rewrites it in place), its hit record
// at hit[p]; what a phase hands to the next is a LIST of entries p | flags in the block's range of two of the queue arrays:
//    kHasRay    the path has a ray of the next bounce to trace (its record's origin / direction)
//    kSun       its sun ray was not answered by the grid (border cell, long list - or no grid): the tree's
//    kLight     it asks a light (record: f in throughput.w, light index in radiance.w)
// Trace phase: the block's waves are persistent over the list (the Feeder of the traversal kernels, chunks from a cursor in LDS); a
// lane takes an entry, walks the path's shadow rays first - sun, then light: their results are added to the path's radiance in the
// reference's order (rgen:63-122) -, then its bounce ray, and leaves the hit record. Shadow rays go through the closest-hit walk
// beside the other lanes' bounce rays (occluded <=> the closest hit lies within the limit; the walk stops at the first hit inside).
// Shading phase: k_shade_hit's, over the block's list - hits compacted per wave in LDS and shaded 64 at a time, the sun grid asked on
// the spot; the paths whose ray left the scene gathered per wave too and the sky integrated for 64 of them at a time (reference.rmiss;
// what is left of the list waits in two registers per lane for the next shading phase); ended paths' radiance to the per-id array.
// Bounce 0's sun rays are asked here as well (sun_of_bounce0), so the frame is k_generate, the camera grid, k_shade_hit(0) and
// k_shade_miss(0), this kernel, k_finish_sample.
// Same words per path as the wavefront: the per-path arithmetic is shared (surface_normal_uv .. select_light, make_shadow_ray,
// tri_compute), a path's random numbers depend on nothing but the path, and shadow rays are predicates.
// Round 5 built this kernel four ways, all bit-identical, measured on the same frames (profiles/README.md "One frame per call"): a
// path per LANE (the lane parked at its hit until 32 lanes of the wave stood at one, then the wave shaded them: lane utilisation 0.40,
// 2.6 ms for the four bounces); THIS one (2.29 ms; the wavefront's launches span 2.44); a pipeline per WAVE (hits gathered in LDS and
// shaded 64 at a time between walking steps, the next bounce's rays walked beside this one's stragglers: no drains, lane utilisation
// 0.54 - but a walk's state stays live through the shading: 168 registers, three waves per SIMD, 2.65 ms); a pipeline per BLOCK of
// three walking waves and one shading wave with rings in LDS (101 registers; 2.45-2.54 ms: the shading wave is busy 0.97 of its
// clock). What they show: the walk's rate follows the number of waves that walk - a kernel that also shades has 16, the wavefront's
// traversal kernel 20 to 24 -, and a lone frame's work cannot be had at the batched rate in one kernel. (Two rays per lane in the
// trace phases, both records asked for before either is used: 40 % slower - the walk does not wait for latency.)
// ------------------------------------------------------------------------------------------
// one sun ray through the grid (k_trace_sun_grid's walk for one ray): 0 = lit, 1 = occluded, 2 = the grid does not answer (border cell,
// long list): the tree's
template <bool COUNT, bool INLINE>
__device__ __forceinline__ int sun_grid_query(const SunGridDev& g, const float4* __restrict__ packets, V3 o, V3 d, uint32_t& n_tris, uint32_t& n_covered) {
   typedef float f4_t __attribute__((ext_vector_type(4)));
   const f4_t* __restrict__ recs = reinterpret_cast<const f4_t*>(g.recs);
   const f4_t* __restrict__ tris = reinterpret_cast<const f4_t*>(packets);
   const uint2* __restrict__ entries = reinterpret_cast<const uint2*>(g.entries);
   const uint4* __restrict__ cells = reinterpret_cast<const uint4*>(g.cell_start);
   const float pu = dot_fma(v3(g.U[0], g.U[1], g.U[2]), o), pv = dot_fma(v3(g.V[0], g.V[1], g.V[2]), o), pw = dot_fma(v3(g.W[0], g.W[1], g.W[2]), o);
   uint32_t cx, cy;
   sun_cell_of(g, pu, pv, cx, cy);
   if (g.coarse) {
      const float cw = g.coarse[(cy >> g.coarse_shift) * g.coarse_nx + (cx >> g.coarse_shift)];
      if (sun_coarse_covered(cw, pw)) {
         if (COUNT) n_covered++;
         return 1;
      }
   }
   const uint32_t cell = cy * g.nx + cx;
   const uint4 cs = cells[cell];
   const uint32_t end = cells[cell + 1].x;
   const float cover = __uint_as_float(cs.y);
   if (pw < cover && cover - pw < kSunCoverReach) {
      if (COUNT) n_covered++;
      return 1;
   }
   if (cx == 0 || cy == 0 || cx + 1 == g.nx || cy + 1 == g.ny || end - cs.x > g.max_walk) return 2;
   if (cs.x >= end) return 0;
   Hit best;
   best.t = 10000.0f;  // tmax (rgen:66)
   best.u = best.v = 0.0f;
   best.idx = kEmptyRef;
   best.key = 0xffffffffu;
   uint32_t e = cs.x;
   if constexpr (INLINE) {
      const f4_t* r = recs + 4 * (size_t)e;
      for (;;) {
         const f4_t na = r[0], nb = r[1], nc = r[2];
         if (nc.z < pw) return 0;  // from here on every packet ends behind the origin
         if (COUNT) n_tris++;
         if (tri_compute<true>(make_float4(na.x, na.y, na.z, na.w), make_float4(nb.x, nb.y, nb.z, nb.w), make_float4(nc.x, nc.y, nc.z, nc.w), 0u, o, d, 0.001f, INFINITY, best)) return 1;
         e++;
         if (e >= end || nc.w < pw) return 0;
         r += 4;
      }
   } else {
      uint2 cur = make_uint2(cs.z, cs.w);  // the list's first entry came with the cell record
      for (;;) {
         if (__uint_as_float(cur.y) < pw) return 0;  // sorted by far depth, descending
         const f4_t* r = tris + kTriStride16 * (size_t)cur.x;
         const f4_t na = r[0], nb = r[1], nc = r[2];
         e++;
         const uint2 nx = e < end ? entries[e] : make_uint2(0u, 0u);  // in flight with the packet
         if (COUNT) n_tris++;
         if (tri_compute<true>(make_float4(na.x, na.y, na.z, na.w), make_float4(nb.x, nb.y, nb.z, nb.w), make_float4(nc.x, nc.y, nc.z, nc.w), cur.x, o, d, 0.001f, INFINITY, best)) return 1;
         if (e >= end) return 0;
         cur = nx;
      }
   }
}

#ifndef UH_FUSED_STAGGER
#define UH_FUSED_STAGGER 0     // 1: the blocks of a CU offset against each other by quarters of a phase (k_path_fused STAGGER; measured 2.545 against 2.505 ms per frame: off)
#endif
#ifndef UH_FUSED_BLOCKS
#define UH_FUSED_BLOCKS 4      // blocks per CU the kernel's registers and LDS are sized for
#endif
struct FusedTraceLds {
   uint32_t stack[kWavesPerBlock][kLdsStack][64];
   RayPool<2> pool[kWavesPerBlock];
};
struct FusedShadeLds {
   uint32_t list[kWavesPerBlock][6][128];  // per wave: path ids, positions, the hit record's four words
   uint32_t miss[kWavesPerBlock][2][128];  // per wave: (position, id) of the paths that missed
};
template <bool COUNT, bool INLINE>
__global__ __launch_bounds__(kBlock, UH_FUSED_BLOCKS) void k_path_fused(SceneDev sc, FrameParams fp, PathState ps, Control* ctl, DeviceStats* stats, SunGridDev g, bool use_grid,
                                                                         bool sun_of_bounce0, uint32_t stagger_unit) {
   constexpr uint32_t kFirst = 1;  // the paths are those of bounce 1's ray queue, their state lies in set 1 at their positions there
   // (positions fit 23 bits: the host fuses only when shard_cap < 2^23; bits 23..28: how many bounces follow the entry's ray)
   constexpr uint32_t kHasRay = 1u << 31, kSun = 1u << 30, kLight = 1u << 29, kLeftShift = 23, kLeftMask = 63u << kLeftShift, kPosMask = (1u << kLeftShift) - 1u;
   __shared__ float s_lut[256];
#if UH_FUSED_BLOCKS >= 6
   constexpr uint32_t kLdsMeshes = 4, kLdsTextures = 2;    // (26.6 KiB of LDS per block: the mesh and texture records from global memory)
#elif UH_FUSED_BLOCKS >= 5
   constexpr uint32_t kLdsMeshes = 48, kLdsTextures = 32;  // (32 KiB of LDS per block)
#else
   constexpr uint32_t kLdsMeshes = 128, kLdsTextures = 64;
#endif
   __shared__ MeshShade s_mesh[kLdsMeshes];
   __shared__ TexInfo s_tex[kLdsTextures];
   __shared__ union {
      FusedTraceLds t;
      FusedShadeLds s;
   } u;  // the phases alternate
   __shared__ uint32_t s_cursor, s_count[2];
   const uint32_t n_lds_mesh = sc.num_meshes < kLdsMeshes ? sc.num_meshes : kLdsMeshes;
   const uint32_t n_lds_tex = sc.num_textures < kLdsTextures ? sc.num_textures : kLdsTextures;
   s_lut[threadIdx.x] = sc.unorm_lut[threadIdx.x];
   if (threadIdx.x < n_lds_mesh) s_mesh[threadIdx.x] = sc.meshes[threadIdx.x];
   if (threadIdx.x < n_lds_tex) s_tex[threadIdx.x] = sc.textures[threadIdx.x];
   const uint32_t lane = lane_id();
   const uint32_t wave = threadIdx.x >> 6;
   const ShardCtx sx = shard_ctx();
   const uint32_t seg = sx.shard * ps.shard_cap;
   const PathRecs rec = ps.set[kFirst & 1];
   const uint32_t* __restrict__ ids = ps.queue[kFirst & 1] + seg;  // path id by position
   // the block's range of positions [lo, hi)
   const uint32_t count1 = ctl->q_count[qc_index(kFirst, Q_RAY, sx.shard)];
   const uint32_t per = (((count1 + sx.nb - 1) / sx.nb) + 63u) & ~63u;
   const uint32_t lo = sx.lb * per < count1 ? sx.lb * per : count1, hi = lo + per < count1 ? lo + per : count1;
   uint32_t* lists[2] = {ps.queue[0] + seg + lo, ps.queue[3] + seg + lo};  // at most hi - lo entries each (one per path of the block)
   const uint4* __restrict__ nodes = sc.nodes;
   const float4* __restrict__ tris = sc.tris;
   const V3 sun_d = v3(fp.sun_dir[0], fp.sun_dir[1], fp.sun_dir[2]);
   uint32_t n_nodes = 0, n_tris = 0, n_snodes = 0, n_stris = 0, n_lnodes = 0, n_ltris = 0, n_covered = 0;  // per lane (COUNT only)
   uint32_t w_rays = 0, w_hits = 0, w_sun = 0, w_sun_tree = 0, w_light = 0, w_miss = 0;                      // per wave
   // bounce 1's list: every position of the range, a ray each. sun_of_bounce0: bounce 0's sun rays (rgen:63-79 for the paths
   // k_shade_hit(0) scattered) are asked here instead of by a k_trace_sun_grid / k_trace_shadow pair in front of this kernel - the grid
   // on the spot, what it does not answer as the entry's sun ray in the first trace phase. (Not when lights are on: bounce 0's light
   // rays are the wavefront's, and they come after the sun rays.)
   // STAGGER: the blocks of a CU would run their phases in step - all trace, all drain their last rays, all shade. The block of
   // dispatch round r (blockIdx / stagger_unit: a CU's resident blocks come from different rounds) therefore puts only the first
   // 1 - (r mod 4) / 4 of its range into the first list and the rest straight into the second: its phases are [part], [all], ..,
   // [rest] - one more than the others', offset against theirs by a quarter, a half, three quarters of a phase - and one block's
   // drain runs under the others' full lists
   const uint32_t n_range = hi - lo, round = (blockIdx.x / (stagger_unit ? stagger_unit : 1u)) & 3u;
   const uint32_t n_first = stagger_unit ? ((n_range * (4u - round) / 4u + 63u) & ~63u) < n_range ? ((n_range * (4u - round) / 4u + 63u) & ~63u) : n_range : n_range;
   const uint32_t left1 = (fp.num_bounces - 2u) << kLeftShift;  // bounce 1's rays
   for (uint32_t i0 = 0; i0 < n_range; i0 += kBlock) {
      const uint32_t i = i0 + threadIdx.x;
      const bool valid = i < n_range;
      uint32_t e = (lo + i) | kHasRay | left1;
      bool to_tree = false;
      if (valid && sun_of_bounce0) {
         int r = 2;
         if (use_grid) {
            const float4 ro = ld_rec(rec_quad(rec, seg + lo + i, REC_ORIGIN));
            r = sun_grid_query<COUNT, INLINE>(g, tris, v3(ro.x, ro.y, ro.z), sun_d, n_stris, n_covered);
            if (COUNT) n_snodes++;
         }
         if (r == 0) {  // rgen:69-78
            const float4 t4 = ld_rec(rec_quad(rec, seg + lo + i, REC_THR)), r4 = ld_rec(rec_quad(rec, seg + lo + i, REC_RAD));
            st_rec(rec_quad(rec, seg + lo + i, REC_RAD), make_float4(r4.x + t4.x, r4.y + t4.y, r4.z + t4.z, r4.w));
         }
         if (r == 2) e |= kSun;
         to_tree = r == 2 && use_grid;
      }
      if (valid) {
         if (i < n_first) lists[0][i] = e;
         else lists[1][i - n_first] = e;
      }
      if (sun_of_bounce0) {
         w_sun += (uint32_t)__popcll(__ballot(valid));
         w_sun_tree += (uint32_t)__popcll(__ballot(to_tree));
      }
   }
   if (threadIdx.x == 0) {
      s_count[0] = n_first;
      s_count[1] = n_range - n_first;
      s_cursor = 0;
   }
   __syncthreads();

   // ---- trace phase over lists[which]
   auto trace_phase = [&](uint32_t which) {
      uint32_t* lds_col = &u.t.stack[wave][0][lane];
      RayPool<2>& pool = u.t.pool[wave];
      RaySource src;
      src.queue = lists[which];
      src.count = s_count[which];
      src.cursor = &s_cursor;
      src.wave_index = src.num_waves = 0;
      auto source_of = [&](int a, uint32_t e) { return (const float4*)rec_quad(rec, seg + (e & kPosMask), a == 0 ? REC_ORIGIN : REC_DIR); };
      Feeder<2> f;
      Trav t;
      t.cur = kEmptyRef;
      t.sp = 0;
      enum : uint32_t { BOUNCE_RAY = 0, SUN_RAY = 1, LIGHT_RAY = 2 };
      uint32_t kind = BOUNCE_RAY, entry = 0, rng_x = 0, mark_nodes = 0, mark_tris = 0;
      V3 thr = v3(0, 0, 0), rad = v3(0, 0, 0), scatter = v3(0, 0, 0);
      float lf = 0.0f;
      uint32_t light_bits = 0;
      bool dirty = false;
      uint32_t spill[kSpillStack];
      // the entry's next ray (origin = t.o): sun, light, then the bounce ray - or, behind the last bounce, the path's radiance to the
      // per-id array (what k_flush_survivors writes)
      auto next_ray = [&]() {
         const float4 o4 = make_float4(t.o.x, t.o.y, t.o.z, 0.0f);
         if (entry & kSun) {
            entry &= ~kSun;
            const ShadowRay s = make_shadow_ray<false>(sc, fp, o4, make_float4(thr.x, thr.y, thr.z, lf), make_float4(rad.x, rad.y, rad.z, 0.0f));
            trav_init(t, s.ro, s.rd, s.ro.w, s.rd.w, s.tlimit);
            kind = SUN_RAY;
         } else if (entry & kLight) {
            entry &= ~kLight;
            const ShadowRay s = make_shadow_ray<true>(sc, fp, o4, make_float4(thr.x, thr.y, thr.z, lf), make_float4(rad.x, rad.y, rad.z, __uint_as_float(light_bits)));
            trav_init(t, s.ro, s.rd, s.ro.w, s.rd.w, s.tlimit);
            kind = LIGHT_RAY;
         } else {
            const uint32_t p = entry & kPosMask;
            if (entry & kHasRay) {
               if (dirty) st_rec(rec_quad(rec, seg + p, REC_RAD), make_float4(rad.x, rad.y, rad.z, __uint_as_float(light_bits)));
               trav_init(t, o4, make_float4(scatter.x, scatter.y, scatter.z, 0.0f), 0.001f, 10000.0f, INFINITY);  // rgen:61, :45-47
               kind = BOUNCE_RAY;
            } else {  // rgen:127 after the last bounce
               st_stream(ps.radf + ld_stream(ids + p), make_float4(rad.x, rad.y, rad.z, __uint_as_float(rng_x)));
               t.cur = kEmptyRef;
            }
         }
         if (COUNT) {
            mark_nodes = n_nodes;
            mark_tris = n_tris;
         }
      };
      auto take = [&](uint32_t slot) {
         entry = pool.id[slot];
         const float4 ro = pool.v[0][slot], rd = pool.v[1][slot];
         if (entry & (kSun | kLight)) {
            const uint32_t p = entry & kPosMask;
            const float4 t4 = ld_rec(rec_quad(rec, seg + p, REC_THR)), r4 = ld_rec(rec_quad(rec, seg + p, REC_RAD));
            thr = v3(t4.x, t4.y, t4.z);
            lf = t4.w;
            rad = v3(r4.x, r4.y, r4.z);
            light_bits = __float_as_uint(r4.w);
            scatter = v3(rd.x, rd.y, rd.z);
            rng_x = __float_as_uint(ro.w);
            dirty = false;
            t.o = v3(ro.x, ro.y, ro.z);
            next_ray();
         } else {
            trav_init(t, ro, rd, 0.001f, 10000.0f, INFINITY);  // rgen:45-47
            kind = BOUNCE_RAY;
         }
      };
      while (refill_lanes<2>(f, src, pool, t.cur == kEmptyRef, source_of, take)) {
         if (t.cur != kEmptyRef) {
            bool occluded = false;
            bool ended = trav_step<false, COUNT, true>(nodes, tris, t, lds_col, spill, occluded, n_nodes, n_tris);
            // a shadow ray is a predicate: occluded <=> some triangle accepts it within (tmin, tmax) and the light's distance <=> the
            // closest such hit lies within it - the walk can stop at the first hit it finds there (rgen:69, :118-119 read nothing else)
            const bool blocked = t.best.idx != kEmptyRef && t.best.t <= t.tlimit;
            if (kind != BOUNCE_RAY && blocked) ended = true;
            if (ended) {
               if (kind == BOUNCE_RAY) {
                  st_rec(ps.hit + seg + (entry & kPosMask), make_float4(t.best.t, t.best.u, t.best.v, __uint_as_float(t.best.idx)));
                  t.cur = kEmptyRef;
               } else {
                  // rgen:69-78 / :118-122: an unoccluded ray adds the path's throughput (x the light's weight) to its radiance
                  if (!blocked) {
                     rad = kind == SUN_RAY ? v3(rad.x + thr.x, rad.y + thr.y, rad.z + thr.z) : v3(rad.x + thr.x * lf, rad.y + thr.y * lf, rad.z + thr.z * lf);
                     dirty = true;
                  }
                  if (COUNT) {  // the walk's visits belong to the shadow counters
                     const uint32_t dn = n_nodes - mark_nodes, dt = n_tris - mark_tris;
                     n_nodes = mark_nodes;
                     n_tris = mark_tris;
                     if (kind == SUN_RAY) {
                        n_snodes += dn;
                        n_stris += dt;
                     } else {
                        n_lnodes += dn;
                        n_ltris += dt;
                     }
                  }
                  next_ray();  // (t.o is still the point the path's rays leave from)
               }
            }
         }
      }
   };

   uint32_t carry_n = 0, carry_pos = 0, carry_id = 0;
   // ---- shading phase over lists[which] (the entries with a bounce ray: its hit record lies at hit[p]); the scattered paths' entries
   // go to lists[which ^ 1]
   auto shade_phase = [&](uint32_t which) {
      const uint32_t count = s_count[which];
      const uint32_t* __restrict__ cur_list = lists[which];
      uint32_t* nxt_list = lists[which ^ 1];
      uint32_t(*list)[128] = u.s.list[wave];
      uint32_t(*missed)[128] = u.s.miss[wave];
      uint32_t n_list = 0, n_missed = carry_n, n_rays = 0;  // wave-uniform
      if (lane < carry_n) {  // the misses the last shading phase left (fewer than 64: the sky integral runs on full waves)
         missed[0][lane] = carry_pos;
         missed[1][lane] = carry_id;
      }
      __builtin_amdgcn_wave_barrier();
      auto shade = [&](uint32_t id, uint32_t pl, float4 hr, bool valid) {  // pl: the path's position | the bounces left behind this ray
         bool scattered = false, want_light = false, to_tree = false, keep = false;
         uint32_t flags = 0;
         const uint32_t p = pl & kPosMask;
         if (valid) {
            const uint32_t left = (pl & kLeftMask) >> kLeftShift;
            const bool last = left == 0u;
            const uint32_t pk = __float_as_uint(hr.w);
            const float4 ro = ld_rec(rec_quad(rec, seg + p, REC_ORIGIN)), rd = ld_rec(rec_quad(rec, seg + p, REC_DIR)), thr4 = ld_rec(rec_quad(rec, seg + p, REC_THR)),
                         rad4 = ld_rec(rec_quad(rec, seg + p, REC_RAD));
            uint2 rng = make_uint2(__float_as_uint(ro.w), __float_as_uint(rd.w));
            const V3 ray_dir = v3(rd.x, rd.y, rd.z);
            const float4* sp = sc.shade + 4 * (size_t)pk;
            const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
            const uint32_t mesh_index = __float_as_uint(s3.w);
            MeshShade ms = lds_fetch(s_mesh + (mesh_index < kLdsMeshes ? mesh_index : kLdsMeshes - 1));  // rchit:22-23
            if (mesh_index >= n_lds_mesh) ms = sc.meshes[mesh_index];
            V3 normal;
            float uu, vv;
            surface_normal_uv(s0, s1, s2, s3, hr.y, hr.z, normal, uu, vv);                // rchit:30-31, :39
            const V3 world_normal = world_normal_of(ms, normal, ray_dir);                 // rchit:32-37
            V3 origin = v3(ro.x, ro.y, ro.z) + hr.x * ray_dir;                            // rgen:59
            origin = offset_ray(origin, world_normal);                                    // rgen:60
            scattered = path_scatters(ms, ray_dir, world_normal);
            V3 color = sample_texture(sc, s_lut, ms.diffuse_map, uu, vv, s_tex, n_lds_tex);  // rchit:40
            color = color * v3(ms.base_color[0], ms.base_color[1], ms.base_color[2]);    // rchit:41
            uint32_t seed = rng.y;
            const V3 scatter = material_scatter(ms, ray_dir, world_normal, color, seed);  // rchit:47-89
            rng.y = seed;                                                                 // rchit:91
            const V3 thr = v3(thr4.x, thr4.y, thr4.z) * color;                            // rgen:48
            V3 rad = v3(rad4.x, rad4.y, rad4.z);
            if (!scattered) {                                                             // rgen:53-57: the path ends here
               st_stream(ps.radf + id, make_float4(rad.x + thr.x, rad.y + thr.y, rad.z + thr.z, __uint_as_float(rng.x)));
            } else {
               float lf = 0.0f;
               int light_index = 0;
               if (fp.lights_enabled == 1) want_light = select_light(fp, sc, id, rng.x, origin, lf, light_index);  // rgen:81-121
               if (fp.sun_shadow_enabled == 1) {                                          // rgen:63-79
                  int r = use_grid ? sun_grid_query<COUNT, INLINE>(g, tris, origin, sun_d, n_stris, n_covered) : 2;
                  if (COUNT && use_grid) n_snodes++;  // every sun ray looked one cell up
                  if (r == 0) rad = v3(rad.x + thr.x, rad.y + thr.y, rad.z + thr.z);       // rgen:69-78
                  if (r == 2) flags |= kSun;
                  to_tree = r == 2 && use_grid;
               }
               if (want_light) flags |= kLight;
               if (!last) flags |= kHasRay | ((left - 1u) << kLeftShift);
               keep = (flags & (kHasRay | kSun | kLight)) != 0u;
               if (!keep) {  // behind the last bounce with no shadow ray out: rgen:127 (what k_flush_survivors writes)
                  st_stream(ps.radf + id, make_float4(rad.x, rad.y, rad.z, __uint_as_float(rng.x)));
               } else {  // the path's state, in place
                  st_rec(rec_quad(rec, seg + p, REC_ORIGIN), make_float4(origin.x, origin.y, origin.z, __uint_as_float(rng.x)));
                  st_rec(rec_quad(rec, seg + p, REC_DIR), make_float4(scatter.x, scatter.y, scatter.z, __uint_as_float(rng.y)));  // rgen:61
                  st_rec(rec_quad(rec, seg + p, REC_THR), make_float4(thr.x, thr.y, thr.z, lf));
                  st_rec(rec_quad(rec, seg + p, REC_RAD), make_float4(rad.x, rad.y, rad.z, __uint_as_float((uint32_t)light_index)));
               }
            }
         }
         const uint32_t slot = wave_append(&s_count[which ^ 1], keep);
         if (keep) nxt_list[slot] = p | flags;
         if (fp.sun_shadow_enabled == 1) w_sun += (uint32_t)__popcll(__ballot(scattered));
         w_sun_tree += (uint32_t)__popcll(__ballot(to_tree));
         w_light += (uint32_t)__popcll(__ballot(want_light));
      };
      // reference.rmiss for `n` paths from the front of the wave's miss list: their state is where shade_miss_path reads it (their
      // records of set 1, untouched since their ray was made)
      auto flush_misses = [&](uint32_t n) {
         if (lane < n) shade_miss_path(fp, ps, seg + missed[0][lane], missed[1][lane], kFirst);
         w_miss += n;
      };
      auto entry_of = [&](uint32_t k) { return make_float4(__uint_as_float(list[2][k]), __uint_as_float(list[3][k]), __uint_as_float(list[4][k]), __uint_as_float(list[5][k])); };
      const uint32_t rounds = (count + kBlock - 1) / kBlock;
      for (uint32_t r = 0; r < rounds; r++) {
         const uint32_t i = r * kBlock + threadIdx.x;
         uint32_t id = 0, p = 0;
         float4 hr = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kEmptyRef));
         bool has_ray = false;
         if (i < count) {
            const uint32_t e = cur_list[i];
            has_ray = (e & kHasRay) != 0u;  // (an entry without one was a shadow ray behind the last bounce: the trace phase finished it)
            p = e & (kPosMask | kLeftMask);
            if (has_ray) {
               id = ld_stream(ids + (e & kPosMask));
               hr = ld_rec(ps.hit + seg + (e & kPosMask));
            }
         }
         const bool is_hit = __float_as_uint(hr.w) != kEmptyRef;
         const unsigned long long mask = __ballot(is_hit);
         const unsigned long long mmask = __ballot(has_ray && !is_hit);
         n_rays += (uint32_t)__popcll(__ballot(has_ray));
         if (mmask) {
            // reference.rmiss for the paths whose ray left the scene, 64 at a time
            const uint32_t mp = __builtin_amdgcn_mbcnt_hi((uint32_t)(mmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mmask, 0u));
            if (has_ray && !is_hit) {
               missed[0][n_missed + mp] = p & kPosMask;
               missed[1][n_missed + mp] = id;
            }
            n_missed += (uint32_t)__popcll(mmask);
            __builtin_amdgcn_wave_barrier();
            if (n_missed >= 64u) {
               flush_misses(64u);
               const uint32_t rest = n_missed - 64u;
               uint32_t tmp0 = 0, tmp1 = 0;
               if (lane < rest) {
                  tmp0 = missed[0][64u + lane];
                  tmp1 = missed[1][64u + lane];
               }
               __builtin_amdgcn_wave_barrier();
               if (lane < rest) {
                  missed[0][lane] = tmp0;
                  missed[1][lane] = tmp1;
               }
               __builtin_amdgcn_wave_barrier();
               n_missed = rest;
            }
         }
         if (mask == 0ull) continue;
         const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
         if (is_hit) {
            list[0][n_list + prefix] = id;
            list[1][n_list + prefix] = p;
            list[2][n_list + prefix] = __float_as_uint(hr.x);
            list[3][n_list + prefix] = __float_as_uint(hr.y);
            list[4][n_list + prefix] = __float_as_uint(hr.z);
            list[5][n_list + prefix] = __float_as_uint(hr.w);
         }
         n_list += (uint32_t)__popcll(mask);
         __builtin_amdgcn_wave_barrier();
         if (n_list >= 64u) {
            shade(list[0][lane], list[1][lane], entry_of(lane), true);
            w_hits += 64u;
            const uint32_t rest = n_list - 64u;
            uint32_t tmp[6] = {0, 0, 0, 0, 0, 0};
            if (lane < rest)
               for (int k = 0; k < 6; k++) tmp[k] = list[k][64u + lane];
            __builtin_amdgcn_wave_barrier();
            if (lane < rest)
               for (int k = 0; k < 6; k++) list[k][lane] = tmp[k];
            __builtin_amdgcn_wave_barrier();
            n_list = rest;
         }
      }
      if (n_list) {
         shade(lane < n_list ? list[0][lane] : 0u, lane < n_list ? list[1][lane] : 0u, lane < n_list ? entry_of(lane) : make_float4(0.0f, 0.0f, 0.0f, 0.0f), lane < n_list);
         w_hits += n_list;
      }
      // the rest waits in registers for the next shading phase (the list's LDS is the trace phase's stacks)
      carry_n = n_missed;
      if (lane < n_missed) {
         carry_pos = missed[0][lane];
         carry_id = missed[1][lane];
      }
      w_rays += n_rays;
   };

   uint32_t which = 0;
#ifdef UH_FUSED_PROFILE  // (measurement only, with count_visits: the waves' clock in the trace phases / at the barriers / in the shading phases, in the light counters)
   unsigned long long c_trace = 0, c_wait = 0, c_shade = 0;
#define UH_TICK(acc)                                  \
   {                                                  \
      const unsigned long long now = wall_clock64();  \
      acc += now - c_last;                            \
      c_last = now;                                   \
   }
   unsigned long long c_last = wall_clock64();
#else
#define UH_TICK(acc)
#endif
   // phases until a shading phase leaves no entry (every phase takes its entries one bounce on: at most num_bounces + 1 rounds)
   for (uint32_t round_no = 0; round_no < kMaxBounces + 2u; round_no++) {
      if (s_count[which] == 0u && s_count[which ^ 1] == 0u) break;  // (block-uniform: read behind a barrier)
      trace_phase(which);
      UH_TICK(c_trace)
      __syncthreads();  // (workgroup-scope release / acquire: the hit records and radiance the block's waves wrote are visible to all of them)
      UH_TICK(c_wait)
      shade_phase(which);
      UH_TICK(c_shade)
      __syncthreads();
      UH_TICK(c_wait)
      if (threadIdx.x == 0) {
         s_count[which] = 0;
         s_cursor = 0;
      }
      which ^= 1;
      __syncthreads();
   }
   if (carry_n) {  // the last misses: reference.rmiss on a partial wave, once
      if (lane < carry_n) shade_miss_path(fp, ps, seg + carry_pos, carry_id, kFirst);
      w_miss += carry_n;
   }
   if (lane == 0) {
      if (w_rays) atomicAdd(&stats->rays[UH_RAY_BOUNCE], (unsigned long long)w_rays);
      if (w_hits) atomicAdd(&stats->closest_hits, (unsigned long long)w_hits);
      if (w_sun) atomicAdd(&stats->rays[UH_RAY_SUN_SHADOW], (unsigned long long)w_sun);
      if (w_sun_tree) atomicAdd(&stats->sun_tree_rays, (unsigned long long)w_sun_tree);
      if (w_light) atomicAdd(&stats->rays[UH_RAY_LIGHT_SHADOW], (unsigned long long)w_light);
      if (w_miss) atomicAdd(&stats->misses, (unsigned long long)w_miss);
   }
   if (COUNT) {
      atomicAdd(&stats->nodes_visited, (unsigned long long)n_nodes);
      atomicAdd(&stats->tris_tested, (unsigned long long)n_tris);
      atomicAdd(&stats->shadow_nodes_visited, (unsigned long long)n_snodes);
      atomicAdd(&stats->shadow_tris_tested, (unsigned long long)n_stris);
      atomicAdd(&stats->light_nodes_visited, (unsigned long long)n_lnodes);
      atomicAdd(&stats->light_tris_tested, (unsigned long long)n_ltris);
      atomicAdd(&stats->sun_covered_rays, (unsigned long long)n_covered);
#ifdef UH_FUSED_PROFILE
      if (lane == 0) {
         atomicAdd(&stats->light_nodes_visited, c_trace);
         atomicAdd(&stats->light_tris_tested, c_wait);
         atomicAdd(&stats->sun_covered_rays, c_shade);
      }
#endif
   }
#undef UH_TICK
}

// ------------------------------------------------------------------------------------------
// finish_sample — reference.rgen:127 (pixelColor += radiance) and, after the last sample of the
// frame, rgen:130-144 (accumulate, sRGB, store both images)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uchar4 resolve_color(float4 acc, uint32_t total_samples, uint32_t limit) {
   float denom = (float)min(total_samples, limit);
   V3 c = v3(acc.x / denom, acc.y / denom, acc.z / denom);                               // rgen:140
   c = v3(linear_to_srgb(c.x), linear_to_srgb(c.y), linear_to_srgb(c.z));               // rgen:141
   return make_uchar4((unsigned char)unorm8(c.z), (unsigned char)unorm8(c.y), (unsigned char)unorm8(c.x), 0);  // B8G8R8A8, alpha 0
}

__global__ __launch_bounds__(kBlock) void k_finish_sample(FrameParams fp, PathState ps, Images im, uint32_t sample, bool last) {
   for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < fp.n_owned; k += gridDim.x * kBlock) {
      const uint32_t pix = fp.owned_pixels ? fp.owned_pixels[k] : k;
      float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      bool acc_loaded = false;
      uint32_t total = fp.total_samples;
      // frames of the batch in order: each applies the reference's accumulate tail with ITS total_samples
      for (uint32_t f = 0; f < fp.batch_frames; f++) {
         const uint32_t id = f * fp.n_owned + k;
         float4 pc = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rad = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
         if (sample != 0) pc = ps.pixcol[id];
         if (fp.num_bounces != 0) rad = ps.radf[id];  // where the path ended (miss, absorption, or the flush after the last bounce); with zero bounces no kernel ever wrote one
         pc = make_float4(pc.x + rad.x, pc.y + rad.y, pc.z + rad.z, 0.0f);                  // rgen:127
         if (!last) {
            ps.pixcol[id] = pc;
            continue;
         }
         total = fp.total_samples_of[f];
         if (total != fp.samples_per_frame) {                                               // rgen:131-134
            if (!acc_loaded) acc = im.accumulation[pix];
         } else {
            acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
         }
         acc_loaded = true;
         if (total <= fp.accumulation_limit) acc = make_float4(acc.x + pc.x, acc.y + pc.y, acc.z + pc.z, 0.0f);  // rgen:136-138
         acc.w = 0.0f;
      }
      if (!last) continue;
      im.accumulation[pix] = acc;                                                            // rgen:143
      im.output[pix] = resolve_color(acc, total, fp.accumulation_limit);                     // rgen:144
   }
}

__global__ __launch_bounds__(kBlock) void k_resolve(Images im, uint32_t n, uint32_t total_samples, uint32_t limit) {
   for (uint32_t id = blockIdx.x * kBlock + threadIdx.x; id < n; id += gridDim.x * kBlock)
      im.output[id] = resolve_color(im.accumulation[id], total_samples, limit);
}

// ------------------------------------------------------------------------------------------
// G-buffer position by primary-ray cast (replaces the raster gbuffer_pass for this input only:
// utopian/src/renderers/gbuffer.rs:11-52, shaders/gbuffer/gbuffer.frag:47; clear colour
// (1,1,1,0): utopian/src/pass.rs:210-214)
// ------------------------------------------------------------------------------------------
// ray j of the cast is the primary ray of pixel spans.pixel_of(j): the rays of a rank's rows lie densely in the ray arrays
__global__ __launch_bounds__(kBlock) void k_gbuffer_generate(FrameParams fp, RawRays ps, RowSpans spans) {
   const uint32_t n = spans.total();
   for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
      const uint32_t id = spans.pixel_of(j);
      V3 o, d;
      primary_ray(fp, id % fp.W, id / fp.W, 0.5f, 0.5f, o, d);
      ps.ray_o[j] = make_float4(o.x, o.y, o.z, 0.001f);
      ps.ray_d[j] = make_float4(d.x, d.y, d.z, 10000.0f);
   }
}
__global__ __launch_bounds__(kBlock) void k_gbuffer_resolve(FrameParams fp, RawRays ps, Images im, DeviceStats* stats, RowSpans spans, uint32_t counted) {
   const uint32_t n = spans.total();
   for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
      const uint32_t id = spans.pixel_of(j);
      float4 h = ps.hit[j];
      if (__float_as_uint(h.w) != kEmptyRef) {
         float4 ro = ps.ray_o[j], rd = ps.ray_d[j];
         V3 p = v3(ro.x, ro.y, ro.z) + h.x * v3(rd.x, rd.y, rd.z);
         im.gbuffer_pos[id] = make_float4(p.x, p.y, p.z, 1.0f);
      } else {
         im.gbuffer_pos[id] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
      }
   }
   if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats->rays[UH_RAY_GBUFFER], (unsigned long long)counted);
}

// texture(in_gbuffer_position, vec2(px) / vec2(size)) through the LINEAR + MIRRORED_REPEAT sampler
// (restir/initial_ris.rgen:22-23): the texel corner, i.e. the mean of the 2x2 texels up-left
__device__ __forceinline__ V3 gbuffer_fetch(const float4* __restrict__ g, uint32_t W, uint32_t px, uint32_t py) {
   uint32_t x0 = px == 0 ? 0 : px - 1, y0 = py == 0 ? 0 : py - 1;
   float4 a = g[(size_t)y0 * W + x0], b = g[(size_t)y0 * W + px], c = g[(size_t)py * W + x0], d = g[(size_t)py * W + px];
   return ((xyz(a) + xyz(b)) + (xyz(c) + xyz(d))) * 0.25f;
}

// ------------------------------------------------------------------------------------------
// ReSTIR passes. The light table (pos + intensity, 32 B/light, <= 32 KiB) is staged in LDS.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kMaxLdsLights = UH_MAX_GPU_LIGHTS;

__device__ __forceinline__ void stage_lights(float4* s_lights, const SceneDev& sc) {
   for (uint32_t i = threadIdx.x; i < 2 * sc.num_lights; i += blockDim.x) s_lights[i] = sc.lights[i];
   __syncthreads();
}

// restir/reset_reservoirs.comp:24-45
__global__ __launch_bounds__(kBlock) void k_reset_reservoirs(Images im, RowSpans spans) {
   const uint32_t n = spans.total();
   for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
      const uint32_t id = spans.pixel_of(j);
      UhReservoir z = {-1, 0.0f, 0.0f, 0};
      im.reservoirs[0][id] = z;
      im.reservoirs[1][id] = z;
   }
}

// restir/initial_ris.rgen:19-39 + restir_sampling.glsl:96-131 (resample, 32 candidates)
__global__ __launch_bounds__(kBlock) void k_initial_ris(FrameParams fp, SceneDev sc, Images im, RowSpans spans) {
   __shared__ float4 s_lights[2 * kMaxLdsLights];
   stage_lights(s_lights, sc);
   const uint32_t work = spans.total();
   for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < work; j += gridDim.x * kBlock) {
      const uint32_t id = spans.pixel_of(j);
      uint32_t px = id % fp.W, py = id / fp.W;
      uint32_t rng = init_rng(px, py, fp.W, fp.frame_number);
      V3 hit_position = gbuffer_fetch(im.gbuffer_pos, fp.W, px, py);
      UhReservoir r = {-1, 0.0f, 0.0f, 0};
      for (int i = 0; i < 32; i++) {
         int cand;
         float p;
         sample_light_uniform(fp.num_lights_used, rng, cand, p);
         float m_i = 1.0f / 32.0f;
         float p_hat = target_function(s_lights, sc.num_lights, cand, hit_position);
         float W_Xi = 1.0f / p;
         float w_i = m_i * p_hat * W_Xi;
         update_reservoir(rng, r, cand, w_i, 1);
      }
      r.M = 1;
      if (r.Y != -1) finalize_resampling(r, target_function(s_lights, sc.num_lights, r.Y, hit_position));
      UhReservoir nr = {-1, 0.0f, 0.0f, 0};
      update_reservoir(rng, nr, r.Y, r.W_sum * (float)r.M, r.M);
      finalize_resampling(nr, target_function(s_lights, sc.num_lights, nr.Y, hit_position));
      im.reservoirs[0][id] = nr;
   }
}

// restir/temporal_reuse.rgen:35-119
__global__ __launch_bounds__(kBlock) void k_temporal_reuse(FrameParams fp, SceneDev sc, Images im, RowSpans spans) {
   __shared__ float4 s_lights[2 * kMaxLdsLights];
   stage_lights(s_lights, sc);
   const uint32_t n = fp.W * fp.H, work = spans.total();
   for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < work; j += gridDim.x * kBlock) {
      const uint32_t id = spans.pixel_of(j);
      if (fp.temporal_enabled == 0) {
         im.reservoirs[1][id] = im.reservoirs[0][id];
         continue;
      }
      uint32_t px = id % fp.W, py = id / fp.W;
      uint32_t rng = init_rng(px, py, fp.W, fp.frame_number);
      V3 hit_position = gbuffer_fetch(im.gbuffer_pos, fp.W, px, py);
      UhReservoir nr = {-1, 0.0f, 0.0f, 0};
      UhReservoir ir = im.reservoirs[0][id];
      float p_hat = target_function(s_lights, sc.num_lights, ir.Y, hit_position);
      update_reservoir(rng, nr, ir.Y, p_hat * ir.W_X * (float)ir.M, ir.M);
      UhReservoir pr = {-1, 0.0f, 0.0f, 0};
      float4 puv = mat4_mul(fp.prev_pv, hit_position.x, hit_position.y, hit_position.z, 1.0f);
      float ux = puv.x / puv.w, uy = puv.y / puv.w;
      ux = ux * 0.5f + 0.5f;
      uy = uy * 0.5f + 0.5f;
      uy = 1.0f - uy;
      if (ux >= 0.0f && ux <= 1.0f && uy >= 0.0f && uy <= 1.0f) {
         int ix = (int)(ux * (float)fp.W + 0.5f), iy = (int)(uy * (float)fp.H + 0.5f);
         uint32_t ti = (uint32_t)iy * fp.W + (uint32_t)ix;  // may be one past the end in the reference (y == H)
         if (ti > n - 1) ti = n - 1;
         pr = im.prev_spatial[ti];  // last frame's spatial_reuse_reservoirs (renderers/mod.rs:294)
      }
      p_hat = pr.Y == -1 ? 0.0f : target_function(s_lights, sc.num_lights, pr.Y, hit_position);
      pr.M = min(20 * ir.M, pr.M);
      update_reservoir(rng, nr, pr.Y, p_hat * pr.W_X * (float)pr.M, pr.M);
      if (nr.Y != -1) finalize_resampling(nr, target_function(s_lights, sc.num_lights, nr.Y, hit_position));
      im.reservoirs[1][id] = nr;
   }
}

// restir/spatial_reuse.rgen:23-73
__global__ __launch_bounds__(kBlock) void k_spatial_reuse(FrameParams fp, SceneDev sc, Images im, RowSpans spans) {
   __shared__ float4 s_lights[2 * kMaxLdsLights];
   stage_lights(s_lights, sc);
   const uint32_t work = spans.total();
   const UhReservoir* __restrict__ temporal = im.reservoirs[1];
   for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < work; j += gridDim.x * kBlock) {
      const uint32_t id = spans.pixel_of(j);
      if (fp.spatial_enabled == 0) {
         im.reservoirs[2][id] = temporal[id];
         continue;
      }
      uint32_t px = id % fp.W, py = id / fp.W;
      uint32_t rng = init_rng(px, py, fp.W, fp.frame_number);
      V3 hit_position = gbuffer_fetch(im.gbuffer_pos, fp.W, px, py);
      UhReservoir nr = {-1, 0.0f, 0.0f, 0};
      UhReservoir tr = temporal[id];
      float p_hat = target_function(s_lights, sc.num_lights, tr.Y, hit_position);
      update_reservoir(rng, nr, tr.Y, p_hat * tr.W_X * (float)tr.M, tr.M);
      for (int i = 0; i < 5; i++) {
         float ox = random_float(rng) * 2.0f - 1.0f, oy = random_float(rng) * 2.0f - 1.0f;
         ox *= 30.0f;
         oy *= 30.0f;
         // uvec2(offset) of a negative float: pinned as (uint)(int)trunc(x); clamp(uvec2) then
         // sends a wrapped-negative coordinate to size-1
         uint32_t nx = px + (uint32_t)(int)ox, ny = py + (uint32_t)(int)oy;
         nx = min(nx, fp.W - 1);
         ny = min(ny, fp.H - 1);
         UhReservoir nb = temporal[(size_t)ny * fp.W + nx];
         float ph = target_function(s_lights, sc.num_lights, nb.Y, hit_position);
         update_reservoir(rng, nr, nb.Y, ph * nb.W_X * (float)nb.M, nb.M);
      }
      if (nr.Y != -1) finalize_resampling(nr, target_function(s_lights, sc.num_lights, nr.Y, hit_position));
      im.reservoirs[2][id] = nr;
   }
}

// ------------------------------------------------------------------------------------------
// multi-GPU tile pack / unpack of the RGBA32F accumulation image
// ------------------------------------------------------------------------------------------
template <bool PACK>
__global__ __launch_bounds__(kBlock) void k_tiles(float4* acc, float4* packed, uint32_t W, uint32_t H, uint32_t rank, uint32_t world, uint32_t tile) {
   const uint32_t tiles_x = (W + tile - 1) / tile, tiles_y = (H + tile - 1) / tile;
   const uint32_t num_tiles = tiles_x * tiles_y;
   const uint32_t owned = num_tiles > rank ? (num_tiles - rank + world - 1) / world : 0;
   const uint64_t total = (uint64_t)owned * tile * tile;
   for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock) {
      uint32_t local = (uint32_t)(i / (tile * tile)), within = (uint32_t)(i % (tile * tile));
      uint32_t t = rank + local * world;
      uint32_t x = (t % tiles_x) * tile + within % tile, y = (t / tiles_x) * tile + within / tile;
      bool inside = x < W && y < H;
      if (PACK)
         packed[i] = inside ? acc[(size_t)y * W + x] : make_float4(0, 0, 0, 0);
      else if (inside)
         acc[(size_t)y * W + x] = packed[i];
   }
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
static inline dim3 stream_grid(const LaunchCfg& c, uint32_t n) {
   uint32_t blocks = (n + kBlock - 1) / kBlock;
   uint32_t cap = c.num_cus * 8;
   return dim3(blocks < cap ? (blocks ? blocks : 1) : cap);
}
// grids of sharded kernels are whole multiples of kShards (blockIdx % kShards = shard)
static inline dim3 sharded_grid(uint32_t blocks) {
   uint32_t g = (blocks / kShards) * kShards;
   return dim3(g < kShards ? kShards : g);
}
static inline dim3 closest_grid(const LaunchCfg& c) { return sharded_grid(c.num_cus * c.closest_blocks_per_cu); }
static inline dim3 shadow_grid(const LaunchCfg& c) { return sharded_grid(c.num_cus * c.shadow_blocks_per_cu); }
static inline dim3 shade_grid(const LaunchCfg& c, uint32_t n) {
   uint32_t blocks = (n + kBlock - 1) / kBlock, cap = c.num_cus * 8;
   return sharded_grid(blocks < cap ? blocks : cap);
}

uint32_t query_trace_occupancy() {
   int a = 0, b = 0;
   if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_trace_closest<false>, kBlock, 0) != hipSuccess) a = 4;
   if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_trace_shadow<false, false>, kBlock, 0) != hipSuccess) b = 4;
   int m = a < b ? a : b;
   if (m < 1) m = 1;
   if (m > 8) m = 8;
   return (uint32_t)m;
}

void launch_generate(const LaunchCfg& c, const FrameParams& fp, const PathState& ps, Control* ctl, uint32_t sample) {
   k_generate<<<stream_grid(c, fp.n_owned * fp.batch_frames), kBlock, 0, c.stream>>>(fp, ps, ctl, sample);
}

// closest-hit traversal over a sharded queue of path ids (sharded) or over n raw rays
static void launch_closest(const LaunchCfg& c, dim3 grid, const SceneDev& sc, bool sharded, const float4* ray_o, const float4* ray_d, float4* hit,
                           uint32_t shard_cap, Control* ctl, DeviceStats* stats, uint32_t bounce, uint32_t cursor_slot, int ray_kind, uint32_t n,
                           const uint32_t* listed = nullptr) {
   if (c.count_visits && sharded)
      k_trace_closest<true><<<grid, kBlock, 0, c.stream>>>(sc, sharded, ray_o, ray_d, hit, shard_cap, ctl, stats, bounce, cursor_slot, ray_kind, n, listed);
   else
      k_trace_closest<false><<<grid, kBlock, 0, c.stream>>>(sc, sharded, ray_o, ray_d, hit, shard_cap, ctl, stats, bounce, cursor_slot, ray_kind, n, listed);
}

void launch_trace_closest(const LaunchCfg& c, const SceneDev& sc, const PathState& ps, Control* ctl, DeviceStats* stats, uint32_t bounce,
                          uint32_t cursor_slot, int ray_kind) {
   const PathRecs& rec = ps.set[bounce & 1];
   launch_closest(c, closest_grid(c), sc, true, rec_quad(rec, 0, REC_ORIGIN), rec_quad(rec, 0, REC_DIR), ps.hit, ps.shard_cap, ctl, stats, bounce, cursor_slot, ray_kind, 0);
}

// bounce 0 of the path tracer through the camera grid, then the tree walk for the rays of the pixels with long lists
void launch_trace_camera_grid(const LaunchCfg& c, const FrameParams& fp, const SceneDev& sc, const PathState& ps, Control* ctl, DeviceStats* stats, uint32_t cursor_slot_grid,
                              uint32_t cursor_slot_tree, const SunGridDev& g, bool leftovers_possible) {
   const dim3 grid = sharded_grid(c.num_cus * 8);
   if (c.count_visits)
      k_trace_camera_grid<true><<<grid, kBlock, 0, c.stream>>>(sc, fp, ps, ctl, stats, cursor_slot_grid, g);
   else
      k_trace_camera_grid<false><<<grid, kBlock, 0, c.stream>>>(sc, fp, ps, ctl, stats, cursor_slot_grid, g);
   if (!leftovers_possible) return;  // no pixel lists more than the grid kernel walks itself
   const PathRecs& rec = ps.set[0];
   launch_closest(c, closest_grid(c), sc, true, rec_quad(rec, 0, REC_ORIGIN), rec_quad(rec, 0, REC_DIR), ps.hit, ps.shard_cap, ctl, stats, 0, cursor_slot_tree, UH_RAY_PRIMARY, 0, ps.queue[3]);
}

void launch_shade_miss(const LaunchCfg& c, const FrameParams& fp, const PathState& ps, Control* ctl, DeviceStats* stats, uint32_t bounce) {
   // the bounce's misses are a dense queue (Q_MISS, written by k_shade_hit): every lane shades one; the count is on the
   // device, so the grid is sized for the paths the wavefront started with
   const dim3 full = shade_grid(c, fp.n_owned * fp.batch_frames), lean = sharded_grid(c.num_cus * 8);  // blocks per CU 2 / 4 / 6 / 8: 7,599 / 7,613-7,656 / 7,699 / 7,676-7,678 Mrays/s (round 4)
   k_shade_miss<<<full.x < lean.x ? full : lean, kBlock, 0, c.stream>>>(fp, ps, ctl, stats, bounce);
}

void launch_shade_hit(const LaunchCfg& c, const FrameParams& fp, const SceneDev& sc, const PathState& ps, const Images& im, Control* ctl,
                      DeviceStats* stats, uint32_t bounce) {
   k_shade_hit<<<shade_grid(c, fp.W * fp.H), kBlock, 0, c.stream>>>(fp, sc, ps, ctl, stats, bounce);
}
void launch_flush_survivors(const LaunchCfg& c, const FrameParams& fp, const PathState& ps, Control* ctl) {
   k_flush_survivors<<<shade_grid(c, fp.n_owned * fp.batch_frames), kBlock, 0, c.stream>>>(fp, ps, ctl);
}

// bounces 1 .. of a lone frame in one persistent kernel (k_path_fused); g: the sun grid when use_grid
void launch_path_fused(const LaunchCfg& c, const FrameParams& fp, const SceneDev& sc, const PathState& ps, Control* ctl, DeviceStats* stats, const SunGridDev& g,
                       bool use_grid, bool sun_of_bounce0) {
   const dim3 grid = sharded_grid(c.num_cus * c.fused_blocks_per_cu);
#define UH_FUSED(COUNT, INLINE) k_path_fused<COUNT, INLINE><<<grid, kBlock, 0, c.stream>>>(sc, fp, ps, ctl, stats, g, use_grid, sun_of_bounce0, UH_FUSED_STAGGER ? c.num_cus : 0u)
   if (use_grid && g.recs) {
      if (c.count_visits) UH_FUSED(true, true);
      else UH_FUSED(false, true);
   } else {
      if (c.count_visits) UH_FUSED(true, false);
      else UH_FUSED(false, false);
   }
#undef UH_FUSED
}

void launch_trace_shadow(const LaunchCfg& c, const FrameParams& fp, const SceneDev& sc, const PathState& ps, Control* ctl, DeviceStats* stats,
                         uint32_t bounce, uint32_t cursor_slot, bool light, bool sun_leftovers) {
   // sun_leftovers: what the sun grid handed to the tree (queue 3)
#define UH_SHADOW(COUNT, LIGHT) k_trace_shadow<COUNT, LIGHT><<<shadow_grid(c), kBlock, 0, c.stream>>>(sc, fp, ps, ctl, stats, bounce, cursor_slot, sun_leftovers)
   if (light) {
      if (c.count_visits) UH_SHADOW(true, true);
      else UH_SHADOW(false, true);
   } else {
      if (c.count_visits) UH_SHADOW(true, false);
      else UH_SHADOW(false, false);
   }
#undef UH_SHADOW
}

void launch_trace_sun_grid(const LaunchCfg& c, const FrameParams& fp, const SceneDev& sc, const PathState& ps, Control* ctl, DeviceStats* stats, uint32_t bounce,
                           uint32_t cursor_slot, const SunGridDev& g) {
   const dim3 grid = sharded_grid(c.num_cus * 8);
#define UH_SUN_GRID(COUNT, INLINE) k_trace_sun_grid<COUNT, INLINE><<<grid, kBlock, 0, c.stream>>>(sc, fp, ps, ctl, stats, bounce, cursor_slot, g)
   if (g.recs) {
      if (c.count_visits) UH_SUN_GRID(true, true);
      else UH_SUN_GRID(false, true);
   } else {
      if (c.count_visits) UH_SUN_GRID(true, false);
      else UH_SUN_GRID(false, false);
   }
#undef UH_SUN_GRID
}

void launch_finish_sample(const LaunchCfg& c, const FrameParams& fp, const PathState& ps, const Images& im, uint32_t sample, bool last) {
   k_finish_sample<<<stream_grid(c, fp.n_owned), kBlock, 0, c.stream>>>(fp, ps, im, sample, last);
}

void launch_resolve(const LaunchCfg& c, const Images& im, uint32_t W, uint32_t H, uint32_t total_samples, uint32_t limit) {
   k_resolve<<<stream_grid(c, W * H), kBlock, 0, c.stream>>>(im, W * H, total_samples, limit);
}

void launch_gbuffer(const LaunchCfg& c, const FrameParams& fp, const SceneDev& sc, const RawRays& ps, const Images& im, DeviceStats* stats, const RowSpans& spans, uint32_t counted,
                    const SunGridDev* camera_grid) {
   const uint32_t n = spans.total();
   if (n == 0) return;
   k_gbuffer_generate<<<stream_grid(c, n), kBlock, 0, c.stream>>>(fp, ps, spans);
   if (camera_grid)
      k_gbuffer_camera_grid<<<stream_grid(c, n), kBlock, 0, c.stream>>>(sc, ps, spans, fp.W, *camera_grid);
   else {
      const uint32_t full = c.num_cus * c.closest_blocks_per_cu, need = (n + kBlock - 1) / kBlock;
      launch_closest(c, dim3(need < full ? need : full), sc, false, ps.ray_o, ps.ray_d, ps.hit, 0, nullptr, stats, 0, 0, 0, n);
   }
   k_gbuffer_resolve<<<stream_grid(c, n), kBlock, 0, c.stream>>>(fp, ps, im, stats, spans, counted);
}

// the reservoir kernels stage the light table per block: no more blocks than the rows at hand can feed
static inline dim3 reservoir_grid(const LaunchCfg& c, const RowSpans& spans) {
   const uint32_t full = c.num_cus * 4, need = (spans.total() + kBlock - 1) / kBlock;
   return dim3(need < full ? (need ? need : 1) : full);
}
void launch_reset_reservoirs(const LaunchCfg& c, const FrameParams& fp, const Images& im, const RowSpans& spans) {
   if (spans.total()) k_reset_reservoirs<<<stream_grid(c, spans.total()), kBlock, 0, c.stream>>>(im, spans);
}
void launch_initial_ris(const LaunchCfg& c, const FrameParams& fp, const SceneDev& sc, const Images& im, const RowSpans& spans) {
   if (spans.total()) k_initial_ris<<<reservoir_grid(c, spans), kBlock, 0, c.stream>>>(fp, sc, im, spans);
}
void launch_temporal_reuse(const LaunchCfg& c, const FrameParams& fp, const SceneDev& sc, const Images& im, const RowSpans& spans) {
   if (spans.total()) k_temporal_reuse<<<reservoir_grid(c, spans), kBlock, 0, c.stream>>>(fp, sc, im, spans);
}
void launch_spatial_reuse(const LaunchCfg& c, const FrameParams& fp, const SceneDev& sc, const Images& im, const RowSpans& spans) {
   if (spans.total()) k_spatial_reuse<<<reservoir_grid(c, spans), kBlock, 0, c.stream>>>(fp, sc, im, spans);
}

void launch_trace_closest_raw(const LaunchCfg& c, const SceneDev& sc, const float4* ray_o, const float4* ray_d, float4* hit, uint32_t n) {
   // a grid that fills the chip once; smaller queries get one block per 256 rays
   const uint32_t full = c.num_cus * c.closest_blocks_per_cu, need = (n + kBlock - 1) / kBlock;
   launch_closest(c, dim3(need < full ? (need ? need : 1) : full), sc, false, ray_o, ray_d, hit, 0, nullptr, nullptr, 0, 0, 0, n);
}
void launch_trace_any_raw(const LaunchCfg& c, const SceneDev& sc, const float4* ray_o, const float4* ray_d, uint32_t* occluded, uint32_t n) {
   k_trace_any_raw<<<stream_grid(c, n), kBlock, 0, c.stream>>>(sc, ray_o, ray_d, occluded, n);
}

// the root's composition in one pass over the frame: a pixel of another rank's tile comes out of that rank's packed buffer
// (`all` holds world buffers of `stride` pixels each, rank r's at r * stride, laid out as k_tiles<true> packs them), the
// root's own pixels stay, and pt_output_image is recomputed for every pixel (rgen:140-144)
__global__ __launch_bounds__(kBlock) void k_compose_tiles(Images im, const float4* __restrict__ all, uint64_t stride, uint32_t W, uint32_t H, uint32_t rank, uint32_t world,
                                                           uint32_t tile, uint32_t total_samples, uint32_t limit) {
   const uint32_t tiles_x = (W + tile - 1) / tile, n = W * H;
   for (uint32_t pix = blockIdx.x * kBlock + threadIdx.x; pix < n; pix += gridDim.x * kBlock) {
      const uint32_t x = pix % W, y = pix / W;
      const uint32_t t = (y / tile) * tiles_x + x / tile, owner = t % world;
      float4 acc;
      if (owner == rank)
         acc = im.accumulation[pix];
      else {
         acc = all[owner * stride + (uint64_t)(t / world) * tile * tile + (y % tile) * tile + x % tile];
         im.accumulation[pix] = acc;
      }
      im.output[pix] = resolve_color(acc, total_samples, limit);
   }
}
void launch_compose_tiles(const LaunchCfg& c, const Images& im, const float4* all, uint64_t stride, uint32_t W, uint32_t H, uint32_t rank, uint32_t world, uint32_t tile,
                          uint32_t total_samples, uint32_t limit) {
   k_compose_tiles<<<stream_grid(c, W * H), kBlock, 0, c.stream>>>(im, all, stride, W, H, rank, world, tile, total_samples, limit);
}

void launch_pack_tiles(const LaunchCfg& c, const float4* acc, float4* out, uint32_t W, uint32_t H, uint32_t rank, uint32_t world, uint32_t tile) {
   k_tiles<true><<<stream_grid(c, W * H), kBlock, 0, c.stream>>>(const_cast<float4*>(acc), out, W, H, rank, world, tile);
}
void launch_unpack_tiles(const LaunchCfg& c, float4* acc, const float4* in, uint32_t W, uint32_t H, uint32_t rank, uint32_t world, uint32_t tile) {
   k_tiles<false><<<stream_grid(c, W * H), kBlock, 0, c.stream>>>(acc, const_cast<float4*>(in), W, H, rank, world, tile);
}

}  // namespace uh
