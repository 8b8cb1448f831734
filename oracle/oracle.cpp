// oracle/oracle.cpp — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
//
// A CPU restatement of the reference's path-tracing + ReSTIR hot path (simplerr/rust-renderer,
// Rust + GLSL on Vulkan KHR ray tracing). It is the checker for the HIP path and the "port" CPU
// baseline of bench.py; nothing under rust-renderer_amd/ links, imports or calls it. Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
//
// PARITY STATUS: the reference holds no tests, golden vectors or fixtures and can
// not be compiled or run in this environment (no cargo/rustc/glslang/Vulkan), so
// this oracle is pinned by (a) the RNG known-answer vectors derived from the GLSL text
// (tests/golden/rng_kat.json), (b) analytic identities of the shaders (furnace test, 1/d^2 target
// function, sRGB continuity, RT-Gems offsetRay identities) and (c) code review against the cited
// lines. BVH build / traversal / ray-triangle intersection live in the Vulkan driver, outside the
// reference checkout: for hit t / barycentrics / tie-breaks parity is UNPINNED; the arithmetic
// used here is this repo's own contract (DESIGN.md "Arithmetic contract").
//
// Every function cites the reference file:line it follows (paths relative to the reference root,
// shaders under utopian/shaders/).
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -mfma; fmaf() is used only where the
// contract says so, so results do not depend on the compiler's contraction choices).

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <string>
#include <sched.h>
#include <thread>
#include <vector>

#include "../include/utopian_hip.h"

namespace {

// ------------------------------------------------------------------------------------------
// small vector helpers — plain IEEE f32 ops, evaluated left to right, never contracted
// ------------------------------------------------------------------------------------------
struct V3 {
   float x, y, z;
};
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
static inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
// GLSL dot(): (x*x' + y*y') + z*z', unfused
static inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float length(V3 a) { return std::sqrt(dot(a, a)); }
// GLSL normalize(): v * (1 / sqrt(dot(v,v))) with correctly rounded div and sqrt
static inline V3 normalize(V3 a) {
   float inv = 1.0f / std::sqrt(dot(a, a));
   return a * inv;
}
static inline V3 vmin(V3 a, float s) { return v3(std::fmin(a.x, s), std::fmin(a.y, s), std::fmin(a.z, s)); }

// fused forms, used ONLY by the ray/triangle test (contract: DESIGN.md "Arithmetic contract")
static inline float dot_fma(V3 a, V3 b) { return std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)); }
static inline V3 cross_fma(V3 a, V3 b) {
   return v3(std::fmaf(a.y, b.z, -(a.z * b.y)), std::fmaf(a.z, b.x, -(a.x * b.z)), std::fmaf(a.x, b.y, -(a.y * b.x)));
}

// column-major mat4 (glam) times vec4: ((c0*x + c1*y) + c2*z) + c3*w
struct V4 {
   float x, y, z, w;
};
static inline V4 mat4_mul(const float* m, V4 v) {
   V4 r;
   r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
   r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
   r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
   r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
   return r;
}

static inline uint32_t f2u(float f) {
   uint32_t u;
   std::memcpy(&u, &f, 4);
   return u;
}
static inline float u2f(uint32_t u) {
   float f;
   std::memcpy(&f, &u, 4);
   return f;
}

// ------------------------------------------------------------------------------------------
// A1 — RNG (include/random.glsl)
// ------------------------------------------------------------------------------------------
// random.glsl:5-12
static inline uint32_t jenkinsHash(uint32_t x) {
   x += x << 10;
   x ^= x >> 6;
   x += x << 3;
   x ^= x >> 11;
   x += x << 15;
   return x;
}
// random.glsl:14-18 — dot(uvec2, uvec2) is evaluated in float (exact below 2^24)
static inline uint32_t initRNG(uint32_t px, uint32_t py, uint32_t resx, uint32_t frame) {
   float d = (float)px * 1.0f + (float)py * (float)resx;
   uint32_t seed = (uint32_t)d ^ jenkinsHash(frame);
   return jenkinsHash(seed);
}
// random.glsl:21-24
static inline uint32_t stepRNG(uint32_t s) { return s * 747796405u + 1u; }
// random.glsl:27-34 — float(word) / 4294967295.0f; the divisor rounds to 2^32 in f32
static inline float randomFloat(uint32_t& s) {
   s = stepRNG(s);
   uint32_t word = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
   word = (word >> 22) ^ word;
   return (float)word / 4294967296.0f;
}
// random.glsl:36-46 — rejection sampling, 3 draws per trial
static inline V3 randomPointInUnitSphere(uint32_t& s) {
   for (;;) {
      float a = randomFloat(s), b = randomFloat(s), c = randomFloat(s);
      V3 p = v3(2.0f * a - 1.0f, 2.0f * b - 1.0f, 2.0f * c - 1.0f);
      if (dot(p, p) < 1.0f) return p;
   }
}
// reference.rgen:24 (and the three ReSTIR raygens): int(float(total_samples) + time * 10000.0)
static inline uint32_t frameNumber(const UhViewUniformData& v) {
   float f = (float)v.total_samples + v.time * 10000.0f;
   return (uint32_t)(int32_t)f;
}

// ------------------------------------------------------------------------------------------
// A11 — view.glsl helpers
// ------------------------------------------------------------------------------------------
// view.glsl:46-50
static inline float luminance(V3 rgb) { return dot(rgb, v3(0.2126f, 0.7152f, 0.0722f)); }
// view.glsl:52-60
static inline float linearToSrgb1(float c) {
   if (c < 0.0031308f) return c * 12.92f;
   return 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f;
}
// view.glsl:92-108 (Ray Tracing Gems ch. 6)
static inline V3 offsetRay(V3 p, V3 n) {
   const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
   int32_t ox = (int32_t)(int_scale * n.x), oy = (int32_t)(int_scale * n.y), oz = (int32_t)(int_scale * n.z);
   float pix = u2f(f2u(p.x) + (uint32_t)((p.x < 0) ? -ox : ox));
   float piy = u2f(f2u(p.y) + (uint32_t)((p.y < 0) ? -oy : oy));
   float piz = u2f(f2u(p.z) + (uint32_t)((p.z < 0) ? -oz : oz));
   return v3(std::fabs(p.x) < origin ? p.x + float_scale * n.x : pix,
             std::fabs(p.y) < origin ? p.y + float_scale * n.y : piy,
             std::fabs(p.z) < origin ? p.z + float_scale * n.z : piz);
}

// ------------------------------------------------------------------------------------------
// A4 — sky (include/atmosphere.glsl), all f32
// ------------------------------------------------------------------------------------------
namespace sky {
const float PLANET_RADIUS = 6371000.0f;
const float ATMOSPHERE_HEIGHT = 100000.0f;
const float RAYLEIGH_HEIGHT = ATMOSPHERE_HEIGHT * 0.08f;
const float MIE_HEIGHT = ATMOSPHERE_HEIGHT * 0.012f;
const float PI = 3.14159265359f;
static inline V3 C_RAYLEIGH() { return v3(5.802f, 13.558f, 33.100f) * 1e-6f; }
static inline V3 C_MIE() { return v3(3.996f, 3.996f, 3.996f) * 1e-6f; }
static inline V3 C_OZONE() { return v3(0.650f, 1.881f, 0.085f) * 1e-6f; }
static inline V3 PLANET_CENTER() { return v3(0, -PLANET_RADIUS, 0); }

// atmosphere.glsl:53-69
static inline void SphereIntersection(V3 rayStart, V3 rayDir, V3 c, float radius, float& t0, float& t1) {
   rayStart = rayStart - c;
   float a = dot(rayDir, rayDir);
   float b = 2.0f * dot(rayStart, rayDir);
   float cc = dot(rayStart, rayStart) - (radius * radius);
   float d = b * b - 4.0f * a * cc;
   if (d < 0) {
      t0 = -1;
      t1 = -1;
   } else {
      d = std::sqrt(d);
      t0 = (-b - d) / (2.0f * a);
      t1 = (-b + d) / (2.0f * a);
   }
}
// atmosphere.glsl:74-77
static inline void AtmosphereIntersection(V3 s, V3 d, float& t0, float& t1) {
   SphereIntersection(s, d, PLANET_CENTER(), PLANET_RADIUS + ATMOSPHERE_HEIGHT, t0, t1);
}
// atmosphere.glsl:81-84
static inline float PhaseRayleigh(float costh) { return 3.0f * (1.0f + costh * costh) / (16.0f * PI); }
// atmosphere.glsl:85-91
static inline float PhaseMie(float costh, float g) {
   g = std::fmin(g, 0.9381f);
   float k = 1.55f * g - 0.55f * g * g * g;
   float kcosth = k * costh;
   return (1.0f - k * k) / ((4.0f * PI) * (1.0f - kcosth) * (1.0f - kcosth));
}
// atmosphere.glsl:95-98
static inline float AtmosphereHeight(V3 p) { return length(p - PLANET_CENTER()) - PLANET_RADIUS; }
// atmosphere.glsl:99-115
static inline V3 AtmosphereDensity(float h) {
   float r = std::exp(-std::fmax(0.0f, h / RAYLEIGH_HEIGHT));
   float m = std::exp(-std::fmax(0.0f, h / MIE_HEIGHT));
   float o = std::fmax(0.0f, 1.0f - std::fabs(h - 25000.0f) / 15000.0f);
   return v3(r, m, o);
}
// atmosphere.glsl:123-143
static inline V3 IntegrateOpticalDepth(V3 rayStart, V3 rayDir) {
   float t0, t1;
   AtmosphereIntersection(rayStart, rayDir, t0, t1);
   float rayLength = t1;
   const int sampleCount = 8;
   float stepSize = rayLength / (float)sampleCount;
   V3 opticalDepth = v3(0, 0, 0);
   for (int i = 0; i < sampleCount; i++) {
      V3 localPosition = rayStart + rayDir * ((float)i + 0.5f) * stepSize;
      float localHeight = AtmosphereHeight(localPosition);
      V3 localDensity = AtmosphereDensity(localHeight);
      opticalDepth = opticalDepth + localDensity * stepSize;
   }
   return opticalDepth;
}
// atmosphere.glsl:146-150
static inline V3 Absorb(V3 od) {
   V3 a = (od.x * C_RAYLEIGH() + od.y * C_MIE() * 1.1f + od.z * C_OZONE()) * 1.0f;
   return v3(std::exp(-a.x), std::exp(-a.y), std::exp(-a.z));
}
// atmosphere.glsl:154-214
static inline V3 IntegrateScattering(V3 rayStart, V3 rayDir, float rayLength, V3 lightDir, V3 lightColor) {
   float rayHeight = AtmosphereHeight(rayStart);
   float c = 1.0f - rayHeight / ATMOSPHERE_HEIGHT;
   c = std::fmin(std::fmax(c, 0.0f), 1.0f);
   float sampleDistributionExponent = 1.0f + c * 8.0f;

   float i0, i1;
   AtmosphereIntersection(rayStart, rayDir, i0, i1);
   rayLength = std::fmin(rayLength, i1);
   if (i0 > 0) {
      rayStart = rayStart + rayDir * i0;
      rayLength -= i0;
   }
   float costh = dot(rayDir, lightDir);
   float phaseR = PhaseRayleigh(costh);
   float phaseM = PhaseMie(costh, 0.85f);
   const int sampleCount = 16;
   V3 opticalDepth = v3(0, 0, 0), rayleigh = v3(0, 0, 0), mie = v3(0, 0, 0);
   float prevRayTime = 0;
   for (int i = 0; i < sampleCount; i++) {
      float rayTime = std::pow((float)i / (float)sampleCount, sampleDistributionExponent) * rayLength;
      float stepSize = (rayTime - prevRayTime);
      V3 localPosition = rayStart + rayDir * rayTime;
      float localHeight = AtmosphereHeight(localPosition);
      V3 localDensity = AtmosphereDensity(localHeight);
      opticalDepth = opticalDepth + localDensity * stepSize;
      V3 viewTransmittance = Absorb(opticalDepth);
      V3 opticalDepthlight = IntegrateOpticalDepth(localPosition, lightDir);
      V3 lightTransmittance = Absorb(opticalDepthlight);
      rayleigh = rayleigh + viewTransmittance * lightTransmittance * phaseR * localDensity.x * stepSize;
      mie = mie + viewTransmittance * lightTransmittance * phaseM * localDensity.y * stepSize;
      prevRayTime = rayTime;
   }
   V3 color = (rayleigh * C_RAYLEIGH() + mie * C_MIE()) * lightColor * 20.0f;
   return color;
}
}  // namespace sky

// ------------------------------------------------------------------------------------------
// scene
// ------------------------------------------------------------------------------------------
struct Texture {
   uint32_t w, h;
   std::vector<uint8_t> px;
};
struct MeshRec {
   std::vector<UhVertex> vertices;
   std::vector<uint32_t> indices;
   UhGpuMaterial material;
   float o2w[12];  // row-major 3x4
   float w2o[9];   // row-major inverse of the upper 3x3
   uint32_t first_tri;
};
struct Tri {  // baked world-space triangle
   V3 v0, e1, e2;
   uint32_t mesh, prim;
};
struct Hit {
   float t, u, v;
   uint32_t mesh, prim;  // mesh == 0xffffffff: miss
};
struct BNode {  // oracle's own binary BVH (independent of the product's BVH4)
   float bmin[3], bmax[3];
   uint32_t left, right;  // interior: child node indices; leaf: right == 0xffffffff, left = first
   uint32_t count;
};

struct Payload {  // pathtrace_reference/payload.glsl:2-8
   V3 color;
   float distance;
   V3 scatter;
   float scattered;
   V3 normal;
   uint32_t seed;
};

struct Counters {
   std::atomic<uint64_t> rays[UH_RAY_KINDS];
   std::atomic<uint64_t> nodes, tris, closest_hits, misses;
};
// per-thread tallies, flushed into Oracle::ctr once per image row (no shared-line ping-pong in
// the timed CPU baseline)
struct LocalCounters {
   uint64_t rays[UH_RAY_KINDS] = {0, 0, 0, 0, 0};
   uint64_t nodes = 0, tris = 0, closest_hits = 0, misses = 0;
};
static thread_local LocalCounters tl_ctr;

struct Oracle {
   uint32_t W, H;
   std::vector<Texture> textures;
   std::vector<MeshRec> meshes;
   std::vector<UhGpuLight> lights;
   std::vector<Tri> tris;
   std::vector<BNode> nodes;
   std::vector<uint32_t> tri_order;
   bool built = false, ever_built = false;
   bool brute_force = false;
   bool full_frame_restir = false;
   bool furnace = false;  // reference.rmiss built with FURNACE_TEST (rmiss:14-28 compiled out): a miss returns white
   int num_threads = 0;
   // graph resources (renderers/mod.rs:199-244)
   std::vector<float> accumulation;  // RGBA32F
   std::vector<uint8_t> output;      // BGRA8
   std::vector<float> gbuffer_pos;   // RGBA32F, un-filtered texels
   std::vector<UhReservoir> reservoirs[3];
   // tile partition
   uint32_t tp_rank = 0, tp_world = 1, tp_tile = 64;
   // reservoir passes by bands of rows (orc_set_restir_partition: the checker's twin of uh_set_restir_partition)
   uint32_t rp_rank = 0, rp_world = 1;
   UhRestirExchangeFn rp_exchange = nullptr;
   void* rp_user = nullptr;
   Counters ctr;
   std::string err;
};

// 3x3 inverse by cofactors; row-major. Contract: cofactor / det, plain ops.
static void invert3x3(const float* m /*3x4 row-major*/, float* inv /*9*/) {
   float a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
   float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
   float det = (a * A + b * B) + c * C;
   float id = 1.0f / det;
   inv[0] = A * id;
   inv[1] = -(b * i - c * h) * id;
   inv[2] = (b * f - c * e) * id;
   inv[3] = B * id;
   inv[4] = (a * i - c * g) * id;
   inv[5] = -(a * f - c * d) * id;
   inv[6] = C * id;
   inv[7] = -(a * h - b * g) * id;
   inv[8] = (a * e - b * d) * id;
}
static bool is_identity3x4(const float* m) {
   static const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
   return std::memcmp(m, I, sizeof(I)) == 0;
}
// object→world point: ((m0*x + m1*y) + m2*z) + m3 per row
static inline V3 xform_point(const float* m, V3 p) {
   return v3(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
             ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]);
}

// ------------------------------------------------------------------------------------------
// A10 — acceleration structure. The reference delegates this to the Vulkan driver
// (raytracing.rs:113-217 BLAS, :219-277 instances, :279-398 TLAS); instance transform, opaque
// geometry, no face culling, mask 0xff are the only semantics visible to the shaders. Here:
// triangles are baked to world space, closest hit = min t over ALL triangles with ties broken by
// the smaller (mesh, primitive) — a definition that does not depend on any BVH.
// ------------------------------------------------------------------------------------------
static void bake_triangles(Oracle& o) {
   o.tris.clear();
   for (uint32_t mi = 0; mi < o.meshes.size(); mi++) {
      MeshRec& m = o.meshes[mi];
      m.first_tri = (uint32_t)o.tris.size();
      bool ident = is_identity3x4(m.o2w);
      uint32_t nt = (uint32_t)m.indices.size() / 3;
      for (uint32_t p = 0; p < nt; p++) {
         V3 a[3];
         for (int k = 0; k < 3; k++) {
            const UhVertex& vx = m.vertices[m.indices[p * 3 + k]];
            V3 q = v3(vx.pos[0], vx.pos[1], vx.pos[2]);
            a[k] = ident ? q : xform_point(m.o2w, q);
         }
         Tri t;
         t.v0 = a[0];
         t.e1 = a[1] - a[0];
         t.e2 = a[2] - a[0];
         t.mesh = mi;
         t.prim = p;
         o.tris.push_back(t);
      }
   }
}

// Ray/triangle test (Möller-Trumbore on precomputed v0,e1,e2). Contract: dot/cross fused as
// dot_fma/cross_fma, everything else plain; accept tmin < t < tbest, or t == tbest with a smaller
// (mesh,prim) key.
static inline bool tri_test(const Tri& tr, V3 o, V3 d, float tmin, Hit& best) {
   V3 p = cross_fma(d, tr.e2);
   float det = dot_fma(tr.e1, p);
   if (det == 0.0f) return false;
   float inv = 1.0f / det;
   V3 tv = o - tr.v0;
   float u = dot_fma(tv, p) * inv;
   if (!(u >= 0.0f && u <= 1.0f)) return false;
   V3 q = cross_fma(tv, tr.e1);
   float v = dot_fma(d, q) * inv;
   if (!(v >= 0.0f && u + v <= 1.0f)) return false;
   float t = dot_fma(tr.e2, q) * inv;
   if (!(t > tmin)) return false;
   if (t < best.t || (t == best.t && (tr.mesh < best.mesh || (tr.mesh == best.mesh && tr.prim < best.prim)))) {
      best.t = t;
      best.u = u;
      best.v = v;
      best.mesh = tr.mesh;
      best.prim = tr.prim;
      return true;
   }
   return false;
}

static void build_bvh(Oracle& o) {
   o.nodes.clear();
   uint32_t n = (uint32_t)o.tris.size();
   o.tri_order.resize(n);
   for (uint32_t i = 0; i < n; i++) o.tri_order[i] = i;
   if (n == 0) return;
   std::vector<V3> cen(n), lo(n), hi(n);
   for (uint32_t i = 0; i < n; i++) {
      const Tri& t = o.tris[i];
      V3 a = t.v0, b = t.v0 + t.e1, c = t.v0 + t.e2;
      lo[i] = v3(std::fmin(a.x, std::fmin(b.x, c.x)), std::fmin(a.y, std::fmin(b.y, c.y)), std::fmin(a.z, std::fmin(b.z, c.z)));
      hi[i] = v3(std::fmax(a.x, std::fmax(b.x, c.x)), std::fmax(a.y, std::fmax(b.y, c.y)), std::fmax(a.z, std::fmax(b.z, c.z)));
      cen[i] = (lo[i] + hi[i]) * 0.5f;
   }
   struct Job {
      uint32_t node, first, count;
   };
   std::vector<Job> stack;
   o.nodes.push_back(BNode());
   stack.push_back(Job{0, 0, n});
   while (!stack.empty()) {
      Job j = stack.back();
      stack.pop_back();
      float bmin[3] = {INFINITY, INFINITY, INFINITY}, bmax[3] = {-INFINITY, -INFINITY, -INFINITY};
      float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (uint32_t k = j.first; k < j.first + j.count; k++) {
         uint32_t i = o.tri_order[k];
         const float l[3] = {lo[i].x, lo[i].y, lo[i].z}, h[3] = {hi[i].x, hi[i].y, hi[i].z}, c[3] = {cen[i].x, cen[i].y, cen[i].z};
         for (int a = 0; a < 3; a++) {
            bmin[a] = std::fmin(bmin[a], l[a]);
            bmax[a] = std::fmax(bmax[a], h[a]);
            cmin[a] = std::fmin(cmin[a], c[a]);
            cmax[a] = std::fmax(cmax[a], c[a]);
         }
      }
      BNode nd;
      // conservative padding: the slab test must never cull a triangle tri_test() would accept
      for (int a = 0; a < 3; a++) {
         float pad = 1e-4f + 1e-5f * std::fmax(std::fabs(bmin[a]), std::fabs(bmax[a]));
         nd.bmin[a] = bmin[a] - pad;
         nd.bmax[a] = bmax[a] + pad;
      }
      int axis = 0;
      float ext = cmax[0] - cmin[0];
      for (int a = 1; a < 3; a++)
         if (cmax[a] - cmin[a] > ext) {
            ext = cmax[a] - cmin[a];
            axis = a;
         }
      if (j.count <= 4 || !(ext > 0)) {
         nd.left = j.first;
         nd.right = 0xffffffffu;
         nd.count = j.count;
         o.nodes[j.node] = nd;
         continue;
      }
      uint32_t mid = j.first + j.count / 2;
      auto key = [&](uint32_t i) { return axis == 0 ? cen[i].x : (axis == 1 ? cen[i].y : cen[i].z); };
      std::nth_element(o.tri_order.begin() + j.first, o.tri_order.begin() + mid, o.tri_order.begin() + j.first + j.count,
                       [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
      nd.left = (uint32_t)o.nodes.size();
      nd.right = nd.left + 1;
      nd.count = 0;
      o.nodes[j.node] = nd;
      o.nodes.push_back(BNode());
      o.nodes.push_back(BNode());
      stack.push_back(Job{nd.left, j.first, mid - j.first});
      stack.push_back(Job{nd.right, mid, j.first + j.count - mid});
   }
}

static inline bool slab(const BNode& n, V3 o, V3 id, float tmin, float tmax) {
   float t0x = (n.bmin[0] - o.x) * id.x, t1x = (n.bmax[0] - o.x) * id.x;
   float t0y = (n.bmin[1] - o.y) * id.y, t1y = (n.bmax[1] - o.y) * id.y;
   float t0z = (n.bmin[2] - o.z) * id.z, t1z = (n.bmax[2] - o.z) * id.z;
   float tn = std::fmax(std::fmax(std::fmin(t0x, t1x), std::fmin(t0y, t1y)), std::fmax(std::fmin(t0z, t1z), tmin));
   float tf = std::fmin(std::fmin(std::fmax(t0x, t1x), std::fmax(t0y, t1y)), std::fmin(std::fmax(t0z, t1z), tmax));
   // robust: widen the far bound by 4 ulp-ish (Ize, "Robust BVH Ray Traversal")
   return tn <= tf * 1.0000005f + 1e-30f;
}

// closest hit over (tmin, tmax). Returns mesh == 0xffffffff on miss.
static Hit trace_closest(Oracle& o, V3 org, V3 dir, float tmin, float tmax, bool count) {
   Hit best;
   best.t = tmax;
   best.u = best.v = 0;
   best.mesh = best.prim = 0xffffffffu;
   uint64_t nn = 0, nt = 0;
   if (o.brute_force || o.nodes.empty()) {
      for (const Tri& t : o.tris) tri_test(t, org, dir, tmin, best);
      nt = o.tris.size();
   } else {
      V3 id = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
      uint32_t stack[128];
      int sp = 0;
      stack[sp++] = 0;
      while (sp) {
         const BNode& n = o.nodes[stack[--sp]];
         nn++;
         if (!slab(n, org, id, tmin, best.t)) continue;
         if (n.right == 0xffffffffu) {
            for (uint32_t k = 0; k < n.count; k++) tri_test(o.tris[o.tri_order[n.left + k]], org, dir, tmin, best);
            nt += n.count;
         } else {
            stack[sp++] = n.left;
            stack[sp++] = n.right;
         }
      }
   }
   if (count) {
      tl_ctr.nodes += nn;
      tl_ctr.tris += nt;
   }
   return best;
}

// ------------------------------------------------------------------------------------------
// A3 — closest-hit shader (pathtrace_reference/reference.rchit:20-92)
// ------------------------------------------------------------------------------------------
// texture.rs:85-98: RGBA8 UNORM, LINEAR mag/min, MIRRORED_REPEAT, LOD 0. Contract: texel
// coordinate = uv*size - 0.5, float weights (not the 8-bit fixed-point of real samplers).
static inline int mirror(int i, int n) {
   int period = 2 * n;
   int m = i % period;
   if (m < 0) m += period;
   return m < n ? m : period - 1 - m;
}
static V3 sample_texture(const Texture& t, float u, float v) {
   float x = u * (float)t.w - 0.5f, y = v * (float)t.h - 0.5f;
   if (!(std::fabs(x) < 1e9f) || !(std::fabs(y) < 1e9f)) return v3(0, 0, 0);
   float fx = std::floor(x), fy = std::floor(y);
   float ax = x - fx, ay = y - fy;
   int x0 = mirror((int)fx, (int)t.w), x1 = mirror((int)fx + 1, (int)t.w);
   int y0 = mirror((int)fy, (int)t.h), y1 = mirror((int)fy + 1, (int)t.h);
   auto tx = [&](int xx, int yy) {
      const uint8_t* p = &t.px[4 * ((size_t)yy * t.w + xx)];
      return v3((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f);
   };
   V3 t00 = tx(x0, y0), t10 = tx(x1, y0), t01 = tx(x0, y1), t11 = tx(x1, y1);
   V3 a = t00 * (1.0f - ax) + t10 * ax;
   V3 b = t01 * (1.0f - ax) + t11 * ax;
   return a * (1.0f - ay) + b * ay;
}
// rchit:12-18. Contract: pow(x, 5.0) is evaluated as ((x*x)*(x*x))*x
static inline float schlick_reflectance(float cosine, float ref_idx) {
   float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
   r0 = r0 * r0;
   float x = 1.0f - cosine;
   float x5 = ((x * x) * (x * x)) * x;
   return r0 + (1.0f - r0) * x5;
}
static inline V3 reflect(V3 I, V3 N) { return I - N * (2.0f * dot(N, I)); }
static inline V3 refract(V3 I, V3 N, float eta) {
   float dn = dot(N, I);
   float k = 1.0f - eta * eta * (1.0f - dn * dn);
   if (k < 0.0f) return v3(0, 0, 0);
   return I * eta - N * (eta * dn + std::sqrt(k));
}

// Material type 4 - an EXTENSION that no reference scene uses: the Cook-Torrance BRDF of
// include/pbr_lighting.glsl:20-79 / include/brdf.glsl:3-36,82-85 evaluated for the Lambertian-style scatter
// direction and returned as BRDF * cos / pdf (pdf = cos/pi): kD * baseColor + specular * pi.
static V3 pbr_weight(V3 N, V3 V, V3 L, V3 base, float metallic, float roughness) {
   const float PI = 3.14159265359f;
   const V3 H = normalize(V + L);
   const float a = roughness * roughness, a2 = a * a;                                   // brdf.glsl:5-6
   const float NdotH = std::fmax(dot(N, H), 0.0f), NdotH2 = NdotH * NdotH;               // :7-8
   float denom = NdotH2 * (a2 - 1.0f) + 1.0f;                                            // :11
   denom = (PI * denom) * denom;                                                         // :12
   const float NDF = a2 / denom;                                                         // :14
   const float NdotV = std::fmax(dot(N, V), 0.0f), NdotL = std::fmax(dot(N, L), 0.0f);   // :30-31
   const float r = roughness + 1.0f, k = (r * r) / 8.0f;                                 // :19-20
   const float gV = NdotV / (NdotV * (1.0f - k) + k), gL = NdotL / (NdotL * (1.0f - k) + k);  // :22-25
   const float G = gL * gV;                                                              // :35
   const float c = std::fmin(std::fmax(1.0f - std::fmax(dot(H, V), 0.0f), 0.0f), 1.0f);  // :84
   const float c5 = ((c * c) * (c * c)) * c;
   const float om = 1.0f - metallic;
   const V3 F0 = v3(0.04f * om + base.x * metallic, 0.04f * om + base.y * metallic, 0.04f * om + base.z * metallic);  // pbr_lighting.glsl:29-30
   const V3 F = v3(F0.x + (1.0f - F0.x) * c5, F0.y + (1.0f - F0.y) * c5, F0.z + (1.0f - F0.z) * c5);
   const V3 kD = v3((1.0f - F.x) * om, (1.0f - F.y) * om, (1.0f - F.z) * om);            // :65-67
   const float den = (4.0f * NdotV) * NdotL + 0.0001f;                                   // :70
   const float dg = NDF * G;
   const V3 spec = v3((dg * F.x) / den, (dg * F.y) / den, (dg * F.z) / den);             // :69-71
   return v3(kD.x * base.x + spec.x * PI, kD.y * base.y + spec.y * PI, kD.z * base.z + spec.z * PI);  // :75 times pi / NdotL
}

static void closest_hit_shader(Oracle& o, const Hit& h, V3 rayDir, Payload& pl) {
   const MeshRec& mesh = o.meshes[h.mesh];                                    // rchit:22
   const UhGpuMaterial& material = mesh.material;                             // rchit:23
   const uint32_t* idx = &mesh.indices[(size_t)h.prim * 3];                   // rchit:25
   const UhVertex &v0 = mesh.vertices[idx[0]], &v1 = mesh.vertices[idx[1]], &v2 = mesh.vertices[idx[2]];  // rchit:26-28
   float bx = 1.0f - h.u - h.v, by = h.u, bz = h.v;                           // rchit:30
   V3 n0 = v3(v0.normal[0], v0.normal[1], v0.normal[2]), n1 = v3(v1.normal[0], v1.normal[1], v1.normal[2]),
      n2 = v3(v2.normal[0], v2.normal[1], v2.normal[2]);
   V3 normal = (n0 * bx + n1 * by) + n2 * bz;                                 // rchit:31
   // rchit:32 — vec3(normal * gl_WorldToObjectEXT): component j = dot(normal, column j of W2O)
   const float* wi = mesh.w2o;
   V3 wn = v3((normal.x * wi[0] + normal.y * wi[3]) + normal.z * wi[6], (normal.x * wi[1] + normal.y * wi[4]) + normal.z * wi[7],
              (normal.x * wi[2] + normal.y * wi[5]) + normal.z * wi[8]);
   V3 world_normal = normalize(wn);
   if (dot(world_normal, rayDir) > 0.0f) world_normal = neg(world_normal);   // rchit:35-37
   float uu = (v0.uv[0] * bx + v1.uv[0] * by) + v2.uv[0] * bz;                // rchit:39
   float vv = (v0.uv[1] * bx + v1.uv[1] * by) + v2.uv[1] * bz;
   V3 color = v3(1, 1, 1);
   if (material.diffuse_map < o.textures.size()) color = sample_texture(o.textures[material.diffuse_map], uu, vv);  // rchit:40
   color = color * v3(material.base_color_factor[0], material.base_color_factor[1], material.base_color_factor[2]);  // rchit:41

   V3 scatter = v3(0, 0, 0);
   bool isScattered = false;
   float type = material.raytrace_properties[0];
   if (type == 0.0f) {                                                        // rchit:47-50 Lambertian
      scatter = world_normal + randomPointInUnitSphere(pl.seed);
      isScattered = dot(rayDir, world_normal) < 0.0f;
   } else if (type == 1.0f) {                                                 // rchit:52-59 Metal
      scatter = reflect(normalize(rayDir), world_normal);
      scatter = scatter + material.raytrace_properties[1] * randomPointInUnitSphere(pl.seed);
      isScattered = true;
      color = v3(1, 1, 1);
   } else if (type == 2.0f) {                                                 // rchit:61-83 Dielectric
      V3 nd = normalize(rayDir);
      float dnd = dot(nd, world_normal);
      V3 outward = dnd > 0 ? neg(world_normal) : world_normal;
      float ratio = material.raytrace_properties[1];
      ratio = dnd > 0 ? ratio : 1.0f / ratio;
      float cos_theta = std::fmin(dot(-1.0f * nd, outward), 1.0f);
      float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
      bool cannot_refract = ratio * sin_theta > 1.0f;
      float reflectance = schlick_reflectance(cos_theta, ratio);
      if (cannot_refract || reflectance > randomFloat(pl.seed))
         scatter = reflect(nd, outward);
      else
         scatter = refract(nd, outward, ratio);
      isScattered = true;
      color = v3(1, 1, 1);
   } else if (type == 4.0f) {
      // EXTENSION (never produced by the reference's scenes): Cook-Torrance, see pbr_weight()
      scatter = world_normal + randomPointInUnitSphere(pl.seed);
      isScattered = dot(rayDir, world_normal) < 0.0f;
      color = pbr_weight(world_normal, -1.0f * normalize(rayDir), normalize(scatter), color, material.metallic_factor, material.roughness_factor);
   } else {                                                                   // rchit:85-89 DiffuseLight
      isScattered = false;
      color = v3(1, 1, 1);
   }
   pl.color = color;                                                          // rchit:91
   pl.distance = h.t;
   pl.scatter = scatter;
   pl.scattered = isScattered ? 1.0f : 0.0f;
   pl.normal = world_normal;
}

// A4 — miss shader (pathtrace_reference/reference.rmiss:10-31). `want_color` = false for shadow
// rays: the raygen only reads .w == -1 from them (rgen:69,118-119), so the sky integral is dead.
static void miss_shader(const UhViewUniformData& view, V3 org, V3 dir, bool want_color, bool furnace, Payload& pl) {
   V3 sky_color = v3(1, 1, 1);  // rmiss:12
   if (furnace) {
      // #ifndef FURNACE_TEST (rmiss:14-28) compiled out: white stays
   } else if (view.sky_enabled == 1) {
      if (want_color) {
         V3 light_dir = normalize(v3(view.sun_dir[0], view.sun_dir[1], view.sun_dir[2]));
         sky_color = sky::IntegrateScattering(org, dir, 999999999.0f, light_dir, v3(1, 1, 1));
         sky_color = vmin(sky_color, 1.0f);
      }
   } else
      sky_color = v3(0, 0, 0);
   pl.color = sky_color;
   pl.distance = -1.0f;
   pl.scatter = v3(0, 0, 0);
   pl.scattered = 0;
   pl.normal = v3(0, 0, 0);
   pl.seed = 0;
}

// traceRayEXT(topLevelAS, opaque, 0xff, 0,0,0, origin, tmin, dir, tmax, payload)
static void trace_ray(Oracle& o, const UhViewUniformData& view, V3 org, V3 dir, float tmin, float tmax, Payload& pl, int kind) {
   tl_ctr.rays[kind]++;
   bool path_ray = (kind == UH_RAY_PRIMARY || kind == UH_RAY_BOUNCE);
   Hit h = trace_closest(o, org, dir, tmin, tmax, path_ray);
   if (h.mesh != 0xffffffffu) {
      if (path_ray) {
         tl_ctr.closest_hits++;
         closest_hit_shader(o, h, dir, pl);
      } else {
         pl.distance = h.t;  // shadow payload: only .w is read (rgen:69,118)
      }
   } else {
      if (path_ray) tl_ctr.misses++;
      miss_shader(view, org, dir, path_ray, o.furnace, pl);
   }
}

// ------------------------------------------------------------------------------------------
// A5 — reservoir math (include/restir_sampling.glsl)
// ------------------------------------------------------------------------------------------
// restir_sampling.glsl:59-69. Contract: pow(d, 2.0) = d*d; an out-of-range light index (the
// reference reads out of bounds for Y = -1 and for idx = n when xi == 1.0) has p_hat = 0.
static inline float target_function(const Oracle& o, int light_index, V3 hit_position) {
   if (light_index < 0 || (size_t)light_index >= o.lights.size()) return 0.0f;
   const UhGpuLight& l = o.lights[light_index];
   float d = length(v3(l.position[0], l.position[1], l.position[2]) - hit_position);
   float d2 = d * d;
   return luminance(v3(l.intensity[0] / d2, l.intensity[1] / d2, l.intensity[2] / d2));
}
// restir_sampling.glsl:71-77
static inline void sample_light_uniform(const UhViewUniformData& view, uint32_t& rng, int& idx, float& w) {
   uint32_t n = std::min(view.num_lights, view.max_num_lights_used);
   idx = (int)(randomFloat(rng) * (float)n);
   w = 1.0f / (float)n;
}
// restir_sampling.glsl:79-82
static inline void finalize_resampling(UhReservoir& r, float p_hat) {
   r.W_X = (p_hat == 0.0f) ? 0.0f : (1.0f / p_hat) * r.W_sum / (float)r.M;
}
// restir_sampling.glsl:85-94
static inline void updateReservoir(uint32_t& rng, UhReservoir& r, int Xi, float w_i, int M) {
   r.W_sum += w_i;
   r.M += M;
   if (randomFloat(rng) * r.W_sum < w_i) r.Y = Xi;
}
// restir_sampling.glsl:96-131
static UhReservoir resample(const Oracle& o, const UhViewUniformData& view, uint32_t& rng, V3 hit_position) {
   UhReservoir r = {-1, 0.0f, 0.0f, 0};
   const int M = 32;
   for (int i = 0; i < M; i++) {
      int cand;
      float p;
      sample_light_uniform(view, rng, cand, p);
      float m_i = 1.0f / (float)M;
      float p_hat = target_function(o, cand, hit_position);
      float W_Xi = 1.0f / p;
      float w_i = m_i * p_hat * W_Xi;
      updateReservoir(rng, r, cand, w_i, 1);
   }
   r.M = 1;
   if (r.Y != -1) finalize_resampling(r, target_function(o, r.Y, hit_position));
   return r;
}

// ------------------------------------------------------------------------------------------
// A2 — raygen (pathtrace_reference/reference.rgen:22-145)
// ------------------------------------------------------------------------------------------
static inline void primary_ray(const UhViewUniformData& view, uint32_t W, uint32_t H, uint32_t px, uint32_t py, float jx, float jy, V3& org, V3& dir) {
   float cx = (float)px + jx, cy = (float)py + jy;                       // rgen:31
   float u = cx / (float)W, v = cy / (float)H;                           // rgen:32
   v = 1.0f - v;                                                         // rgen:33
   float dx = u * 2.0f - 1.0f, dy = v * 2.0f - 1.0f;                     // rgen:34
   V4 o4 = mat4_mul(view.inverse_view, V4{0, 0, 0, 1});                  // rgen:36
   V4 tg = mat4_mul(view.inverse_projection, V4{dx, dy, 1, 1});          // rgen:37
   V3 nt = normalize(v3(tg.x, tg.y, tg.z));
   V4 d4 = mat4_mul(view.inverse_view, V4{nt.x, nt.y, nt.z, 0});         // rgen:38
   org = v3(o4.x, o4.y, o4.z);
   dir = v3(d4.x, d4.y, d4.z);
}

static inline bool owns_pixel(const Oracle& o, uint32_t x, uint32_t y) {
   if (o.tp_world <= 1) return true;
   uint32_t tiles_x = (o.W + o.tp_tile - 1) / o.tp_tile;
   uint32_t tile = (y / o.tp_tile) * tiles_x + (x / o.tp_tile);
   return tile % o.tp_world == o.tp_rank;
}

static inline uint8_t unorm8(float x) {
   if (!(x > 0.0f)) x = 0.0f;  // also maps NaN to 0
   if (x > 1.0f) x = 1.0f;
   return (uint8_t)std::nearbyint(x * 255.0f);
}

static void resolve_pixel(Oracle& o, uint32_t total_samples, uint32_t limit, size_t pi) {
   float denom = (float)std::min(total_samples, limit);
   float* acc = &o.accumulation[pi * 4];
   V3 c = v3(acc[0] / denom, acc[1] / denom, acc[2] / denom);           // rgen:140
   c = v3(linearToSrgb1(c.x), linearToSrgb1(c.y), linearToSrgb1(c.z));  // rgen:141
   // rgen:144 imageStore(vec4(pixelColor, 0)) on a B8G8R8A8_UNORM image (renderers/mod.rs:199-203)
   o.output[pi * 4 + 0] = unorm8(c.z);
   o.output[pi * 4 + 1] = unorm8(c.y);
   o.output[pi * 4 + 2] = unorm8(c.x);
   o.output[pi * 4 + 3] = 0;
}

static void raygen_pixel(Oracle& o, const UhViewUniformData& view, uint32_t px, uint32_t py) {
   const uint32_t W = o.W, H = o.H;
   uint32_t rngState = initRNG(px, py, W, frameNumber(view));            // rgen:24
   V3 pixelColor = v3(0, 0, 0);
   Payload rayPayload, shadowRayPayload;
   std::memset(&rayPayload, 0, sizeof(rayPayload));
   std::memset(&shadowRayPayload, 0, sizeof(shadowRayPayload));
   const V3 sun_dir = normalize(v3(view.sun_dir[0], view.sun_dir[1], view.sun_dir[2]));
   for (uint32_t s = 0; s < view.samples_per_frame; s++) {               // rgen:28
      rayPayload.seed = rngState;                                        // rgen:30 (copy BEFORE the jitter draws)
      float jx = randomFloat(rngState), jy = randomFloat(rngState);      // rgen:31
      V3 origin, direction;
      primary_ray(view, W, H, px, py, jx, jy, origin, direction);
      V3 radiance = v3(0, 0, 0), throughput = v3(1, 1, 1);
      for (uint32_t b = 0; b < view.num_bounces; b++) {                  // rgen:42
         const float tmin = 0.001f, tmax = 10000.0f;
         trace_ray(o, view, origin, direction, tmin, tmax, rayPayload, b == 0 ? UH_RAY_PRIMARY : UH_RAY_BOUNCE);  // rgen:47
         throughput = throughput * rayPayload.color;                     // rgen:48
         float hitDistance = rayPayload.distance;
         bool isScattered = rayPayload.scattered != 0.0f;
         if (hitDistance < 0 || !isScattered) {                          // rgen:53-57
            radiance = radiance + throughput;
            break;
         }
         origin = origin + hitDistance * direction;                      // rgen:59
         origin = offsetRay(origin, rayPayload.normal);                  // rgen:60
         direction = rayPayload.scatter;                                 // rgen:61 (un-normalised)
         if (view.sun_shadow_enabled == 1) {                             // rgen:63-79
            trace_ray(o, view, origin, sun_dir, tmin, tmax, shadowRayPayload, UH_RAY_SUN_SHADOW);
            if (shadowRayPayload.distance == -1.0f) radiance = radiance + throughput;
         }
         if (view.lights_enabled == 1) {                                 // rgen:81-124
            int light_index = 0;
            float light_sample_weight = 0.0f, total_weights = 1.0f;
            bool use_reservoir = (px > W / 2 || o.full_frame_restir) && view.use_ris_light_sampling == 1;  // rgen:87 (:90-96 is dead code)
            if (use_reservoir) {
               UhReservoir r = o.reservoirs[2][(size_t)py * W + px];    // rgen:98 (binding 1 = spatial_reuse_reservoirs, mod.rs:354)
               light_sample_weight = r.W_X;
               total_weights = r.W_sum;
               light_index = r.Y;
            } else {
               sample_light_uniform(view, rngState, light_index, light_sample_weight);  // rgen:107
               light_sample_weight = 1.0f / light_sample_weight;                          // rgen:108
            }
            if (total_weights != 0.0f) {                                 // rgen:112
               // out-of-range light (Y = -1, or idx = n when xi == 1.0): zero light at the origin
               V3 lpos = v3(0, 0, 0);
               if (light_index >= 0 && (size_t)light_index < o.lights.size())
                  lpos = v3(o.lights[light_index].position[0], o.lights[light_index].position[1], o.lights[light_index].position[2]);
               V3 light_dir = normalize(lpos - origin);                  // rgen:113
               float distance_to_light = length(lpos - origin);          // rgen:114
               trace_ray(o, view, origin, light_dir, tmin, tmax, shadowRayPayload, UH_RAY_LIGHT_SHADOW);  // rgen:115
               if (shadowRayPayload.distance > distance_to_light || shadowRayPayload.distance == -1.0f) {  // rgen:118-119
                  float f = target_function(o, light_index, origin) * light_sample_weight;                 // rgen:121
                  radiance = radiance + throughput * f;
               }
            }
         }
      }
      pixelColor = pixelColor + radiance;                                // rgen:127
   }
   size_t pi = (size_t)py * W + px;
   V3 acc = v3(0, 0, 0);
   if (view.total_samples != view.samples_per_frame)                      // rgen:131-134
      acc = v3(o.accumulation[pi * 4 + 0], o.accumulation[pi * 4 + 1], o.accumulation[pi * 4 + 2]);
   if (view.total_samples <= view.accumulation_limit) acc = acc + pixelColor;  // rgen:136-138
   o.accumulation[pi * 4 + 0] = acc.x;                                   // rgen:143
   o.accumulation[pi * 4 + 1] = acc.y;
   o.accumulation[pi * 4 + 2] = acc.z;
   o.accumulation[pi * 4 + 3] = 0.0f;
   resolve_pixel(o, view.total_samples, view.accumulation_limit, pi);    // rgen:140-144
}

// ------------------------------------------------------------------------------------------
// A13 — G-buffer position (renderers/gbuffer.rs:11-52, gbuffer.vert:29-46, gbuffer.frag:47):
// world position of the primary-visible surface at each pixel centre, clear colour (1,1,1,0)
// (pass.rs:210-214). Produced here by an un-jittered primary ray instead of rasterisation.
// ------------------------------------------------------------------------------------------
static void gbuffer_pixel(Oracle& o, const UhViewUniformData& view, uint32_t px, uint32_t py, bool counted = true) {
   V3 org, dir;
   primary_ray(view, o.W, o.H, px, py, 0.5f, 0.5f, org, dir);
   if (counted) tl_ctr.rays[UH_RAY_GBUFFER]++;  // a rank counts the rays of its own band, not the rows it casts again for its neighbourhood
   Hit h = trace_closest(o, org, dir, 0.001f, 10000.0f, false);
   float* g = &o.gbuffer_pos[((size_t)py * o.W + px) * 4];
   if (h.mesh != 0xffffffffu) {
      V3 p = org + h.t * dir;
      g[0] = p.x;
      g[1] = p.y;
      g[2] = p.z;
      g[3] = 1.0f;
   } else {
      g[0] = g[1] = g[2] = 1.0f;
      g[3] = 0.0f;
   }
}
// texture(in_gbuffer_position, vec2(px)/vec2(size)) with a LINEAR + MIRRORED_REPEAT sampler
// (initial_ris.rgen:22-23): uv is the texel corner, so the value is the mean of texels
// (x-1..x, y-1..y) with index -1 mirrored to 0. Contract: ((a+b)+(c+d))*0.25 per channel.
static inline V3 gbuffer_fetch(const Oracle& o, uint32_t px, uint32_t py) {
   uint32_t x0 = px == 0 ? 0 : px - 1, y0 = py == 0 ? 0 : py - 1;
   auto tx = [&](uint32_t x, uint32_t y) {
      const float* g = &o.gbuffer_pos[((size_t)y * o.W + x) * 4];
      return v3(g[0], g[1], g[2]);
   };
   V3 a = tx(x0, y0), b = tx(px, y0), c = tx(x0, py), d = tx(px, py);
   return ((a + b) + (c + d)) * 0.25f;
}

// ------------------------------------------------------------------------------------------
// A6-A9 — ReSTIR passes
// ------------------------------------------------------------------------------------------
// restir/reset_reservoirs.comp:24-45 (spatial is NOT reset: it is the temporal history)
static void reset_pixel(Oracle& o, uint32_t px, uint32_t py) {
   size_t i = (size_t)py * o.W + px;
   o.reservoirs[0][i] = UhReservoir{-1, 0.0f, 0.0f, 0};
   o.reservoirs[1][i] = UhReservoir{-1, 0.0f, 0.0f, 0};
}
// restir/initial_ris.rgen:19-39
static void initial_ris_pixel(Oracle& o, const UhViewUniformData& view, uint32_t px, uint32_t py) {
   uint32_t rng = initRNG(px, py, o.W, frameNumber(view));
   V3 hit_position = gbuffer_fetch(o, px, py);
   UhReservoir nr = {-1, 0.0f, 0.0f, 0};
   UhReservoir r = resample(o, view, rng, hit_position);
   updateReservoir(rng, nr, r.Y, r.W_sum * (float)r.M, r.M);
   float p_hat = target_function(o, nr.Y, hit_position);
   finalize_resampling(nr, p_hat);
   o.reservoirs[0][(size_t)py * o.W + px] = nr;
}
// restir/temporal_reuse.rgen:35-119
static void temporal_pixel(Oracle& o, const UhViewUniformData& view, uint32_t px, uint32_t py) {
   const uint32_t W = o.W, H = o.H;
   size_t index = (size_t)py * W + px;
   uint32_t rng = initRNG(px, py, W, frameNumber(view));
   V3 hit_position = gbuffer_fetch(o, px, py);
   if (view.temporal_reuse_enabled == 0) {
      o.reservoirs[1][index] = o.reservoirs[0][index];
      return;
   }
   UhReservoir nr = {-1, 0.0f, 0.0f, 0};
   UhReservoir ir = o.reservoirs[0][index];
   float p_hat = target_function(o, ir.Y, hit_position);
   float initial_weight = p_hat * ir.W_X * (float)ir.M;
   updateReservoir(rng, nr, ir.Y, initial_weight, ir.M);
   UhReservoir pr = {-1, 0.0f, 0.0f, 0};
   V4 puv = mat4_mul(view.prev_frame_projection_view, V4{hit_position.x, hit_position.y, hit_position.z, 1.0f});
   float ux = puv.x / puv.w, uy = puv.y / puv.w;
   ux = ux * 0.5f + 0.5f;
   uy = uy * 0.5f + 0.5f;
   uy = 1.0f - uy;
   if (ux >= 0.0f && ux <= 1.0f && uy >= 0.0f && uy <= 1.0f) {
      int32_t ix = (int32_t)(ux * (float)W + 0.5f), iy = (int32_t)(uy * (float)H + 0.5f);
      // temporal_reuse.rgen:97: uint index = y*W + x may be one past the end (y == H) — clamped here
      uint32_t ti = (uint32_t)iy * W + (uint32_t)ix;
      if (ti > W * H - 1) ti = W * H - 1;
      pr = o.reservoirs[2][ti];  // prev frame = last frame's spatial_reuse_reservoirs (mod.rs:294)
   }
   p_hat = pr.Y == -1 ? 0.0f : target_function(o, pr.Y, hit_position);
   pr.M = std::min(20 * ir.M, pr.M);
   float prev_weight = p_hat * pr.W_X * (float)pr.M;
   updateReservoir(rng, nr, pr.Y, prev_weight, pr.M);
   if (nr.Y != -1) finalize_resampling(nr, target_function(o, nr.Y, hit_position));
   o.reservoirs[1][index] = nr;
}
// restir/spatial_reuse.rgen:23-73. Writes into `out` (the pass reads temporal, writes spatial).
static void spatial_pixel(Oracle& o, const UhViewUniformData& view, uint32_t px, uint32_t py) {
   const uint32_t W = o.W, H = o.H;
   size_t index = (size_t)py * W + px;
   uint32_t rng = initRNG(px, py, W, frameNumber(view));
   V3 hit_position = gbuffer_fetch(o, px, py);
   if (view.spatial_reuse_enabled == 0) {
      o.reservoirs[2][index] = o.reservoirs[1][index];
      return;
   }
   UhReservoir nr = {-1, 0.0f, 0.0f, 0};
   UhReservoir tr = o.reservoirs[1][index];
   float p_hat = target_function(o, tr.Y, hit_position);
   updateReservoir(rng, nr, tr.Y, p_hat * tr.W_X * (float)tr.M, tr.M);
   for (int i = 0; i < 5; i++) {
      float ox = randomFloat(rng) * 2.0f - 1.0f, oy = randomFloat(rng) * 2.0f - 1.0f;
      ox *= 30.0f;
      oy *= 30.0f;
      // uvec2(offset) of a negative float is undefined in GLSL; pinned as (uint)(int)trunc(x)
      uint32_t nx = px + (uint32_t)(int32_t)ox, ny = py + (uint32_t)(int32_t)oy;
      nx = std::min(nx, W - 1);  // clamp(uvec2, 0, size-1): a wrapped-negative coordinate clamps to size-1
      ny = std::min(ny, H - 1);
      UhReservoir nb = o.reservoirs[1][(size_t)ny * W + nx];
      float ph = target_function(o, nb.Y, hit_position);
      updateReservoir(rng, nr, nb.Y, ph * nb.W_X * (float)nb.M, nb.M);
   }
   if (nr.Y != -1) finalize_resampling(nr, target_function(o, nr.Y, hit_position));
   o.reservoirs[2][index] = nr;
}

static void flush_counters(Oracle& o) {
   for (int i = 0; i < UH_RAY_KINDS; i++) o.ctr.rays[i] += tl_ctr.rays[i];
   o.ctr.nodes += tl_ctr.nodes;
   o.ctr.tris += tl_ctr.tris;
   o.ctr.closest_hits += tl_ctr.closest_hits;
   o.ctr.misses += tl_ctr.misses;
   tl_ctr = LocalCounters();
}

// the cores this process may actually run on (a container's CPU set), not every core of the host
static int default_threads() {
   int n = (int)std::thread::hardware_concurrency();
#ifdef __linux__
   cpu_set_t set;
   if (sched_getaffinity(0, sizeof(set), &set) == 0) {
      int c = CPU_COUNT(&set);
      if (c > 0 && c < n) n = c;
   }
#endif
   return n > 64 ? 64 : n;
}

template <typename F>
static void parallel_rows(Oracle& o, F f, const std::vector<uint8_t>* rows = nullptr) {
   int nt = o.num_threads > 0 ? o.num_threads : default_threads();
   if (nt < 1) nt = 1;
   std::atomic<uint32_t> next(0);
   auto worker = [&]() {
      for (;;) {
         uint32_t y = next.fetch_add(1);
         if (y >= o.H) break;
         if (rows && !(*rows)[y]) continue;
         for (uint32_t x = 0; x < o.W; x++) f(x, y);
         flush_counters(o);
      }
   };
   if (nt == 1) {
      worker();
      return;
   }
   std::vector<std::thread> th;
   for (int i = 0; i < nt; i++) th.emplace_back(worker);
   for (auto& t : th) t.join();
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C interface for ctypes (mirrors include/utopian_hip.h so the parity tests call both the same way)
// ------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------
// N3 — marching cubes (utopian/shaders/marching_cubes/marching_cubes.comp), the checker of csrc/isosurface.hip.
// Runs on the reference's own lookup tables, handed in as data (tests/golden/mc_reference_tables.npz: edgeTable[256],
// triangleTable[256][16] of shaders/marching_cubes/tables.glsl:4-293). The density field is the shader's (:61-103), with
// the product's placement of the shapes in a [lo, hi]^3 domain: toOrigin(v) = v + (6, 0, 6), so that the torus sits at
// (16, 20, 16), the box at (16, 10, 16), the sphere at (16, 26, 16) - the convention of uh_add_isosurface_mesh.
// ------------------------------------------------------------------------------------------
namespace mc {
static inline float sdSphere(V3 p, float s) { return length(p) - s; }                         // marching_cubes.comp:59-62
static inline float sdTorus(V3 p, float tx, float ty) {                                       // :64-68
   const float qx = std::sqrt(p.x * p.x + p.z * p.z) - tx, qy = p.y;
   return std::sqrt(qx * qx + qy * qy) - ty;
}
static inline float sdBox(V3 p, V3 b) {                                                       // :77-81
   const V3 d = v3(std::fabs(p.x) - b.x, std::fabs(p.y) - b.y, std::fabs(p.z) - b.z);
   const V3 m = v3(std::fmax(d.x, 0.0f), std::fmax(d.y, 0.0f), std::fmax(d.z, 0.0f));
   return std::fmin(std::fmax(d.x, std::fmax(d.y, d.z)), 0.0f) + length(m);
}
// density(vec3) :92-119 = addShapes(pos, -1) :83-90; the noise branches are compiled out (#if 0) in the reference
static inline float density(V3 pos, float sphere_radius) {
   float d = std::fmax(-sdTorus(pos - v3(16.0f, 20.0f, 16.0f), 5.0f, 3.0f), -1.0f);
   d = std::fmax(-sdBox(pos - v3(16.0f, 10.0f, 16.0f), v3(5.0f, 5.0f, 5.0f)), d);
   d = std::fmax(-sdSphere(pos - v3(16.0f, 26.0f, 16.0f), sphere_radius), d);  // :87 (radius 8 |sin(0.3 time)|, 0 at time 0)
   return d;
}
// marching_cubes.rs:23-32: corner offsets of a voxel (x voxelSize)
static const int kCorner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
// the corner pairs of vertList[0..11], in the order main() interpolates them (:203-226)
static const int kEdgeCorners[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
}  // namespace mc

extern "C" {

struct orc_ctx {
   Oracle o;
};

int orc_create(uint32_t width, uint32_t height, orc_ctx** out) {
   if (!out || width == 0 || height == 0) return UH_ERR_INVALID_ARGUMENT;
   orc_ctx* c = new orc_ctx();
   c->o.W = width;
   c->o.H = height;
   size_t n = (size_t)width * height;
   c->o.accumulation.assign(n * 4, 0.0f);
   c->o.output.assign(n * 4, 0);
   c->o.gbuffer_pos.assign(n * 4, 0.0f);
   for (int i = 0; i < 3; i++) c->o.reservoirs[i].assign(n, UhReservoir{0, 0.0f, 0.0f, 0});
   for (auto& r : c->o.ctr.rays) r = 0;
   c->o.ctr.nodes = c->o.ctr.tris = c->o.ctr.closest_hits = c->o.ctr.misses = 0;
   *out = c;
   return UH_OK;
}
void orc_destroy(orc_ctx* c) { delete c; }

int orc_add_texture_rgba8(orc_ctx* c, const uint8_t* px, uint32_t w, uint32_t h, uint32_t* out_index) {
   if (!c || !px || !w || !h) return UH_ERR_INVALID_ARGUMENT;
   Texture t;
   t.w = w;
   t.h = h;
   t.px.assign(px, px + (size_t)w * h * 4);
   c->o.textures.push_back(std::move(t));
   if (out_index) *out_index = (uint32_t)c->o.textures.size() - 1;
   return UH_OK;
}
int orc_add_mesh(orc_ctx* c, const UhVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, const UhGpuMaterial* mat,
                 const float world3x4[12], uint32_t* out_mesh_index) {
   if (!c || !v || !idx || !mat || !world3x4 || ni % 3) return UH_ERR_INVALID_ARGUMENT;
   if (c->o.meshes.size() >= UH_MAX_GPU_MESHES) return UH_ERR_CAPACITY;
   for (uint32_t i = 0; i < ni; i++)
      if (idx[i] >= nv) return UH_ERR_INVALID_ARGUMENT;
   for (uint32_t i = 0; i < nv; i++)
      if (!std::isfinite(v[i].pos[0]) || !std::isfinite(v[i].pos[1]) || !std::isfinite(v[i].pos[2])) return UH_ERR_INVALID_ARGUMENT;
   for (int i = 0; i < 12; i++)
      if (!std::isfinite(world3x4[i])) return UH_ERR_INVALID_ARGUMENT;
   MeshRec m;
   m.vertices.assign(v, v + nv);
   m.indices.assign(idx, idx + ni);
   m.material = *mat;
   std::memcpy(m.o2w, world3x4, sizeof(m.o2w));
   if (is_identity3x4(m.o2w)) {
      static const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      std::memcpy(m.w2o, I, sizeof(I));
   } else
      invert3x3(m.o2w, m.w2o);
   m.first_tri = 0;
   c->o.meshes.push_back(std::move(m));
   c->o.built = c->o.ever_built = false;
   if (out_mesh_index) *out_mesh_index = (uint32_t)c->o.meshes.size() - 1;
   return UH_OK;
}
int orc_add_light(orc_ctx* c, const UhGpuLight* l, uint32_t* out_index) {
   if (!c || !l) return UH_ERR_INVALID_ARGUMENT;
   if (c->o.lights.size() >= UH_MAX_GPU_LIGHTS) return UH_ERR_CAPACITY;
   c->o.lights.push_back(*l);
   if (out_index) *out_index = (uint32_t)c->o.lights.size() - 1;
   return UH_OK;
}
int orc_set_instance_transform(orc_ctx* c, uint32_t mesh_index, const float world3x4[12]) {
   if (!c || mesh_index >= c->o.meshes.size() || !world3x4) return UH_ERR_INVALID_ARGUMENT;
   for (int i = 0; i < 12; i++)
      if (!std::isfinite(world3x4[i])) return UH_ERR_INVALID_ARGUMENT;
   MeshRec& m = c->o.meshes[mesh_index];
   std::memcpy(m.o2w, world3x4, sizeof(m.o2w));
   if (is_identity3x4(m.o2w)) {
      static const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      std::memcpy(m.w2o, I, sizeof(I));
   } else
      invert3x3(m.o2w, m.w2o);
   c->o.built = false;
   return UH_OK;
}
int orc_build_acceleration(orc_ctx* c) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   bake_triangles(c->o);
   build_bvh(c->o);
   c->o.built = c->o.ever_built = true;
   return UH_OK;
}
// raytracing.rs:400-459 rebuild_tlas. The oracle simply rebuilds: results do not depend on the tree.
int orc_refit_acceleration(orc_ctx* c) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   if (!c->o.ever_built) return UH_ERR_NOT_BUILT;
   return orc_build_acceleration(c);
}
// option names: "brute_force", "threads", "full_frame_restir", "furnace"
int orc_set_option(orc_ctx* c, const char* name, int value) {
   if (!c || !name) return UH_ERR_INVALID_ARGUMENT;
   std::string n(name);
   if (n == "brute_force")
      c->o.brute_force = value != 0;
   else if (n == "threads")
      c->o.num_threads = value;
   else if (n == "full_frame_restir")
      c->o.full_frame_restir = value != 0;
   else if (n == "furnace")
      c->o.furnace = value != 0;
   else if (n == "count_visits" || n == "time_kernels")
      ;
   else
      return UH_ERR_INVALID_ARGUMENT;
   return UH_OK;
}
int orc_set_tile_partition(orc_ctx* c, uint32_t rank, uint32_t world, uint32_t tile) {
   if (!c || world == 0 || rank >= world || tile == 0) return UH_ERR_INVALID_ARGUMENT;
   c->o.tp_rank = rank;
   c->o.tp_world = world;
   c->o.tp_tile = tile;
   return UH_OK;
}

// The reservoir passes of one rank of a job that partitions them by rows (the product's uh_set_restir_partition, restated from
// what the shaders read rather than from the product's interval arithmetic): the spatial pass runs on the rank's band; it
// gathers temporal reservoirs from rows y + dy, |dy| < 30, a negative row wrapping to the last one (spatial_reuse.rgen:48-56
// with the uvec2 conversion pinned in DESIGN.md section 2); temporal, initial and reset run on those rows; each of them reads
// the G-buffer at its row and the one above (initial_ris.rgen:22-23 through the LINEAR sampler).
int orc_set_restir_partition(orc_ctx* c, uint32_t rank, uint32_t world, UhRestirExchangeFn exchange, void* user) {
   if (!c || world == 0 || rank >= world) return UH_ERR_INVALID_ARGUMENT;
   Oracle& o = c->o;
   o.rp_rank = rank;
   o.rp_world = world;
   o.rp_exchange = world > 1 ? exchange : nullptr;
   o.rp_user = user;
   const uint32_t band = (o.H + world - 1) / world;
   const size_t need = std::max((size_t)o.W * o.H, (size_t)band * world * o.W);  // the frame padded to equal bands: what an in-place all-gather addresses
   if (o.reservoirs[2].size() < need) o.reservoirs[2].resize(need, UhReservoir{0, 0.0f, 0.0f, 0});
   return UH_OK;
}
struct RestirRowMasks {
   std::vector<uint8_t> band, reuse, cast;
   uint32_t rows_per_band = 0;
};
static RestirRowMasks restir_row_masks(const Oracle& o) {
   RestirRowMasks m;
   const uint32_t H = o.H;
   m.rows_per_band = (H + o.rp_world - 1) / o.rp_world;
   m.band.assign(H, 0);
   m.reuse.assign(H, 0);
   m.cast.assign(H, 0);
   for (uint32_t y = 0; y < H; y++) {
      if (y / m.rows_per_band != o.rp_rank) continue;
      m.band[y] = 1;
      for (int dy = -30; dy <= 30; dy++) {
         const int64_t ny = (int64_t)y + dy;
         m.reuse[ny < 0 || ny > (int64_t)H - 1 ? H - 1 : (uint32_t)ny] = 1;
      }
   }
   for (uint32_t y = 0; y < H; y++)
      if (m.reuse[y]) m.cast[y] = m.cast[y ? y - 1 : 0] = 1;
   return m;
}
int orc_get_restir_rows(orc_ctx* c, UhRestirRows* out) {
   if (!c || !out) return UH_ERR_INVALID_ARGUMENT;
   const Oracle& o = c->o;
   const RestirRowMasks m = restir_row_masks(o);
   *out = UhRestirRows{};
   out->rows_per_band = o.rp_world > 1 ? m.rows_per_band : o.H;
   auto interval = [&](const std::vector<uint8_t>& rows, uint32_t& r0, uint32_t& n, uint32_t& e0, uint32_t& en) {
      uint32_t y = 0;
      while (y < o.H && !rows[y]) y++;
      r0 = y < o.H ? y : 0;
      while (y < o.H && rows[y]) y++, n++;
      while (y < o.H && !rows[y]) y++;
      e0 = y < o.H ? y : 0;
      while (y < o.H && rows[y]) y++, en++;
   };
   uint32_t none0 = 0, none = 0;
   interval(m.band, out->band_row0, out->band_rows, none0, none);
   interval(m.reuse, out->reuse_row0, out->reuse_rows, out->reuse_extra_row0, out->reuse_extra_rows);
   interval(m.cast, out->cast_row0, out->cast_rows, out->cast_extra_row0, out->cast_extra_rows);
   return UH_OK;
}

// renderers/mod.rs:246-358 pass order; frame protocol of prototype/src/main.rs:460-471
int orc_render_frame(orc_ctx* c, const UhViewUniformData* view, uint32_t pass_mask) {
   if (!c || !view) return UH_ERR_INVALID_ARGUMENT;
   Oracle& o = c->o;
   if (!o.built && o.ever_built && view->rebuild_tlas == 1) orc_build_acceleration(c);  // main.rs:392,526
   if (!o.built) return UH_ERR_NOT_BUILT;
   const UhViewUniformData v = *view;
   if (o.rp_world > 1 && (pass_mask & UH_PASS_RESTIR)) {
      const RestirRowMasks m = restir_row_masks(o);
      if (pass_mask & UH_PASS_GBUFFER) parallel_rows(o, [&](uint32_t x, uint32_t y) { gbuffer_pixel(o, v, x, y, m.band[y] != 0); }, &m.cast);
      if (pass_mask & UH_PASS_RESET_RESERVOIRS) parallel_rows(o, [&](uint32_t x, uint32_t y) { reset_pixel(o, x, y); }, &m.reuse);
      if (pass_mask & UH_PASS_INITIAL_RIS) parallel_rows(o, [&](uint32_t x, uint32_t y) { initial_ris_pixel(o, v, x, y); }, &m.reuse);
      if (pass_mask & UH_PASS_TEMPORAL_REUSE) parallel_rows(o, [&](uint32_t x, uint32_t y) { temporal_pixel(o, v, x, y); }, &m.reuse);
      if (pass_mask & UH_PASS_SPATIAL_REUSE) {
         parallel_rows(o, [&](uint32_t x, uint32_t y) { spatial_pixel(o, v, x, y); }, &m.band);
         // the other ranks' bands: an in-place all-gather over equal bands, done by the caller (synchronously: there is no stream here)
         if (o.rp_exchange)
            if (o.rp_exchange(o.rp_user, nullptr, o.reservoirs[2].data(), (uint64_t)m.rows_per_band * o.W * sizeof(UhReservoir), o.rp_rank, o.rp_world) != 0) return UH_ERR_HIP;
      }
   } else {
      if (pass_mask & UH_PASS_GBUFFER) parallel_rows(o, [&](uint32_t x, uint32_t y) { gbuffer_pixel(o, v, x, y); });
      if (pass_mask & UH_PASS_RESET_RESERVOIRS) parallel_rows(o, [&](uint32_t x, uint32_t y) { reset_pixel(o, x, y); });
      if (pass_mask & UH_PASS_INITIAL_RIS) parallel_rows(o, [&](uint32_t x, uint32_t y) { initial_ris_pixel(o, v, x, y); });
      if (pass_mask & UH_PASS_TEMPORAL_REUSE) parallel_rows(o, [&](uint32_t x, uint32_t y) { temporal_pixel(o, v, x, y); });
      // the spatial pass reads temporal_reuse_reservoirs and writes spatial_reuse_reservoirs: no hazard
      if (pass_mask & UH_PASS_SPATIAL_REUSE) parallel_rows(o, [&](uint32_t x, uint32_t y) { spatial_pixel(o, v, x, y); });
   }
   if (pass_mask & UH_PASS_REFERENCE_PT)
      parallel_rows(o, [&](uint32_t x, uint32_t y) {
         if (owns_pixel(o, x, y)) raygen_pixel(o, v, x, y);
      });
   return UH_OK;
}
int orc_reset_accumulation(orc_ctx* c) {
   if (!c) return UH_ERR_INVALID_ARGUMENT;
   std::fill(c->o.accumulation.begin(), c->o.accumulation.end(), 0.0f);
   std::fill(c->o.output.begin(), c->o.output.end(), 0);
   return UH_OK;
}
int orc_read_accumulation(orc_ctx* c, float* out) {
   std::memcpy(out, c->o.accumulation.data(), c->o.accumulation.size() * 4);
   return UH_OK;
}
int orc_read_output_bgra8(orc_ctx* c, uint8_t* out) {
   std::memcpy(out, c->o.output.data(), c->o.output.size());
   return UH_OK;
}
int orc_read_reservoirs(orc_ctx* c, int which, UhReservoir* out) {
   if (which < 0 || which > 2) return UH_ERR_INVALID_ARGUMENT;
   std::memcpy(out, c->o.reservoirs[which].data(), (size_t)c->o.W * c->o.H * sizeof(UhReservoir));
   return UH_OK;
}
int orc_write_reservoirs(orc_ctx* c, int which, const UhReservoir* in) {
   if (which < 0 || which > 2) return UH_ERR_INVALID_ARGUMENT;
   std::memcpy(c->o.reservoirs[which].data(), in, (size_t)c->o.W * c->o.H * sizeof(UhReservoir));
   return UH_OK;
}
// test hook (the twin of uh_write_gbuffer_position): the G-buffer the reservoir passes read, set from outside - the known-answer
// chains of tests/golden/shading_kat.npz run the passes on synthetic positions, without a cast
int orc_write_gbuffer_position(orc_ctx* c, const float* in) {
   if (!c || !in) return UH_ERR_INVALID_ARGUMENT;
   std::memcpy(c->o.gbuffer_pos.data(), in, c->o.gbuffer_pos.size() * sizeof(float));
   return UH_OK;
}
int orc_read_gbuffer_position(orc_ctx* c, float* out) {
   std::memcpy(out, c->o.gbuffer_pos.data(), c->o.gbuffer_pos.size() * 4);
   return UH_OK;
}
int orc_resolve_output(orc_ctx* c, uint32_t total_samples, uint32_t limit) {
   Oracle& o = c->o;
   parallel_rows(o, [&](uint32_t x, uint32_t y) { resolve_pixel(o, total_samples, limit, (size_t)y * o.W + x); });
   return UH_OK;
}
int orc_trace_closest(orc_ctx* c, const float* rays, uint32_t n, float* out_tuv, uint32_t* out_mesh, uint32_t* out_prim) {
   if (!c || !c->o.built) return UH_ERR_NOT_BUILT;
   Oracle& o = c->o;
   int nt = o.num_threads > 0 ? o.num_threads : default_threads();
   std::atomic<uint32_t> next(0);
   auto worker = [&]() {
      for (;;) {
         uint32_t b = next.fetch_add(256);
         if (b >= n) break;
         for (uint32_t i = b; i < std::min(n, b + 256); i++) {
            const float* r = rays + (size_t)i * 8;
            Hit h = trace_closest(o, v3(r[0], r[1], r[2]), v3(r[4], r[5], r[6]), r[3], r[7], false);
            bool hit = h.mesh != 0xffffffffu;
            out_tuv[i * 3 + 0] = hit ? h.t : -1.0f;
            out_tuv[i * 3 + 1] = hit ? h.u : 0.0f;
            out_tuv[i * 3 + 2] = hit ? h.v : 0.0f;
            out_mesh[i] = h.mesh;
            out_prim[i] = h.prim;
         }
      }
   };
   std::vector<std::thread> th;
   for (int i = 0; i < std::max(1, nt); i++) th.emplace_back(worker);
   for (auto& t : th) t.join();
   return UH_OK;
}
int orc_trace_any(orc_ctx* c, const float* rays, uint32_t n, uint8_t* out_occluded) {
   if (!c || !c->o.built) return UH_ERR_NOT_BUILT;
   Oracle& o = c->o;
   for (uint32_t i = 0; i < n; i++) {
      const float* r = rays + (size_t)i * 8;
      Hit h = trace_closest(o, v3(r[0], r[1], r[2]), v3(r[4], r[5], r[6]), r[3], r[7], false);
      out_occluded[i] = h.mesh != 0xffffffffu;
   }
   return UH_OK;
}
int orc_get_stats(orc_ctx* c, UhStats* out) {
   std::memset(out, 0, sizeof(*out));
   for (int i = 0; i < UH_RAY_KINDS; i++) out->rays[i] = c->o.ctr.rays[i];
   out->nodes_visited = c->o.ctr.nodes;
   out->tris_tested = c->o.ctr.tris;
   out->closest_hits = c->o.ctr.closest_hits;
   out->misses = c->o.ctr.misses;
   out->bvh_nodes = (uint32_t)c->o.nodes.size();
   out->bvh_triangles = (uint32_t)c->o.tris.size();
   return UH_OK;
}
int orc_reset_stats(orc_ctx* c) {
   for (auto& r : c->o.ctr.rays) r = 0;
   c->o.ctr.nodes = c->o.ctr.tris = c->o.ctr.closest_hits = c->o.ctr.misses = 0;
   return UH_OK;
}
int orc_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

// marching_cubes.comp:179-254 main(), for every voxel of a res^3 grid over [lo, hi]^3, voxels in x-fastest order.
// Per voxel: cubeIndex (:185-190: bit i set when density(corner i) < 0), the crossed edges from edgeTable (:201-226), their
// vertices by vertexInterp (:134-137: mix(p1, p2, (iso - v1) / (v2 - v1)) in the order the shader names the corners), the
// triangles of triangleTable up to the first -1 (:231-251). Nothing is dropped or reordered.
//   order = 0: the shader's own corner order per edge; 1: the endpoint with the smaller grid index first - what
//   isosurface.hip does so that the cells sharing an edge produce the same bits (its one arithmetic difference; the two
//   agree to a rounding of the interpolation).
// Outputs (each may be null): cube_index[res^3]; tri_count[res^3]; positions: 9 floats per triangle, `cap_triangles` at most.
// Returns the number of triangles the grid yields (also when it exceeds the capacity: then only the first are written).
uint64_t orc_marching_cubes(uint32_t res, float lo, float hi, float time, const int32_t* edge_table, const int32_t* triangle_table, int order, uint8_t* cube_index,
                            uint8_t* tri_count, float* positions, uint64_t cap_triangles) {
   using namespace mc;
   const float h = (hi - lo) / (float)res;
   const float radius = 8.0f * std::fabs(std::sin(time * 0.3f));  // :87
   uint64_t total = 0;
   for (uint32_t iz = 0; iz < res; iz++)
      for (uint32_t iy = 0; iy < res; iy++)
         for (uint32_t ix = 0; ix < res; ix++) {
            const uint64_t cell = ((uint64_t)iz * res + iy) * res + ix;
            V3 p[8];
            float v[8];
            int cube = 0;
            for (int i = 0; i < 8; i++) {
               p[i] = v3(lo + h * (float)(ix + kCorner[i][0]), lo + h * (float)(iy + kCorner[i][1]), lo + h * (float)(iz + kCorner[i][2]));
               v[i] = density(p[i], radius);
               if (v[i] < 0.0f) cube |= 1 << i;  // isoLevel = 0 (:182)
            }
            if (cube_index) cube_index[cell] = (uint8_t)cube;
            uint32_t n = 0;
            if (edge_table[cube] != 0) {
               V3 vert[12];
               for (int e = 0; e < 12; e++) {
                  if (!(edge_table[cube] & (1 << e))) continue;
                  int a = kEdgeCorners[e][0], b = kEdgeCorners[e][1];
                  if (order == 1) {
                     // smaller grid index first: along the edge's axis the corner with offset 0
                     const int sa = kCorner[a][0] + kCorner[a][1] + kCorner[a][2], sb = kCorner[b][0] + kCorner[b][1] + kCorner[b][2];
                     if (sb < sa) std::swap(a, b);
                  }
                  const float t = (0.0f - v[a]) / (v[b] - v[a]);                          // vertexInterp :136
                  vert[e] = v3(p[a].x + t * (p[b].x - p[a].x), p[a].y + t * (p[b].y - p[a].y), p[a].z + t * (p[b].z - p[a].z));  // mix(x, y, a) = x + a (y - x)
               }
               for (int i = 0; triangle_table[16 * cube + i] != -1; i += 3, n++) {
                  if (positions && total + n < cap_triangles) {
                     float* o = positions + 9 * (total + n);
                     for (int k = 0; k < 3; k++) {
                        const V3 q = vert[triangle_table[16 * cube + i + k]];
                        o[3 * k] = q.x;
                        o[3 * k + 1] = q.y;
                        o[3 * k + 2] = q.z;
                     }
                  }
               }
            }
            if (tri_count) tri_count[cell] = (uint8_t)n;
            total += n;
         }
   return total;
}
// generateNormal (marching_cubes.comp:160-177): central differences of the density at distance d = 1, negated and normalised
void orc_mc_normal(const float p[3], float time, float out[3]) {
   const float r = 8.0f * std::fabs(std::sin(time * 0.3f)), d = 1.0f / 1.0f;
   const V3 q = v3(p[0], p[1], p[2]);
   const V3 g = v3(mc::density(q + v3(d, 0, 0), r) - mc::density(q + v3(-d, 0, 0), r), mc::density(q + v3(0, d, 0), r) - mc::density(q + v3(0, -d, 0), r),
                   mc::density(q + v3(0, 0, d), r) - mc::density(q + v3(0, 0, -d), r));
   const V3 n = neg(normalize(g));
   out[0] = n.x;
   out[1] = n.y;
   out[2] = n.z;
}
float orc_mc_density(float x, float y, float z, float time) { return mc::density(v3(x, y, z), 8.0f * std::fabs(std::sin(time * 0.3f))); }

// ---- unit entry points for the known-answer tests ---------------------------------------
uint32_t orc_jenkins_hash(uint32_t x) { return jenkinsHash(x); }
uint32_t orc_init_rng(uint32_t px, uint32_t py, uint32_t resx, uint32_t frame) { return initRNG(px, py, resx, frame); }
float orc_random_float(uint32_t* state) { return randomFloat(*state); }
void orc_random_point_in_unit_sphere(uint32_t* state, float out[3]) {
   V3 p = randomPointInUnitSphere(*state);
   out[0] = p.x;
   out[1] = p.y;
   out[2] = p.z;
}
uint32_t orc_frame_number(const UhViewUniformData* v) { return frameNumber(*v); }
void orc_offset_ray(const float p[3], const float n[3], float out[3]) {
   V3 r = offsetRay(v3(p[0], p[1], p[2]), v3(n[0], n[1], n[2]));
   out[0] = r.x;
   out[1] = r.y;
   out[2] = r.z;
}
float orc_linear_to_srgb(float x) { return linearToSrgb1(x); }
float orc_luminance(const float rgb[3]) { return luminance(v3(rgb[0], rgb[1], rgb[2])); }
void orc_sky(const float o[3], const float d[3], const float sun[3], float out[3]) {
   V3 c = sky::IntegrateScattering(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), 999999999.0f, normalize(v3(sun[0], sun[1], sun[2])), v3(1, 1, 1));
   out[0] = c.x;
   out[1] = c.y;
   out[2] = c.z;
}
float orc_target_function(orc_ctx* c, int light_index, const float p[3]) { return target_function(c->o, light_index, v3(p[0], p[1], p[2])); }
void orc_primary_ray(const UhViewUniformData* v, uint32_t W, uint32_t H, uint32_t px, uint32_t py, float jx, float jy, float out[6]) {
   V3 o, d;
   primary_ray(*v, W, H, px, py, jx, jy, o, d);
   out[0] = o.x;
   out[1] = o.y;
   out[2] = o.z;
   out[3] = d.x;
   out[4] = d.y;
   out[5] = d.z;
}
void orc_sample_texture(orc_ctx* c, uint32_t tex, float u, float v, float out[3]) {
   V3 r = sample_texture(c->o.textures[tex], u, v);
   out[0] = r.x;
   out[1] = r.y;
   out[2] = r.z;
}
// closest-hit shader on an explicit hit (mesh, prim, u, v, t) with an explicit seed: returns the payload
void orc_closest_hit_shader(orc_ctx* c, uint32_t mesh, uint32_t prim, float t, float u, float v, const float dir[3], uint32_t* seed,
                            float out[11]) {
   Hit h{t, u, v, mesh, prim};
   Payload pl;
   std::memset(&pl, 0, sizeof(pl));
   pl.seed = *seed;
   closest_hit_shader(c->o, h, v3(dir[0], dir[1], dir[2]), pl);
   *seed = pl.seed;
   float r[11] = {pl.color.x, pl.color.y, pl.color.z, pl.distance, pl.scatter.x, pl.scatter.y, pl.scatter.z, pl.scattered, pl.normal.x, pl.normal.y, pl.normal.z};
   std::memcpy(out, r, sizeof(r));
}

}  // extern "C"
