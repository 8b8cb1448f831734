// device_math.h — gfx950 device functions for the shading side of the path: RNG, view helpers,
// sky, textures, closest-hit material evaluation, reservoir math. Each function cites the GLSL it
// replaces (paths relative to the reference's utopian/shaders/). Arithmetic follows DESIGN.md
// "Arithmetic contract": plain IEEE f32 ops in the written order (the library is compiled with
// -ffp-contract=off), fused multiply-adds only where fmaf() is spelled out, correctly rounded
// division and sqrt; only the sky's exp/pow use the hardware approximations.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace uh {

struct V3 {
   float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 vneg(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float length3(V3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ V3 normalize3(V3 a) {
   float inv = 1.0f / sqrtf(dot3(a, a));
   return a * inv;
}
__device__ __forceinline__ V3 xyz(float4 v) { return v3(v.x, v.y, v.z); }

// column-major mat4 * vec4: ((c0*x + c1*y) + c2*z) + c3*w
__device__ __forceinline__ float4 mat4_mul(const float* m, float x, float y, float z, float w) {
   float4 r;
   r.x = ((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w;
   r.y = ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w;
   r.z = ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w;
   r.w = ((m[3] * x + m[7] * y) + m[11] * z) + m[15] * w;
   return r;
}

// ---- include/random.glsl -----------------------------------------------------------------
__device__ __forceinline__ uint32_t jenkins_hash(uint32_t x) {  // random.glsl:5-12
   x += x << 10;
   x ^= x >> 6;
   x += x << 3;
   x ^= x >> 11;
   x += x << 15;
   return x;
}
__device__ __forceinline__ uint32_t init_rng(uint32_t px, uint32_t py, uint32_t resx, uint32_t frame) {  // random.glsl:14-18
   float d = (float)px * 1.0f + (float)py * (float)resx;
   return jenkins_hash((uint32_t)d ^ jenkins_hash(frame));
}
__device__ __forceinline__ float random_float(uint32_t& s) {  // random.glsl:21-34
   s = s * 747796405u + 1u;
   uint32_t word = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
   word = (word >> 22) ^ word;
   return (float)word * 2.3283064365386963e-10f;  // exact: / 2^32 (4294967295.0f rounds to 2^32)
}
__device__ __forceinline__ V3 random_point_in_unit_sphere(uint32_t& s) {  // random.glsl:36-46
   for (;;) {
      float a = random_float(s), b = random_float(s), c = random_float(s);
      V3 p = v3(2.0f * a - 1.0f, 2.0f * b - 1.0f, 2.0f * c - 1.0f);
      if (dot3(p, p) < 1.0f) return p;
   }
}

// ---- include/view.glsl -------------------------------------------------------------------
__device__ __forceinline__ float luminance(V3 c) { return dot3(c, v3(0.2126f, 0.7152f, 0.0722f)); }  // view.glsl:46-50
__device__ __forceinline__ float linear_to_srgb(float c) {                                            // view.glsl:52-60
   if (c < 0.0031308f) return c * 12.92f;
   return 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
}
__device__ __forceinline__ V3 offset_ray(V3 p, V3 n) {  // view.glsl:92-108
   const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
   int ox = (int)(int_scale * n.x), oy = (int)(int_scale * n.y), oz = (int)(int_scale * n.z);
   float pix = __uint_as_float(__float_as_uint(p.x) + (uint32_t)((p.x < 0) ? -ox : ox));
   float piy = __uint_as_float(__float_as_uint(p.y) + (uint32_t)((p.y < 0) ? -oy : oy));
   float piz = __uint_as_float(__float_as_uint(p.z) + (uint32_t)((p.z < 0) ? -oz : oz));
   return v3(fabsf(p.x) < origin ? p.x + float_scale * n.x : pix, fabsf(p.y) < origin ? p.y + float_scale * n.y : piy,
             fabsf(p.z) < origin ? p.z + float_scale * n.z : piz);
}
__device__ __forceinline__ uint32_t unorm8(float x) {
   if (!(x > 0.0f)) x = 0.0f;
   if (x > 1.0f) x = 1.0f;
   return (uint32_t)rintf(x * 255.0f);
}

// ---- include/atmosphere.glsl ---------------------------------------------------------------
// exp/pow use the hardware exp2/log2 path (v_exp_f32 / v_log_f32): the sky is a smooth integrand,
// the ~1e-6 relative difference to libm is far inside the 1e-3 parity tolerance.
namespace sky {
constexpr float PLANET_RADIUS = 6371000.0f;
constexpr float ATMOSPHERE_HEIGHT = 100000.0f;
constexpr float RAYLEIGH_HEIGHT = ATMOSPHERE_HEIGHT * 0.08f;
constexpr float MIE_HEIGHT = ATMOSPHERE_HEIGHT * 0.012f;
constexpr float PI = 3.14159265359f;
__device__ __forceinline__ V3 C_RAYLEIGH() { return v3(5.802f, 13.558f, 33.100f) * 1e-6f; }
__device__ __forceinline__ V3 C_MIE() { return v3(3.996f, 3.996f, 3.996f) * 1e-6f; }
__device__ __forceinline__ V3 C_OZONE() { return v3(0.650f, 1.881f, 0.085f) * 1e-6f; }
__device__ __forceinline__ V3 PLANET_CENTER() { return v3(0.0f, -PLANET_RADIUS, 0.0f); }

// The sky is the one part of the path that is compared with a tolerance instead of bit for bit (device exp / pow
// already differ from libm by an ulp): inside it, divisions by constants are reciprocal multiplications and square
// roots / reciprocals use the 1-ulp hardware instructions. ~2.5x fewer instructions in k_shade_miss; the result moves
// by ~1e-6 relative (tests: per-pixel L2 <= 1e-3, sampled max |diff| <= 1e-4).
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ void atmosphere_intersection(V3 s, V3 d, float& t0, float& t1) {  // atmosphere.glsl:53-77
   const float radius = PLANET_RADIUS + ATMOSPHERE_HEIGHT;
   s = s - PLANET_CENTER();
   float a = dot3(d, d);
   float b = 2.0f * dot3(s, d);
   float c = dot3(s, s) - (radius * radius);
   float disc = b * b - 4.0f * a * c;
   if (disc < 0) {
      t0 = -1.0f;
      t1 = -1.0f;
   } else {
      disc = fast_sqrt(disc);
      const float inv = fast_rcp(2.0f * a);
      t0 = (-b - disc) * inv;
      t1 = (-b + disc) * inv;
   }
}
__device__ __forceinline__ float atmosphere_height(V3 p) {  // :95-98
   V3 q = p - PLANET_CENTER();
   return fast_sqrt(dot3(q, q)) - PLANET_RADIUS;
}
__device__ __forceinline__ V3 atmosphere_density(float h) {  // :99-115
   float r = __expf(-fmaxf(0.0f, h * (1.0f / RAYLEIGH_HEIGHT)));
   float m = __expf(-fmaxf(0.0f, h * (1.0f / MIE_HEIGHT)));
   float o = fmaxf(0.0f, 1.0f - fabsf(h - 25000.0f) * (1.0f / 15000.0f));
   return v3(r, m, o);
}
__device__ __forceinline__ V3 absorb(V3 od) {  // :146-150
   V3 a = (od.x * C_RAYLEIGH() + od.y * C_MIE() * 1.1f + od.z * C_OZONE()) * 1.0f;
   return v3(__expf(-a.x), __expf(-a.y), __expf(-a.z));
}
__device__ __forceinline__ V3 integrate_optical_depth(V3 start, V3 dir) {  // :123-143
   float t0, t1;
   atmosphere_intersection(start, dir, t0, t1);
   float step = t1 / 8.0f;
   V3 od = v3(0, 0, 0);
#pragma unroll 2
   for (int i = 0; i < 8; i++) {
      V3 p = start + dir * ((float)i + 0.5f) * step;
      od = od + atmosphere_density(atmosphere_height(p)) * step;
   }
   return od;
}
__device__ __noinline__ V3 integrate_scattering(V3 start, V3 dir, float ray_length, V3 light_dir) {  // :154-214 (lightColor = 1)
   float ray_height = atmosphere_height(start);
   float c = 1.0f - ray_height * (1.0f / ATMOSPHERE_HEIGHT);
   c = fminf(fmaxf(c, 0.0f), 1.0f);
   float exponent = 1.0f + c * 8.0f;
   float i0, i1;
   atmosphere_intersection(start, dir, i0, i1);
   ray_length = fminf(ray_length, i1);
   if (i0 > 0) {
      start = start + dir * i0;
      ray_length -= i0;
   }
   float costh = dot3(dir, light_dir);
   float phaseR = 3.0f * (1.0f + costh * costh) / (16.0f * PI);
   float g = fminf(0.85f, 0.9381f);
   float k = 1.55f * g - 0.55f * g * g * g;
   float kcosth = k * costh;
   float phaseM = (1.0f - k * k) / ((4.0f * PI) * (1.0f - kcosth) * (1.0f - kcosth));
   V3 od = v3(0, 0, 0), rayleigh = v3(0, 0, 0), mie = v3(0, 0, 0);
   float prev = 0.0f;
   // log2(i / 16), i = 1..15: pow(i / 16, exponent) = exp2(exponent * log2(i / 16)) - one transcendental instead of two (i = 0: 0)
   const float kLog2[16] = {0.0f,          -4.0f,         -3.0f,          -2.4150374993f, -2.0f,          -1.6780719051f, -1.4150374993f, -1.1926450779f,
                            -1.0f,         -0.8300749986f, -0.6780719051f, -0.5405683814f, -0.4150374993f, -0.2995602819f, -0.1926450779f, -0.0931094044f};
   for (int i = 0; i < 16; i++) {
      float ray_time = i == 0 ? 0.0f : __builtin_amdgcn_exp2f(exponent * kLog2[i]) * ray_length;
      float step = ray_time - prev;
      V3 p = start + dir * ray_time;
      V3 dens = atmosphere_density(atmosphere_height(p));
      od = od + dens * step;
      // viewTransmittance * lightTransmittance (atmosphere.glsl:203-206) = absorb(od_view) * absorb(od_light) = absorb(od_view + od_light):
      // absorb is exp of a linear form - three exponentials per step instead of six (the sky is compared with a tolerance, see above)
      V3 both_t = absorb(od + integrate_optical_depth(p, light_dir));
      rayleigh = rayleigh + both_t * phaseR * dens.x * step;
      mie = mie + both_t * phaseM * dens.y * step;
      prev = ray_time;
   }
   return (rayleigh * C_RAYLEIGH() + mie * C_MIE()) * v3(1, 1, 1) * 20.0f;
}
}  // namespace sky

// ---- explicit LDS reads of whole records ------------------------------------------------------
// `cond ? lds_table[i] : global_table[i]` - however it is written - ends as flat loads of a selected pointer, and a flat
// load takes the vector-memory path (the busiest unit of the shading kernel) even when it resolves to LDS. These read the
// LDS copy with ds_read instructions whatever the surrounding code looks like.
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ MeshShade lds_fetch(const MeshShade* lds) {
   const uint32_t at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const MeshShade*)lds;
   u4_t q[5];
   asm volatile(
      "ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\tds_read_b128 %3, %5 offset:48\n\t"
      "ds_read_b128 %4, %5 offset:64\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4])
      : "v"(at)
      : "memory");
   MeshShade m;
   static_assert(sizeof(m) == sizeof(q), "five quads");
   __builtin_memcpy(&m, q, sizeof(m));
   return m;
}
__device__ __forceinline__ TexInfo lds_fetch(const TexInfo* lds) {
   const uint32_t at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const TexInfo*)lds;
   struct {
      u4_t a;
      u2_t b;
   } q;
   asm volatile("ds_read_b128 %0, %2\n\tds_read_b64 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q.a), "=&v"(q.b) : "v"(at) : "memory");
   TexInfo t;
   static_assert(sizeof(TexInfo) == 24, "texture descriptor: a quad and a pair");
   __builtin_memcpy(&t, &q, sizeof(t));
   return t;
}

// ---- texture sampling (utopian/src/texture.rs:85-98: RGBA8 UNORM, LINEAR, MIRRORED_REPEAT) ---
__device__ __forceinline__ int mirror_index(int i, int n) {
   int period = 2 * n;
   int m = i % period;
   if (m < 0) m += period;
   return m < n ? m : period - 1 - m;
}
// lds_tex (may be null): the first n_lds_tex texture descriptors staged in LDS by the caller
// `pre` is called exactly once, right before the four texels are requested (or before an early return): what it requests
// is in flight TOGETHER with the texels. k_shade_hit asks for its scattered paths' queue positions there (a returning atomic) and,
// fused, for the sun grid's coarse cover: requested any earlier, the wait for a texture descriptor from global memory (vmcnt
// counts in order) made everything wait for them first - three round trips in a row instead of two.
template <typename Pre>
__device__ __forceinline__ V3 sample_texture_pre(const SceneDev& sc, const float* __restrict__ lut, uint32_t index, float u, float v, const TexInfo* lds_tex, uint32_t n_lds_tex,
                                                 Pre&& pre) {
   if (index >= sc.num_textures) {
      pre();
      return v3(1, 1, 1);
   }
   TexInfo t;
   if (lds_tex) {
      // unconditional LDS read + one-armed replacement: see the mesh record in k_shade_hit (no flat loads)
      t = lds_fetch(lds_tex + (index < n_lds_tex ? index : 0u));
      if (index >= n_lds_tex) t = sc.textures[index];
   } else {
      t = sc.textures[index];
   }
   float x = u * (float)t.w - 0.5f, y = v * (float)t.h - 0.5f;
   if (!(fabsf(x) < 1e9f) || !(fabsf(y) < 1e9f)) {
      pre();
      return v3(0, 0, 0);
   }
   float fx = floorf(x), fy = floorf(y);
   float ax = x - fx, ay = y - fy;
   int x0 = mirror_index((int)fx, (int)t.w), x1 = mirror_index((int)fx + 1, (int)t.w);
   int y0 = mirror_index((int)fy, (int)t.h), y1 = mirror_index((int)fy + 1, (int)t.h);
   // the four texel addresses first (branch-free: a branch per texel put a wait behind every fetch and the four
   // fetches ran one after the other), then the four fetches together, then the table look-ups
   const bool tiled = t.tiles_x != 0;
   auto address = [&](int xx, int yy) {
      const uint32_t in_tiles = ((uint32_t)((yy >> 3) * (int)t.tiles_x + (xx >> 3)) << 6) + (uint32_t)(((yy & 7) << 3) + (xx & 7));
      const uint32_t in_rows = (uint32_t)yy * t.w + (uint32_t)xx;
      return tiled ? in_tiles : in_rows;
   };
   const uchar4* q00 = t.texels + address(x0, y0);
   const uchar4* q10 = t.texels + address(x1, y0);
   const uchar4* q01 = t.texels + address(x0, y1);
   const uchar4* q11 = t.texels + address(x1, y1);
   // one block: as four C++ loads the compiler still made the third wait for the first (register reuse), and as loads
   // through the descriptor's generic pointer they were flat loads
   uint32_t w00, w10, w01, w11;
   pre();
   asm volatile(
      "global_load_dword %0, %4, off\n\tglobal_load_dword %1, %5, off\n\tglobal_load_dword %2, %6, off\n\tglobal_load_dword %3, %7, off\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(w00), "=&v"(w10), "=&v"(w01), "=&v"(w11)
      : "v"(q00), "v"(q10), "v"(q01), "v"(q11)
      : "memory");
   auto texel = [](uint32_t w) { return make_uchar4((unsigned char)(w & 0xffu), (unsigned char)((w >> 8) & 0xffu), (unsigned char)((w >> 16) & 0xffu), (unsigned char)(w >> 24)); };
   const uchar4 p00 = texel(w00), p10 = texel(w10), p01 = texel(w01), p11 = texel(w11);
   const V3 t00 = v3(lut[p00.x], lut[p00.y], lut[p00.z]), t10 = v3(lut[p10.x], lut[p10.y], lut[p10.z]);
   const V3 t01 = v3(lut[p01.x], lut[p01.y], lut[p01.z]), t11 = v3(lut[p11.x], lut[p11.y], lut[p11.z]);
   V3 a = t00 * (1.0f - ax) + t10 * ax;
   V3 b = t01 * (1.0f - ax) + t11 * ax;
   return a * (1.0f - ay) + b * ay;
}
__device__ __forceinline__ V3 sample_texture(const SceneDev& sc, const float* __restrict__ lut, uint32_t index, float u, float v, const TexInfo* lds_tex = nullptr,
                                             uint32_t n_lds_tex = 0) {
   return sample_texture_pre(sc, lut, index, u, v, lds_tex, n_lds_tex, [] {});
}

// ---- include/restir_sampling.glsl ------------------------------------------------------------
// lights: 2 float4 per light (pos, intensity). Out-of-range index (Y = -1, or idx = n when
// xi == 1.0 — both out-of-bounds reads in the reference): p_hat = 0.
template <typename LightPtr>
__device__ __forceinline__ float target_function(LightPtr lights, uint32_t num_lights, int light_index, V3 hit_position) {  // :59-69
   if (light_index < 0 || (uint32_t)light_index >= num_lights) return 0.0f;
   float4 lp = lights[2 * light_index], li = lights[2 * light_index + 1];
   float d = length3(xyz(lp) - hit_position);
   float d2 = d * d;  // pow(d, 2.0)
   return luminance(v3(li.x / d2, li.y / d2, li.z / d2));
}
__device__ __forceinline__ void sample_light_uniform(uint32_t num_used, uint32_t& rng, int& idx, float& w) {  // :71-77
   idx = (int)(random_float(rng) * (float)num_used);
   w = 1.0f / (float)num_used;
}
__device__ __forceinline__ void finalize_resampling(UhReservoir& r, float p_hat) {  // :79-82
   r.W_X = (p_hat == 0.0f) ? 0.0f : (1.0f / p_hat) * r.W_sum / (float)r.M;
}
__device__ __forceinline__ void update_reservoir(uint32_t& rng, UhReservoir& r, int Xi, float w_i, int M) {  // :85-94
   r.W_sum += w_i;
   r.M += M;
   if (random_float(rng) * r.W_sum < w_i) r.Y = Xi;
}

// ---- pathtrace_reference/reference.rgen:31-38 ---------------------------------------------
__device__ __forceinline__ void primary_ray(const FrameParams& fp, uint32_t px, uint32_t py, float jx, float jy, V3& org, V3& dir) {
   float cx = (float)px + jx, cy = (float)py + jy;
   float u = cx / (float)fp.W, v = cy / (float)fp.H;
   v = 1.0f - v;
   float dx = u * 2.0f - 1.0f, dy = v * 2.0f - 1.0f;
   float4 o4 = mat4_mul(fp.inv_view, 0.0f, 0.0f, 0.0f, 1.0f);
   float4 tg = mat4_mul(fp.inv_proj, dx, dy, 1.0f, 1.0f);
   V3 nt = normalize3(v3(tg.x, tg.y, tg.z));
   float4 d4 = mat4_mul(fp.inv_view, nt.x, nt.y, nt.z, 0.0f);
   org = v3(o4.x, o4.y, o4.z);
   dir = v3(d4.x, d4.y, d4.z);
}

// the whole state k_generate gives a path of a frame's first sample, from its id (rgen:24-38): origin | raygen RNG word, direction |
// payload seed - the same functions on the same inputs, so the words are those k_generate would have stored
__device__ __forceinline__ void primary_state(const FrameParams& fp, uint32_t id, float4& ro, float4& rd) {
   const uint32_t f = id / fp.n_owned, k = id - f * fp.n_owned;
   const uint32_t pix = fp.owned_pixels ? fp.owned_pixels[k] : k;
   const uint32_t px = pix % fp.W, py = pix / fp.W;
   uint32_t rng = init_rng(px, py, fp.W, fp.frame_numbers[f]);                                 // rgen:24
   const uint32_t seed = rng;                                                                  // rgen:30
   const float jx = random_float(rng), jy = random_float(rng);                                 // rgen:31
   V3 o, d;
   primary_ray(fp, px, py, jx, jy, o, d);
   ro = make_float4(o.x, o.y, o.z, __uint_as_float(rng));
   rd = make_float4(d.x, d.y, d.z, __uint_as_float(seed));
}

__device__ __forceinline__ bool owns_pixel(const FrameParams& fp, uint32_t x, uint32_t y) {
   if (fp.tp_world <= 1) return true;
   uint32_t tile = (y / fp.tp_tile) * fp.tiles_x + (x / fp.tp_tile);
   return tile % fp.tp_world == fp.tp_rank;
}

// ---- streaming (non-temporal) access to per-path state -----------------------------------------
// Path state is touched once per kernel and is hundreds of MB per wavefront of frames; the BVH (nodes + triangle packets,
// tens of MB) is what the traversal kernels re-read. `nt` loads / stores ask the caches not to keep the stream, so that it
// does not push the tree out of the 4 MiB L2 of each XCD (+1-1.5 % frame rate, profiles/README.md). A BUILD-time switch:
// as a run-time flag every access was a branch with its own wait behind it, and the independent loads of a kernel's
// prologue ran one after the other (k_shade_hit: eight dependent round trips instead of four).
#ifndef UH_STREAM_NT
#define UH_STREAM_NT 1
#endif
constexpr bool kStreamNt = UH_STREAM_NT != 0;
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef uint32_t v2u_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
   if (kStreamNt) {
      const v4f_t v = __builtin_nontemporal_load((const v4f_t*)p);
      return make_float4(v.x, v.y, v.z, v.w);
   }
   return *p;
}
__device__ __forceinline__ void st_stream(float4* p, float4 x) {
   if (kStreamNt) {
      const v4f_t v = {x.x, x.y, x.z, x.w};
      __builtin_nontemporal_store(v, (v4f_t*)p);
   } else {
      *p = x;
   }
}
__device__ __forceinline__ uint2 ld_stream(const uint2* p) {
   if (kStreamNt) {
      const v2u_t v = __builtin_nontemporal_load((const v2u_t*)p);
      return make_uint2(v.x, v.y);
   }
   return *p;
}
__device__ __forceinline__ void st_stream(uint2* p, uint2 x) {
   if (kStreamNt) {
      const v2u_t v = {x.x, x.y};
      __builtin_nontemporal_store(v, (v2u_t*)p);
   } else {
      *p = x;
   }
}
__device__ __forceinline__ uint32_t ld_stream(const uint32_t* p) { return kStreamNt ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void st_stream(uint32_t* p, uint32_t x) {
   if (kStreamNt)
      __builtin_nontemporal_store(x, p);
   else
      *p = x;
}

// ---- path state planes (device_types.h PathState): streamed like the other per-path arrays -------------------------------
__device__ __forceinline__ float4 ld_rec(const float4* p) { return ld_stream(p); }
__device__ __forceinline__ void st_rec(float4* p, float4 x) { st_stream(p, x); }

// ---- wave64 helpers ---------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Wave-aggregated queue append: one atomic per wave, lanes get consecutive slots (ballot +
// popcount prefix). Must be reached by all lanes of the wave that are still in the loop.
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool pred) {
   unsigned long long mask = __ballot(pred);
   uint32_t n = (uint32_t)__popcll(mask);
   uint32_t base = 0;
   if (n) {
      int leader = __ffsll((long long)mask) - 1;
      if ((int)lane_id() == leader) base = atomicAdd(counter, n);
      base = __shfl(base, leader);
   }
   uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
   return base + prefix;
}

}  // namespace uh
