// utopian_jpeg.hpp — JPEG decoding for the C++ host mirror's glTF loader. The reference loads
// textures through `gltf::import` -> the `image` crate (utopian/src/gltf_loader.rs:168-196); 65 of Sponza's 69 images
// (prototype/data/models/Sponza/glTF/*.jpg) are baseline 8-bit 4:4:4 JPEGs. This decoder, written here from the JPEG standard
// (ITU-T T.81) and the published IJG algorithms, covers what that crate covers for 8-bit Huffman JPEGs:
//   * baseline (SOF0), extended sequential (SOF1) and progressive (SOF2) DCT, 8-bit, 1 or 3 components (4-component CMYK /
//     YCCK, arithmetic coding, lossless and 12-bit are refused), interleaved and non-interleaved scans, restart intervals,
//     8- and 16-bit quantisation tables, any 1..4 sampling factors;
//   * the arithmetic after entropy decoding is the IJG reference decoder's, integer for integer, so that results are
//     reproducible across the two host languages (rust-renderer_amd/image_decode.py restates it) and equal libjpeg's:
//     "islow" inverse DCT (13-bit constants, two passes), "fancy" triangle up-sampling for 2:1 chroma (h2v1, h2v2, h1v2),
//     replication for other ratios, fixed-point YCbCr -> RGB (JFIF), Adobe APP14 transform 0 = RGB.
// Output: 8-bit grey (channels 1) or RGB (channels 3), row-major.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "utopian_host.hpp"

namespace utopian {
namespace jpeg {

struct Image {
   uint32_t width = 0, height = 0, channels = 0;
   std::vector<uint8_t> pixels;
   bool progressive = false;
};

namespace detail {

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

[[noreturn]] inline void fail(const std::string& what) { throw Error(UH_ERR_INVALID_ARGUMENT, "JPEG: " + what); }

struct Huffman {
   bool present = false;
   uint8_t values[256];
   int32_t mincode[17], maxcode[18], valptr[17];
   uint16_t fast[512];  // 9-bit look-ahead: (length << 8) | symbol, 0 = longer code

   void build(const uint8_t counts[16], const uint8_t* vals, int n) {
      std::memcpy(values, vals, (size_t)n);
      int code = 0, k = 0;
      std::memset(fast, 0, sizeof(fast));
      for (int len = 1; len <= 16; len++) {
         valptr[len] = k;
         mincode[len] = code;
         for (int i = 0; i < counts[len - 1]; i++, k++, code++) {
            if (len <= 9) {
               const int first = code << (9 - len);
               for (int f = 0; f < (1 << (9 - len)); f++) fast[first + f] = (uint16_t)((len << 8) | values[k]);
            }
         }
         maxcode[len] = counts[len - 1] ? code - 1 : -1;
         if (code > (1 << len)) fail("Huffman table assigns more codes than its lengths allow");
         code <<= 1;
      }
      maxcode[17] = 0x7fffffff;
      present = true;
   }
};

// entropy-coded segment reader: 0xFF00 un-stuffing; at a marker the stream ends and zero bits are supplied (T.81 F.2.2.5)
struct BitReader {
   const uint8_t* p;
   const uint8_t* end;
   uint32_t acc = 0;
   int bits = 0;
   bool hit_marker = false;

   void fill() {
      while (bits <= 24) {
         uint32_t byte = 0;
         if (!hit_marker && p < end) {
            if (p[0] == 0xff) {
               if (p + 1 < end && p[1] == 0x00) {
                  byte = 0xff;
                  p += 2;
               } else {
                  hit_marker = true;  // leave p at the marker
               }
            } else {
               byte = *p++;
            }
         } else {
            hit_marker = true;
         }
         acc |= byte << (24 - bits);
         bits += 8;
      }
   }
   inline uint32_t peek(int n) {
      if (bits < n) fill();
      return acc >> (32 - n);
   }
   inline void skip(int n) {
      acc <<= n;
      bits -= n;
   }
   inline int get(int n) {
      if (n == 0) return 0;
      const uint32_t v = peek(n);
      skip(n);
      return (int)v;
   }
   inline int bit() { return get(1); }
   void reset() {
      acc = 0;
      bits = 0;
      hit_marker = false;
   }
   int decode(const Huffman& h) {
      if (bits < 16) fill();
      const uint16_t f = h.fast[acc >> 23];
      if (f) {
         skip(f >> 8);
         return f & 0xff;
      }
      int code = (int)(acc >> 22);  // 10 bits
      for (int len = 10; len <= 16; len++) {
         if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) {
            skip(len);
            return h.values[h.valptr[len] + code - h.mincode[len]];
         }
         code = (int)(acc >> (31 - len));
      }
      fail("bad Huffman code");
   }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Component {
   int id = 0, h = 1, v = 1, tq = 0;
   int dc_table = 0, ac_table = 0;
   int width = 0, height = 0;           // in samples: ceil(W * h / hmax), ceil(H * v / vmax)
   int blocks_w = 0, blocks_h = 0;      // blocks a non-interleaved scan visits
   int stride_blocks = 0, rows_blocks = 0;  // allocated (whole MCUs)
   std::vector<int16_t> coef;           // [rows_blocks][stride_blocks][64], natural order
   int last_dc = 0;
};

// IJG jidctint.c (jpeg_idct_islow): CONST_BITS 13, PASS1_BITS 2. 64-bit intermediates: the same values as the 32-bit original on
// every valid stream, and no signed overflow on a corrupt one
inline void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int out_stride) {
   constexpr int CB = 13, P1 = 2;
   constexpr int64_t F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373, F_1_175875602 = 9633,
                     F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;
   auto descale = [](int64_t x, int n) { return (x + ((int64_t)1 << (n - 1))) >> n; };
   int64_t ws[64];
   for (int c = 0; c < 8; c++) {
      auto d = [&](int r) { return (int64_t)in[8 * r + c] * (int64_t)q[8 * r + c]; };
      int64_t z2 = d(2), z3 = d(6);
      int64_t z1 = (z2 + z3) * F_0_541196100;
      int64_t tmp2 = z1 + z3 * (-F_1_847759065), tmp3 = z1 + z2 * F_0_765366865;
      z2 = d(0);
      z3 = d(4);
      int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
      const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = d(7);
      tmp1 = d(5);
      tmp2 = d(3);
      tmp3 = d(1);
      z1 = tmp0 + tmp3;
      z2 = tmp1 + tmp2;
      z3 = tmp0 + tmp2;
      int64_t z4 = tmp1 + tmp3;
      const int64_t z5 = (z3 + z4) * F_1_175875602;
      tmp0 *= F_0_298631336;
      tmp1 *= F_2_053119869;
      tmp2 *= F_3_072711026;
      tmp3 *= F_1_501321110;
      z1 *= -F_0_899976223;
      z2 *= -F_2_562915447;
      z3 *= -F_1_961570560;
      z4 *= -F_0_390180644;
      z3 += z5;
      z4 += z5;
      tmp0 += z1 + z3;
      tmp1 += z2 + z4;
      tmp2 += z2 + z3;
      tmp3 += z1 + z4;
      ws[8 * 0 + c] = descale(tmp10 + tmp3, CB - P1);
      ws[8 * 7 + c] = descale(tmp10 - tmp3, CB - P1);
      ws[8 * 1 + c] = descale(tmp11 + tmp2, CB - P1);
      ws[8 * 6 + c] = descale(tmp11 - tmp2, CB - P1);
      ws[8 * 2 + c] = descale(tmp12 + tmp1, CB - P1);
      ws[8 * 5 + c] = descale(tmp12 - tmp1, CB - P1);
      ws[8 * 3 + c] = descale(tmp13 + tmp0, CB - P1);
      ws[8 * 4 + c] = descale(tmp13 - tmp0, CB - P1);
   }
   for (int r = 0; r < 8; r++) {
      const int64_t* w = ws + 8 * r;
      int64_t z2 = w[2], z3 = w[6];
      int64_t z1 = (z2 + z3) * F_0_541196100;
      int64_t tmp2 = z1 + z3 * (-F_1_847759065), tmp3 = z1 + z2 * F_0_765366865;
      int64_t tmp0 = (w[0] + w[4]) * (1 << CB), tmp1 = (w[0] - w[4]) * (1 << CB);
      const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = w[7];
      tmp1 = w[5];
      tmp2 = w[3];
      tmp3 = w[1];
      z1 = tmp0 + tmp3;
      z2 = tmp1 + tmp2;
      z3 = tmp0 + tmp2;
      int64_t z4 = tmp1 + tmp3;
      const int64_t z5 = (z3 + z4) * F_1_175875602;
      tmp0 *= F_0_298631336;
      tmp1 *= F_2_053119869;
      tmp2 *= F_3_072711026;
      tmp3 *= F_1_501321110;
      z1 *= -F_0_899976223;
      z2 *= -F_2_562915447;
      z3 *= -F_1_961570560;
      z4 *= -F_0_390180644;
      z3 += z5;
      z4 += z5;
      tmp0 += z1 + z3;
      tmp1 += z2 + z4;
      tmp2 += z2 + z3;
      tmp3 += z1 + z4;
      auto put = [&](int c, int64_t x) {
         int64_t v = descale(x, CB + P1 + 3) + 128;
         out[out_stride * r + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
      };
      put(0, tmp10 + tmp3);
      put(7, tmp10 - tmp3);
      put(1, tmp11 + tmp2);
      put(6, tmp11 - tmp2);
      put(2, tmp12 + tmp1);
      put(5, tmp12 - tmp1);
      put(3, tmp13 + tmp0);
      put(4, tmp13 - tmp0);
   }
}

}  // namespace detail

inline Image decode(const uint8_t* data, size_t size) {
   using namespace detail;
   if (size < 4 || data[0] != 0xff || data[1] != 0xd8) fail("no SOI marker");
   uint16_t qt[4][64];
   bool have_qt[4] = {false, false, false, false};
   Huffman dc_tab[4], ac_tab[4];
   std::vector<Component> comp;
   int W = 0, H = 0, hmax = 1, vmax = 1, restart_interval = 0;
   bool progressive = false, have_frame = false, jfif = false, adobe = false;
   int adobe_transform = -1;
   size_t pos = 2;
   auto be16 = [&](size_t at) {
      if (at + 2 > size) fail("truncated");
      return (int)((data[at] << 8) | data[at + 1]);
   };

   // ---- one scan -----------------------------------------------------------------------------------------------------------
   auto decode_scan = [&](size_t header, size_t entropy_start) -> size_t {
      const int ns = data[header];
      if (ns < 1 || ns > 4 || header + 1 + 2 * (size_t)ns + 3 > size) fail("bad SOS");
      std::vector<Component*> sc;
      for (int i = 0; i < ns; i++) {
         const int id = data[header + 1 + 2 * i], tables = data[header + 2 + 2 * i];
         Component* c = nullptr;
         for (Component& k : comp)
            if (k.id == id) c = &k;
         if (!c) fail("scan names an unknown component");
         c->dc_table = tables >> 4;
         c->ac_table = tables & 15;
         if (c->dc_table > 3 || c->ac_table > 3) fail("Huffman table index");
         sc.push_back(c);
      }
      const int Ss = data[header + 1 + 2 * ns], Se = data[header + 2 + 2 * ns], Ah = data[header + 3 + 2 * ns] >> 4, Al = data[header + 3 + 2 * ns] & 15;
      if (progressive) {
         if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13) fail("bad progressive scan parameters");
      } else if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) {
         // IJG tolerates this with a warning; a sequential scan always carries the whole block
      }
      for (Component* c : sc) {
         if ((!progressive || Ss == 0) && !(progressive && Ah != 0) && !dc_tab[c->dc_table].present) fail("missing DC Huffman table");
         if ((!progressive || Ss > 0) && !ac_tab[c->ac_table].present) fail("missing AC Huffman table");
         c->last_dc = 0;
      }
      BitReader br{data + entropy_start, data + size};
      int eobrun = 0;
      const bool interleaved = ns > 1;
      const int mcus_x = interleaved ? (W + 8 * hmax - 1) / (8 * hmax) : sc[0]->blocks_w;
      const int mcus_y = interleaved ? (H + 8 * vmax - 1) / (8 * vmax) : sc[0]->blocks_h;
      int until_restart = restart_interval, next_rst = 0;

      auto block_sequential = [&](Component& c, int16_t* b) {
         int s = br.decode(dc_tab[c.dc_table]);
         if (s > 15) fail("bad DC category");
         int diff = s ? extend(br.get(s), s) : 0;
         c.last_dc += diff;
         b[0] = (int16_t)c.last_dc;
         const Huffman& ac = ac_tab[c.ac_table];
         for (int k = 1; k < 64;) {
            const int rs = br.decode(ac), r = rs >> 4;
            s = rs & 15;
            if (s) {
               k += r;
               if (k > 63) fail("AC run past the end of the block");
               b[kZigzag[k]] = (int16_t)extend(br.get(s), s);
               k++;
            } else {
               if (r != 15) break;
               k += 16;
            }
         }
      };
      auto block_dc_first = [&](Component& c, int16_t* b) {
         const int s = br.decode(dc_tab[c.dc_table]);
         if (s > 15) fail("bad DC category");
         const int diff = s ? extend(br.get(s), s) : 0;
         c.last_dc += diff;
         b[0] = (int16_t)(c.last_dc * (1 << Al));
      };
      auto block_dc_refine = [&](int16_t* b) {
         if (br.bit()) b[0] = (int16_t)(b[0] | (1 << Al));
      };
      auto block_ac_first = [&](Component& c, int16_t* b) {
         if (eobrun > 0) {
            eobrun--;
            return;
         }
         const Huffman& ac = ac_tab[c.ac_table];
         for (int k = Ss; k <= Se; k++) {
            const int rs = br.decode(ac), r = rs >> 4, s = rs & 15;
            if (s) {
               k += r;
               if (k > 63) fail("AC run past the end of the block");
               b[kZigzag[k]] = (int16_t)(extend(br.get(s), s) * (1 << Al));
            } else if (r == 15) {
               k += 15;
            } else {
               eobrun = 1 << r;
               if (r) eobrun += br.get(r);
               eobrun--;
               break;
            }
         }
      };
      auto block_ac_refine = [&](Component& c, int16_t* b) {
         const int p1 = 1 << Al, m1 = -(1 << Al);
         const Huffman& ac = ac_tab[c.ac_table];
         int k = Ss;
         auto correct = [&](int16_t* coef) {
            if (br.bit() && (*coef & p1) == 0) *coef = (int16_t)(*coef + (*coef >= 0 ? p1 : m1));
         };
         if (eobrun == 0) {
            for (; k <= Se; k++) {
               const int rs = br.decode(ac);
               int r = rs >> 4, s = rs & 15;
               if (s) {
                  s = br.bit() ? p1 : m1;  // the magnitude of a newly non-zero coefficient is always 1
               } else if (r != 15) {
                  eobrun = 1 << r;
                  if (r) eobrun += br.get(r);
                  break;  // end of band for this block too
               }
               // skip the already non-zero coefficients (each takes a correction bit) and r still-zero ones
               do {
                  int16_t* coef = b + kZigzag[k];
                  if (*coef != 0) {
                     correct(coef);
                  } else if (--r < 0) {
                     break;
                  }
                  k++;
               } while (k <= Se);
               if (s && k <= Se) b[kZigzag[k]] = (int16_t)s;
            }
         }
         if (eobrun > 0) {
            for (; k <= Se; k++) {
               int16_t* coef = b + kZigzag[k];
               if (*coef != 0) correct(coef);
            }
            eobrun--;
         }
      };
      auto one_block = [&](Component& c, int by, int bx) {
         int16_t* b = &c.coef[((size_t)by * c.stride_blocks + bx) * 64];
         if (!progressive)
            block_sequential(c, b);
         else if (Ss == 0)
            Ah == 0 ? block_dc_first(c, b) : block_dc_refine(b);
         else
            Ah == 0 ? block_ac_first(c, b) : block_ac_refine(c, b);
      };

      for (int my = 0; my < mcus_y; my++)
         for (int mx = 0; mx < mcus_x; mx++) {
            if (restart_interval && until_restart == 0) {
               // RSTn: byte-align, expect the marker, reset the predictors
               br.reset();
               const uint8_t* q = br.p;
               while (q + 1 < br.end && !(q[0] == 0xff && q[1] != 0x00 && q[1] != 0xff)) q++;
               if (q + 1 < br.end && q[1] == 0xd0 + next_rst) {
                  br.p = q + 2;
               } else if (q + 1 < br.end && q[1] >= 0xd0 && q[1] <= 0xd7) {
                  br.p = q + 2;  // out of sequence: resynchronise on it
               } else {
                  br.p = q;
                  br.hit_marker = true;
               }
               next_rst = (next_rst + 1) & 7;
               until_restart = restart_interval;
               eobrun = 0;
               for (Component* c : sc) c->last_dc = 0;
            }
            if (interleaved) {
               for (Component* c : sc)
                  for (int v = 0; v < c->v; v++)
                     for (int h = 0; h < c->h; h++) one_block(*c, my * c->v + v, mx * c->h + h);
            } else {
               one_block(*sc[0], my, mx);
            }
            if (restart_interval) until_restart--;
         }
      // the next marker: where the bit reader stopped, or the first marker after it
      const uint8_t* q = br.p;
      while (q + 1 < data + size && !(q[0] == 0xff && q[1] != 0x00 && q[1] != 0xff && !(q[1] >= 0xd0 && q[1] <= 0xd7))) q++;
      return (size_t)(q - data);
   };

   // ---- markers ------------------------------------------------------------------------------------------------------------
   bool done = false;
   while (!done) {
      if (pos + 4 > size) {
         if (have_frame) break;  // truncated file: decode what was read (IJG: premature end of file)
         fail("truncated before the frame header");
      }
      if (data[pos] != 0xff) {
         pos++;
         continue;
      }
      const int m = data[pos + 1];
      if (m == 0xff) {
         pos++;
         continue;
      }
      if (m == 0xd9) break;
      if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) {
         pos += 2;
         continue;
      }
      const int L = be16(pos + 2);
      if (L < 2 || pos + 2 + (size_t)L > size) fail("marker segment reaches past the end of the data");
      const size_t seg = pos + 4, seg_end = pos + 2 + (size_t)L;
      switch (m) {
         case 0xc0:
         case 0xc1:
         case 0xc2: {
            if (have_frame) fail("more than one frame");
            progressive = m == 0xc2;
            if (L < 8) fail("short SOF");
            if (data[seg] != 8) fail("only 8-bit precision is supported");
            H = be16(seg + 1);
            W = be16(seg + 3);
            const int nc = data[seg + 5];
            if (W == 0 || H == 0) fail("empty image (DNL is not supported)");
            if (nc != 1 && nc != 3) fail(nc == 4 ? "4-component (CMYK / YCCK) images are not supported" : "component count");
            if (L < 8 + 3 * nc) fail("short SOF");
            comp.resize((size_t)nc);
            for (int i = 0; i < nc; i++) {
               comp[i].id = data[seg + 6 + 3 * i];
               comp[i].h = data[seg + 7 + 3 * i] >> 4;
               comp[i].v = data[seg + 7 + 3 * i] & 15;
               comp[i].tq = data[seg + 8 + 3 * i];
               if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) fail("bad component parameters");
               hmax = comp[i].h > hmax ? comp[i].h : hmax;
               vmax = comp[i].v > vmax ? comp[i].v : vmax;
            }
            if (nc == 1) comp[0].h = comp[0].v = hmax = vmax = 1;  // a single component is never subsampled (T.81 A.2.2)
            const int mcus_x = (W + 8 * hmax - 1) / (8 * hmax), mcus_y = (H + 8 * vmax - 1) / (8 * vmax);
            for (Component& c : comp) {
               c.width = (W * c.h + hmax - 1) / hmax;
               c.height = (H * c.v + vmax - 1) / vmax;
               c.blocks_w = (c.width + 7) / 8;
               c.blocks_h = (c.height + 7) / 8;
               c.stride_blocks = mcus_x * c.h;
               c.rows_blocks = mcus_y * c.v;
               if ((uint64_t)c.stride_blocks * c.rows_blocks > (1u << 24)) fail("image too large");
               c.coef.assign((size_t)c.stride_blocks * c.rows_blocks * 64, 0);
            }
            have_frame = true;
            break;
         }
         case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
            fail("lossless, hierarchical and arithmetic-coded JPEGs are not supported");
         case 0xc4: {
            size_t at = seg;
            while (at < seg_end) {
               if (at + 17 > seg_end) fail("short DHT");
               const int tc = data[at] >> 4, th = data[at] & 15;
               if (tc > 1 || th > 3) fail("bad DHT class / index");
               int n = 0;
               for (int i = 0; i < 16; i++) n += data[at + 1 + i];
               if (n > 256 || at + 17 + (size_t)n > seg_end) fail("bad DHT size");
               (tc ? ac_tab[th] : dc_tab[th]).build(data + at + 1, data + at + 17, n);
               at += 17 + (size_t)n;
            }
            break;
         }
         case 0xdb: {
            size_t at = seg;
            while (at < seg_end) {
               const int pq = data[at] >> 4, tq = data[at] & 15;
               if (pq > 1 || tq > 3 || at + 1 + (size_t)64 * (pq + 1) > seg_end) fail("bad DQT");
               for (int i = 0; i < 64; i++) qt[tq][kZigzag[i]] = (uint16_t)(pq ? be16(at + 1 + 2 * i) : data[at + 1 + i]);
               have_qt[tq] = true;
               at += 1 + (size_t)64 * (pq + 1);
            }
            break;
         }
         case 0xdd:
            if (L != 4) fail("bad DRI");
            restart_interval = be16(seg);
            break;
         case 0xe0:
            if (L >= 7 && std::memcmp(data + seg, "JFIF\0", 5) == 0) jfif = true;
            break;
         case 0xee:
            if (L >= 14 && std::memcmp(data + seg, "Adobe", 5) == 0) {
               adobe = true;
               adobe_transform = data[seg + 11];
            }
            break;
         case 0xda: {
            if (!have_frame) fail("SOS before SOF");
            pos = decode_scan(seg, seg_end);
            continue;
         }
         default:
            break;  // APPn, COM, DNL ...: skipped
      }
      pos = seg_end;
   }
   if (!have_frame) fail("no frame");

   // ---- coefficients -> samples (dequantise + inverse DCT), per component at its own resolution --------------------------------
   std::vector<std::vector<uint8_t>> plane(comp.size());
   for (size_t ci = 0; ci < comp.size(); ci++) {
      Component& c = comp[ci];
      if (!have_qt[c.tq]) fail("missing quantisation table");
      const int pw = c.stride_blocks * 8;
      plane[ci].assign((size_t)pw * c.rows_blocks * 8, 0);
      for (int by = 0; by < c.rows_blocks; by++)
         for (int bx = 0; bx < c.stride_blocks; bx++) idct_islow(&c.coef[((size_t)by * c.stride_blocks + bx) * 64], qt[c.tq], &plane[ci][(size_t)by * 8 * pw + (size_t)bx * 8], pw);
      c.coef.clear();
      c.coef.shrink_to_fit();
   }

   // ---- up-sampling to the frame's resolution (IJG jdsample.c) ------------------------------------------------------------------
   auto upsample = [&](size_t ci) {
      const Component& c = comp[ci];
      const int pw = c.stride_blocks * 8, dw = c.width, dh = c.height;
      const uint8_t* src = plane[ci].data();
      const int hr = hmax / c.h, vr = vmax / c.v;
      if (hmax % c.h || vmax % c.v) fail("fractional sampling ratios are not supported");
      std::vector<uint8_t> out((size_t)W * H);
      auto row = [&](int y) { return src + (size_t)(y < 0 ? 0 : (y >= dh ? dh - 1 : y)) * pw; };
      if (hr == 1 && vr == 1) {
         for (int y = 0; y < H; y++) std::memcpy(&out[(size_t)y * W], row(y), (size_t)W);
      } else if (hr == 2 && vr == 1 && dw > 2) {  // h2v1_fancy_upsample
         std::vector<uint8_t> line((size_t)2 * dw);
         for (int y = 0; y < H; y++) {
            const uint8_t* in = row(y);
            line[0] = in[0];
            line[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
            for (int x = 1; x < dw - 1; x++) {
               const int v = in[x] * 3;
               line[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2);
               line[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2);
            }
            line[2 * dw - 2] = (uint8_t)((in[dw - 1] * 3 + in[dw - 2] + 1) >> 2);
            line[2 * dw - 1] = in[dw - 1];
            std::memcpy(&out[(size_t)y * W], line.data(), (size_t)W);
         }
      } else if (hr == 2 && vr == 2 && dw > 2) {  // h2v2_fancy_upsample
         std::vector<uint8_t> line((size_t)2 * dw);
         for (int y = 0; y < H; y++) {
            const int sy = y >> 1;
            const uint8_t* in0 = row(sy);                          // nearer row
            const uint8_t* in1 = row((y & 1) ? sy + 1 : sy - 1);  // farther row
            int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
            line[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
            line[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol;
            thiscol = nextcol;
            for (int x = 1; x < dw - 1; x++) {
               nextcol = in0[x + 1] * 3 + in1[x + 1];
               line[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
               line[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
               lastcol = thiscol;
               thiscol = nextcol;
            }
            line[2 * dw - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
            line[2 * dw - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
            std::memcpy(&out[(size_t)y * W], line.data(), (size_t)W);
         }
      } else if (hr == 1 && vr == 2) {  // h1v2_fancy_upsample (libjpeg-turbo)
         for (int y = 0; y < H; y++) {
            const int sy = y >> 1;
            const uint8_t* in0 = row(sy);
            const uint8_t* in1 = row((y & 1) ? sy + 1 : sy - 1);
            const int bias = (y & 1) ? 2 : 1;
            for (int x = 0; x < W; x++) out[(size_t)y * W + x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
         }
      } else {  // replication (int_upsample)
         for (int y = 0; y < H; y++) {
            const uint8_t* in = row(y / vr);
            for (int x = 0; x < W; x++) out[(size_t)y * W + x] = in[(x / hr) < dw ? (x / hr) : dw - 1];
         }
      }
      return out;
   };

   Image img;
   img.width = (uint32_t)W;
   img.height = (uint32_t)H;
   img.progressive = progressive;
   if (comp.size() == 1) {
      img.channels = 1;
      img.pixels = upsample(0);
      return img;
   }
   const std::vector<uint8_t> c0 = upsample(0), c1 = upsample(1), c2 = upsample(2);
   img.channels = 3;
   img.pixels.resize((size_t)W * H * 3);
   // colour space (IJG jdapimin.c default_decompress_parms): JFIF -> YCbCr; Adobe -> by its transform flag; neither -> RGB when
   // the component ids spell 'R','G','B', YCbCr otherwise
   bool ycc = true;
   if (jfif)
      ycc = true;
   else if (adobe)
      ycc = adobe_transform != 0;
   else if (comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B')
      ycc = false;
   if (!ycc) {
      for (size_t i = 0; i < (size_t)W * H; i++) {
         img.pixels[3 * i] = c0[i];
         img.pixels[3 * i + 1] = c1[i];
         img.pixels[3 * i + 2] = c2[i];
      }
      return img;
   }
   // jdcolor.c build_ycc_rgb_table / ycc_rgb_convert: 16-bit fixed point
   int32_t cr_r[256], cb_b[256], cr_g[256], cb_g[256];
   for (int i = 0; i < 256; i++) {
      const int32_t x = i - 128;
      cr_r[i] = (91881 * x + 32768) >> 16;
      cb_b[i] = (116130 * x + 32768) >> 16;
      cr_g[i] = -46802 * x;
      cb_g[i] = -22554 * x + 32768;
   }
   auto clamp8 = [](int32_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
   for (size_t i = 0; i < (size_t)W * H; i++) {
      const int y = c0[i], cb = c1[i], cr = c2[i];
      img.pixels[3 * i] = clamp8(y + cr_r[cr]);
      img.pixels[3 * i + 1] = clamp8(y + ((cb_g[cb] + cr_g[cr]) >> 16));
      img.pixels[3 * i + 2] = clamp8(y + cb_b[cb]);
   }
   return img;
}

inline Image decode(const std::vector<uint8_t>& data) { return decode(data.data(), data.size()); }

}  // namespace jpeg
}  // namespace utopian
