// utopian_gltf.hpp — glTF 2.0 ingestion for the C++ host mirror: turns a .gltf file into the
// utopian::Model that Renderer::add_model uploads, with the semantics of the reference's loader
// (utopian/src/gltf_loader.rs:47-218):
//   * scenes' nodes are walked depth first, children BEFORE the node's own mesh (gltf_loader.rs:57-63); the node transform
//     is parent * local; one Mesh + one transform per primitive;
//   * Vertex{pos.w = 0, normal.w = 0, uv (0,0) if absent, color (1,1,1,1) if absent, tangent 0 if absent};
//   * material: base colour factor / metallic / roughness factors; diffuse_map = the glTF *texture* index, used to index the
//     model's *image* list (the reference's own quirk, gltf_loader.rs:103-107 vs :183-207); Lambertian, property 0
//     (callers override, e.g. prototype/src/scenes.rs:116-121);
//   * images become RGBA8: RGB8 is expanded with alpha 255, RGBA8 passes, anything else is the reference's
//     "Unsupported image format!" (gltf_loader.rs:179-198). PNG is decoded here (zlib's inflate + the five scanline
//     filters; palette images expand to RGB8 / RGBA8 as the `image` crate does), JPEG by utopian_jpeg.hpp (baseline,
//     extended sequential and progressive Huffman JPEGs: 65 of Sponza's 69 images are baseline JPEGs).
// Buffers and images may be base64 data URIs, files next to the .gltf, or buffer views.
// The Python twin (rust-renderer_amd/gltf.py + image_decode.py) is what the parity tests use; tests/test_gltf_cpp.py
// holds the two against each other. Header-only; link with -lz.
#pragma once
#include <zlib.h>

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "utopian_host.hpp"
#include "utopian_jpeg.hpp"

namespace utopian {
namespace gltf {

// ---- a small JSON reader (objects, arrays, strings, numbers, true / false / null) ---------------------------------
struct Json {
   enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
   bool b = false;
   double num = 0.0;
   std::string str;
   std::vector<Json> arr;
   std::map<std::string, Json> obj;

   bool has(const std::string& k) const { return kind == Object && obj.count(k) != 0; }
   const Json& operator[](const std::string& k) const {
      static const Json none;
      auto it = obj.find(k);
      return it == obj.end() ? none : it->second;
   }
   const Json& operator[](size_t i) const { return arr.at(i); }
   size_t size() const { return kind == Array ? arr.size() : 0; }
   double number(double fallback) const { return kind == Number ? num : fallback; }
   uint32_t index() const {
      if (kind != Number || num < 0) throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: expected an index");
      return (uint32_t)num;
   }
};

class JsonParser {
  public:
   explicit JsonParser(const std::string& text) : s_(text) {}
   Json parse() {
      Json v = value();
      ws();
      if (p_ != s_.size()) fail("trailing characters");
      return v;
   }

  private:
   const std::string& s_;
   size_t p_ = 0;
   [[noreturn]] void fail(const char* what) const { throw Error(UH_ERR_INVALID_ARGUMENT, std::string("glTF JSON: ") + what + " at byte " + std::to_string(p_)); }
   void ws() {
      while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\n' || s_[p_] == '\r' || s_[p_] == '\t')) p_++;
   }
   bool eat(const char* lit) {
      size_t n = std::strlen(lit);
      if (s_.compare(p_, n, lit) != 0) return false;
      p_ += n;
      return true;
   }
   Json value() {
      ws();
      if (p_ >= s_.size()) fail("unexpected end");
      Json v;
      char c = s_[p_];
      if (c == '{') {
         v.kind = Json::Object;
         p_++;
         ws();
         if (p_ < s_.size() && s_[p_] == '}') {
            p_++;
            return v;
         }
         for (;;) {
            ws();
            std::string key = string();
            ws();
            if (p_ >= s_.size() || s_[p_++] != ':') fail("':' expected");
            v.obj[key] = value();
            ws();
            if (p_ < s_.size() && s_[p_] == ',') {
               p_++;
               continue;
            }
            if (p_ < s_.size() && s_[p_] == '}') {
               p_++;
               return v;
            }
            fail("',' or '}' expected");
         }
      }
      if (c == '[') {
         v.kind = Json::Array;
         p_++;
         ws();
         if (p_ < s_.size() && s_[p_] == ']') {
            p_++;
            return v;
         }
         for (;;) {
            v.arr.push_back(value());
            ws();
            if (p_ < s_.size() && s_[p_] == ',') {
               p_++;
               continue;
            }
            if (p_ < s_.size() && s_[p_] == ']') {
               p_++;
               return v;
            }
            fail("',' or ']' expected");
         }
      }
      if (c == '"') {
         v.kind = Json::String;
         v.str = string();
         return v;
      }
      if (eat("true")) {
         v.kind = Json::Bool;
         v.b = true;
         return v;
      }
      if (eat("false")) {
         v.kind = Json::Bool;
         return v;
      }
      if (eat("null")) return v;
      size_t start = p_;
      while (p_ < s_.size() && (std::isdigit((unsigned char)s_[p_]) || s_[p_] == '-' || s_[p_] == '+' || s_[p_] == '.' || s_[p_] == 'e' || s_[p_] == 'E')) p_++;
      if (p_ == start) fail("value expected");
      v.kind = Json::Number;
      v.num = std::strtod(s_.substr(start, p_ - start).c_str(), nullptr);
      return v;
   }
   std::string string() {
      if (p_ >= s_.size() || s_[p_] != '"') fail("string expected");
      p_++;
      std::string out;
      while (p_ < s_.size() && s_[p_] != '"') {
         char c = s_[p_++];
         if (c != '\\') {
            out.push_back(c);
            continue;
         }
         if (p_ >= s_.size()) fail("unterminated escape");
         char e = s_[p_++];
         switch (e) {
            case 'n': out.push_back('\n'); break;
            case 't': out.push_back('\t'); break;
            case 'r': out.push_back('\r'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'u': {
               if (p_ + 4 > s_.size()) fail("short \\u escape");
               unsigned cp = (unsigned)std::strtoul(s_.substr(p_, 4).c_str(), nullptr, 16);
               p_ += 4;
               if (cp < 0x80)
                  out.push_back((char)cp);
               else if (cp < 0x800) {
                  out.push_back((char)(0xC0 | (cp >> 6)));
                  out.push_back((char)(0x80 | (cp & 0x3F)));
               } else {
                  out.push_back((char)(0xE0 | (cp >> 12)));
                  out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
                  out.push_back((char)(0x80 | (cp & 0x3F)));
               }
               break;
            }
            default: out.push_back(e);  // \" \\ \/
         }
      }
      if (p_ >= s_.size()) fail("unterminated string");
      p_++;
      return out;
   }
};

// ---- bytes ----------------------------------------------------------------------------------------------------------
inline std::vector<uint8_t> base64_decode(const std::string& in, size_t from) {
   std::vector<uint8_t> out;
   out.reserve((in.size() - from) * 3 / 4);
   uint32_t acc = 0;
   int bits = 0;
   for (size_t i = from; i < in.size(); i++) {
      const char c = in[i];
      int v;
      if (c >= 'A' && c <= 'Z') v = c - 'A';
      else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
      else if (c >= '0' && c <= '9') v = c - '0' + 52;
      else if (c == '+' || c == '-') v = 62;
      else if (c == '/' || c == '_') v = 63;
      else continue;  // padding, line breaks
      acc = (acc << 6) | (uint32_t)v;
      bits += 6;
      if (bits >= 8) {
         bits -= 8;
         out.push_back((uint8_t)(acc >> bits));
      }
   }
   return out;
}

inline std::vector<uint8_t> read_file(const std::string& path) {
   std::ifstream f(path, std::ios::binary);
   if (!f) throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: cannot open " + path);
   return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

inline std::vector<uint8_t> load_uri(const std::string& uri, const std::string& base_dir) {
   if (uri.compare(0, 5, "data:") == 0) {
      size_t comma = uri.find(',');
      if (comma == std::string::npos) throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: malformed data URI");
      return base64_decode(uri, comma + 1);
   }
   return read_file(base_dir.empty() ? uri : base_dir + "/" + uri);
}

// ---- PNG --------------------------------------------------------------------------------------------------------------
struct DecodedImage {
   uint32_t width = 0, height = 0, channels = 0, depth = 8;  // as the gltf crate reports it: R8 / R8G8 / R8G8B8 / R8G8B8A8 / R16...
   std::vector<uint8_t> pixels;                               // depth 8 only (deeper images are rejected by the policy below anyway)
};

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline DecodedImage decode_png(const std::vector<uint8_t>& data) {
   static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
   if (data.size() < 8 || std::memcmp(data.data(), magic, 8) != 0) throw Error(UH_ERR_INVALID_ARGUMENT, "not a PNG");
   uint32_t w = 0, h = 0;
   int depth = 0, ctype = 0, interlace = 0;
   std::vector<uint8_t> idat, palette, trns;
   bool have_trns = false;
   for (size_t pos = 8; pos + 12 <= data.size();) {
      const uint32_t len = be32(&data[pos]);
      const char* kind = reinterpret_cast<const char*>(&data[pos + 4]);
      if (pos + 12 + (size_t)len > data.size()) throw Error(UH_ERR_INVALID_ARGUMENT, "PNG: truncated chunk");
      const uint8_t* body = &data[pos + 8];
      if (!std::memcmp(kind, "IHDR", 4) && len >= 13) {
         w = be32(body);
         h = be32(body + 4);
         depth = body[8];
         ctype = body[9];
         interlace = body[12];
      } else if (!std::memcmp(kind, "PLTE", 4)) {
         palette.assign(body, body + len);
      } else if (!std::memcmp(kind, "tRNS", 4)) {
         trns.assign(body, body + len);
         have_trns = true;
      } else if (!std::memcmp(kind, "IDAT", 4)) {
         idat.insert(idat.end(), body, body + len);
      } else if (!std::memcmp(kind, "IEND", 4)) {
         break;
      }
      pos += 12 + (size_t)len;
   }
   if (!w || !h) throw Error(UH_ERR_INVALID_ARGUMENT, "PNG: no IHDR");
   if (interlace) throw Error(UH_ERR_INVALID_ARGUMENT, "interlaced PNG");
   int channels;
   switch (ctype) {
      case 0: channels = 1; break;
      case 2: channels = 3; break;
      case 3: channels = 1; break;
      case 4: channels = 2; break;
      case 6: channels = 4; break;
      default: throw Error(UH_ERR_INVALID_ARGUMENT, "PNG: colour type");
   }
   if (depth != 8 && depth != 16 && !((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))) throw Error(UH_ERR_INVALID_ARGUMENT, "PNG bit depth");
   const size_t bits = (size_t)channels * depth, stride = ((size_t)w * bits + 7) / 8, bpp = bits / 8 ? bits / 8 : 1;
   std::vector<uint8_t> raw((stride + 1) * (size_t)h);
   {
      uLongf out_len = (uLongf)raw.size();
      int z = uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size());
      if (z != Z_OK || out_len != raw.size()) throw Error(UH_ERR_INVALID_ARGUMENT, "PNG: inflate failed");
   }
   // scanline filters 0-4 (None, Sub, Up, Average, Paeth)
   std::vector<uint8_t> rows(stride * (size_t)h);
   std::vector<uint8_t> zero(stride, 0);
   for (uint32_t y = 0; y < h; y++) {
      const uint8_t ft = raw[(stride + 1) * (size_t)y];
      const uint8_t* line = &raw[(stride + 1) * (size_t)y + 1];
      uint8_t* cur = &rows[stride * (size_t)y];
      const uint8_t* prev = y ? &rows[stride * (size_t)(y - 1)] : zero.data();
      for (size_t i = 0; i < stride; i++) {
         const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
         int pred;
         switch (ft) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: {
               const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
               pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
               break;
            }
            default: throw Error(UH_ERR_INVALID_ARGUMENT, "PNG filter type");
         }
         cur[i] = (uint8_t)(line[i] + pred);
      }
   }
   DecodedImage img;
   img.width = w;
   img.height = h;
   img.depth = depth == 16 ? 16 : 8;
   std::vector<uint8_t> samples;  // one byte per sample for depths <= 8
   if (depth < 8) {
      samples.resize((size_t)w * h);
      const int maxv = (1 << depth) - 1;
      for (uint32_t y = 0; y < h; y++)
         for (uint32_t x = 0; x < w; x++) {
            const size_t bit = (size_t)x * depth;
            const int v = (rows[stride * (size_t)y + bit / 8] >> (8 - depth - (bit % 8))) & maxv;
            samples[(size_t)y * w + x] = (uint8_t)(ctype == 3 ? v : v * (255 / maxv));
         }
   } else if (depth == 8) {
      samples = rows;
   } else {
      img.channels = (uint32_t)channels;
      return img;  // 16-bit: reported, never used (the loader's policy rejects it)
   }
   if (ctype == 3) {  // palette -> RGB8 (RGBA8 with a tRNS chunk), as the image crate expands it
      img.channels = have_trns ? 4 : 3;
      img.pixels.resize((size_t)w * h * img.channels);
      for (size_t i = 0; i < (size_t)w * h; i++) {
         const size_t idx = samples[i];
         if (idx * 3 + 2 >= palette.size()) throw Error(UH_ERR_INVALID_ARGUMENT, "PNG: palette index out of range");
         uint8_t* px = &img.pixels[i * img.channels];
         px[0] = palette[idx * 3];
         px[1] = palette[idx * 3 + 1];
         px[2] = palette[idx * 3 + 2];
         if (have_trns) px[3] = idx < trns.size() ? trns[idx] : 255;
      }
      return img;
   }
   img.channels = (uint32_t)channels;
   img.pixels = std::move(samples);
   return img;
}

// gltf_loader.rs:179-198: RGB8 -> RGBA8 with alpha 255, RGBA8 as it is, anything else "Unsupported image format!"
inline Texture load_image_rgba8(const std::vector<uint8_t>& data) {
   DecodedImage img;
   if (data.size() >= 2 && data[0] == 0xff && data[1] == 0xd8) {
      jpeg::Image j = jpeg::decode(data);
      img.width = j.width;
      img.height = j.height;
      img.channels = j.channels;  // 1 = L8: "Unsupported image format!" below, as in the reference
      img.depth = 8;
      img.pixels = std::move(j.pixels);
   } else {
      img = decode_png(data);
   }
   Texture t;
   t.width = img.width;
   t.height = img.height;
   if (img.depth == 8 && img.channels == 4) {
      t.rgba = std::move(img.pixels);
   } else if (img.depth == 8 && img.channels == 3) {
      t.rgba.resize((size_t)img.width * img.height * 4);
      for (size_t i = 0; i < (size_t)img.width * img.height; i++) {
         t.rgba[4 * i] = img.pixels[3 * i];
         t.rgba[4 * i + 1] = img.pixels[3 * i + 1];
         t.rgba[4 * i + 2] = img.pixels[3 * i + 2];
         t.rgba[4 * i + 3] = 255;
      }
   } else {
      throw Error(UH_ERR_INVALID_ARGUMENT, "Unsupported image format! (the reference loader panics on anything but RGB8 / RGBA8)");
   }
   return t;
}

// ---- accessors --------------------------------------------------------------------------------------------------------
struct Document {
   Json json;
   std::vector<std::vector<uint8_t>> buffers;
};

// `count` elements of `n` components, converted to float (normalised integers are scaled as glTF prescribes) or left as
// unsigned integers (indices): out[i * n + c]
inline void read_accessor(const Document& d, uint32_t index, std::vector<float>* as_float, std::vector<uint32_t>* as_uint, uint32_t* components) {
   const Json& acc = d.json["accessors"][index];
   if (acc.has("sparse")) throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: sparse accessors are not supported");
   const uint32_t ctype = acc["componentType"].index(), count = acc["count"].index();
   const std::string& type = acc["type"].str;
   const uint32_t n = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : type == "MAT4" ? 16 : 0;
   if (!n) throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: accessor type " + type);
   size_t item;
   switch (ctype) {
      case 5120: case 5121: item = 1; break;
      case 5122: case 5123: item = 2; break;
      case 5125: case 5126: item = 4; break;
      default: throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: component type");
   }
   *components = n;
   if (as_float) as_float->assign((size_t)count * n, 0.0f);
   if (as_uint) as_uint->assign((size_t)count * n, 0u);
   if (!acc.has("bufferView")) return;  // all zeros
   const Json& bv = d.json["bufferViews"][acc["bufferView"].index()];
   const std::vector<uint8_t>& raw = d.buffers.at(bv["buffer"].index());
   auto checked = [](double v, const char* what) {
      if (!(v >= 0.0) || !(v < 9.0e15)) throw Error(UH_ERR_INVALID_ARGUMENT, std::string("glTF: ") + what + " is negative, not finite or absurdly large");
      return (size_t)v;
   };
   const size_t start = checked(bv["byteOffset"].number(0), "bufferView.byteOffset") + checked(acc["byteOffset"].number(0), "accessor.byteOffset");
   size_t stride = checked(bv["byteStride"].number(0), "bufferView.byteStride");
   if (stride > 252) throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: bufferView.byteStride above 252 (the format allows 4..252)");
   if (!stride) stride = item * n;
   // no multiplication that could wrap: the last element must start early enough for its item * n bytes
   if (count) {
      if (start > raw.size() || item * n > raw.size() - start) throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: accessor reaches past its buffer");
      if ((size_t)(count - 1) > (raw.size() - start - item * n) / stride) throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: accessor reaches past its buffer");
   }
   const bool normalized = acc["normalized"].kind == Json::Bool && acc["normalized"].b;
   for (uint32_t i = 0; i < count; i++)
      for (uint32_t c = 0; c < n; c++) {
         const uint8_t* p = &raw[start + stride * i + item * c];
         double v;
         float scale = 0.0f;
         switch (ctype) {
            case 5120: { int8_t x; std::memcpy(&x, p, 1); v = x; scale = 127.0f; break; }
            case 5121: { uint8_t x; std::memcpy(&x, p, 1); v = x; scale = 255.0f; break; }
            case 5122: { int16_t x; std::memcpy(&x, p, 2); v = x; scale = 32767.0f; break; }
            case 5123: { uint16_t x; std::memcpy(&x, p, 2); v = x; scale = 65535.0f; break; }
            case 5125: { uint32_t x; std::memcpy(&x, p, 4); v = x; break; }
            default: { float x; std::memcpy(&x, p, 4); v = x; break; }
         }
         if (as_uint) (*as_uint)[(size_t)i * n + c] = (uint32_t)v;
         if (as_float) (*as_float)[(size_t)i * n + c] = ctype == 5126 ? (float)v : (normalized && scale != 0.0f ? (float)v / scale : (float)v);
      }
}

// ---- nodes ------------------------------------------------------------------------------------------------------------
inline Mat4 node_matrix(const Json& node) {
   Mat4 m;
   if (node.has("matrix")) {  // column-major in the file, as Mat4 stores it
      for (int i = 0; i < 16; i++) m.m[i] = (float)node["matrix"][i].num;
      return m;
   }
   float q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1}, t[3] = {0, 0, 0};
   if (node.has("rotation")) for (int i = 0; i < 4; i++) q[i] = (float)node["rotation"][i].num;
   if (node.has("scale")) for (int i = 0; i < 3; i++) s[i] = (float)node["scale"][i].num;
   if (node.has("translation")) for (int i = 0; i < 3; i++) t[i] = (float)node["translation"][i].num;
   const float x = q[0], y = q[1], z = q[2], w = q[3];
   const float x2 = x + x, y2 = y + y, z2 = z + z;
   const float xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2, wx = w * x2, wy = w * y2, wz = w * z2;
   const float r[3][3] = {{1 - (yy + zz), xy - wz, xz + wy}, {xy + wz, 1 - (xx + zz), yz - wx}, {xz - wy, yz + wx, 1 - (xx + yy)}};
   for (int row = 0; row < 3; row++) {
      for (int col = 0; col < 3; col++) m.at(row, col) = r[row][col] * s[col];
      m.at(row, 3) = t[row];
   }
   return m;
}

inline std::string dirname_of(const std::string& path) {
   size_t slash = path.find_last_of("/\\");
   return slash == std::string::npos ? std::string() : path.substr(0, slash);
}

// utopian::gltf_loader::load_gltf
inline Model load_gltf(const std::string& path, std::vector<std::string>* mesh_names = nullptr) {
   const std::string base_dir = dirname_of(path);
   std::string text;
   {
      std::vector<uint8_t> bytes = read_file(path);
      text.assign(bytes.begin(), bytes.end());
   }
   Document d;
   d.json = JsonParser(text).parse();
   for (size_t i = 0; i < d.json["buffers"].size(); i++) d.buffers.push_back(load_uri(d.json["buffers"][i]["uri"].str, base_dir));
   Model model;
   for (size_t i = 0; i < d.json["images"].size(); i++) {
      const Json& image = d.json["images"][i];
      std::vector<uint8_t> data;
      if (image.has("uri")) {
         data = load_uri(image["uri"].str, base_dir);
      } else {
         const Json& bv = d.json["bufferViews"][image["bufferView"].index()];
         const std::vector<uint8_t>& raw = d.buffers.at(bv["buffer"].index());
         const double foff = bv["byteOffset"].number(0), flen = bv["byteLength"].number(0);
         if (!(foff >= 0.0) || !(flen >= 0.0) || !(foff + flen <= (double)raw.size())) throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: image reaches past its buffer");
         const size_t off = (size_t)foff, len = (size_t)flen;
         data.assign(raw.begin() + (long)off, raw.begin() + (long)(off + len));
      }
      model.textures.push_back(load_image_rgba8(data));
   }
   struct Walker {
      const Document& d;
      Model& model;
      std::vector<std::string>* names;
      int depth = 0;
      void node(uint32_t index, const Mat4& parent) {
         // a node hierarchy is a forest (glTF 2.0, 3.5.2): a file whose children loop back would recurse for ever
         if (++depth > 256) throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: node hierarchy deeper than 256 levels (cyclic children?)");
         struct Leave {
            int& d;
            ~Leave() { d--; }
         } leave{depth};
         const Json& n = d.json["nodes"][index];
         const Mat4 transform = parent * node_matrix(n);
         for (size_t c = 0; c < n["children"].size(); c++) node(n["children"][c].index(), transform);  // children first (gltf_loader.rs:57-63)
         if (!n.has("mesh")) return;
         const Json& prims = d.json["meshes"][n["mesh"].index()]["primitives"];
         for (size_t pi = 0; pi < prims.size(); pi++) {
            const Json& prim = prims[pi];
            const Json& attrs = prim["attributes"];
            std::vector<float> pos, nrm, uv, tan, col;
            std::vector<uint32_t> idx;
            uint32_t n3 = 0, nuv = 0, ntan = 0, ncol = 0, nidx = 0;
            read_accessor(d, attrs["POSITION"].index(), &pos, nullptr, &n3);
            if (n3 != 3) throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: POSITION must be VEC3");
            read_accessor(d, attrs["NORMAL"].index(), &nrm, nullptr, &n3);
            read_accessor(d, prim["indices"].index(), nullptr, &idx, &nidx);
            if (attrs.has("TEXCOORD_0")) read_accessor(d, attrs["TEXCOORD_0"].index(), &uv, nullptr, &nuv);
            if (attrs.has("TANGENT")) read_accessor(d, attrs["TANGENT"].index(), &tan, nullptr, &ntan);
            if (attrs.has("COLOR_0")) read_accessor(d, attrs["COLOR_0"].index(), &col, nullptr, &ncol);
            Mesh mesh;
            const size_t nv = pos.size() / 3;
            // every attribute accessor must hold one element per POSITION (the reference's reader zips them and would panic)
            if (n3 != 3 || nrm.size() != 3 * nv || (nuv && uv.size() != (size_t)nuv * nv) || (ntan && tan.size() != (size_t)ntan * nv) ||
                (ncol && col.size() != (size_t)ncol * nv))
               throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: attribute accessors of one primitive differ in element count (or NORMAL is not VEC3)");
            for (uint32_t i : idx)
               if (i >= nv) throw Error(UH_ERR_INVALID_ARGUMENT, "glTF: index beyond the primitive's vertices");
            mesh.primitive.vertices.resize(nv);
            for (size_t v = 0; v < nv; v++) {
               Vertex& o = mesh.primitive.vertices[v];
               std::memset(&o, 0, sizeof(o));
               for (int k = 0; k < 3; k++) {
                  o.pos[k] = pos[3 * v + k];
                  o.normal[k] = nrm[3 * v + k];
               }
               if (nuv >= 2) {
                  o.uv[0] = uv[nuv * v];
                  o.uv[1] = uv[nuv * v + 1];
               }
               for (uint32_t k = 0; k < 4 && k < ntan; k++) o.tangent[k] = tan[ntan * v + k];
               for (int k = 0; k < 4; k++) o.color[k] = 1.0f;
               for (uint32_t k = 0; k < 4 && k < ncol; k++) o.color[k] = col[ncol * v + k];
            }
            mesh.primitive.indices = std::move(idx);
            const Json& mat = prim.has("material") ? d.json["materials"][prim["material"].index()] : Json();
            const Json& pbr = mat["pbrMetallicRoughness"];
            if (pbr.has("baseColorFactor"))
               for (int k = 0; k < 4; k++) mesh.material.base_color_factor[k] = (float)pbr["baseColorFactor"][k].num;
            mesh.material.metallic_factor = (float)pbr["metallicFactor"].number(1.0);
            mesh.material.roughness_factor = (float)pbr["roughnessFactor"].number(1.0);
            if (pbr.has("baseColorTexture")) mesh.material.diffuse_map = pbr["baseColorTexture"]["index"].index();  // the texture index (see the header comment)
            model.meshes.push_back(std::move(mesh));
            model.transforms.push_back(transform);
            if (names) names->push_back(mat.has("name") ? mat["name"].str : (n.has("name") ? n["name"].str : std::string()));
         }
      }
   } walker{d, model, mesh_names};
   for (size_t s = 0; s < d.json["scenes"].size(); s++)
      for (size_t k = 0; k < d.json["scenes"][s]["nodes"].size(); k++) walker.node(d.json["scenes"][s]["nodes"][k].index(), Mat4::identity());
   return model;
}

}  // namespace gltf
}  // namespace utopian
