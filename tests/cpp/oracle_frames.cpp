// oracle_frames.cpp — runs the CPU oracle (oracle/oracle.cpp, compiled into this binary with
// -fsanitize=address,undefined) on a scene blob written by tests/test_cpp_host.py::write_blob and
// dumps the accumulation image. Test infrastructure only.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <vector>

#include "utopian_hip.h"

extern "C" {
struct orc_ctx;
int orc_create(uint32_t, uint32_t, orc_ctx**);
void orc_destroy(orc_ctx*);
int orc_add_texture_rgba8(orc_ctx*, const uint8_t*, uint32_t, uint32_t, uint32_t*);
int orc_add_mesh(orc_ctx*, const UhVertex*, uint32_t, const uint32_t*, uint32_t, const UhGpuMaterial*, const float*, uint32_t*);
int orc_add_light(orc_ctx*, const UhGpuLight*, uint32_t*);
int orc_build_acceleration(orc_ctx*);
int orc_render_frame(orc_ctx*, const UhViewUniformData*, uint32_t);
int orc_read_accumulation(orc_ctx*, float*);
int orc_set_option(orc_ctx*, const char*, int);
}

template <typename T>
static T rd(std::ifstream& f) {
   T v;
   f.read(reinterpret_cast<char*>(&v), sizeof(T));
   if (!f) throw std::runtime_error("truncated scene blob");
   return v;
}

int main(int argc, char** argv) {
   if (argc < 3) return 2;
   std::ifstream f(argv[1], std::ios::binary);
   if (!f || rd<uint32_t>(f) != 0x43534855u) return 2;
   const uint32_t W = rd<uint32_t>(f), H = rd<uint32_t>(f), frames = rd<uint32_t>(f), pass_mask = rd<uint32_t>(f);
   UhViewUniformData view = rd<UhViewUniformData>(f);
   for (int i = 0; i < 9; i++) rd<float>(f);
   orc_ctx* c = nullptr;
   if (orc_create(W, H, &c)) return 1;
   orc_set_option(c, "threads", 3);
   uint32_t white_index = 0;
   const uint8_t white[4] = {255, 255, 255, 255};
   orc_add_texture_rgba8(c, white, 1, 1, &white_index);
   const uint32_t ntex = rd<uint32_t>(f);
   std::vector<uint32_t> tex_index(ntex);
   for (uint32_t i = 0; i < ntex; i++) {
      uint32_t w = rd<uint32_t>(f), h = rd<uint32_t>(f);
      std::vector<uint8_t> px((size_t)w * h * 4);
      f.read(reinterpret_cast<char*>(px.data()), (std::streamsize)px.size());
      orc_add_texture_rgba8(c, px.data(), w, h, &tex_index[i]);
   }
   const uint32_t nmesh = rd<uint32_t>(f);
   for (uint32_t i = 0; i < nmesh; i++) {
      const uint32_t nv = rd<uint32_t>(f), ni = rd<uint32_t>(f);
      std::vector<UhVertex> v(nv);
      std::vector<uint32_t> idx(ni);
      f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(nv * sizeof(UhVertex)));
      f.read(reinterpret_cast<char*>(idx.data()), (std::streamsize)(ni * 4));
      UhGpuMaterial m;
      std::memset(&m, 0, sizeof(m));
      const int32_t tex = rd<int32_t>(f);
      m.diffuse_map = tex < 0 ? white_index : tex_index[(size_t)tex];
      for (float& x : m.base_color_factor) x = rd<float>(f);
      m.metallic_factor = m.roughness_factor = 1.0f;
      m.raytrace_properties[0] = (float)rd<uint32_t>(f);
      m.raytrace_properties[1] = rd<float>(f);
      float mat4[16], w3x4[12];
      for (float& x : mat4) x = rd<float>(f);
      for (int r = 0; r < 3; r++)
         for (int col = 0; col < 4; col++) w3x4[r * 4 + col] = mat4[col * 4 + r];
      if (orc_add_mesh(c, v.data(), nv, idx.data(), ni, &m, w3x4, nullptr)) return 1;
   }
   const uint32_t nl = rd<uint32_t>(f);
   for (uint32_t i = 0; i < nl; i++) {
      UhGpuLight l;
      std::memset(&l, 0, sizeof(l));
      for (int a = 0; a < 3; a++) l.position[a] = rd<float>(f);
      l.color[0] = l.color[1] = l.color[2] = 1.0f;
      l.range = 1.0f;
      l.attenuation[2] = 0.1f;
      l.light_type = 1.0f;
      l.intensity[0] = l.intensity[1] = l.intensity[2] = 1.0f;
      orc_add_light(c, &l, nullptr);
   }
   if (orc_build_acceleration(c)) return 1;
   for (uint32_t i = 0; i < frames; i++) {
      view.total_samples += view.samples_per_frame;
      if (orc_render_frame(c, &view, pass_mask)) return 1;
      // prev_frame_projection_view = projection * view (column-major), main.rs:545-546
      float pv[16];
      for (int col = 0; col < 4; col++)
         for (int r = 0; r < 4; r++) {
            float s = 0;
            for (int k = 0; k < 4; k++) s += view.projection[k * 4 + r] * view.view[col * 4 + k];
            pv[col * 4 + r] = s;
         }
      std::memcpy(view.prev_frame_projection_view, pv, sizeof(pv));
   }
   std::vector<float> acc((size_t)W * H * 4);
   orc_read_accumulation(c, acc.data());
   std::ofstream out(argv[2], std::ios::binary);
   out.write(reinterpret_cast<const char*>(acc.data()), (std::streamsize)(acc.size() * 4));
   orc_destroy(c);
   std::printf("ok\n");
   return 0;
}
