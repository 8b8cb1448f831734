// host_frames.cpp — drives the HIP path through the C++ host mirror (include/utopian_host.hpp):
// Renderer::add_model / add_light / Raytracing::initialize, build_path_tracing_render_graph, and
// the Application frame protocol. The scene arrives as a blob written by tests/test_cpp_host.py so
// the same bytes can be rendered through the ctypes path and compared bit for bit.
//   usage: host_frames <scene.blob> <out_accumulation.f32>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "utopian_host.hpp"

using namespace utopian;

template <typename T>
static T rd(std::ifstream& f) {
   T v;
   f.read(reinterpret_cast<char*>(&v), sizeof(T));
   if (!f) throw std::runtime_error("truncated scene blob");
   return v;
}

int main(int argc, char** argv) {
   if (argc < 3) {
      std::fprintf(stderr, "usage: %s scene.blob out.f32\n", argv[0]);
      return 2;
   }
   try {
      std::ifstream f(argv[1], std::ios::binary);
      if (!f) throw std::runtime_error("cannot open scene blob");
      if (rd<uint32_t>(f) != 0x43534855u) throw std::runtime_error("bad magic");
      const uint32_t W = rd<uint32_t>(f), H = rd<uint32_t>(f), frames = rd<uint32_t>(f), pass_mask = rd<uint32_t>(f);
      ViewUniformData ref_view = rd<ViewUniformData>(f);
      float cam[9];
      for (float& c : cam) c = rd<float>(f);

      Renderer renderer(0, W, H);  // throws utopian::Error(UH_ERR_NO_DEVICE) when there is no GPU
      renderer.initialize();

      Model model;
      const uint32_t ntex = rd<uint32_t>(f);
      for (uint32_t i = 0; i < ntex; i++) {
         Texture t;
         t.width = rd<uint32_t>(f);
         t.height = rd<uint32_t>(f);
         t.rgba.resize((size_t)t.width * t.height * 4);
         f.read(reinterpret_cast<char*>(t.rgba.data()), (std::streamsize)t.rgba.size());
         model.textures.push_back(std::move(t));
      }
      const uint32_t nmesh = rd<uint32_t>(f);
      for (uint32_t i = 0; i < nmesh; i++) {
         Mesh mesh;
         const uint32_t nv = rd<uint32_t>(f), ni = rd<uint32_t>(f);
         mesh.primitive.vertices.resize(nv);
         mesh.primitive.indices.resize(ni);
         f.read(reinterpret_cast<char*>(mesh.primitive.vertices.data()), (std::streamsize)(nv * sizeof(Vertex)));
         f.read(reinterpret_cast<char*>(mesh.primitive.indices.data()), (std::streamsize)(ni * 4));
         const int32_t tex = rd<int32_t>(f);
         mesh.material.diffuse_map = tex < 0 ? DEFAULT_TEXTURE_MAP : (uint32_t)tex;
         for (float& c : mesh.material.base_color_factor) c = rd<float>(f);
         mesh.material.material_type = (MaterialType)rd<uint32_t>(f);
         mesh.material.material_property = rd<float>(f);
         Mat4 t;
         for (float& c : t.m) c = rd<float>(f);
         model.meshes.push_back(std::move(mesh));
         model.transforms.push_back(t);
      }
      renderer.add_model(std::move(model), Mat4::identity());
      const uint32_t nlights = rd<uint32_t>(f);
      for (uint32_t i = 0; i < nlights; i++) {
         float x = rd<float>(f), y = rd<float>(f), z = rd<float>(f);
         renderer.add_light({x, y, z}, {1, 1, 1}, 1.0f);
      }
      renderer.initialize_raytracing();

      Camera camera({cam[0], cam[1], cam[2]}, {cam[3], cam[4], cam[5]}, cam[6], (float)W / (float)H, cam[7], cam[8]);
      Application app(renderer, camera);
      // the C++ camera must agree with the caller's matrices (glam semantics) ...
      float max_diff = 0.0f;
      const float* mine[4] = {app.view_data.view, app.view_data.projection, app.view_data.inverse_view, app.view_data.inverse_projection};
      const float* theirs[4] = {ref_view.view, ref_view.projection, ref_view.inverse_view, ref_view.inverse_projection};
      for (int k = 0; k < 4; k++)
         for (int i = 0; i < 16; i++) max_diff = std::fmax(max_diff, std::fabs(mine[k][i] - theirs[k][i]));
      // ... and the frames are rendered from the caller's exact view block so results can be compared bit for bit
      app.view_data = ref_view;
      if (pass_mask != UH_PASS_ALL) {
         app.graph.clear();
         app.graph.add_pass("reference_pt_pass", [pass_mask](Renderer& r, const ViewUniformData& v) { r.check(uh_render_frame(r.handle(), &v, pass_mask), "render_func"); });
      }
      // the first half frame by frame through the graph, the rest as one static-camera run (uh_render_frames batches it)
      const uint32_t by_graph = frames / 2;
      for (uint32_t i = 0; i < by_graph; i++) app.frame();
      app.frames(frames - by_graph, pass_mask == UH_PASS_REFERENCE_PT);
      std::vector<float> acc = renderer.read_accumulation();
      std::ofstream out(argv[2], std::ios::binary);
      out.write(reinterpret_cast<const char*>(acc.data()), (std::streamsize)(acc.size() * 4));
      UhStats s = renderer.get_stats();
      std::printf("ok passes=%zu total_samples=%u rays=%llu camera_max_diff=%g\n", app.graph.passes.size(), app.view_data.total_samples,
                  (unsigned long long)(s.rays[0] + s.rays[1] + s.rays[2] + s.rays[3]), max_diff);
      return 0;
   } catch (const Error& e) {
      std::printf("utopian::Error status=%d %s\n", e.status, e.what());
      return e.status == UH_ERR_NO_DEVICE ? 3 : 1;
   } catch (const std::exception& e) {
      std::printf("error: %s\n", e.what());
      return 1;
   }
}
