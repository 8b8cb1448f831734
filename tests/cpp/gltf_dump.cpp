// gltf_dump.cpp — loads a .gltf with the C++ loader (include/utopian_gltf.hpp) and writes what it produced as a flat binary
// that tests/test_gltf_cpp.py compares with the Python loader's result.  usage: gltf_dump <in.gltf> <out.bin>
// Layout (little endian): u32 n_meshes, u32 n_textures; per mesh: u32 nv, u32 ni, u32 diffuse_map, f32 base_color[4],
// f32 metallic, f32 roughness, f32 transform[16] (column-major), u32 name_len, name bytes, vertices (80 B each), indices;
// per texture: u32 w, u32 h, rgba bytes. Needs no GPU: the loader only fills host structures.
#include <cstdio>
#include <fstream>

#include "utopian_gltf.hpp"

int main(int argc, char** argv) {
   if (argc < 3) return 2;
   try {
      std::vector<std::string> names;
      utopian::Model m = utopian::gltf::load_gltf(argv[1], &names);
      std::ofstream out(argv[2], std::ios::binary);
      auto w32 = [&](uint32_t v) { out.write(reinterpret_cast<const char*>(&v), 4); };
      w32((uint32_t)m.meshes.size());
      w32((uint32_t)m.textures.size());
      for (size_t i = 0; i < m.meshes.size(); i++) {
         const utopian::Mesh& mesh = m.meshes[i];
         w32((uint32_t)mesh.primitive.vertices.size());
         w32((uint32_t)mesh.primitive.indices.size());
         w32(mesh.material.diffuse_map);
         out.write(reinterpret_cast<const char*>(mesh.material.base_color_factor), 16);
         out.write(reinterpret_cast<const char*>(&mesh.material.metallic_factor), 4);
         out.write(reinterpret_cast<const char*>(&mesh.material.roughness_factor), 4);
         out.write(reinterpret_cast<const char*>(m.transforms[i].m), 64);
         w32((uint32_t)names[i].size());
         out.write(names[i].data(), (std::streamsize)names[i].size());
         out.write(reinterpret_cast<const char*>(mesh.primitive.vertices.data()), (std::streamsize)(mesh.primitive.vertices.size() * sizeof(utopian::Vertex)));
         out.write(reinterpret_cast<const char*>(mesh.primitive.indices.data()), (std::streamsize)(mesh.primitive.indices.size() * 4));
      }
      for (const utopian::Texture& t : m.textures) {
         w32(t.width);
         w32(t.height);
         out.write(reinterpret_cast<const char*>(t.rgba.data()), (std::streamsize)t.rgba.size());
      }
      std::printf("ok meshes=%zu textures=%zu\n", m.meshes.size(), m.textures.size());
      return 0;
   } catch (const utopian::Error& e) {
      std::printf("utopian::Error status=%d %s\n", e.status, e.what());
      return 1;
   }
}
