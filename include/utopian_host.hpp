// utopian_host.hpp — C++ host-side mirror of the reference's plugin surface for the path-tracing
// + ReSTIR path, layered on the C ABI of utopian_hip.h (header-only, C++17).
//
// The reference's host code is Rust; there is no Rust toolchain in the build image, so the host
// layer a Rust caller would write is provided in C++ with the same names and
// argument meaning:
//   utopian::Camera                    utopian/src/camera.rs:90-107 (look_at_rh / perspective_rh)
//   utopian::Material / Mesh / Model   utopian/src/gltf_loader.rs:11-45, primitive.rs:9-24
//   utopian::Renderer                  utopian/src/renderer.rs:123-412 (new, initialize, add_model,
//                                      add_light, get_num_lights) + Raytracing::initialize
//                                      (raytracing.rs:89) + rebuild_tlas (raytracing.rs:400)
//   utopian::Graph / build_path_tracing_render_graph
//                                      utopian/src/renderers/mod.rs:189-375: the same named passes in
//                                      the same order; each pass's render closure is one
//                                      uh_render_frame call with that pass's bit
//   utopian::Application               the per-frame protocol of prototype/src/main.rs:460-471,545-546
// Error behaviour: the reference panics on every failure (graph.rs:253, raytracing.rs:178); here a
// non-zero status throws utopian::Error carrying uh_last_error().
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "utopian_hip.h"

namespace utopian {

struct Error : std::runtime_error {
   int status;
   Error(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

// ---- glam-like math (column-major Mat4, f32) ---------------------------------------------
struct Vec3 {
   float x = 0, y = 0, z = 0;
};
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec3 normalize(Vec3 a) {
   float l = std::sqrt(dot(a, a));
   return {a.x / l, a.y / l, a.z / l};
}

struct Mat4 {
   float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // m[c*4 + r]
   float& at(int r, int c) { return m[c * 4 + r]; }
   float at(int r, int c) const { return m[c * 4 + r]; }
   static Mat4 identity() { return Mat4(); }
   static Mat4 from_diagonal(float a, float b, float c, float d) {
      Mat4 r;
      r.at(0, 0) = a;
      r.at(1, 1) = b;
      r.at(2, 2) = c;
      r.at(3, 3) = d;
      return r;
   }
   static Mat4 from_scale_translation(Vec3 s, Vec3 t) {
      Mat4 r;
      r.at(0, 0) = s.x;
      r.at(1, 1) = s.y;
      r.at(2, 2) = s.z;
      r.at(0, 3) = t.x;
      r.at(1, 3) = t.y;
      r.at(2, 3) = t.z;
      return r;
   }
   static Mat4 from_translation(Vec3 t) { return from_scale_translation({1, 1, 1}, t); }
   // glam Mat4::look_at_rh
   static Mat4 look_at_rh(Vec3 eye, Vec3 center, Vec3 up) {
      Vec3 f = normalize(center - eye), s = normalize(cross(f, up)), u = cross(s, f);
      Mat4 r;
      r.at(0, 0) = s.x;
      r.at(0, 1) = s.y;
      r.at(0, 2) = s.z;
      r.at(1, 0) = u.x;
      r.at(1, 1) = u.y;
      r.at(1, 2) = u.z;
      r.at(2, 0) = -f.x;
      r.at(2, 1) = -f.y;
      r.at(2, 2) = -f.z;
      r.at(0, 3) = -dot(s, eye);
      r.at(1, 3) = -dot(u, eye);
      r.at(2, 3) = dot(f, eye);
      return r;
   }
   // glam Mat4::perspective_rh (depth 0..1)
   static Mat4 perspective_rh(float fov_y_radians, float aspect, float z_near, float z_far) {
      float h = std::cos(0.5f * fov_y_radians) / std::sin(0.5f * fov_y_radians);
      float w = h / aspect, r = z_far / (z_near - z_far);
      Mat4 p;
      std::memset(p.m, 0, sizeof(p.m));
      p.at(0, 0) = w;
      p.at(1, 1) = h;
      p.at(2, 2) = r;
      p.at(3, 2) = -1.0f;
      p.at(2, 3) = r * z_near;
      return p;
   }
   Mat4 operator*(const Mat4& b) const {
      Mat4 r;
      for (int c = 0; c < 4; c++)
         for (int rr = 0; rr < 4; rr++) {
            float s = 0;
            for (int k = 0; k < 4; k++) s += at(rr, k) * b.at(k, c);
            r.at(rr, c) = s;
         }
      return r;
   }
   Mat4 inverse() const {  // Gauss-Jordan in double
      double a[4][8];
      for (int r = 0; r < 4; r++)
         for (int c = 0; c < 4; c++) {
            a[r][c] = at(r, c);
            a[r][4 + c] = r == c ? 1.0 : 0.0;
         }
      for (int col = 0; col < 4; col++) {
         int piv = col;
         for (int r = col + 1; r < 4; r++)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
         for (int c = 0; c < 8; c++) std::swap(a[col][c], a[piv][c]);
         double d = a[col][col];
         for (int c = 0; c < 8; c++) a[col][c] /= d;
         for (int r = 0; r < 4; r++)
            if (r != col) {
               double f = a[r][col];
               for (int c = 0; c < 8; c++) a[r][c] -= f * a[col][c];
            }
      }
      Mat4 out;
      for (int r = 0; r < 4; r++)
         for (int c = 0; c < 4; c++) out.at(r, c) = (float)a[r][4 + c];
      return out;
   }
   // row-major 3x4 (VkTransformMatrixKHR layout) of the affine part
   std::array<float, 12> to_3x4() const {
      std::array<float, 12> r{};
      for (int rr = 0; rr < 3; rr++)
         for (int c = 0; c < 4; c++) r[rr * 4 + c] = at(rr, c);
      return r;
   }
};

// ---- camera.rs ----------------------------------------------------------------------------
class Camera {
  public:
   Camera(Vec3 pos, Vec3 target, float fov_degrees, float aspect_ratio, float z_near, float z_far)
       : pos_(pos), target_(target), fov_(fov_degrees), aspect_(aspect_ratio), near_(z_near), far_(z_far) {}
   void set_position_target(Vec3 pos, Vec3 target) {
      pos_ = pos;
      target_ = target;
   }
   Mat4 get_view() const { return Mat4::look_at_rh(pos_, target_, {0, 1, 0}); }
   Mat4 get_projection() const { return Mat4::perspective_rh(fov_ * 3.14159265358979323846f / 180.0f, aspect_, near_, far_); }
   Vec3 get_position() const { return pos_; }

  private:
   Vec3 pos_, target_;
   float fov_, aspect_, near_, far_;
};

// ---- gltf_loader.rs / primitive.rs --------------------------------------------------------
using Vertex = UhVertex;
using ViewUniformData = UhViewUniformData;
using Reservoir = UhReservoir;
constexpr uint32_t DEFAULT_TEXTURE_MAP = 0xffffffffu;  // gltf_loader.rs:9
enum class MaterialType : uint32_t { Lambertian = 0, Metal = 1, Dielectric = 2, DiffuseLight = 3, CookTorrance = 4 /* extension, utopian_hip.h */ };

struct Texture {
   uint32_t width = 0, height = 0;
   std::vector<uint8_t> rgba;
};
struct Material {
   uint32_t diffuse_map = DEFAULT_TEXTURE_MAP;
   float base_color_factor[4] = {1, 1, 1, 1};
   float metallic_factor = 1.0f, roughness_factor = 1.0f;
   MaterialType material_type = MaterialType::Lambertian;
   float material_property = 0.0f;  // metal -> fuzz, dielectric -> index of refraction
};
struct Primitive {
   std::vector<Vertex> vertices;
   std::vector<uint32_t> indices;
};
struct Mesh {
   Primitive primitive;
   Material material;
   uint32_t gpu_mesh = 0;
};
struct Model {
   std::vector<Mesh> meshes;
   std::vector<Mat4> transforms;  // one per mesh (node transforms flattened, gltf_loader.rs:47-63)
   std::vector<Texture> textures;
};
struct ModelInstance {
   Model model;
   Mat4 transform;
};

// ---- renderer.rs + raytracing.rs ------------------------------------------------------------
class Renderer {
  public:
   Renderer(int device_ordinal, uint32_t width, uint32_t height) : width_(width), height_(height) {
      int st = uh_create(device_ordinal, width, height, &ctx_);
      if (st != UH_OK) throw Error(st, std::string("Renderer::new: ") + uh_last_error(nullptr));
   }
   ~Renderer() { uh_destroy(ctx_); }
   Renderer(const Renderer&) = delete;
   Renderer& operator=(const Renderer&) = delete;

   // Renderer::initialize (renderer.rs:202-220): default 1x1 maps
   void initialize() {
      const uint8_t white[4] = {255, 255, 255, 255};
      check(uh_add_texture_rgba8(ctx_, white, 1, 1, &default_diffuse_map_index_), "initialize");
   }
   uint32_t add_bindless_texture(const Texture& t) {
      uint32_t idx = 0;
      check(uh_add_texture_rgba8(ctx_, t.rgba.data(), t.width, t.height, &idx), "add_bindless_texture");
      return idx;
   }
   // Renderer::add_model (renderer.rs:222-299): bindless remap of the diffuse map, one GpuMesh per mesh
   void add_model(Model model, const Mat4& transform) {
      std::vector<uint32_t> remap(model.textures.size(), DEFAULT_TEXTURE_MAP);
      for (size_t i = 0; i < model.meshes.size(); i++) {
         Mesh& mesh = model.meshes[i];
         UhGpuMaterial gm;
         std::memset(&gm, 0, sizeof(gm));
         if (mesh.material.diffuse_map == DEFAULT_TEXTURE_MAP) {
            gm.diffuse_map = default_diffuse_map_index_;
         } else {
            uint32_t& slot = remap.at(mesh.material.diffuse_map);
            if (slot == DEFAULT_TEXTURE_MAP) slot = add_bindless_texture(model.textures[mesh.material.diffuse_map]);
            gm.diffuse_map = slot;
         }
         std::memcpy(gm.base_color_factor, mesh.material.base_color_factor, sizeof(gm.base_color_factor));
         gm.metallic_factor = mesh.material.metallic_factor;
         gm.roughness_factor = mesh.material.roughness_factor;
         gm.raytrace_properties[0] = (float)(uint32_t)mesh.material.material_type;
         gm.raytrace_properties[1] = mesh.material.material_property;
         Mat4 world = transform * (i < model.transforms.size() ? model.transforms[i] : Mat4::identity());
         auto w = world.to_3x4();
         check(uh_add_mesh(ctx_, mesh.primitive.vertices.data(), (uint32_t)mesh.primitive.vertices.size(), mesh.primitive.indices.data(),
                           (uint32_t)mesh.primitive.indices.size(), &gm, w.data(), &mesh.gpu_mesh),
               "add_model");
      }
      instances.push_back(ModelInstance{std::move(model), transform});
   }
   // Renderer::add_light (renderer.rs:391-410)
   uint32_t add_light(Vec3 position, Vec3 color, float range) {
      UhGpuLight l;
      std::memset(&l, 0, sizeof(l));
      l.color[0] = color.x;
      l.color[1] = color.y;
      l.color[2] = color.z;
      l.position[0] = position.x;
      l.position[1] = position.y;
      l.position[2] = position.z;
      l.range = range;
      l.attenuation[2] = 0.1f;
      l.light_type = 1.0f;
      l.intensity[0] = l.intensity[1] = l.intensity[2] = 1.0f;
      uint32_t idx = 0;
      check(uh_add_light(ctx_, &l, &idx), "add_light");
      return idx;
   }
   uint32_t get_num_lights() const {
      uint32_t n = 0;
      uh_get_num_lights(ctx_, &n);
      return n;
   }
   // Raytracing::initialize (raytracing.rs:89-111)
   void initialize_raytracing() { check(uh_build_acceleration(ctx_), "Raytracing::initialize"); }
   // gizmo edit + Raytracing::rebuild_tlas (raytracing.rs:400-459)
   void set_instance_transform(uint32_t gpu_mesh, const Mat4& world) {
      auto w = world.to_3x4();
      check(uh_set_instance_transform(ctx_, gpu_mesh, w.data()), "set_instance_transform");
   }
   void rebuild_tlas() { check(uh_refit_acceleration(ctx_), "Raytracing::rebuild_tlas"); }

   std::vector<float> read_accumulation() {
      std::vector<float> out((size_t)width_ * height_ * 4);
      check(uh_read_accumulation(ctx_, out.data()), "read_accumulation");
      return out;
   }
   std::vector<uint8_t> read_output_bgra8() {
      std::vector<uint8_t> out((size_t)width_ * height_ * 4);
      check(uh_read_output_bgra8(ctx_, out.data()), "read_output_bgra8");
      return out;
   }
   UhStats get_stats() {
      UhStats s;
      check(uh_get_stats(ctx_, &s), "get_stats");
      return s;
   }
   // uh_set_option: "device_build", "frames_in_flight", ... (DESIGN.md "Options")
   void set_option(const char* name, int value) { check(uh_set_option(ctx_, name, value), name); }
   // marching_cubes.rs:17-83 / marching_cubes.comp: the density field's iso-surface, extracted on the GPU and added
   // as a mesh of the scene; returns the triangle count (0: nothing crosses the iso value, no mesh added)
   uint32_t add_isosurface_mesh(uint32_t resolution, float lo, float hi, float time, const Material& material, const Mat4& world = Mat4::identity()) {
      UhGpuMaterial m;
      std::memset(&m, 0, sizeof(m));
      m.diffuse_map = default_diffuse_map_index_;
      std::memcpy(m.base_color_factor, material.base_color_factor, sizeof(m.base_color_factor));
      m.metallic_factor = material.metallic_factor;
      m.roughness_factor = material.roughness_factor;
      m.raytrace_properties[0] = (float)(uint32_t)material.material_type;
      m.raytrace_properties[1] = material.material_property;
      auto w = world.to_3x4();
      uint32_t mesh = 0, tris = 0;
      check(uh_add_isosurface_mesh(ctx_, resolution, lo, hi, time, &m, w.data(), &mesh, &tris), "add_isosurface_mesh");
      return tris;
   }
   // ---- one process per GPU (the reference is single-device: utopian/src/device.rs:45) ----
   // path tracing: tiles t % world == rank; reservoir passes: this rank's band of rows, exchanged over RCCL inside the library
   // (rank 0 makes the id, the launcher hands the 128 bytes to every rank); composition on the root: compose_tiles
   void set_tile_partition(uint32_t rank, uint32_t world, uint32_t tile_size = 64) { check(uh_set_tile_partition(ctx_, rank, world, tile_size), "set_tile_partition"); }
   static std::array<uint8_t, 128> rccl_unique_id() {
      std::array<uint8_t, 128> id{};
      if (uh_rccl_unique_id(id.data()) != UH_OK) throw Error(UH_ERR_HIP, "uh_rccl_unique_id: librccl is not loadable");
      return id;
   }
   void rccl_attach(uint32_t rank, uint32_t world, const std::array<uint8_t, 128>& id) { check(uh_rccl_attach(ctx_, rank, world, id.data()), "rccl_attach"); }
   void set_restir_partition(uint32_t rank, uint32_t world, UhRestirExchangeFn exchange = nullptr, void* user = nullptr) {
      check(uh_set_restir_partition(ctx_, rank, world, exchange, user), "set_restir_partition");
   }
   void pack_tiles(void* device_out, uint64_t capacity_pixels) { check(uh_pack_tiles(ctx_, device_out, capacity_pixels), "pack_tiles"); }
   void compose_tiles(const void* device_all, uint64_t stride_pixels, uint32_t total_samples, uint32_t accumulation_limit = 999999) {
      check(uh_compose_tiles(ctx_, device_all, stride_pixels, total_samples, accumulation_limit), "compose_tiles");
   }
   uh_ctx* handle() { return ctx_; }
   uint32_t width() const { return width_; }
   uint32_t height() const { return height_; }
   void check(int st, const char* where) {
      if (st != UH_OK) throw Error(st, std::string(where) + ": " + uh_last_error(ctx_));
   }

   std::vector<ModelInstance> instances;

  private:
   uh_ctx* ctx_ = nullptr;
   uint32_t width_, height_;
   uint32_t default_diffuse_map_index_ = 0;
};

// ---- graph.rs (only what the path-tracing recipe needs) -------------------------------------
struct RenderPass {
   std::string name;
   std::function<void(Renderer&, const ViewUniformData&)> render_func;  // graph.rs:126-127
};
class Graph {
  public:
   void clear() { passes.clear(); }
   void add_pass(std::string name, std::function<void(Renderer&, const ViewUniformData&)> f) { passes.push_back({std::move(name), std::move(f)}); }
   // Graph::render (graph.rs:703-1065): passes in submission order on the single render thread
   void render(Renderer& renderer, const ViewUniformData& view) {
      for (auto& p : passes) p.render_func(renderer, view);
   }
   std::vector<RenderPass> passes;
};

// renderers::build_path_tracing_render_graph (renderers/mod.rs:189-375): same pass names, same order
inline void build_path_tracing_render_graph(Graph& graph) {
   auto node = [](uint32_t mask) {
      return [mask](Renderer& r, const ViewUniformData& v) { r.check(uh_render_frame(r.handle(), &v, mask), "render_func"); };
   };
   graph.add_pass("gbuffer_pass", node(UH_PASS_GBUFFER));
   graph.add_pass("reset_reservoirs_pass", node(UH_PASS_RESET_RESERVOIRS));
   graph.add_pass("initial_ris_pass", node(UH_PASS_INITIAL_RIS));
   graph.add_pass("temporal_reuse_pass", node(UH_PASS_TEMPORAL_REUSE));
   graph.add_pass("spatial_reuse_pass", node(UH_PASS_SPATIAL_REUSE));
   graph.add_pass("reference_pt_pass", node(UH_PASS_REFERENCE_PT));
}

// ---- prototype/src/main.rs: view defaults (:55-86) and the frame protocol (:460-471, :545-546) ----
inline ViewUniformData default_view_data(const Camera& camera, uint32_t width, uint32_t height) {
   ViewUniformData v;
   std::memset(&v, 0, sizeof(v));
   Mat4 view = camera.get_view(), proj = camera.get_projection(), iv = view.inverse(), ip = proj.inverse();
   Mat4 prev = Mat4::from_diagonal(-1, -1, -1, -1);
   std::memcpy(v.view, view.m, 64);
   std::memcpy(v.projection, proj.m, 64);
   std::memcpy(v.inverse_view, iv.m, 64);
   std::memcpy(v.inverse_projection, ip.m, 64);
   std::memcpy(v.prev_frame_projection_view, prev.m, 64);
   Vec3 e = camera.get_position();
   v.eye_pos[0] = e.x;
   v.eye_pos[1] = e.y;
   v.eye_pos[2] = e.z;
   v.samples_per_frame = 1;
   v.total_samples = 0;
   v.num_bounces = 5;
   v.viewport_width = width;
   v.viewport_height = height;
   Vec3 sun = normalize({0.0f, 0.9f, 0.15f});
   v.sun_dir[0] = sun.x;
   v.sun_dir[1] = sun.y;
   v.sun_dir[2] = sun.z;
   v.shadows_enabled = v.ssao_enabled = v.fxaa_enabled = v.cubemap_enabled = v.ibl_enabled = 1;
   v.sky_enabled = v.sun_shadow_enabled = v.lights_enabled = 1;
   v.max_num_lights_used = 10000;
   v.temporal_reuse_enabled = v.spatial_reuse_enabled = 1;
   v.rebuild_tlas = 0;
   v.accumulation_limit = 999999;
   v.use_ris_light_sampling = 1;
   v.raytracing_supported = 1;
   return v;
}

class Application {
  public:
   Application(Renderer& renderer, const Camera& camera) : renderer_(renderer), view_data(default_view_data(camera, renderer.width(), renderer.height())) {
      build_path_tracing_render_graph(graph);
   }
   void frame() {
      view_data.total_samples += view_data.samples_per_frame;  // main.rs:467-469, BEFORE the frame
      view_data.num_lights = renderer_.get_num_lights();        // main.rs:471
      graph.render(renderer_, view_data);
      Mat4 proj, view;
      std::memcpy(proj.m, view_data.projection, 64);
      std::memcpy(view.m, view_data.view, 64);
      Mat4 pv = proj * view;                                    // main.rs:545-546, AFTER the frame
      std::memcpy(view_data.prev_frame_projection_view, pv.m, 64);
   }
   // `count` frames of a static camera, handed to the library in one call (uh_render_frames batches frames into shared
   // wavefronts and keeps several in flight). With the reservoir passes the first frame goes through the graph alone: it is
   // the one that may still see another prev_frame_projection_view (main.rs:545-546 sets it AFTER a frame).
   void frames(uint32_t count, bool path_tracing_only) {
      if (!path_tracing_only && count > 0) {
         frame();
         count--;
      }
      if (count < 2) {
         for (uint32_t i = 0; i < count; i++) frame();
         return;
      }
      view_data.num_lights = renderer_.get_num_lights();
      view_data.total_samples += view_data.samples_per_frame;
      renderer_.check(uh_render_frames(renderer_.handle(), &view_data, path_tracing_only ? UH_PASS_REFERENCE_PT : UH_PASS_ALL, count), "render_frames");
      view_data.total_samples += (count - 1) * view_data.samples_per_frame;
   }
   Graph graph;

  private:
   Renderer& renderer_;

  public:
   ViewUniformData view_data;
};

}  // namespace utopian
