// bvh.h — memory layout of the acceleration structure shared by the host builder (bvh_build.cpp)
// and the gfx950 traversal kernels (kernels.hip). Replaces the driver-private BLAS/TLAS of
// VK_KHR_acceleration_structure (reference: utopian/src/raytracing.rs:113-398).
//
// One flattened BVH4 over world-space triangles (instance transforms baked at build time):
//   node     128 B = one gfx950 L2 line: lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] child[4] meta[4]
//   TriPacket 48 B: v0, e1 = v1-v0, e2 = v2-v0, key = mesh<<22 | primitive, 2 spare dwords
//   ShadePacket 64 B (same order as TriPacket): object-space vertex normals, uvs, mesh index
#pragma once
#include <cstdint>
#include <vector>

namespace uh {

constexpr uint32_t kLeafBit = 0x80000000u;    // child ref: bit31 = leaf
constexpr uint32_t kEmptyRef = 0xffffffffu;   // unused child slot / empty stack
constexpr uint32_t kLeafCountShift = 27;      // leaf ref: bits 27..30 = triangle count (1..15)
constexpr uint32_t kLeafFirstMask = 0x07ffffffu;
constexpr uint32_t kMaxLeafTris = 4;
constexpr uint32_t kPrimBits = 22;            // key = mesh << 22 | prim  (mesh < 1024, prim < 4 Mi)
constexpr uint32_t kPrimMask = (1u << kPrimBits) - 1;

struct alignas(16) Node4 {
   float lox[4], loy[4], loz[4];
   float hix[4], hiy[4], hiz[4];
   uint32_t child[4];
   uint32_t meta[4];  // meta[0] = number of used child slots
};
static_assert(sizeof(Node4) == 128, "one node = one 128-byte line");

// Device node: the same 4 children with boxes quantised to 8 bits per plane relative to the node's
// own box (origin + 2^e * q), rounded outwards, so the slab test stays conservative and the hit
// result does not change. 64 B = four dwordx4 loads per visit instead of seven, half the bytes:
// the traversal kernels are bound by the L1/L2/Infinity-Cache gather of node lines, not by VALU.
struct alignas(16) Node4Q {
   float origin[3];    // lower corner of the node's own (padded) box
   float scale_x;      // per-axis quantisation step, a power of two
   float scale_yz[2];
   uint32_t qlo[3];    // per axis: child k's quantised lower plane in byte k
   uint32_t qhi[3];    // per axis: upper plane. Empty slot: qlo = 255, qhi = 0 (inverted box, never hit)
   uint32_t child[4];
};
static_assert(sizeof(Node4Q) == 64, "quantised node = half a 128-byte line");

struct alignas(16) TriPacket {
   float v0[3];
   float e1x;
   float e1yz[2];
   float e2[2];  // e2.x, e2.y
   float e2z;
   uint32_t key;
   uint32_t pad[2];
};
static_assert(sizeof(TriPacket) == 48, "triangle packet");

struct alignas(16) ShadePacket {
   float n0[3], n1[3], n2[3];  // object-space vertex normals
   float uv0[2], uv1[2], uv2[2];
   uint32_t mesh;
};
static_assert(sizeof(ShadePacket) == 64, "shading packet");

struct BuildInput {
   // world-space triangle corners, 9 floats per triangle, and its key
   const float* corners;
   const uint32_t* keys;
   uint32_t count;
};

struct BuildOutput {
   std::vector<Node4> nodes;         // BFS order, node 0 = root (full-precision, padded child boxes)
   std::vector<Node4Q> qnodes;       // the same tree, quantised (what the kernels traverse)
   std::vector<uint32_t> tri_order;  // packet i holds input triangle tri_order[i]
   uint32_t max_depth = 0;
   std::vector<uint32_t> level_start;  // BFS level l = nodes [level_start[l], level_start[l+1]); children always lie in a later level
};

// Binned-SAH BVH2 build, collapsed to BVH4, emitted breadth-first. Host-side, multi-threaded.
void build_bvh4(const BuildInput& in, BuildOutput& out, int num_threads, uint32_t max_leaf_tris = kMaxLeafTris, float sah_traversal_cost = 1.0f);

}  // namespace uh
