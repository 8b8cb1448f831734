// bvh.h — memory layout of the acceleration structure shared by the host builder (bvh_build.cpp), the device builder
// and refit (lbvh.hip, refit.hip) and the gfx950 traversal kernels (kernels.hip). Replaces the driver-private BLAS/TLAS of
// VK_KHR_acceleration_structure (reference: utopian/src/raytracing.rs:113-398).
//
// One flattened BVH4 over world-space triangles (instance transforms baked at build time). Every leaf is ONE triangle.
//   node        48 B = three 16-byte loads per visit (Node4C below)
//   TriPacket   48 B = three loads per triangle test: v0, e1 = v1-v0, e2 = v2-v0, key = mesh<<22 | primitive, 2 spare dwords
//   ShadePacket 64 B (same order as TriPacket): object-space vertex normals, uvs, mesh index
// Why 48-byte nodes with implicit child addresses: the traversal kernels are bound by the CU's vector-memory issue
// rate - one <=16-byte lane load per clock, whatever the address or the cache level that serves it
// (profiles/r02_microbench_rates.txt, profiles/r02g_counters.json: TA busy 79-87 %). A node visit therefore costs its
// number of 16-byte loads, and four explicit 32-bit child references were a whole load of the former 64-byte node.
// The children of a node are stored contiguously instead - its triangle children in the packet array from `tri_base`,
// its node children in the node array from `child_base` - and a slot's reference is base + slot.
#pragma once
#include <cstdint>
#include <vector>

namespace uh {

constexpr uint32_t kLeafBit = 0x80000000u;    // child ref / stack entry: bit31 = triangle, low bits = packet index
constexpr uint32_t kEmptyRef = 0xffffffffu;   // unused child slot / empty stack
constexpr uint32_t kMaxTriangles = 0x7ffffffeu;
constexpr uint32_t kPrimBits = 22;            // key = mesh << 22 | prim  (mesh < 1024, prim < 4 Mi)
constexpr uint32_t kPrimMask = (1u << kPrimBits) - 1;

constexpr int kMaxWidth = 8;
// full-precision node (host builder output; tests and the quantiser read it): padded child boxes + explicit child refs.
// Up to `width` (4 or 8) children are used. Slot order: triangles, then nodes, then empty.
struct NodeW {
   float lo[3][kMaxWidth], hi[3][kMaxWidth];
   uint32_t child[kMaxWidth];  // kLeafBit | packet, node index, or kEmptyRef
   uint32_t count;             // number of used child slots
};

// Device node. The children's boxes are quantised to 8 bits per plane in the node's FRAME (plane = origin + 2^(e-127) * q), rounded
// outwards, so the slab test stays conservative and hits do not change.
// Slots [0, n_tri) are triangles: packet tri_base + slot. Slots [n_tri, n_child) are nodes: node child_base + (slot - n_tri).
// The remaining slots are empty: an inverted box (qlo = 255, qhi = 0) that no ray hits.
// The first 16 bytes are the node's frame; everything else a visit needs lies in the other 32 (n_tri rides in the top bits of
// child_base), so that a build with UH_INHERIT_FRAME = 1 - the frame a function of the parent's frame and of this node's quantised box
// there, node_quant.h qn_inherit - can derive the frame in registers and load 32 bytes only. Measured slower (the derivation's ~45 VALU
// instructions cost more than the load they save): the default build gives every node its own frame.
struct alignas(16) Node4C {
   float origin[3];    // the frame's origin: at or below the lower corner of the node's own (padded) box
   uint32_t meta;      // bits 0-7 / 8-15 / 16-23: biased exponent of the x / y / z quantisation step (a power of two, byte >= 1);
                       // bits 24-26: n_tri; bits 28-30: n_child (both for the builders and the refit: the traversal reads neither here)
   uint32_t qlo[3];    // per axis: child slot k's quantised lower plane in byte k
   uint32_t qhi[3];    // per axis: upper plane
   uint32_t child_base;  // bits 0-28: first node child; bits 29-31: n_tri
   uint32_t tri_base;
};
static_assert(sizeof(Node4C) == 48, "device node = three 16-byte loads (two when the frame is inherited)");
constexpr uint32_t kMetaTriShift = 24, kMetaChildShift = 28;
// Build-time experiment (round 5, measured SLOWER and therefore off: profiles/README.md "inherited frames"): 1 = a node's frame is inherited
// from its parent (node_quant.h) and a traversal that descends into a child derives it in registers instead of loading the child's
// first quad. 0 = every node's frame is its own and every visit loads the three quads.
#ifndef UH_INHERIT_FRAME
#define UH_INHERIT_FRAME 0
#endif
constexpr uint32_t kChildBaseBits = 29, kChildBaseMask = (1u << kChildBaseBits) - 1;  // (node counts stay below 2^29: kMaxTriangles nodes at most, far fewer in practice)
// Stride of the device arrays in 16-byte units. A 48-byte record at a 48-byte stride straddles two 64-byte cache
// sectors half of the time; at a 64-byte stride (the last 16 bytes unused) every record is one sector - more bytes of
// working set against fewer sector fetches per record. Measured (profiles/README.md "record stride"): neutral for both
// arrays in round 2; with the sun grid, whose rays fetch packets and nothing else of the tree, triangle packets at 64 bytes
// are worth +3 % (config 1 7,620 -> 7,841 Mrays/s, 4K Bistro-class 6,230 -> 6,429, config 2 +1.9 %), nodes at 64 bytes -3 %.
#ifndef UH_NODE_STRIDE16
#define UH_NODE_STRIDE16 3
#endif
#ifndef UH_TRI_STRIDE16
#define UH_TRI_STRIDE16 4
#endif
constexpr uint32_t kNodeStride16 = UH_NODE_STRIDE16, kTriStride16 = UH_TRI_STRIDE16;

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
   uint32_t width = 4;               // children per node
   std::vector<NodeW> nodes;         // BFS order, node 0 = root (full-precision, padded child boxes)
   std::vector<Node4C> cnodes;       // width 4: the same tree, quantised (what the kernels traverse)
   std::vector<uint32_t> tri_order;  // packet i holds input triangle tri_order[i]; the triangle children of a node are consecutive packets
   uint32_t max_depth = 0;           // depth of the 4-wide tree (root = 0)
   std::vector<uint32_t> level_start;  // BFS level l = nodes [level_start[l], level_start[l+1]); children always lie in a later level
};

// Binned-SAH BVH2 down to single triangles, collapsed to BVH4, emitted breadth-first. Host-side, multi-threaded.
// balanced: median splits only (depth ceil(log2 n)) - the fallback for geometry whose SAH tree is deeper than the
// traversal stack holds (kMaxTreeLevels).
void build_bvh4(const BuildInput& in, BuildOutput& out, int num_threads, bool balanced = false, uint32_t width = 4);

// Binned-SAH binary tree over `count` boxes (6 floats each: lo xyz, hi xyz) - the top of a device-built tree, over the
// clusters its bottom-up rounds stopped at (lbvh.hip). Node 0 is the root (count >= 2); a child reference is a node
// index, or kLeafBit | box index. Serial: meant for thousands of boxes.
struct TopNode {
   uint32_t left, right;
   float half_area;
};
void build_sah_top(const float* boxes6, uint32_t count, std::vector<TopNode>& out);

// A traversal pushes at most 3 entries per level; the kernels' stack holds 16 (LDS) + 96 (scratch) entries per ray.
constexpr uint32_t kTraversalStackEntries = 16 + 96;
constexpr uint32_t kMaxTreeLevels = kTraversalStackEntries / 3;

// the full-precision tree (BFS order, node 0 the root) -> the device nodes: every node's frame is its own (UH_INHERIT_FRAME = 1: only the
// root's, the others' inherited from their parents - node_quant.h, shared with the refit kernels); children's planes rounded outwards in
// the node's frame
void quantise_tree(const std::vector<NodeW>& nodes, std::vector<Node4C>& out);

}  // namespace uh
