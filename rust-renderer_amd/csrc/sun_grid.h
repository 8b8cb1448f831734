// sun_grid.h — visibility structure for the sun shadow rays of reference.rgen:63-79.
//
// Every sun shadow ray of a frame has the SAME direction (normalize(view.sun_dir), rgen:64) - 48 % of all path rays of the
// headline workload. In a frame whose w axis is that direction such a ray is a point (u, v) plus a start depth w0, so "is
// any triangle in front of it" needs no tree: the triangles are binned ONCE per sun direction into a uniform 2-D grid over
// the (u, v) plane - each cell lists the triangle packets whose (conservatively dilated) projection overlaps it, sorted by
// the packet's far depth, descending - and a ray tests the packets of its one cell, front (sun side) to back, until one
// occludes it or the list falls behind its origin. The test itself is the very tri_compute<ANY> of the tree walk on the same
// 48-byte packets, so the result is the same predicate - "some triangle accepts the ray in (tmin, tmax)" - bit for bit;
// only the set of triangles that gets ASKED shrinks, and it shrinks conservatively (sun_grid.cpp "margins").
// Replaces, for this ray class, the any-hit tree walk of k_trace_shadow<sun> (12.6 steps per ray) by about one cell
// look-up and a few triangle tests.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace uh {

// The cover shortcut and the rays' tmax (reference.rgen:45,66-67: tmax = 10000). A ray that starts below its cell's cover depth has
// the covering packet in front of it at t = w_packet(u, v) - w0 > tmin; the packet only occludes it when also t < tmax. The builder
// admits a packet as a cover only if its depth over the cell, margins included, stays within kSunCoverSlack of the stored cover
// depth, and the kernel takes the shortcut only while cover - w0 < kSunCoverReach: then t < kSunCoverReach + kSunCoverSlack <
// tmax. Rays deeper than that below their cover (scenes more than 9,000 units deep along the sun) ask the cell's packets or the
// tree like any other ray.
constexpr float kSunCoverReach = 9000.0f;
constexpr double kSunCoverSlack = 900.0;
// The coarse cover: one depth per block of 2^shift x 2^shift cells = the LOWEST cover depth of the block's cells (-inf when a cell of
// the block has none, or when the cells' cover depths lie more than kSunCoarseSpread apart). A ray below it is below its own cell's
// cover depth c as well, and c - w0 <= coarse + kSunCoarseSpread - w0 < kSunCoarseReach + kSunCoarseSpread = kSunCoverReach: the
// cell's own shortcut would have answered the same. It only exists to be SMALL: 4 x 4 cells per word are 0.7 MB for the 2.9 M
// cells of the config-1 scene - resident in each XCD's 4 MiB L2, where the 23 MB of cell records are not -, and about 85 % of the
// rays a cell's cover depth answers start below their block's coarse cover too.
constexpr float kSunCoarseSpread = 100.0f;
constexpr float kSunCoarseReach = kSunCoverReach - kSunCoarseSpread;

constexpr uint32_t kSunCellWords = 4;  // words per cell record of the sun grid (device side)

struct SunGridEntry {
   uint32_t packet;  // triangle packet index (TriPacket array, leaf order)
   float wmax;       // far end of the packet's depth range along the sun direction, padded
};

// what the kernel reads (device pointers)
struct SunGridDev {
   float U[3], V[3], W[3];  // orthonormal frame, W = the sun direction exactly as FrameParams::sun_dir
   float u0, v0, inv_cell;  // cell (ix, iy) covers u0 + [ix, ix+1) / inv_cell, v0 + [iy, iy+1) / inv_cell
   uint32_t nx, ny;
   uint32_t max_walk;            // cells with a longer list, and the border cells, hand their rays to the tree walk
   // camera grid only: a list longer than max_walk (hence not sorted) but no longer than this is walked WHOLE, without the early
   // exit - the same hit, a few more tests - and only longer ones go to the tree; when the grid's longest list fits (the usual
   // case: a few dozen pixels see more than max_walk packets) the tree-walk launch behind the grid kernel is not made at all
   uint32_t walk_whole;
   const uint32_t* cell_start;   // sun grid: nx * ny + 1 records of kSunCellWords words: offset into entries | cover depth (float bits) | the first entry's packet | its far depth; camera grid: plain offsets
                                 // cell that starts below the cover depth is occluded - some packet spans the whole cell in front of it
   const SunGridEntry* entries;
   // the lists once more, as 64-byte records that carry their packet (null: not built): sixteen floats per entry, in the entries'
   // order - v0.xyz e1.x | e1.yz e2.xy | e2.z key, this entry's wmax, the next entry's wmax | unused. One sector and one round
   // trip per triangle test instead of two of each (entry, then packet). Built by build_sun_inline_records.
   const float* recs;
   // the coarse cover (null: none): coarse_nx x ceil(ny >> coarse_shift) depths, block (cx >> coarse_shift, cy >> coarse_shift)
   const float* coarse;
   uint32_t coarse_shift, coarse_nx;
};

// 64 bytes per entry (SunGridDev::recs) from the entries and the packets where they lie; `out` holds 64 * num_entries bytes.
// Enqueued on `hip_stream`, not waited for. Returns a hipError_t as int.
int build_sun_inline_records(void* hip_stream, const void* d_packets, const SunGridEntry* d_entries, uint64_t num_entries, void* out);

// the coarse cover of a grid's cell records (two words per cell: offset | cover depth), `out`: ((nx + b - 1) / b) x ((ny + b - 1) / b)
// floats with b = 1 << shift. Enqueued on `hip_stream`, not waited for. Returns a hipError_t as int.
int build_sun_coarse_cover(void* hip_stream, const uint32_t* d_cells, uint32_t nx, uint32_t ny, uint32_t shift, float* out);

struct SunGridHost {
   float U[3], V[3], W[3];
   float u0 = 0, v0 = 0, inv_cell = 0;
   uint32_t nx = 0, ny = 0;
   std::vector<uint32_t> cell_start;
   // per cell: the depth below which every ray of the cell is occluded (some packet whose projection, eroded by twice the
   // margins, contains the whole cell lies in front of it by more than the rays' tmin); -inf: no such packet
   std::vector<float> cell_cover;
   uint64_t covered_cells = 0;
   std::vector<SunGridEntry> entries;
   // quality figures (what a ray can expect): entries per non-empty cell, the longest list, cells
   double mean_list = 0.0;
   uint32_t max_list = 0;
   double fallback_area = 0.0;  // share of the triangles' surface area whose cell sends its rays to the tree (border or long list)
   double build_ms = 0.0;
   std::string why_not;  // non-empty: the grid was not built (degenerate direction, over budget, lists too long)
};

struct SunGridLimits {
   uint64_t max_entries = 96ull << 20;  // 8 bytes each
   uint64_t max_cells = 24ull << 20;
   double entries_per_triangle = 96.0;  // target density: the cell size is the finest that keeps the estimate under it (MI355X, config 1, round 4: 24 / 48 / 96 / 200 = 0.347 / 0.315 / 0.291 / 0.285 ms of sun rays per frame at 2.9 / 3.4 / 4.5 / 6.8 ms of build; round 3, lists and packets apart: 0.590 / 0.515 / 0.490 / 0.479, tree walk 0.632)
   // beyond this the grid does not beat the tree walk. Round 4 (lists with their packets inline, two rays per lane): Sponza-class
   // atrium 4.1 entries per occupied cell, sun rays -50 % against the tree walk; Bistro-class street (config 3) 8.2 entries per
   // occupied cell: the frame +5 % (7,080 -> 7,430-7,480 Mrays/s at 4K) with lists of up to 48 / 64 / 96 walked.
   // (Round 3, entries and packets apart: 4.5 -> -37 %; 7.4 -> -9 % at 555 ms of host build, +2 % without the cover depth: limit 8.)
   double max_mean_list = 12.0;
   // a ray whose cell lists more than this (or is a border cell) walks the tree instead (k_trace_sun_grid's fallback queue).
   // Config 1: 32 / 48 / 64 = 9,905-9,941 / 9,932-9,984 / 9,868-9,870 Mrays/s
   uint32_t max_walk = 48;
   double max_fallback_area = 0.2;      // share of the scene's surface area that may lie in such cells; above it the grid is refused as a whole
};

// the grid's frame and raster: what a builder chooses and the kernel is told
struct SunGridParams {
   float U[3] = {0, 0, 0}, V[3] = {0, 0, 0}, W[3] = {0, 0, 0};
   float u0 = 0, v0 = 0, inv_cell = 0;
   uint32_t nx = 0, ny = 0;
};
// the orthonormal frame both builders use: W = the direction as given, U and V complete it (rounded to float)
void sun_grid_frame(const float sun_dir[3], SunGridParams& out);

// packets: n x 12 floats in TriPacket layout (v0.xyz e1.x | e1.yz e2.xy | e2.z key pad pad). sun_dir: the frame's
// normalised direction (the floats the kernels use). Returns false (and says why in out.why_not) when the direction is not
// finite, the scene is empty or a limit is exceeded; the caller then keeps the tree walk.
// forced: the raster (u0, v0, inv_cell, nx, ny) to bin into instead of the builder's own choice (tests hold the two builders against
// each other on the same raster)
bool build_sun_grid(const float* packets12, uint32_t n, const float sun_dir[3], const SunGridLimits& lim, int num_threads, SunGridHost& out, const SunGridParams* forced = nullptr);

// The same grid built on the device (sun_grid_build.hip) from the packets where they lie (d_packets: n records of kTriStride16 x 16
// bytes): a few milliseconds instead of the host builder's 130-550, nothing but a few dozen numbers over PCIe. `cells` (two words
// per cell, nx * ny + 1 of them: offset | cover depth) and `entries` are hipMalloc'ed and owned by the result. The lists of the
// cells a ray may walk (interior, at most lim.max_walk entries) are sorted as the host builder sorts them; the others - whose rays
// k_trace_sun_grid hands to the tree - are left in arrival order. Runs on `hip_stream` and waits for it.
struct SunGridDevice {
   uint32_t* cells = nullptr;
   SunGridEntry* entries = nullptr;
   SunGridParams params;
   uint64_t num_entries = 0;
   double mean_list = 0.0, fallback_area = 0.0, build_ms = 0.0;
   uint32_t max_list = 0;
   uint32_t max_list_interior = 0;  // the longest list of a cell a ray can be in (the border ring left out)
   std::string why_not;  // non-empty: refused, as the host builder refuses
   void release();
};
bool build_sun_grid_device(void* hip_stream, const void* d_packets, uint32_t n, const float sun_dir[3], const SunGridLimits& lim, const SunGridParams* forced, SunGridDevice& out);

// The camera grid: the same structure for the rays that leave ONE POINT - the primary rays of reference.rgen:31-47 and the
// G-buffer cast (renderers/gbuffer.rs:11-52) of a camera at rest. The raster is the frame: one cell per pixel, cell (px + 1, py + 1)
// of a (W + 2) x (H + 2) grid (the border ring takes what projects outside the frame); a cell lists every packet some ray through
// the pixel's square can be accepted by (margins: sun_grid_build.hip k_pg_project), sorted by a LOWER BOUND of the hit distance,
// ascending - stored as SunGridEntry::wmax = -bound, so the shared sort (descending) serves. k_trace_camera_grid walks a pixel's
// list front to back with the closest-hit test of the tree walk (tri_compute<false>: same t, same tie-break) and stops when the
// next bound exceeds the best hit: the same hit record, bit for bit, without a tree walk. inverse_view / inverse_projection:
// the 16 floats of UhViewUniformData (column-major), the very numbers primary_ray reads.
// (The camera grid's `cells` are nx * ny + 1 plain offsets - it has no cover depths -, the sun grid's are (offset, cover) pairs.)
bool build_camera_grid_device(void* hip_stream, const void* d_packets, uint32_t n, const float inverse_view[16], const float inverse_projection[16], uint32_t W, uint32_t H,
                              const SunGridLimits& lim, SunGridDevice& out);

}  // namespace uh
