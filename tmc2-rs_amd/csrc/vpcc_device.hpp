// vpcc_device.hpp — structures shared by the host runtime and the gfx950 kernels.
// Internal to libvpcc_recon.so; the public boundary is include/vpcc_recon.h.
#pragma once

#include <stdint.h>

#include "vpcc_recon.h"

namespace vpcc {

// Patch as the kernels see it.  patch_to_canvas_helper (reference
// src/decoder.rs:853-867) is affine in (u, v) for every orientation, so the
// host folds orientation, uv0 and the reference's "size_uv0 stays in blocks"
// quirk into integer coefficients: x = ax_u*u + ax_v*v + cx, y likewise.
struct DevPatch {
  int32_t ax_u, ax_v, cx;
  int32_t ay_u, ay_v, cy;
  uint32_t u1, v1, d1;
  uint32_t lod_x, lod_y;
  uint8_t normal_axis, tangent_axis, bitangent_axis, projection_mode;
  uint32_t size_u0, size_v0;
  uint32_t vb_base;   // first virtual block of this patch (emission order)
  uint32_t bc;        // the BLOCK map's constants, cx | cy << 16 (patch_affine at resolution 1: the a-coefficients are the
                      // pixel map's, the constants are not — size_uv0 stays in blocks at every resolution): canvas block of
                      // the patch's block (u0, v0) = (ax_u*u0 + ax_v*v0 + cx, ay_u*u0 + ay_v*v0 + cy); both < 32768
};
static_assert(sizeof(DevPatch) == 64, "DevPatch is one 64-B record");

// One virtual block = (patch, v0, u0) in the reference's emission order
// (src/codec.rs:352-385): patch ascending, v0 outer, u0 inner — with everything of its patch that the evaluation of its
// pixels needs folded in (round 5): the general sequence reads ONE 32-byte record per block where it read a 16-byte block,
// then — a dependent round trip later — the 64-byte patch.
struct alignas(16) VBlock {
  uint32_t canvas_block;  // patch_block_to_canvas_block(u0, v0)
  uint16_t patch;
  uint8_t coef;           // patch_to_canvas (src/decoder.rs:853-867) is x = cu_x * u + cv_x * v + ..., coefficients in {-1, 0, 1}:
                          //   (cu_x + 1) | (cv_x + 1) << 2 | (cu_y + 1) << 4 | (cv_y + 1) << 6
  uint8_t axes_mode;      // normal | tangent << 2 | bitangent << 4 | projection_mode << 6
  uint16_t x0, y0;        // canvas pixel of the block's first pixel (u, v) = (u0 * R, v0 * R): inside the canvas (validate_frame)
  uint32_t t0, b0;        // its tangent / bitangent coordinate: u0 * R * lod_x + u1, v0 * R * lod_y + v1 (mod 2^32; src/decoder.rs:875-876)
  uint16_t lod_x, lod_y;  // level of detail mod 2^16 (the coordinates are truncated to u16)
  uint32_t d1;
  uint16_t u0, v0;        // the block of its patch
};
static_assert(sizeof(VBlock) == 32, "VBlock is 32 B");

#if defined(__HIPCC__)
#define VPCC_HD __host__ __device__
#else
#define VPCC_HD
#endif
// The host writes O(patches) per frame: vb_base[p] = sum of size_u0 * size_v0 over the patches before p (n_patches + 1
// entries, the last one the frame's number of virtual blocks).  Which patch a virtual block belongs to, and which block of
// it, is derived where it is needed — on the device (k_plan_tiles, k_plan_vblocks), and by the CPU tests of these two
// functions against the reference's loop nest (tests/fuzz_plan.cpp).
// Patch of virtual block vb < vb_base[n_patches]: the last patch whose base is <= vb (patches without blocks share their
// successor's base and are never the last).
VPCC_HD inline uint32_t patch_of_vblock(const uint32_t* vb_base, uint32_t n_patches, uint32_t vb) {
  uint32_t lo = 0, hi = n_patches;                    // vb_base[lo] <= vb < vb_base[hi]
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (vb_base[mid] <= vb) lo = mid; else hi = mid;
  }
  return lo;
}
// (patch, v0, u0) of a virtual block, its canvas block and the patch's part of its pixels' arithmetic, from the patch's record:
// src/codec.rs:352-385 (v0 outer, u0 inner), patch_block_to_canvas_block and patch_to_canvas, src/decoder.rs:827-867.
VPCC_HD inline VBlock vblock_of(const DevPatch& p, uint32_t patch, uint32_t vb, uint32_t bw, uint32_t R) {
  const uint32_t r = vb - p.vb_base, v0 = r / p.size_u0, u0 = r - v0 * p.size_u0;
  const int32_t bx = p.ax_u * (int32_t)u0 + p.ax_v * (int32_t)v0 + (int32_t)(p.bc & 0xFFFFu);
  const int32_t by = p.ay_u * (int32_t)u0 + p.ay_v * (int32_t)v0 + (int32_t)(p.bc >> 16);
  VBlock b{};
  b.patch = (uint16_t)patch;
  b.u0 = (uint16_t)u0;
  b.v0 = (uint16_t)v0;
  b.canvas_block = (uint32_t)by * bw + (uint32_t)bx;
  b.coef = (uint8_t)((p.ax_u + 1) | ((p.ax_v + 1) << 2) | ((p.ay_u + 1) << 4) | ((p.ay_v + 1) << 6));
  b.axes_mode = (uint8_t)(p.normal_axis | (p.tangent_axis << 2) | (p.bitangent_axis << 4) | (p.projection_mode << 6));
  const int32_t pu = (int32_t)(u0 * R), pv = (int32_t)(v0 * R);
  b.x0 = (uint16_t)(p.ax_u * pu + p.ax_v * pv + p.cx);
  b.y0 = (uint16_t)(p.ay_u * pu + p.ay_v * pv + p.cy);
  b.t0 = u0 * R * p.lod_x + p.u1;
  b.b0 = v0 * R * p.lod_y + p.v1;
  b.lod_x = (uint16_t)p.lod_x;
  b.lod_y = (uint16_t)p.lod_y;
  b.d1 = p.d1;
  return b;
}

// Work item of the single-pass tile kernel: one virtual block that OWNS its canvas block, with the
// patch fields the per-point arithmetic needs folded in (one 32-B scalar load per item).
// The host writes one TEMPLATE per patch — everything of an item that the patch decides — and the planning kernel
// completes a copy of it for every block the patch owns.  What the planning kernel needs to know of a patch beyond that
// travels in the three fields it overwrites: x0, y0 = the patch's origin in BLOCKS (uv0), patch = size_u0.
struct TileItem {
  uint16_t x0, y0;        // canvas pixel origin of the block                      (template: u0, v0 of the patch, in blocks)
  uint16_t patch;         // patch index (partition output)                        (template: size_u0)
  uint8_t flags;          // kTileSwap | kTileMode1
  uint8_t axes;           // normal | tangent << 2 | bitangent << 4
  uint32_t tb, bb;        // tangent / bitangent coordinate of the block's first pixel (mod 2^32):
                          //   u0*R*lod_x + u1 ,  v0*R*lod_y + v1     (src/decoder.rs:875-876)
  uint32_t d1;
  uint16_t lod_x, lod_y;
  uint32_t sel_xy, sel_z; // v_perm_b32 selectors that assemble {x | y << 16} and {z} from {normal | tangent << 16}
                          //   (bytes 0-3) and {bitangent} (bytes 4-5): the axis assignment of src/decoder.rs:871-888
};
static_assert(sizeof(TileItem) == 32, "TileItem is 32 B");
constexpr uint8_t kTileSwap = 1, kTileMode1 = 2;
constexpr uint32_t kTileItemsPerWave = 4;
constexpr uint32_t kTileWaves = 4;        // waves per workgroup of the tile kernel (1, 2 and 8 were measured and dropped: DESIGN.md)
constexpr uint32_t kTileItemsPerGroup = kTileWaves * kTileItemsPerWave;   // one ticket / one look-back word per group
// Look-back words are allocated one per kTileScanGranule items: enough for the finest ticket granularity any
// kernel structure uses (one ticket per wave = 4 items); a coarser structure uses the first words only.
constexpr uint32_t kTileScanGranule = 4;

// Per-frame descriptor, resident in HBM, read by every kernel.
struct DevFrame {
  const uint8_t* occ;
  const uint16_t* geo[2];
  const uint16_t* attr_y[2];
  const uint16_t* attr_u[2];
  const uint16_t* attr_v[2];
  const DevPatch* patches;    // general sequence (and tile frames too large for k_plan_tiles' LDS): host
  VBlock* vblocks;            // ... every (patch, v0, u0) in emission order: written by k_plan_vblocks when the gof is created
  const uint32_t* vb_base;    // n_patches + 1: first virtual block of every patch, then the frame's number of them (host)
  uint32_t* block_to_patch;   // bw*bh, 0 = unowned else patch+1
  uint32_t* vb_count;         // general sequence: the units' status words, 64 bits each: {generation : 30 | status : 2 | value : 32}
  uint32_t* vb_offset;        // (unused)
  vpcc_point3* out_xyz;
  vpcc_color3* out_rgb;
  uint16_t* out_patch;        // optional (partition), may be null
  uint32_t* n_points;         // device counter of this frame
  // single-pass tile kernel (R == 16, Default/Swap patches, aligned planes)
  TileItem* tiles;            // the frame's work items, emission order: written by the planning kernel of every launch
  const TileItem* patch_items;   // one item template per patch (host); null: a frame of the general sequence
  uint64_t* scan_state;       // one {status:2 | value:32} word per group of 16 items
  uint64_t* ticket;           // dynamic group counter {launch generation : 32 | groups drawn : 32} (deadlock-free ordering of
                              // the look-back chain; the generation makes a counter of an earlier launch read as fresh)
  uint32_t* error_flag;       // set when a bounded spin gives up
  uint32_t occ_stride, occ_w, occ_h;
  uint32_t geo_stride[2];
  uint32_t attr_stride[2], attr_cstride[2];
  uint32_t width, height, R, prec;
  uint32_t bw, bh;
  uint32_t n_patches, n_vblocks;
  uint32_t map_count, absolute_d1, has_attr;
  uint32_t capacity;
  uint32_t n_tiles;           // number of work items, written by the planning kernel (the array is padded to a multiple of 16 readable items)
  uint32_t prec_shift;        // log2(prec) when prec is a power of two (tile kernel)
};

// One grid cell of the smoothing filters (oracle/vpcc_smoothing_spec.h): all-zero = empty.
// Every field is a SUM, so that a wave updates a cell with ONE atomic instruction (four 64-bit adds: {count, s0},
// {s1, s2}, sp2, {sp, 0}; no carry crosses a pair's halves): the statistics kernel is bound by the number of atomic
// instructions a CU can issue, not by their lanes.  "The cell holds points of more than one patch" (the spec's
// min patch index != max patch index) is count * sp2 != sp * sp — equality in Cauchy-Schwarz holds exactly when all
// patch indices are equal; count < 2^22, index < 2^16: both sides stay below 2^60.
struct SmoothCell {
  uint32_t count;       // ... with kSmoothCountMixed above it once k_smooth_mark has seen the cell: the filters read the first 16 bytes only
  uint32_t s[3];        // coordinate sums (geometry) or R,G,B sums (colour)
  uint64_t sp2;         // sum of squared patch indices
  uint32_t sp;          // sum of patch indices
  uint32_t mixed;       // kSmoothMixed: the cell holds more than one patch; kSmoothPainted: the flags around it are set
                        // (k_smooth_mark, k_smooth_moved_mark; the statistics kernel adds 0 here)
};
static_assert(sizeof(SmoothCell) == 32, "SmoothCell is 32 B");
constexpr uint32_t kSmoothMixed = 1u, kSmoothPainted = 2u, kSmoothMovedHere = 4u;   // (a moved point left or entered the cell: k_smooth_moved_mark)
constexpr uint32_t kSmoothCountMixed = 1u << 31;
// 65 537 x 65 535 < 2^32: up to this many points in a cell no sum of 16-bit values (coordinates, patch indices) leaves its 32 bits
constexpr uint32_t kSmoothCellMaxPoints = 65537u;
// DevFrame::error_flag bits
constexpr uint32_t kErrorSpinLimit = 1u, kErrorSmoothCellOverflow = 2u;
// R, G, B sums of a cell when ONE statistics pass serves both filters (same grid size): a parallel array.
// What the colour filter needs of a cell, in 16 bytes: the point count with the mixed bit above it, and the sums.
struct SmoothColorCell { uint32_t count, s[3]; };
constexpr uint32_t kColorCellMixed = 1u << 31;
static_assert(sizeof(SmoothColorCell) == 16, "SmoothColorCell is 16 B");


// Scratch of the smoothing filters for a chunk of frames: per frame a dense grid of w^3 cells (all-zero between
// launches), the neighbourhood flags, and the list of the cells every chunk of 256 consecutive points touched (the
// geometry filter moves points in place, so the cells to clear afterwards cannot be recomputed from the positions).
constexpr uint32_t kSmoothListSpan = 1024;        // consecutive points (four chunks of 256: one wave of the statistics kernel) per cell list
constexpr uint32_t kSmoothListLen = 1024;         // entries reserved per list: one per point at worst; a few dozen are used
struct SmoothGrid {
  unsigned char* base;        // frame slot j at base + j * slot_bytes: [w^3 cells | (w + 1)^3 + w^3 flag bytes | w^3 colour cells],
  size_t slot_bytes;          //   all-zero between launches
  size_t flags_offset;        // bytes from the slot's start to its flags: one per 2x2x2 neighbourhood, named by its lower
                              //   corner + 1 in [0, w]^3 — "one of these eight cells mixes patches"
  size_t near_offset;         //   ... and to w^3 more, one per cell: "one of the 27 cells around this one mixes patches"
  uint32_t* list_base;        // cell lists of frame slot j at list_base + j * list_stride: kSmoothListLen entries per span of points —
  size_t list_stride;         //   the cells the span's points fall into (their own allocation: no zero invariant)
  uint32_t* count_base;       //   ... and how many entries each list holds (frame slot j at count_base + j * count_stride)
  size_t count_stride;
  uint32_t* flag_base;        //   ... and whether any of its cells has a mixed cell next to it (k_smooth_spans; same stride)
  uint64_t* painted_base;     //   ... and which of its entries k_smooth_mark painted flags for: a bit per entry (frame slot j at
  size_t painted_stride;      //   painted_base + j * painted_stride)
  size_t color_offset;        // both filters in one pass: bytes from the slot's start to its w^3 colour cells (else 0)
  uint64_t* moved_base;       //   ... which points the geometry filter moved: one bit per point, a 64-bit word per 64
  size_t moved_stride;        //   points (frame slot j at + j * moved_stride; zeroed by vpcc_gof_smooth before every pass),
  uint32_t* oldkey_base;      //   and the cell a moved point was counted in (frame slot j at + j * oldkey_stride; written
  size_t oldkey_stride;       //   for those points only)
  uint64_t* moved_painted_base;   // ... and which moved points k_smooth_moved_mark painted flags for, around the cell they left /
                              //   entered: two words per word of moved bits (frame slot j at + j * 2 * moved_stride)
  VPCC_HD SmoothCell* cells(uint32_t j) const { return reinterpret_cast<SmoothCell*>(base + j * slot_bytes); }
  VPCC_HD SmoothColorCell* color_cells(uint32_t j) const { return reinterpret_cast<SmoothColorCell*>(base + j * slot_bytes + color_offset); }
  VPCC_HD uint64_t* moved(uint32_t j) const { return moved_base + j * moved_stride; }
  VPCC_HD uint32_t* old_keys(uint32_t j) const { return oldkey_base + j * oldkey_stride; }
  VPCC_HD unsigned char* flags(uint32_t j) const { return base + j * slot_bytes + flags_offset; }
  VPCC_HD unsigned char* near(uint32_t j) const { return base + j * slot_bytes + near_offset; }
  VPCC_HD uint32_t* lists(uint32_t j) const { return list_base + j * list_stride; }
  VPCC_HD uint32_t* list_counts(uint32_t j) const { return count_base + j * count_stride; }
  VPCC_HD uint32_t* span_flags(uint32_t j) const { return flag_base + j * count_stride; }
  // "a cell has become mixed behind the geometry filter": the stride's last word (the lists need one word less at least)
  VPCC_HD uint32_t* frame_dirty(uint32_t j) const { return flag_base + j * count_stride + (count_stride - 1); }
  VPCC_HD uint64_t* painted(uint32_t j) const { return painted_base + j * painted_stride; }
  VPCC_HD uint64_t* moved_painted(uint32_t j) const { return moved_painted_base + j * 2 * moved_stride; }
};

void launch_smooth_stats(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg,
                         uint32_t w, uint32_t G, uint32_t mode, void* stream);
// mode of launch_smooth_stats: 0 coordinate sums, 1 colour sums, 2 both (colour sums into the colour cells); `both` of
// the others: the launch belongs to a pass that serves both filters.
void launch_smooth_apply_geometry(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points,
                                  SmoothGrid sg, uint32_t w, uint32_t G, uint32_t T, bool both, void* stream);
void launch_smooth_apply_color(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points,
                               SmoothGrid sg, uint32_t w, uint32_t G, uint32_t Ts, uint32_t Td, bool both, void* stream);
void launch_smooth_moved(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg,
                         uint32_t w, uint32_t G, void* stream);
void launch_smooth_spans(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg, void* stream);
void launch_smooth_mark(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg,
                        uint32_t w, void* stream);
void launch_smooth_clear(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg,
                         uint32_t w, uint32_t G, bool both, void* stream);

// Work lists of the tile kernel for frames [first, first + count): which virtual blocks own their canvas block and hold
// occupancy (generate_block_to_patch_from_occupancy_map_video, src/codec.rs:205-250), in emission order.  Needs the frames'
// occupancy planes in place; writes block_to_patch, tiles and n_tiles.
//   launch_plan_tiles: ONE kernel, a workgroup per frame with the frame's block_to_patch and patch table in LDS — frames of
//     up to kPlanLdsBlocks canvas blocks and kPlanLdsPatches patches (`lds_bytes` from plan_tiles_lds_bytes of the launch's
//     largest frame);
//   launch_plan_tiles_global: any frame — block_to_patch zeroed in global memory (here), k_plan_cover, k_plan_items over
//     the virtual blocks k_plan_vblocks wrote.
constexpr uint32_t kPlanLdsBlocks = 32768, kPlanLdsPatches = 2048;
constexpr size_t kPlanLdsMax = 160u * 1024u - 1024u;              // a workgroup may have all of a CU's 160 KB (less the kernel's static words)
// block_to_patch word per block, {vb_base, origin, size | swap} per patch (+ the 32-byte item templates where they fit)
VPCC_HD inline size_t plan_tiles_lds_bytes(uint32_t blocks, uint32_t patches, bool with_templates) {
  return 4u * (size_t)blocks + 12u * ((size_t)patches + 1u) + 16u + (with_templates ? 32u * (size_t)patches : 0u) + 256u;
}
// what a launch asks for: with the templates when that fits a workgroup's LDS
inline size_t plan_tiles_lds_launch_bytes(uint32_t blocks, uint32_t patches) {
  const size_t with = plan_tiles_lds_bytes(blocks, patches, true);
  return with <= kPlanLdsMax ? with : plan_tiles_lds_bytes(blocks, patches, false);
}
// write_block_to_patch: the frames' block_to_patch leaves the LDS for global memory too (vpcc_gof_block_to_patch asks for it;
// a launch does not: the tile kernel works from the items)
void launch_plan_tiles(DevFrame* d_frames, uint32_t first, uint32_t count, size_t lds_bytes, bool write_block_to_patch, void* stream);
void launch_plan_tiles_global(DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_vb, uint32_t* d_b2p, size_t b2p_words, void* stream);
// The virtual blocks of frames [first, first + count) from their patch tables (once per gof: they do not depend on the planes)
void launch_plan_vblocks(DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_vb, void* stream);
// Kernel launchers (vpcc_kernels.hip).  All asynchronous on `stream`.
// max_samples: the largest rectangle of occupancy samples under a block of the launch's frames, (R / precision + 1)^2
void launch_block_owner(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_vb, uint32_t max_samples,
                        void* stream);
// The general sequence's single pass; a frame has general_units(R, virtual blocks) units of up to 256 pixels each, and one
// 64-bit status word per unit (DevFrame::vb_count).  gen: the gof's launch counter (tags the status words).
#ifndef VPCC_GEN_UNITS
#define VPCC_GEN_UNITS 5
#endif
#ifndef VPCC_GENB_UNITS
#define VPCC_GENB_UNITS 4
#endif
constexpr uint32_t kGenUnitsPerGroup = VPCC_GEN_UNITS;   // units a workgroup of k_general takes: one status word per group
constexpr uint32_t kGenBlockUnits = VPCC_GENB_UNITS;     // ... and of k_general_blocks
#ifndef VPCC_GENB_STAGE
#define VPCC_GENB_STAGE 2
#endif
constexpr uint32_t kGenBlockStage = VPCC_GENB_STAGE;     // units of a group that pass through the LDS together on their way out
VPCC_HD inline uint32_t general_units(uint32_t R, uint32_t n_vblocks) {
  const uint64_t RR = (uint64_t)R * R;
  if (RR >= 256u) return (uint32_t)(n_vblocks * ((RR + 255u) / 256u));
  const uint32_t per = (uint32_t)(256u / RR);
  return (n_vblocks + per - 1u) / per;
}
// Which (frame of the launch, group) workgroup L of the pass takes, and how many workgroups a launch has: of `lanes` lanes — the eight
// XCDs (workgroup L runs on XCD L % 8) for launches of eight frames or more, else one per frame — lane x takes frames x, x + lanes, ...,
// `interleave` of them at a time with their groups in turn (why: vpcc_kernels.hip).  What the look-back relies on: every (frame,
// group) is taken exactly once, and L grows with the group inside a frame.  gen_shape picks lanes and interleave so that a small
// launch has no idle workgroups (a one-frame launch over 8 x 8 slots would dispatch 64 workgroups for every one that works).
struct GenWork { uint32_t frame, group; bool any; };
struct GenShape { uint32_t lanes, interleave, grid; };
VPCC_HD inline GenWork gen_work_of(uint32_t L, uint32_t count, uint32_t groups_per_frame, uint32_t interleave, uint32_t lanes) {
  const uint32_t lane = L % lanes, slot = L / lanes, per_round = interleave * groups_per_frame;
  const uint32_t round = slot / per_round, r = slot - round * per_round, group = r / interleave, fi = r - group * interleave;
  GenWork w;
  w.group = group;
  w.frame = (round * interleave + fi) * lanes + lane;
  w.any = w.frame < count;
  return w;
}
inline GenShape gen_shape(uint32_t count, uint32_t groups_per_frame, uint32_t max_interleave) {
  GenShape s;
  s.lanes = count >= 8u ? 8u : (count ? count : 1u);
  const uint32_t per_lane = (count + s.lanes - 1u) / s.lanes;
  s.interleave = per_lane < max_interleave ? (per_lane ? per_lane : 1u) : max_interleave;
  s.grid = s.lanes * ((per_lane + s.interleave - 1u) / s.interleave) * s.interleave * groups_per_frame;
  return s;
}
// block_units: every frame of the launch has FrameShape::block_units (vpcc_host.hpp) — k_general_blocks, whose units are chunks
// of ONE virtual block each, so that everything a block decides is scalar work; else k_general, which takes any frame.
void launch_general(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_units, uint32_t gen, bool block_units, void* stream);
// Where the workgroups of one tile-kernel launch start (kernel argument, by value).  A workgroup stays with its
// frame; frames differ in size (S-longdress +-5 %, S-owlii +-11 % between the largest frame and the mean), so the
// resident workgroups of an XCD are shared out among its frames in proportion to their tile counts instead of
// equally — with an equal split the largest frame sets the launch time.
constexpr uint32_t kTileMapSlots = 128;   // resident workgroups per XCD on MI355X (32 CUs x 4)
struct TileLaunchMap {
  uint8_t frame_of_slot[8][kTileMapSlots];   // frame = xcd + 8 * value (relative to the launch's first frame); 0xFF: none
  uint8_t wgs_of_slot[8][kTileMapSlots];     // workgroups that frame gets in this launch (planning invariant; the kernel
                                             // needs no head count: ticket counters carry the launch generation)
  uint32_t slots;                            // slots per XCD in use; 0: equal split over max_groups / depth workgroups per frame
};
// weights[i]: tiles of frame first + i.  resident_per_xcd: workgroups of this kernel an XCD holds at a time.
void plan_tile_launch(const uint32_t* tiles, uint32_t count, uint32_t resident_per_xcd, uint32_t depth, TileLaunchMap& map);
void launch_tiles(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_groups, uint32_t gen,
                  const TileLaunchMap& map, uint32_t resident_per_xcd,
                  void* stream);
// One piece of the plane ingest by kernel (k_ingest_planes): `bytes` (<= 64 KB) of page-locked host memory, through its
// device-visible address, to device memory; src and dst are congruent modulo 16.
struct IngestPiece { const void* src; void* dst; uint32_t bytes, pad; };
constexpr uint32_t kIngestPieceBytes = 65536;
void launch_ingest_planes(const IngestPiece* d_pieces, uint32_t n, void* stream);
// up to three arrays (bytes = low half, pad = high half of the size; 0: unused) from device memory to page-locked host memory
void launch_push_results(const IngestPiece pieces[3], void* stream);
// vpcc_ctx_reserve's probe: the tile kernel's output pattern between two arrays (positions: items * 1824 B, colours: items * 912 B)
void launch_probe_outputs(unsigned char* xyz, unsigned char* rgb, uint32_t items, void* stream);
void launch_warm_kernels(void* stream);   // empty kernels: the first launch out of a translation unit loads its code object
void launch_warm_tiles(void* stream);
void launch_upsample_occupancy(const DevFrame* d_frames, uint32_t frame, uint8_t* d_out, uint32_t width,
                               uint32_t height, void* stream);

}  // namespace vpcc
