// vpcc_host.hpp — host-side planning of a gof, all of it pure host logic (no HIP call: built with plain g++ under
// ASan/UBSan by tests/fuzz_plan.cpp): validation (the reference's asserts, evaluated up front), the records the host
// writes per frame — O(patches): which virtual blocks exist, own their canvas block and hold occupancy is decided on the
// device —, the classification of the caller's planes into stretches for the ingest, and the layout of a gof's memory.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <utility>
#include <vector>

#include "vpcc_device.hpp"

namespace vpcc {

// Integer coefficients of Patch::patch_to_canvas_helper (reference src/decoder.rs:853-867):
//   x = ax_u*u + ax_v*v + cx ,  y = ay_u*u + ay_v*v + cy      at resolution `res`
// (res = 1: block map, res = R: pixel map).  size_uv0 stays in BLOCKS at every resolution,
// exactly as the reference computes it.
struct Affine {
  int64_t ax_u, ax_v, cx, ay_u, ay_v, cy;
};
Affine patch_affine(const vpcc_patch& p, int64_t res);

// What validate_frame learns of a frame in its one pass over the patch table.
struct FrameShape {
  uint32_t bw = 0, bh = 0;           // canvas in blocks
  uint32_t n_patches = 0;
  uint32_t n_vblocks = 0;            // sum of size_u0 * size_v0: every (patch, v0, u0) the reference's loops visit
  uint32_t tile_bound = 0;           // upper bound of the frame's work items: min(virtual blocks, canvas blocks); 0: not eligible
  bool tile_eligible = false;        // R == 16, Default/Swap (/MRot270 == Swap) patches only, occupancy precision a power of two <= 16
  bool block_units = false;          // general sequence: k_general_blocks may take the frame — block side a power of two in [16, 256], occupancy
                                     // precision a power of two, plane strides <= 65536 elements, every patch's three axes distinct
  uint64_t plane_bytes = 0;          // occupancy + geometry luma + attribute Y,U,V planes
};

// VPCC_OK or the status that stands for the reference panic this frame would run into; fills *shape (may be null) when OK.
// Deliberately stricter than the reference in two data-dependent places (documented in DESIGN.md):
// geometry/attribute planes must cover the whole canvas, and must be present, regardless of
// whether an occupied pixel would actually touch the missing part.
int validate_frame(const vpcc_frame_desc* f, FrameShape* shape = nullptr);

// The records the host writes for one frame, O(patches) — requires validate_frame(f) == VPCC_OK:
//   vb_base[n_patches + 1]   first virtual block of every patch, then their number;
//   items[n_patches]         (tile path; null: none) one work-item TEMPLATE per patch, see TileItem;
//   patches[n_patches]       (general sequence, and tile frames too large for k_plan_tiles; null: none) the affine patches.
// WHICH virtual blocks own their canvas block and hold any occupancy — generate_block_to_patch_from_occupancy_map_video,
// src/codec.rs:205-250 — is decided on the device, which reads the occupancy plane where it lies, whoever owns it.
void write_frame_records(const vpcc_frame_desc& f, uint32_t* vb_base, TileItem* items, DevPatch* patches);

// Elements of a chroma plane that the reference's flat index (v/2)*cstride + (u/2) can reach (src/decoder.rs:977):
// what the runtime uploads of a U or V plane.
inline size_t chroma_elems(const vpcc_image_u16& a) {
  if (a.width == 0 || a.height == 0) return 1;
  return (size_t)((a.height - 1) / 2) * a.cstride + (a.width - 1) / 2 + 1;
}

// ------------------------------------------------------------------------------------------------ memory of a gof
constexpr int kGofParts = 2;          // the big blocks come in parts by frame: eight frames (one per XCD label) to part 0, the next
inline int gof_part_of(uint32_t frame) { return (int)((frame >> 3) % kGofParts); }   // eight to part 1, ... (DESIGN.md 4.1 "Two homes")

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
struct ArenaLayout {
  size_t total = 0;
  size_t take(size_t bytes) {
    const size_t off = total;
    total = align_up(total + bytes, 256);
    return off;
  }
};

// Plane ingest by extent.  Planes that lie next to each other in the caller's page-locked memory — the output of a decoder
// that allocates from one pool, a decoded-GOF container — keep that arrangement on the device and arrive as ONE copy per
// stretch: with the copy engine moving 146-MB stretches the link runs at 57 GB/s host -> device AND 53 GB/s device -> host
// at the same time; a kernel that pulls the same bytes gets 46 GB/s beside pushed results, 1 280 copies of single planes 34.
struct PlaneSlots { size_t occ = 0, geo[2] = {0, 0}, ay[2] = {0, 0}, au[2] = {0, 0}, av[2] = {0, 0}; };   // offsets into the planes block of the frame's part
struct IngestExtent {
  const char* lo;                    // the stretch in the caller's memory
  size_t bytes;
  size_t dev;                        // where it goes in the planes block of its part (congruent to lo modulo 256)
  int part;
  std::vector<std::pair<const char*, size_t>> pieces;   // one, unless the stretch crosses from one page-locked region into the next
};
// [lo, lo + bytes) as pieces that each lie inside one page-locked region; false: some byte of it lies in none
using PinnedQuery = std::function<bool(const char* lo, size_t bytes, std::vector<std::pair<const char*, size_t>>* pieces)>;
struct GofLayout;
// True: every plane of the frames belongs to a stretch (`extents`, sorted by part and address; layout->f[i].planes says where
// frame i's planes lie; layout->block[2 * part] has grown by the stretches; layout->f must have n_frames entries).  False: the
// planes are not tight, lie scattered (fewer than four planes per stretch on average), or a stretch is not page-locked from
// end to end — the layout is as it was.
bool classify_extents(const vpcc_frame_desc* frames, uint32_t n_frames, const PinnedQuery& pinned, GofLayout* layout,
                      std::vector<IngestExtent>* extents);

// Where everything of a gof lies: byte offsets into its arena (descriptors, control words, work lists) and into its up to
// 2 * kGofParts big blocks ([2 * part]: planes of the frames of the part, gofs that own their planes; [2 * part + 1]:
// positions, colours, partition).
struct FrameOffsets {
  size_t vb_base = 0, patch_items = 0, patches = 0;     // arena, host-written (the head of the arena: ONE copy from the staging buffer)
  size_t vblocks = 0, items = 0, b2p = 0, vb_count = 0, vb_offset = 0;   // arena, device-written
  size_t scan_word = 0;                                 // first look-back word of the frame (index into the scan region)
  size_t xyz = 0, rgb = 0, pidx = 0;                    // output block of the frame's part
  PlaneSlots planes;                                    // planes block of the frame's part
};
struct GofLayout {
  size_t frames = 0;                 // DevFrame[n]
  size_t host_end = 0;               // end of what the host writes
  size_t counts = 0;                 // uint32[n]: point counts
  size_t ctrl_begin = 0, tickets = 0, errors = 0, scan = 0, ctrl_bytes = 0;   // tile-kernel control words, one region
  size_t b2p_begin = 0, b2p_words = 0;                  // block_to_patch of all frames, contiguous
  size_t ingest_pieces = 0;          // IngestPiece[ingest_bound] (plane ingest by kernel)
  size_t ingest_bound = 0;
  size_t arena_bytes = 0;
  size_t stage_bytes = 0;            // page-locked staging: the host-written head + counts and error words coming back
  size_t stage_counts = 0;           // ... where those lie in it
  ArenaLayout block[2 * kGofParts];
  std::vector<FrameOffsets> f;
};
struct GofLayoutRequest {
  const vpcc_frame_desc* frames;
  const FrameShape* shapes;
  uint32_t n_frames;
  uint64_t capacity;                 // points per frame
  bool want_patch_index;
  bool tile_records;                 // item templates and work lists of the tile kernel
  bool general_records;              // DevPatch tables, virtual blocks and the arrays of the general sequence
  bool pull_ingest;                  // page-locked host planes pulled by kernel (k_ingest_planes): a plane whose address is a
                                     //   multiple of eight lies 0 or 8 bytes behind its 256-byte boundary — where its source
                                     //   does modulo 16 —, and the arena has room for the piece list
};
// A gof that keeps a copy of the caller's planes in places of its own (no stretches: classify_extents said no, or was not
// asked): every plane tight, at a 256-byte boundary of the planes block of its frame's part.  Fills layout->f[i].planes and
// layout->block[2 * part]; layout->ingest_bound = the 64-KB pieces a pull of all of them would take.
void place_planes(const GofLayoutRequest& rq, GofLayout* layout);
// The arena and the output blocks.  layout->f[i].planes and layout->block[2 * part] are kept as they are (place_planes /
// classify_extents, or nothing for borrowed planes); layout->f has n_frames entries.
void layout_gof(const GofLayoutRequest& rq, GofLayout* layout);

// Free space of a context's pool (vpcc_ctx_reserve): the classified slabs as RUNS — maximal stretches of granules of one
// kind of VRAM region — and per kind a list of free extents, sorted by address, coalesced within a run (never across a
// change of kind or of slab).  Pure host logic, no HIP: fuzzed under ASan/UBSan by tests/fuzz_plan.cpp.
struct PoolExtents {
  struct Run { char* ptr; size_t bytes; int kind; };
  struct Extent { char* ptr; size_t bytes; uint32_t run; };
  std::vector<Run> runs;
  std::vector<Extent> free_[2];
  size_t in_use[2] = {0, 0};
  uint32_t add_run(char* ptr, size_t bytes, int kind);            // a new run, all of it free; returns its index
  bool take(int kind, size_t bytes, char** ptr_out, uint32_t* run_out);   // first fit; false: no extent of that kind is big enough
  void give_back(uint32_t run, char* ptr, size_t bytes);          // exactly what take() handed out
};

// Alignment preconditions of the tile kernel's vector loads for the planes as the kernels will see
// them (device pointers and strides): 8-B aligned luma rows, 4-B aligned chroma pairs.
bool tile_planes_aligned(const DevFrame& d);

}  // namespace vpcc
