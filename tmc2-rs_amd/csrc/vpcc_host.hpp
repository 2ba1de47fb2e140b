// vpcc_host.hpp — host-side planning of one atlas frame: validation (the reference's asserts,
// evaluated up front) and translation of the patch table into the kernels' work lists.
#pragma once

#include <stddef.h>
#include <stdint.h>

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

struct FramePlan {
  std::vector<DevPatch> patches;
  std::vector<VBlock> vblocks;       // every (patch, v0, u0) in emission order
  std::vector<TileItem> patch_items; // tile kernel: per patch, the fields of a work item that do not depend on the block
                                     //   (k_plan_items fills in x0, y0, tb, bb of every block the patch owns)
  uint32_t tile_bound = 0;           // upper bound of the frame's work items: min(virtual blocks, canvas blocks)
  bool tile_eligible = false;        // R == 16, Default/Swap only, occupancy precision compatible
  uint32_t bw = 0, bh = 0;
  bool simple_orientations = true;   // only Default / Swap (/MRot270 == Swap) patches
  uint64_t plane_bytes = 0;          // occupancy + geometry luma + attribute Y,U,V planes
};

// VPCC_OK or the status that stands for the reference panic this frame would run into.
// Deliberately stricter than the reference in two data-dependent places (documented in DESIGN.md):
// geometry/attribute planes must cover the whole canvas, and must be present, regardless of
// whether an occupied pixel would actually touch the missing part.
int validate_frame(const vpcc_frame_desc* f);

// Requires validate_frame(f) == VPCC_OK.  The host only translates the patch table (affine patches, the list of virtual
// blocks in emission order, one item template per patch); WHICH virtual blocks own their canvas block and hold any
// occupancy — generate_block_to_patch_from_occupancy_map_video, src/codec.rs:205-250 — is decided on the device
// (k_plan_cover / k_plan_items), which reads the occupancy plane where it lies, whoever owns it.
void plan_frame(const vpcc_frame_desc& f, FramePlan* out);

// Elements of a chroma plane that the reference's flat index (v/2)*cstride + (u/2) can reach (src/decoder.rs:977):
// what the runtime uploads of a U or V plane.
inline size_t chroma_elems(const vpcc_image_u16& a) {
  if (a.width == 0 || a.height == 0) return 1;
  return (size_t)((a.height - 1) / 2) * a.cstride + (a.width - 1) / 2 + 1;
}

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
