// vpcc_host.cpp — frame validation and patch-table planning (host side, no GPU calls).
#include "vpcc_host.hpp"

#include <algorithm>
#include <cstring>

namespace vpcc {

Affine patch_affine(const vpcc_patch& p, int64_t res) {
  const int64_t u0 = (int64_t)p.u0 * res, v0 = (int64_t)p.v0 * res;
  const int64_t su = p.size_u0, sv = p.size_v0;   // blocks at every resolution (reference quirk)
  Affine a{};
  switch (p.orientation) {
    case VPCC_ORIENT_DEFAULT: a = {1, 0, u0, 0, 1, v0}; break;
    case VPCC_ORIENT_ROT90:   a = {0, -1, sv - 1 + u0, 1, 0, v0}; break;
    case VPCC_ORIENT_ROT180:  a = {-1, 0, su - 1 + u0, 0, -1, sv - 1 + v0}; break;
    case VPCC_ORIENT_ROT270:  a = {0, 1, u0, -1, 0, su - 1 + v0}; break;
    case VPCC_ORIENT_MIRROR:  a = {-1, 0, su - 1 + u0, 0, 1, v0}; break;
    case VPCC_ORIENT_MROT90:  a = {0, -1, sv - 1 + u0, -1, 0, su - 1 + v0}; break;
    case VPCC_ORIENT_MROT180: a = {1, 0, u0, 0, -1, sv - 1 + v0}; break;
    case VPCC_ORIENT_MROT270:
    case VPCC_ORIENT_SWAP:    a = {0, 1, u0, 1, 0, v0}; break;
    default: break;
  }
  return a;
}

namespace {

// Are all mapped coordinates of u in [0,nu), v in [0,nv) inside [0,w) x [0,h)?  The map is
// affine, so the extremes sit at the corners.  Signed arithmetic: a negative value is what the
// reference's wrapping usize turns into a huge number that fails its assert.
bool extent_inside(const Affine& a, int64_t nu, int64_t nv, int64_t w, int64_t h) {
  if (nu <= 0 || nv <= 0) return true;   // empty loops never evaluate the assert
  const int64_t us[2] = {0, nu - 1}, vs[2] = {0, nv - 1};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) {
      const int64_t x = a.ax_u * us[i] + a.ax_v * vs[j] + a.cx;
      const int64_t y = a.ay_u * us[i] + a.ay_v * vs[j] + a.cy;
      if (x < 0 || y < 0 || x >= w || y >= h) return false;
    }
  return true;
}

}  // namespace

int validate_frame(const vpcc_frame_desc* f, FrameShape* shape) {
  if (!f) return VPCC_ERR_INVALID_ARG;
  if (f->width == 0 || f->height == 0 || f->occupancy_resolution == 0 || f->occupancy_precision == 0)
    return VPCC_ERR_INVALID_ARG;
  if (f->width > 32768 || f->height > 32768) return VPCC_ERR_INVALID_ARG;
  if (f->map_count < 1 || f->map_count > 2) return VPCC_ERR_UNSUPPORTED;       // multiple maps > 2 never produced
  if (f->attribute_count > 1) return VPCC_ERR_UNSUPPORTED;                      // src/decoder.rs:133
  if (f->flags & VPCC_FRAME_RGB444) return VPCC_ERR_UNSUPPORTED;                // unreachable in the reference
  if (f->patch_count && !f->patches) return VPCC_ERR_INVALID_ARG;
  if (f->patch_count > 65535) return VPCC_ERR_INVALID_ARG;
  if (!f->occupancy.y || f->occupancy.stride < f->occupancy.width) return VPCC_ERR_INVALID_ARG;

  const int64_t W = f->width, H = f->height, R = f->occupancy_resolution, prec = f->occupancy_precision;
  const int64_t bw = W / R, bh = H / R;

  // occupancy upsample touches every canvas pixel: Image::get bounds assert, src/decoder.rs:974
  if ((W - 1) / prec >= (int64_t)f->occupancy.width || (H - 1) / prec >= (int64_t)f->occupancy.height)
    return VPCC_ERR_PATCH_OUT_OF_CANVAS;

  // geometry frames f*map_count .. must exist: src/codec.rs:317-321
  uint64_t plane_bytes = (uint64_t)f->occupancy.width * f->occupancy.height;    // B = Wo*Ho + M*W*H*2 +
  for (uint32_t m = 0; m < f->map_count; ++m) {                                 //   M*(W*H*2 + 2*(W/2)*(H/2)*2) (+ 9*N at run time)
    const vpcc_image_u16& G = f->geometry[m];
    if (!G.y) return VPCC_ERR_SHORT_VIDEO;
    if (G.stride < G.width) return VPCC_ERR_INVALID_ARG;
    if ((int64_t)G.width < W || (int64_t)G.height < H) return VPCC_ERR_PATCH_OUT_OF_CANVAS;
    plane_bytes += (uint64_t)G.width * G.height * 2;
  }
  if (f->attribute_count) {
    for (uint32_t m = 0; m < f->map_count; ++m) {
      const vpcc_image_u16& A = f->attribute[m];
      if (!A.y || !A.u || !A.v) return VPCC_ERR_SHORT_VIDEO;                    // src/codec.rs:589-590, 637
      if (A.stride < A.width || A.cstride < A.width / 2) return VPCC_ERR_INVALID_ARG;
      if ((int64_t)A.width < W || (int64_t)A.height < H) return VPCC_ERR_PATCH_OUT_OF_CANVAS;
      plane_bytes += (uint64_t)A.width * A.height * 2 + 2ull * (A.width / 2) * (A.height / 2) * 2;
    }
  }

  uint64_t n_vb = 0;
  bool simple = true;                 // Default / Swap (/ MRot270 == Swap) patches with levels of detail that fit a tile item
  bool distinct_axes = true;          // (the reference's axes are a permutation; the interface takes any three)
  for (uint32_t i = 0; i < f->patch_count; ++i) {
    const vpcc_patch& p = f->patches[i];
    if (p.axis_of_additional_plane != 0) return VPCC_ERR_UNSUPPORTED;           // src/codec.rs:437
    if (p.normal_axis > 2 || p.tangent_axis > 2 || p.bitangent_axis > 2) return VPCC_ERR_INVALID_ARG;
    if (p.projection_mode > 1) return VPCC_ERR_INVALID_ARG;                     // unreachable!() decoder.rs:886
    if (p.orientation > VPCC_ORIENT_MROT270) return VPCC_ERR_INVALID_ARG;
    if (p.lod_x > 65535u || p.lod_y > 65535u) simple = false;
    if (p.normal_axis == p.tangent_axis || p.normal_axis == p.bitangent_axis || p.tangent_axis == p.bitangent_axis) distinct_axes = false;
    if (p.size_u0 == 0 || p.size_v0 == 0) continue;
    if (p.size_u0 > 65535 || p.size_v0 > 65535) return VPCC_ERR_PATCH_OUT_OF_CANVAS;
    if (p.orientation == VPCC_ORIENT_DEFAULT || p.orientation == VPCC_ORIENT_SWAP || p.orientation == VPCC_ORIENT_MROT270) {
      // the two orientations every stream has (use_eight_orientations_flag = 0), without the corner walk: the patch's
      // blocks are [u0, u0 + su) x [v0, v0 + sv) (Default) or [u0, u0 + sv) x [v0, v0 + su) (Swap) of the canvas, and a
      // rectangle of whole blocks that lies inside the canvas in blocks does so in pixels (bw * R <= W)
      const bool swap = p.orientation != VPCC_ORIENT_DEFAULT;
      if ((uint64_t)p.u0 + (swap ? p.size_v0 : p.size_u0) > (uint64_t)bw || (uint64_t)p.v0 + (swap ? p.size_u0 : p.size_v0) > (uint64_t)bh)
        return VPCC_ERR_PATCH_OUT_OF_CANVAS;
    } else {
      simple = false;
      // assert in patch_block_to_canvas_block, src/decoder.rs:835
      if (!extent_inside(patch_affine(p, 1), p.size_u0, p.size_v0, bw, bh)) return VPCC_ERR_PATCH_OUT_OF_CANVAS;
      // assert in patch_to_canvas, src/decoder.rs:848
      if (!extent_inside(patch_affine(p, R), (int64_t)p.size_u0 * R, (int64_t)p.size_v0 * R, W, H))
        return VPCC_ERR_PATCH_OUT_OF_CANVAS;
    }
    n_vb += (uint64_t)p.size_u0 * p.size_v0;
  }
  if (n_vb > 0x7FFFFFFFull) return VPCC_ERR_INVALID_ARG;
  if (shape) {
    shape->bw = (uint32_t)bw;
    shape->bh = (uint32_t)bh;
    shape->n_patches = f->patch_count;
    shape->n_vblocks = (uint32_t)n_vb;
    // Tile kernel (R = 16, Default/Swap patches): the pixels of a virtual block are exactly the pixels of its canvas block,
    // so every covering patch sees the same occupancy and the reference's ascending overwrite (src/codec.rs:217, 242-244)
    // leaves "highest covering patch, if any occupancy"
    shape->tile_eligible = simple && R == 16 && prec <= 16 && (prec & (prec - 1)) == 0;
    shape->tile_bound = shape->tile_eligible ? (uint32_t)std::min<uint64_t>(n_vb, (uint64_t)bw * bh) : 0u;
    uint32_t max_stride = f->occupancy.stride;
    for (uint32_t m = 0; m < f->map_count; ++m) {
      max_stride = std::max(max_stride, f->geometry[m].stride);
      if (f->attribute_count) max_stride = std::max(max_stride, std::max(f->attribute[m].stride, f->attribute[m].cstride));
    }
    shape->block_units = R >= 16 && R <= 256 && (R & (R - 1)) == 0 && (prec & (prec - 1)) == 0 && max_stride <= 65536u && distinct_axes;
    shape->plane_bytes = plane_bytes;
  }
  return VPCC_OK;
}

namespace {
// v_perm_b32 selectors of a tile item by its axes byte (normal | tangent << 2 | bitangent << 4): coordinate a takes
// bitangent if bitangent_axis == a, else tangent, else normal, else 0 — the reference assigns normal, tangent, bitangent
// in this order (src/decoder.rs:874-876)
struct AxisSelectors {
  uint32_t xy[64], z[64];
  AxisSelectors() {
    for (uint32_t axes = 0; axes < 64; ++axes) {
      const uint32_t n = axes & 3u, t = (axes >> 2) & 3u, b = (axes >> 4) & 3u;
      uint32_t sel[3];
      for (uint32_t a = 0; a < 3; ++a) {
        uint32_t v = 0x0C0Cu;
        if (n == a) v = 0x0100u;
        if (t == a) v = 0x0302u;
        if (b == a) v = 0x0504u;
        sel[a] = v;
      }
      xy[axes] = sel[0] | (sel[1] << 16);
      z[axes] = sel[2] | 0x0C0C0000u;
    }
  }
};
const AxisSelectors kAxisSelectors;
}  // namespace

void write_frame_records(const vpcc_frame_desc& f, uint32_t* vb_base, TileItem* items, DevPatch* patches) {
  const int64_t R = f.occupancy_resolution;
  uint32_t base = 0;
  for (uint32_t i = 0; i < f.patch_count; ++i) {
    const vpcc_patch& p = f.patches[i];
    vb_base[i] = base;
    if (items) {
      // one item TEMPLATE per patch: the fields of a work item that the patch decides, and — in the fields the planning kernel
      // overwrites — where the patch lies: x0, y0 = uv0 in blocks, patch = size_u0
      TileItem t{};
      t.x0 = (uint16_t)p.u0;
      t.y0 = (uint16_t)p.v0;
      t.patch = (uint16_t)p.size_u0;
      t.flags = (uint8_t)((p.orientation == VPCC_ORIENT_DEFAULT ? 0 : kTileSwap) | (p.projection_mode ? kTileMode1 : 0));
      t.axes = (uint8_t)(p.normal_axis | (p.tangent_axis << 2) | (p.bitangent_axis << 4));
      t.tb = p.u1;                                     // + u0 * 16 * lod_x of the block   (src/decoder.rs:875-876)
      t.bb = p.v1;                                     // + v0 * 16 * lod_y
      t.d1 = p.d1;
      t.lod_x = (uint16_t)p.lod_x;
      t.lod_y = (uint16_t)p.lod_y;
      t.sel_xy = kAxisSelectors.xy[t.axes];
      t.sel_z = kAxisSelectors.z[t.axes];
      items[i] = t;
    }
    if (patches) {
      const Affine px = patch_affine(p, R), bl = patch_affine(p, 1);
      DevPatch d{};
      d.ax_u = (int32_t)px.ax_u; d.ax_v = (int32_t)px.ax_v; d.cx = (int32_t)px.cx;
      d.ay_u = (int32_t)px.ay_u; d.ay_v = (int32_t)px.ay_v; d.cy = (int32_t)px.cy;
      d.u1 = p.u1; d.v1 = p.v1; d.d1 = p.d1;
      d.lod_x = p.lod_x; d.lod_y = p.lod_y;
      d.normal_axis = p.normal_axis; d.tangent_axis = p.tangent_axis; d.bitangent_axis = p.bitangent_axis;
      d.projection_mode = p.projection_mode;
      d.size_u0 = p.size_u0; d.size_v0 = p.size_v0;
      d.vb_base = base;
      // the image of block (0, 0) lies inside the canvas (validate_frame) — patches without blocks are never looked at
      d.bc = ((uint32_t)bl.cx & 0xFFFFu) | ((uint32_t)bl.cy << 16);
      patches[i] = d;
    }
    base += p.size_u0 * p.size_v0;
  }
  vb_base[f.patch_count] = base;
}

// ------------------------------------------------------------------------------------------------ memory of a gof
namespace {
struct PlaneRef { const char* src; size_t bytes; size_t* slot; int part; };
// every plane the gof ingests, with the slot that will say where it lies on the device
template <class Fn>
void for_each_plane(const vpcc_frame_desc& F, PlaneSlots& o, Fn&& fn) {
  fn(F.occupancy.y, (size_t)F.occupancy.width * F.occupancy.height, &o.occ, F.occupancy.stride == F.occupancy.width);
  for (uint32_t m = 0; m < F.map_count; ++m) {
    fn(F.geometry[m].y, (size_t)F.geometry[m].width * F.geometry[m].height * 2, &o.geo[m], F.geometry[m].stride == F.geometry[m].width);
    if (F.attribute_count) {
      fn(F.attribute[m].y, (size_t)F.attribute[m].width * F.attribute[m].height * 2, &o.ay[m], F.attribute[m].stride == F.attribute[m].width);
      // chroma keeps its source stride: the reference indexes it as a flat array (v/2)*(width/2)+(u/2), src/decoder.rs:977
      fn(F.attribute[m].u, chroma_elems(F.attribute[m]) * 2, &o.au[m], true);
      fn(F.attribute[m].v, chroma_elems(F.attribute[m]) * 2, &o.av[m], true);
    }
  }
}
}  // namespace

bool classify_extents(const vpcc_frame_desc* frames, uint32_t n_frames, const PinnedQuery& pinned, GofLayout* layout,
                      std::vector<IngestExtent>* extents) {
  extents->clear();
  std::vector<PlaneSlots> slots(n_frames);
  std::vector<PlaneRef> refs;
  bool tight = true;
  for (uint32_t i = 0; i < n_frames && tight; ++i)
    for_each_plane(frames[i], slots[i], [&](const void* src, size_t bytes, size_t* slot, bool is_tight) {
      tight = tight && is_tight;
      refs.push_back(PlaneRef{(const char*)src, bytes, slot, gof_part_of(i)});
    });
  if (!tight || refs.empty()) return false;
  std::stable_sort(refs.begin(), refs.end(), [](const PlaneRef& a, const PlaneRef& b) { return a.part != b.part ? a.part < b.part : a.src < b.src; });
  const size_t kGap = 256u << 10;                         // what may lie between two planes of a stretch (patch tables, headers)
  std::vector<std::pair<size_t, size_t>> span;             // [first, last] plane of every stretch
  for (size_t k = 0; k < refs.size(); ++k) {
    const char* end = extents->empty() ? nullptr : extents->back().lo + extents->back().bytes;
    if (!extents->empty() && extents->back().part == refs[k].part && refs[k].src <= end + kGap) {
      extents->back().bytes = std::max<size_t>(extents->back().bytes, (size_t)(refs[k].src + refs[k].bytes - extents->back().lo));
      span.back().second = k;
    } else {
      extents->push_back(IngestExtent{refs[k].src, refs[k].bytes, 0, refs[k].part, {}});
      span.emplace_back(k, k);
    }
  }
  // worth it when stretches are long, and every stretch must be page-locked memory from end to end (what lies between its
  // planes is copied along)
  bool ok = extents->size() * 4 <= refs.size();
  for (size_t e = 0; e < extents->size() && ok; ++e) ok = pinned((*extents)[e].lo, (*extents)[e].bytes, &(*extents)[e].pieces);
  if (!ok) { extents->clear(); return false; }
  for (size_t e = 0; e < extents->size(); ++e) {
    // the device copy lies where the host stretch lies modulo 256: every plane keeps its alignment
    IngestExtent& E = (*extents)[e];
    const size_t shift = (uintptr_t)E.lo & 255u;
    E.dev = layout->block[2 * E.part + 0].take(E.bytes + 256) + shift;
    for (size_t k = span[e].first; k <= span[e].second; ++k) *refs[k].slot = E.dev + (size_t)(refs[k].src - E.lo);
  }
  for (uint32_t i = 0; i < n_frames; ++i) layout->f[i].planes = slots[i];
  return true;
}

void place_planes(const GofLayoutRequest& rq, GofLayout* L) {
  L->ingest_bound = 0;
  for (uint32_t i = 0; i < rq.n_frames; ++i) {
    ArenaLayout& B = L->block[2 * gof_part_of(i) + 0];
    for_each_plane(rq.frames[i], L->f[i].planes, [&](const void* src, size_t bytes, size_t* slot, bool is_tight) {
      // (+ 16: a plane pulled by the ingest kernel starts 0 or 8 bytes behind its 256-byte boundary — where its source does
      // modulo 16, so that 16-byte pieces line up on both sides)
      const size_t shift = rq.pull_ingest && is_tight && ((uintptr_t)src & 7u) == 0 ? ((uintptr_t)src & 15u) : 0u;
      *slot = B.take(bytes + 16) + shift;
      L->ingest_bound += bytes / kIngestPieceBytes + 2;
    });
  }
}

void layout_gof(const GofLayoutRequest& rq, GofLayout* Lp) {
  GofLayout& G = *Lp;
  const uint32_t n = rq.n_frames;
  ArenaLayout L;
  G.frames = L.take(sizeof(DevFrame) * n);
  // what the HOST writes — frame descriptors, vb_base, item templates, patches — lies together at the arena's start: it is
  // put together in a page-locked staging buffer and arrives as ONE copy at the head of the gof's ingest
  for (uint32_t i = 0; i < n; ++i) {
    const size_t P = rq.shapes[i].n_patches;
    G.f[i].vb_base = L.take(sizeof(uint32_t) * (P + 1));
    G.f[i].patch_items = rq.tile_records ? L.take(sizeof(TileItem) * std::max<size_t>(P, 1)) : 0;
    G.f[i].patches = rq.general_records ? L.take(sizeof(DevPatch) * std::max<size_t>(P, 1)) : 0;
  }
  G.host_end = L.total;
  G.counts = L.take(sizeof(uint32_t) * n);
  // control words of the single-pass path: one contiguous region, zeroed once at creation
  size_t scan_words = 0;
  for (uint32_t i = 0; i < n; ++i) {
    G.f[i].scan_word = scan_words;
    scan_words += (rq.shapes[i].tile_bound + kTileScanGranule - 1) / kTileScanGranule;
  }
  G.ctrl_begin = L.total;
  G.tickets = L.total;
  L.total += 256 * (size_t)n;                     // one ticket per 256-B line: same-line atomics serialise
  G.errors = L.total;
  L.total += sizeof(uint32_t) * n;
  L.total = align_up(L.total, 8);
  G.scan = L.total;
  L.total += sizeof(uint64_t) * std::max<size_t>(scan_words, 1);
  // ... and of the general sequence: one 64-bit status word per unit of up to 256 pixels (they carry the launch generation:
  // zeroed once with the rest of the region, they read as "of no launch")
  for (uint32_t i = 0; i < n; ++i) {
    G.f[i].vb_count = L.total;
    if (rq.general_records) L.total += sizeof(uint64_t) * general_units(rq.frames[i].occupancy_resolution, rq.shapes[i].n_vblocks);
  }
  G.ctrl_bytes = L.total - G.ctrl_begin;
  L.total = align_up(L.total, 256);
  // block_to_patch of all frames contiguous: one memset per launch where it is cleared in global memory
  G.b2p_begin = L.total;
  for (uint32_t i = 0; i < n; ++i) {
    G.f[i].b2p = L.total;
    L.total += sizeof(uint32_t) * (size_t)rq.shapes[i].bw * rq.shapes[i].bh;
  }
  G.b2p_words = (L.total - G.b2p_begin) / sizeof(uint32_t);
  L.total = align_up(L.total, 256);
  const size_t cap = (size_t)rq.capacity;
  for (uint32_t i = 0; i < n; ++i) {
    const FrameShape& S = rq.shapes[i];
    FrameOffsets& o = G.f[i];
    const size_t vb = std::max<size_t>(S.n_vblocks, 1);
    o.items = rq.tile_records ? L.take(sizeof(TileItem) * (((S.tile_bound + kTileItemsPerGroup - 1) / kTileItemsPerGroup) * kTileItemsPerGroup + kTileItemsPerGroup)) : 0;
    o.vblocks = rq.general_records ? L.take(sizeof(VBlock) * vb) : 0;
    o.vb_offset = 0;
    // output block: positions, colours, partition; + 4: the smoothing kernels read whole quads of points, so a quad that
    // begins inside an array must end in memory
    ArenaLayout& B = G.block[2 * gof_part_of(i) + 1];
    o.xyz = B.take(sizeof(vpcc_point3) * (cap + 4));
    o.rgb = rq.frames[i].attribute_count ? B.take(sizeof(vpcc_color3) * (cap + 4)) : 0;
    o.pidx = rq.want_patch_index ? B.take(sizeof(uint16_t) * (cap + 4)) : 0;
  }
  G.ingest_pieces = rq.pull_ingest ? L.take(sizeof(IngestPiece) * std::max<size_t>(G.ingest_bound, 1)) : 0;
  G.arena_bytes = L.total;
  G.stage_counts = align_up(G.host_end, 256);
  G.stage_bytes = G.stage_counts + sizeof(uint32_t) * 2 * n;
}

uint32_t PoolExtents::add_run(char* ptr, size_t bytes, int kind) {
  runs.push_back(Run{ptr, bytes, kind});
  const uint32_t run = (uint32_t)runs.size() - 1;
  in_use[kind] += bytes;                                           // (give_back takes it off again)
  give_back(run, ptr, bytes);
  return run;
}

bool PoolExtents::take(int kind, size_t bytes, char** ptr_out, uint32_t* run_out) {
  auto& F = free_[kind];
  for (size_t k = 0; k < F.size(); ++k)
    if (F[k].bytes >= bytes) {
      *ptr_out = F[k].ptr;
      *run_out = F[k].run;
      F[k].ptr += bytes;
      F[k].bytes -= bytes;
      if (!F[k].bytes) F.erase(F.begin() + k);
      in_use[kind] += bytes;
      return true;
    }
  return false;
}

void PoolExtents::give_back(uint32_t run, char* ptr, size_t bytes) {
  const int kind = runs[run].kind;
  in_use[kind] -= bytes;
  auto& F = free_[kind];
  size_t k = 0;
  while (k < F.size() && F[k].ptr < ptr) ++k;
  F.insert(F.begin() + k, Extent{ptr, bytes, run});
  if (k + 1 < F.size() && F[k + 1].run == run && F[k].ptr + F[k].bytes == F[k + 1].ptr) { F[k].bytes += F[k + 1].bytes; F.erase(F.begin() + k + 1); }
  if (k > 0 && F[k - 1].run == run && F[k - 1].ptr + F[k - 1].bytes == F[k].ptr) { F[k - 1].bytes += F[k].bytes; F.erase(F.begin() + k); }
}

bool tile_planes_aligned(const DevFrame& d) {
  auto al = [](const void* p, uintptr_t a) { return ((uintptr_t)p % a) == 0; };
  // the lane's occupancy bytes (4 / 2 / 1 for precision 1 / 2 / >= 4) must lie inside one aligned dword
  if (d.prec_shift == 0 && (!al(d.occ, 4) || d.occ_stride % 4)) return false;
  if (d.prec_shift == 1 && (!al(d.occ, 2) || d.occ_stride % 2)) return false;
  // the kernel addresses every plane with 32-bit byte offsets
  const uint64_t lim = 1ull << 32;
  if ((uint64_t)d.occ_h * d.occ_stride >= lim) return false;
  // row offsets are formed with 24-bit multiplies
  if (d.occ_stride >= (1u << 24) || d.height >= (1u << 24)) return false;
  // the layers of one video share a row pitch: the kernel forms one offset per plane kind
  if (d.map_count > 1 && (d.geo_stride[0] != d.geo_stride[1] ||
                          (d.has_attr && (d.attr_stride[0] != d.attr_stride[1] || d.attr_cstride[0] != d.attr_cstride[1]))))
    return false;
  for (uint32_t m = 0; m < d.map_count; ++m) {
    if ((uint64_t)d.height * d.geo_stride[m] * 2 >= lim) return false;
    if (d.has_attr && ((uint64_t)d.height * d.attr_stride[m] * 2 >= lim || (uint64_t)d.height * d.attr_cstride[m] >= lim))
      return false;
    if (!al(d.geo[m], 8) || d.geo_stride[m] % 4 || d.geo_stride[m] >= (1u << 24)) return false;
    if (d.has_attr && (d.attr_stride[m] >= (1u << 24) || d.attr_cstride[m] >= (1u << 24))) return false;
    if (d.has_attr) {
#ifdef VPCC_LDS_STAGED_ATTRIBUTES
      // (tools/experiments/vpcc_tiles_lds_dma.hip) attribute tiles reach LDS by 16-byte LDS-DMA pieces: half a luma row /
      // a whole chroma row of a block
      if (!al(d.attr_y[m], 16) || d.attr_stride[m] % 8) return false;
      if (!al(d.attr_u[m], 16) || !al(d.attr_v[m], 16) || d.attr_cstride[m] % 8) return false;
#else
      if (!al(d.attr_y[m], 8) || d.attr_stride[m] % 4) return false;
      if (!al(d.attr_u[m], 4) || !al(d.attr_v[m], 4) || d.attr_cstride[m] % 2) return false;
#endif
    }
  }
  return true;
}

// Largest-remainder shares of `resident_per_xcd` workgroups among the frames of each XCD label, at most
// ceil(groups / depth) per frame (a workgroup should have a few groups to pipeline), handed out in a smooth weighted
// round-robin so that the workgroups of one frame start a few slots apart.
void plan_tile_launch(const uint32_t* tiles, uint32_t count, uint32_t resident_per_xcd, uint32_t depth, TileLaunchMap& map) {
  std::memset(&map, 0xFF, sizeof(map));
  map.slots = 0;
  const uint32_t R = resident_per_xcd < kTileMapSlots ? resident_per_xcd : kTileMapSlots;
  if (!R || !count || (count + 7u) / 8u > 254u || (count + 7u) / 8u > R) return;   // equal split
  for (uint32_t x = 0; x < 8; ++x) {
    std::vector<uint32_t> idx, cap, share;
    std::vector<double> want;
    double total = 0;
    for (uint32_t i = x; i < count; i += 8) {
      const uint32_t groups = (tiles[i] + kTileItemsPerGroup - 1u) / kTileItemsPerGroup;
      idx.push_back(i / 8u);
      cap.push_back(groups ? (groups + depth - 1u) / depth : 0u);
      total += tiles[i];
    }
    if (idx.empty() || total == 0) continue;
    share.assign(idx.size(), 0);
    want.assign(idx.size(), 0.0);
    uint32_t left = R;
    // shares; frames that hit their cap give their surplus back to the others
    std::vector<bool> fixed(idx.size(), false);
    for (int pass = 0; pass < 4 && left; ++pass) {
      double open_total = 0;
      for (size_t j = 0; j < idx.size(); ++j) if (!fixed[j]) open_total += tiles[x + 8u * j];
      if (open_total == 0) break;
      bool capped = false;
      for (size_t j = 0; j < idx.size(); ++j) {
        if (fixed[j]) continue;
        want[j] = left * (double)tiles[x + 8u * j] / open_total;
        if (want[j] >= cap[j]) { share[j] = cap[j]; fixed[j] = true; capped = true; }
      }
      if (!capped) break;
      left = R;
      for (size_t j = 0; j < idx.size(); ++j) if (fixed[j]) left -= share[j] < left ? share[j] : left;
    }
    uint32_t given = 0;
    for (size_t j = 0; j < idx.size(); ++j) if (fixed[j]) given += share[j];
    if (given < R) {
      uint32_t open_left = R - given, floor_sum = 0;
      for (size_t j = 0; j < idx.size(); ++j) if (!fixed[j]) { share[j] = (uint32_t)want[j]; floor_sum += share[j]; }
      for (uint32_t extra = open_left > floor_sum ? open_left - floor_sum : 0; extra; --extra) {   // largest remainders
        size_t best = idx.size();
        double best_rem = -1;
        for (size_t j = 0; j < idx.size(); ++j)
          if (!fixed[j] && share[j] < cap[j] && want[j] - share[j] > best_rem) { best_rem = want[j] - share[j]; best = j; }
        if (best == idx.size()) break;
        ++share[best];
      }
    }
    for (size_t j = 0; j < idx.size(); ++j) if (!share[j] && cap[j]) share[j] = 1;                // nobody is left out
    // smooth weighted round-robin
    std::vector<uint32_t> placed(idx.size(), 0);
    uint32_t s = 0;
    for (; s < kTileMapSlots; ++s) {
      size_t best = idx.size();
      double best_frac = 0;
      for (size_t j = 0; j < idx.size(); ++j) {
        if (placed[j] >= share[j]) continue;
        const double frac = (double)(share[j] - placed[j]) / share[j];
        if (frac > best_frac) { best_frac = frac; best = j; }
      }
      if (best == idx.size()) break;
      ++placed[best];
      map.frame_of_slot[x][s] = (uint8_t)idx[best];
      map.wgs_of_slot[x][s] = (uint8_t)share[best];
    }
    bool all = true;
    for (size_t j = 0; j < idx.size(); ++j) all = all && placed[j] == share[j];
    if (!all) { std::memset(&map, 0xFF, sizeof(map)); map.slots = 0; return; }                    // (more shares than slots)
    if (s > map.slots) map.slots = s;
  }
}


}  // namespace vpcc
