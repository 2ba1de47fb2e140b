// Robustness harness for the host-side frame validation and planning (vpcc_host.cpp), CPU only, built with
// -fsanitize=address,undefined by tests/test_plan_fuzz.py: random and adversarial patch tables must either be
// rejected by validate_frame or planned into work lists that stay inside the canvas.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#include "vpcc_host.hpp"

static uint64_t st = 0x243F6A8885A308D3ull;
static uint64_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static uint32_t below(uint32_t n) { return n ? (uint32_t)(rnd() % n) : 0; }

int main(int argc, char** argv) {
  const long iterations = argc > 1 ? std::atol(argv[1]) : 1000;
  static uint8_t dummy8[16];
  static uint16_t dummy16[16];
  long accepted = 0, rejected = 0, items = 0, rotated = 0;
  // The general sequence's (frame, group) of a workgroup (gen_work_of): every pair of the launch exactly once, a frame on ONE XCD
  // (workgroup L runs on XCD L % 8), and inside a frame the workgroup number grows with the group — what the look-back relies on.
  for (long it = 0; it < std::min(iterations, 200L); ++it) {
    const uint32_t count = 1 + below(it % 3 ? 140 : 12), groups = 1 + below(60);
    const vpcc::GenShape shape = vpcc::gen_shape(count, groups, 1 + below(16));
    const uint32_t grid = shape.grid, il = shape.interleave;
    if (shape.lanes != std::min(count, 8u) || (uint64_t)grid > (uint64_t)8 * il * groups * ((count + 8 * il - 1) / (8 * il)) ||
        (count <= 8 * il && grid != shape.lanes * ((count + shape.lanes - 1) / shape.lanes) * groups)) {
      std::fprintf(stderr, "gen_shape: %u frames, %u groups: lanes %u interleave %u grid %u\n", count, groups, shape.lanes, il, grid); return 1; }
    std::vector<int64_t> last((size_t)count, -1), at((size_t)count * groups, -1);
    for (uint32_t L = 0; L < grid; ++L) {
      const vpcc::GenWork w = vpcc::gen_work_of(L, count, groups, il, shape.lanes);
      if (!w.any) continue;
      if (w.frame >= count || w.group >= groups) { std::fprintf(stderr, "gen_work_of: out of range\n"); return 1; }
      if (shape.lanes == 8u && w.frame % 8u != L % 8u) { std::fprintf(stderr, "gen_work_of: frame %u on XCD %u\n", w.frame, L % 8u); return 1; }
      if (at[(size_t)w.frame * groups + w.group] >= 0) { std::fprintf(stderr, "gen_work_of: (%u, %u) twice\n", w.frame, w.group); return 1; }
      at[(size_t)w.frame * groups + w.group] = L;
      if ((int64_t)w.group != last[w.frame] + 1) { std::fprintf(stderr, "gen_work_of: groups of frame %u out of order\n", w.frame); return 1; }
      last[w.frame] = w.group;
    }
    for (uint32_t i = 0; i < count; ++i)
      if (last[i] != (int64_t)groups - 1) { std::fprintf(stderr, "gen_work_of: frame %u misses groups (count %u, groups %u, interleave %u)\n", i, count, groups, il); return 1; }
  }
  for (long it = 0; it < iterations; ++it) {
    vpcc_frame_desc f{};
    const uint32_t R = 1u << (3 + below(3)), prec = 1u << below(3);
    f.occupancy_resolution = R;
    f.occupancy_precision = prec;
    f.width = R * (1 + below(40));
    f.height = R * (1 + below(40));
    if (below(50) == 0) f.width += below(R);                       // ragged canvas
    f.map_count = 1 + below(2);
    f.absolute_d1 = below(2);
    f.attribute_count = below(2);
    f.occupancy = vpcc_image_u8{dummy8, f.width / prec, f.height / prec, f.width / prec};
    for (uint32_t m = 0; m < f.map_count; ++m) {
      f.geometry[m] = vpcc_image_u16{dummy16, nullptr, nullptr, f.width, f.height, f.width, f.width / 2};
      f.attribute[m] = vpcc_image_u16{dummy16, dummy16, dummy16, f.width, f.height, f.width, f.width / 2};
    }
    std::vector<vpcc_patch> patches(below(40));
    const uint32_t bw = f.width / R, bh = f.height / R;
    const bool fitting = below(2) == 0;
    const bool wild = below(4) == 0, exotic = !wild && below(3) == 0;       // exotic: all nine orientations on patches that may fit
    for (vpcc_patch& p : patches) {
      p = vpcc_patch{};
      p.orientation = (uint8_t)below(wild ? 12 : exotic ? 9 : 2);
      p.size_u0 = 1 + below(wild ? 70000 : 6);
      p.size_v0 = 1 + below(wild ? 70000 : 6);
      p.u0 = below(wild ? 0xFFFFFFFFu : bw + 1);
      p.v0 = below(wild ? 0xFFFFFFFFu : bh + 1);
      if (!wild && fitting) {                                         // a square patch somewhere inside the canvas: fits in every orientation
        p.size_u0 = p.size_v0 = 1 + below(std::min(bw, bh) < 4 ? std::min(bw, bh) : 4);
        p.u0 = below(bw - p.size_u0 + 1);
        p.v0 = below(bh - p.size_v0 + 1);
      }
      p.u1 = (uint32_t)rnd(); p.v1 = (uint32_t)rnd(); p.d1 = (uint32_t)rnd();
      p.lod_x = wild ? (uint32_t)rnd() : 1; p.lod_y = wild ? below(3) : 1;
      p.normal_axis = (uint8_t)below(wild ? 5 : 3); p.tangent_axis = (uint8_t)below(3); p.bitangent_axis = (uint8_t)below(3);
      p.projection_mode = (uint8_t)below(wild ? 3 : 2);
      p.axis_of_additional_plane = wild ? (uint8_t)below(2) : 0;
    }
    f.patches = patches.empty() ? nullptr : patches.data();
    f.patch_count = (uint32_t)patches.size();
    vpcc::FrameShape shape;
    if (vpcc::validate_frame(&f, &shape) != VPCC_OK) { ++rejected; continue; }
    ++accepted;
    for (const vpcc_patch& p : patches) rotated += p.orientation >= 2 && p.orientation != 8;
    {   // which frames k_general_blocks may take (FrameShape::block_units): said again here, independently
      bool distinct = true;
      for (const vpcc_patch& p : patches)
        distinct = distinct && p.normal_axis != p.tangent_axis && p.normal_axis != p.bitangent_axis && p.tangent_axis != p.bitangent_axis;
      const bool want = R >= 16 && R <= 256 && (R & (R - 1)) == 0 && (prec & (prec - 1)) == 0 && f.width <= 65536 && distinct;   // (tight rows: pitch = width)
      if (shape.block_units != want) { std::fprintf(stderr, "block_units %d, expected %d (R %u, precision %u)\n", (int)shape.block_units, (int)want, R, prec); return 1; }
    }
    // The host writes O(patches): vb_base, one item template per patch, the affine patches.  The virtual blocks are DERIVED
    // from them on the device (patch_of_vblock + vblock_of in k_plan_vblocks; the templates' origin and size_u0 in
    // k_plan_tiles): both derivations, run here on the CPU, must give the reference's loop nest — patches ascending, v0
    // outer, u0 inner, patch_block_to_canvas_block (src/codec.rs:352-385, src/decoder.rs:827-838).
    std::vector<uint32_t> vb_base(patches.size() + 1);
    std::vector<vpcc::TileItem> templates(patches.size());
    std::vector<vpcc::DevPatch> dev(patches.size());
    vpcc::write_frame_records(f, vb_base.data(), shape.tile_eligible ? templates.data() : nullptr, dev.data());
    uint64_t expect_vb = 0;
    for (const vpcc_patch& p : patches) expect_vb += (uint64_t)p.size_u0 * p.size_v0;
    if (shape.n_vblocks != expect_vb || vb_base.back() != expect_vb || shape.n_patches != patches.size() || shape.bw != bw || shape.bh != bh) { std::fprintf(stderr, "shape\n"); return 1; }
    uint32_t vb = 0;
    for (uint32_t i = 0; i < patches.size(); ++i) {
      const vpcc_patch& p = patches[i];
      if (vb_base[i] != vb) { std::fprintf(stderr, "vb_base\n"); return 1; }
      const vpcc::Affine bl = vpcc::patch_affine(p, 1);
      for (uint32_t v0 = 0; v0 < p.size_v0; ++v0)
        for (uint32_t u0 = 0; u0 < p.size_u0; ++u0, ++vb) {
          const int64_t bx = bl.ax_u * u0 + bl.ax_v * v0 + bl.cx, by = bl.ay_u * u0 + bl.ay_v * v0 + bl.cy;
          const uint64_t cb = (uint64_t)(by * bw + bx);
          if (cb >= (uint64_t)bw * bh) { std::fprintf(stderr, "canvas block out of range\n"); return 1; }
          const uint32_t q = vpcc::patch_of_vblock(vb_base.data(), (uint32_t)patches.size(), vb);
          const vpcc::VBlock b = vpcc::vblock_of(dev[q], q, vb, bw, R);
          {
            // what the record says of the block's pixels: patch_to_canvas of its corners (src/decoder.rs:841-867), the tangent /
            // bitangent coordinates of its first pixel (src/decoder.rs:875-876), the patch's axes
            const vpcc::Affine px = vpcc::patch_affine(p, R);
            const int cux = (b.coef & 3) - 1, cvx = ((b.coef >> 2) & 3) - 1, cuy = ((b.coef >> 4) & 3) - 1, cvy = (b.coef >> 6) - 1;
            for (int c = 0; c < 4; ++c) {
              const int64_t pu = (c & 1) ? R - 1 : 0, pv = (c & 2) ? R - 1 : 0, u = (int64_t)u0 * R + pu, v = (int64_t)v0 * R + pv;
              const int64_t x = px.ax_u * u + px.ax_v * v + px.cx, y = px.ay_u * u + px.ay_v * v + px.cy;
              if (x != (int64_t)b.x0 + cux * pu + cvx * pv || y != (int64_t)b.y0 + cuy * pu + cvy * pv || x < 0 || y < 0 || x >= f.width || y >= f.height) {
                std::fprintf(stderr, "block record: pixel map\n"); return 1;
              }
            }
            if (b.t0 != u0 * R * p.lod_x + p.u1 || b.b0 != v0 * R * p.lod_y + p.v1 || b.lod_x != (uint16_t)p.lod_x || b.lod_y != (uint16_t)p.lod_y || b.d1 != p.d1 ||
                b.axes_mode != (p.normal_axis | (p.tangent_axis << 2) | (p.bitangent_axis << 4) | (p.projection_mode << 6))) {
              std::fprintf(stderr, "block record: coordinates\n"); return 1;
            }
          }
          if (q != i || b.patch != i || b.u0 != u0 || b.v0 != v0 || b.canvas_block != cb) { std::fprintf(stderr, "derived virtual block %u: patch %u (%u) block (%u, %u) canvas block %u (%llu)\n", vb, q, i, b.u0, b.v0, b.canvas_block, (unsigned long long)cb); return 1; }
          if (shape.tile_eligible) {                     // k_plan_tiles' route: the template's origin, size_u0 and Swap flag
            const vpcc::TileItem& t = templates[q];
            const uint32_t r = vb - vb_base[q], tv0 = r / t.patch, tu0 = r - tv0 * t.patch;
            const bool swap = (t.flags & vpcc::kTileSwap) != 0;
            const uint32_t tx = t.x0 + (swap ? tv0 : tu0), ty = t.y0 + (swap ? tu0 : tv0);
            if (tu0 != u0 || tv0 != v0 || (uint64_t)ty * bw + tx != cb) { std::fprintf(stderr, "template route\n"); return 1; }
            if (tx * 16u + 16u > f.width || ty * 16u + 16u > f.height) { std::fprintf(stderr, "tile outside the canvas\n"); return 1; }
          }
        }
    }
    if (shape.tile_eligible) {
      if (shape.tile_bound > expect_vb || shape.tile_bound > (uint64_t)bw * bh) { std::fprintf(stderr, "tile bound\n"); return 1; }
      items += shape.tile_bound;
    } else if (shape.tile_bound) { std::fprintf(stderr, "tile bound of an ineligible frame\n"); return 1; }
  }
  // The memory of a gof (place_planes / classify_extents / layout_gof): random batches of frames whose planes lie in random
  // arrangements — one container, scattered, partly outside page-locked memory.  Nothing overlaps: the regions of the arena,
  // the arrays of an output block, the planes of a planes block; every plane keeps the alignment the ingest gives it; what
  // the host writes lies in front of host_end; a stretch's device copy is congruent to its source modulo 256.
  long layouts = 0, by_extent = 0;
  for (long it = 0; it < iterations / 4 + 4; ++it) {
    const uint32_t n = 1 + below(below(6) == 0 ? 40 : 6);
    std::vector<vpcc_frame_desc> fr(n);
    std::vector<vpcc::FrameShape> sh(n);
    std::vector<std::vector<vpcc_patch>> pt(n);
    // a fake "host memory": addresses only
    const uintptr_t host0 = 0x100000000ull + 8 * below(64);
    uintptr_t at = host0;
    const bool container = below(2) == 0, attr = below(3) != 0;
    const uint32_t prec = 1u << below(3);
    auto put = [&](size_t bytes) { const uintptr_t p = at; at += bytes + (container ? below(4) * 8 : (size_t)below(1u << 20) * 8 + (below(4) == 0 ? (1u << 20) : 0)); return p; };
    for (uint32_t i = 0; i < n; ++i) {
      vpcc_frame_desc& f = fr[i];
      f = vpcc_frame_desc{};
      f.occupancy_resolution = 16; f.occupancy_precision = prec;
      f.width = 16 * (1 + below(12)); f.height = 16 * (1 + below(12));
      f.map_count = 1 + below(2); f.absolute_d1 = 1; f.attribute_count = attr ? 1 : 0;
      const bool padded = below(10) == 0;
      f.occupancy = vpcc_image_u8{(const uint8_t*)put((size_t)(f.width / prec) * (f.height / prec)), f.width / prec, f.height / prec, f.width / prec};
      for (uint32_t m = 0; m < f.map_count; ++m) {
        f.geometry[m] = vpcc_image_u16{(const uint16_t*)put((size_t)f.width * f.height * 2), nullptr, nullptr, f.width, f.height, f.width + (padded ? 8 : 0), f.width / 2};
        if (attr) {
          f.attribute[m] = vpcc_image_u16{(const uint16_t*)put((size_t)f.width * f.height * 2), nullptr, nullptr, f.width, f.height, f.width, f.width / 2};
          f.attribute[m].u = (const uint16_t*)put(vpcc::chroma_elems(f.attribute[m]) * 2);
          f.attribute[m].v = (const uint16_t*)put(vpcc::chroma_elems(f.attribute[m]) * 2);
        }
      }
      pt[i].resize(below(6));
      for (vpcc_patch& p : pt[i]) { p = vpcc_patch{}; p.size_u0 = 1 + below(f.width / 16); p.size_v0 = 1 + below(f.height / 16); p.u0 = below(f.width / 16 - p.size_u0 + 1); p.v0 = below(f.height / 16 - p.size_v0 + 1); p.lod_x = p.lod_y = 1; p.tangent_axis = 1; p.bitangent_axis = 2; }
      f.patches = pt[i].empty() ? nullptr : pt[i].data();
      f.patch_count = (uint32_t)pt[i].size();
      if (vpcc::validate_frame(&f, &sh[i]) != VPCC_OK) { std::fprintf(stderr, "layout: frame rejected\n"); return 1; }
    }
    // page-locked regions: all of the "memory", or with a hole, or in two chunks that meet somewhere
    const uintptr_t end = at + 4096;
    std::vector<std::pair<uintptr_t, uintptr_t>> regions;
    const uint32_t mode = below(4);
    if (mode == 0) regions.push_back({host0 - 64, end});
    else if (mode == 1) { const uintptr_t mid = host0 + below((uint32_t)std::min<uintptr_t>(end - host0, 0x7FFFFFFF)); regions.push_back({host0 - 64, mid}); regions.push_back({mid, end}); }
    else if (mode == 2) { const uintptr_t mid = host0 + below((uint32_t)std::min<uintptr_t>(end - host0, 0x7FFFFFFF)); regions.push_back({host0 - 64, mid}); regions.push_back({mid + 4096, end}); }
    vpcc::PinnedQuery pinned = [&](const char* lo, size_t bytes, std::vector<std::pair<const char*, size_t>>* pieces) {
      pieces->clear();
      uintptr_t cur = (uintptr_t)lo;
      const uintptr_t hi = cur + bytes;
      while (cur < hi) {
        const std::pair<uintptr_t, uintptr_t>* in = nullptr;
        for (const auto& r : regions) if (cur >= r.first && cur < r.second) { in = &r; break; }
        if (!in) { pieces->clear(); return false; }
        const uintptr_t e = std::min(hi, in->second);
        pieces->emplace_back((const char*)cur, (size_t)(e - cur));
        cur = e;
      }
      return true;
    };
    vpcc::GofLayout L;
    L.f.assign(n, vpcc::FrameOffsets{});
    std::vector<vpcc::IngestExtent> extents;
    const bool own = below(5) != 0, pull = below(2) == 0;
    bool ext = false;
    vpcc::GofLayoutRequest rq{};
    rq.frames = fr.data(); rq.shapes = sh.data(); rq.n_frames = n; rq.capacity = 1 + below(5000);
    rq.want_patch_index = below(2) == 0; rq.tile_records = below(4) != 0; rq.general_records = !rq.tile_records || below(3) == 0;
    rq.pull_ingest = own && pull;
    if (own) {
      ext = vpcc::classify_extents(fr.data(), n, pinned, &L, &extents);
      if (!ext) {
        for (const vpcc::FrameOffsets& o : L.f) if (o.planes.occ || o.planes.geo[0]) { std::fprintf(stderr, "layout: a refused classification left slots behind\n"); return 1; }
        for (int j = 0; j < 2 * vpcc::kGofParts; ++j) if (L.block[j].total) { std::fprintf(stderr, "layout: a refused classification left block space behind\n"); return 1; }
        vpcc::place_planes(rq, &L);
      }
    }
    by_extent += ext;
    vpcc::layout_gof(rq, &L);
    ++layouts;
    struct Span { size_t lo, hi; const char* what; };
    auto disjoint = [](std::vector<Span>& v, size_t limit) -> const char* {
      std::sort(v.begin(), v.end(), [](const Span& a, const Span& b) { return a.lo < b.lo; });
      for (size_t k = 0; k < v.size(); ++k) {
        if (v[k].hi > limit) return v[k].what;
        if (k && v[k].lo < v[k - 1].hi) return v[k].what;
      }
      return nullptr;
    };
    std::vector<Span> arena, blocks[2 * vpcc::kGofParts];
    arena.push_back({L.frames, L.frames + sizeof(vpcc::DevFrame) * n, "frames"});
    arena.push_back({L.counts, L.counts + 4 * (size_t)n, "counts"});
    arena.push_back({L.ctrl_begin, L.ctrl_begin + L.ctrl_bytes, "control words"});
    if (L.tickets != L.ctrl_begin || L.errors < L.tickets + 256 * (size_t)n || L.scan < L.errors + 4 * (size_t)n || (L.scan & 7)) { std::fprintf(stderr, "layout: control words\n"); return 1; }
    arena.push_back({L.b2p_begin, L.b2p_begin + 4 * L.b2p_words, "block_to_patch"});
    if (rq.pull_ingest) arena.push_back({L.ingest_pieces, L.ingest_pieces + sizeof(vpcc::IngestPiece) * std::max<size_t>(L.ingest_bound, 1), "ingest pieces"});
    size_t b2p_at = L.b2p_begin;
    for (uint32_t i = 0; i < n; ++i) {
      const vpcc::FrameOffsets& o = L.f[i];
      const size_t P = sh[i].n_patches, vbn = std::max<size_t>(sh[i].n_vblocks, 1);
      arena.push_back({o.vb_base, o.vb_base + 4 * (P + 1), "vb_base"});
      if (o.vb_base + 4 * (P + 1) > L.host_end) { std::fprintf(stderr, "layout: vb_base behind host_end\n"); return 1; }
      if (rq.tile_records) {
        arena.push_back({o.patch_items, o.patch_items + 32 * std::max<size_t>(P, 1), "item templates"});
        if (o.patch_items + 32 * P > L.host_end) { std::fprintf(stderr, "layout: templates behind host_end\n"); return 1; }
        arena.push_back({o.items, o.items + 32 * ((size_t)sh[i].tile_bound + 16), "items"});
      }
      if (rq.general_records) {
        arena.push_back({o.patches, o.patches + 64 * std::max<size_t>(P, 1), "patches"});
        if (o.patches + 64 * P > L.host_end) { std::fprintf(stderr, "layout: patches behind host_end\n"); return 1; }
        arena.push_back({o.vblocks, o.vblocks + sizeof(vpcc::VBlock) * vbn, "vblocks"});
        if (o.vblocks & 15) { std::fprintf(stderr, "layout: virtual blocks not aligned for 16-byte scalar loads\n"); return 1; }
        const size_t units = vpcc::general_units(fr[i].occupancy_resolution, sh[i].n_vblocks);      // status words: inside the control region
        if (o.vb_count < L.scan || o.vb_count + 8 * units > L.ctrl_begin + L.ctrl_bytes || (o.vb_count & 7)) { std::fprintf(stderr, "layout: unit status words\n"); return 1; }
        if (i + 1 < n && L.f[i + 1].vb_count != o.vb_count + 8 * units) { std::fprintf(stderr, "layout: unit status words overlap\n"); return 1; }
      }
      if (o.b2p != b2p_at) { std::fprintf(stderr, "layout: block_to_patch not contiguous\n"); return 1; }
      b2p_at += 4 * (size_t)sh[i].bw * sh[i].bh;
      const int part = vpcc::gof_part_of(i);
      const size_t cap = (size_t)rq.capacity;
      blocks[2 * part + 1].push_back({o.xyz, o.xyz + 6 * (cap + 4), "positions"});
      if (fr[i].attribute_count) blocks[2 * part + 1].push_back({o.rgb, o.rgb + 3 * (cap + 4), "colours"});
      if (rq.want_patch_index) blocks[2 * part + 1].push_back({o.pidx, o.pidx + 2 * (cap + 4), "partition"});
      if ((o.xyz | o.rgb | o.pidx) & 255) { std::fprintf(stderr, "layout: output alignment\n"); return 1; }
      if (own) {
        const vpcc_frame_desc& F = fr[i];
        auto plane = [&](size_t slot, const void* src, size_t bytes) -> bool {
          blocks[2 * part + 0].push_back({slot, slot + bytes, "plane"});
          if (ext) return ((slot ^ (uintptr_t)src) & 255u) == 0;                                  // keeps its alignment
          return (slot & 255u) == (rq.pull_ingest && ((uintptr_t)src & 7u) == 0 ? ((uintptr_t)src & 15u) : 0u) ||
                 (slot & 255u) == 0;                                                              // (a padded plane is copied row by row: no shift)
        };
        bool good = plane(o.planes.occ, F.occupancy.y, (size_t)F.occupancy.width * F.occupancy.height);
        for (uint32_t m = 0; m < F.map_count; ++m) {
          good = good && plane(o.planes.geo[m], F.geometry[m].y, (size_t)F.geometry[m].width * F.geometry[m].height * 2);
          if (F.attribute_count) {
            good = good && plane(o.planes.ay[m], F.attribute[m].y, (size_t)F.attribute[m].width * F.attribute[m].height * 2);
            good = good && plane(o.planes.au[m], F.attribute[m].u, vpcc::chroma_elems(F.attribute[m]) * 2);
            good = good && plane(o.planes.av[m], F.attribute[m].v, vpcc::chroma_elems(F.attribute[m]) * 2);
          }
        }
        if (!good) { std::fprintf(stderr, "layout: a plane lost its alignment (%s)\n", ext ? "by extent" : "placed"); return 1; }
      }
    }
    if (b2p_at != L.b2p_begin + 4 * L.b2p_words) { std::fprintf(stderr, "layout: block_to_patch size\n"); return 1; }
    if (const char* w = disjoint(arena, L.arena_bytes)) { std::fprintf(stderr, "layout: arena region '%s' overlaps or leaves the arena\n", w); return 1; }
    for (int j = 0; j < 2 * vpcc::kGofParts; ++j) {
      if (ext && !(j & 1)) {
        // planes of one stretch may share bytes only if the caller's planes did: compare against the host arrangement instead
        for (const vpcc::IngestExtent& e : extents) {
          if (((e.dev ^ (uintptr_t)e.lo) & 255u) != 0 || e.dev + e.bytes > L.block[2 * e.part].total) { std::fprintf(stderr, "layout: stretch\n"); return 1; }
          size_t sum = 0;
          for (const auto& pc : e.pieces) sum += pc.second;
          if (sum != e.bytes || e.pieces.empty() || e.pieces.front().first != e.lo) { std::fprintf(stderr, "layout: stretch pieces\n"); return 1; }
        }
        for (const Span& sp : blocks[j]) if (sp.hi > L.block[j].total) { std::fprintf(stderr, "layout: plane outside its block\n"); return 1; }
        continue;
      }
      if (const char* w = disjoint(blocks[j], L.block[j].total)) { std::fprintf(stderr, "layout: '%s' overlaps or leaves block %d\n", w, j); return 1; }
    }
    if (L.stage_counts < L.host_end || L.stage_bytes < L.stage_counts + 8 * (size_t)n) { std::fprintf(stderr, "layout: staging\n"); return 1; }
  }
  std::printf("layouts %ld, by extent %ld\n", layouts, by_extent);
  // Shares of the resident workgroups per frame (plan_tile_launch): every frame with tiles gets at least one
  // workgroup, the table's per-slot share equals the number of slots the frame really has (the kernel re-arms a
  // frame's ticket counter after exactly that many workgroups have left it), no slot names a frame outside the launch.
  long maps = 0, equal_split = 0;
  for (long it = 0; it < iterations; ++it) {
    const uint32_t count = 1 + below(below(8) == 0 ? 2100 : 70);
    std::vector<uint32_t> tiles(count);
    const uint32_t scale = 1 + below(9000);
    for (uint32_t& t : tiles) t = below(6) == 0 ? 0 : below(scale) + (below(3) == 0 ? below(40) : 0);
    const uint32_t resident = below(10) == 0 ? below(300) : 128, depth = 1 + below(6);
    vpcc::TileLaunchMap map;
    vpcc::plan_tile_launch(tiles.data(), count, resident, depth, map);
    if (!map.slots) { ++equal_split; continue; }
    ++maps;
    if (map.slots > vpcc::kTileMapSlots) { std::fprintf(stderr, "slots\n"); return 1; }
    for (uint32_t x = 0; x < 8; ++x) {
      std::vector<uint32_t> have((count + 7) / 8, 0);
      for (uint32_t s = 0; s < vpcc::kTileMapSlots; ++s) {
        const uint32_t v = map.frame_of_slot[x][s];
        if (v == 0xFF) continue;
        if (s >= map.slots || x + 8u * v >= count) { std::fprintf(stderr, "slot names a frame outside the launch\n"); return 1; }
        ++have[v];
      }
      for (uint32_t s = 0; s < map.slots; ++s) {
        const uint32_t v = map.frame_of_slot[x][s];
        if (v != 0xFF && map.wgs_of_slot[x][s] != have[v]) { std::fprintf(stderr, "share != slots of the frame\n"); return 1; }
      }
      for (uint32_t i = x; i < count; i += 8)
        if ((tiles[i] != 0) != (have[i / 8] != 0)) { std::fprintf(stderr, "frame %u tiles %u workgroups %u\n", i, tiles[i], have[i / 8]); return 1; }
    }
  }
  // The free space of a context's pool (vpcc::PoolExtents, behind vpcc_ctx_reserve): random runs of two kinds, random takes and
  // returns.  Blocks never overlap, lie inside one run of the kind asked for, the byte accounts add up, and once everything is
  // back every run is ONE free extent again (coalescing never crosses a run).
  long pool_ops = 0;
  for (long it = 0; it < iterations / 8 + 4; ++it) {
    vpcc::PoolExtents E;
    static char arena[1];                                     // addresses only: nothing is dereferenced
    char* base = arena;
    const size_t G = 1u << 20;
    size_t at = 0, total[2] = {0, 0};
    const uint32_t n_runs = 1 + below(6);
    for (uint32_t r = 0; r < n_runs; ++r) {
      const size_t len = (1 + below(48)) * G;
      const int kind = (int)below(2);
      E.add_run(base + at, len, kind);
      total[kind] += len;
      at += len + (below(3) == 0 ? (1 + below(4)) * G : 0);    // sometimes a gap: another slab
    }
    if (E.in_use[0] != 0 || E.in_use[1] != 0) { std::fprintf(stderr, "pool: fresh runs in use\n"); return 1; }
    struct Held { char* p; size_t bytes; uint32_t run; };
    std::vector<Held> held;
    for (int op = 0; op < 200; ++op, ++pool_ops) {
      if (held.empty() || below(3) != 0) {
        const int kind = (int)below(2);
        const size_t bytes = (1 + below(below(4) == 0 ? 40 * 1024 : 4096)) * 1024;
        char* p = nullptr; uint32_t run = 0;
        if (!E.take(kind, bytes, &p, &run)) continue;
        const vpcc::PoolExtents::Run& R = E.runs[run];
        if (R.kind != kind || p < R.ptr || p + bytes > R.ptr + R.bytes) { std::fprintf(stderr, "pool: block outside a run of its kind\n"); return 1; }
        for (const Held& h : held)
          if (p < h.p + h.bytes && h.p < p + bytes) { std::fprintf(stderr, "pool: blocks overlap\n"); return 1; }
        held.push_back(Held{p, bytes, run});
      } else {
        const size_t k = below((uint32_t)held.size());
        E.give_back(held[k].run, held[k].p, held[k].bytes);
        held.erase(held.begin() + k);
      }
      size_t used[2] = {0, 0}, freeb[2] = {0, 0};
      for (const Held& h : held) used[E.runs[h.run].kind] += h.bytes;
      for (int kd = 0; kd < 2; ++kd) {
        for (const auto& x : E.free_[kd]) freeb[kd] += x.bytes;
        if (used[kd] != E.in_use[kd] || used[kd] + freeb[kd] != total[kd]) { std::fprintf(stderr, "pool: byte accounts\n"); return 1; }
      }
    }
    for (const Held& h : held) E.give_back(h.run, h.p, h.bytes);
    size_t extents = 0;
    for (int kd = 0; kd < 2; ++kd) extents += E.free_[kd].size();
    if (extents != E.runs.size() || E.in_use[0] || E.in_use[1]) { std::fprintf(stderr, "pool: not whole again (%zu extents, %zu runs)\n", extents, E.runs.size()); return 1; }
  }
  std::printf("launch maps %ld, equal split %ld, pool operations %ld\n", maps, equal_split, pool_ops);
  std::printf("iterations %ld accepted %ld rejected %ld tile items %ld rotated or mirrored patches %ld\n", iterations, accepted, rejected, items, rotated);
  return accepted > 0 && rejected > 0 ? 0 : 1;
}
