// vpcc_gof_create.hip — vpcc_gof_create, in stages:
//   measure     validate every frame (the reference's asserts, up front) and learn its shape                 vpcc_host.cpp
//   place       where the planes the gof keeps will lie: whole stretches of the caller's page-locked memory
//               (classify_extents) or plane by plane (place_planes)                                           vpcc_host.cpp
//   choose      single-pass tile kernel or general sequence — from the planes as the kernels will see them
//   lay out     arena and output blocks (layout_gof)                                                          vpcc_host.cpp
//   allocate    arena, big blocks (the context's pool), page-locked staging buffer, events
//   describe    what the host writes, O(patches) per frame, straight into the staging buffer
//   enqueue     ONE copy of the staging buffer, the control words, the plane ingest, the virtual blocks (general sequence)
// Everything above "allocate" is pure host logic without a HIP call (fuzzed under ASan/UBSan: tests/fuzz_plan.cpp); what a
// launch needs of the planes' CONTENT — block_to_patch, the tile kernel's work list — is built by every vpcc_gof_reconstruct.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "vpcc_runtime.hpp"

using namespace vpcc;

namespace {

// Copies a (possibly strided) plane of the caller's — host memory, or device memory (VPCC_GOF_COPY_PLANES) — into a
// tight device plane.
int copy_plane(vpcc_ctx* ctx, void* dst, const void* src, size_t elem, uint32_t width, uint32_t height,
               uint32_t stride, hipStream_t s, hipMemcpyKind dir) {
  // A host plane may cross from one page-locked region into the next (an input page-locked chunk by chunk): a copy whose
  // source does is refused by the runtime (invalid argument), so it goes piece by piece.
  std::vector<std::pair<const char*, size_t>> pieces;
  const size_t span = height ? ((size_t)stride * (height - 1) + width) * elem : 0;
  const bool split = dir == hipMemcpyHostToDevice && span && pinned_pieces((const char*)src, span, &pieces) && pieces.size() > 1;
  if (split && stride == width) {
    for (const auto& pc : pieces)
      HIP_TRY(ctx, hipMemcpyAsync((char*)dst + (size_t)(pc.first - (const char*)src), pc.first, pc.second, dir, s));
  } else if (split) {                                       // padded rows across a boundary: row by row, each row piece by piece
    for (uint32_t r = 0; r < height; ++r) {
      const char* row = (const char*)src + (size_t)r * stride * elem;
      if (!pinned_pieces(row, (size_t)width * elem, &pieces)) pieces.assign(1, {row, (size_t)width * elem});
      for (const auto& pc : pieces)
        HIP_TRY(ctx, hipMemcpyAsync((char*)dst + (size_t)r * width * elem + (size_t)(pc.first - row), pc.first, pc.second, dir, s));
    }
  } else if (stride == width) {
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, (size_t)width * height * elem, dir, s));
  } else {
    HIP_TRY(ctx, hipMemcpy2DAsync(dst, (size_t)width * elem, src, (size_t)stride * elem, (size_t)width * elem, height, dir, s));
  }
  return VPCC_OK;
}

struct GofBuild {
  vpcc_ctx* ctx;
  vpcc_gof* g;
  const vpcc_frame_desc* frames;
  uint32_t n;
  vpcc_memory_kind kind;
  uint32_t flags;
  bool own_planes = false;     // the planes are copied into memory of the gof's own: host planes always, device planes on request
  bool pinned = false;         // host planes in page-locked memory (VPCC_GOF_ASYNC_UPLOAD says they are)
  bool by_extent = false;      // ... that arrive as whole stretches
  bool pull = false;           // ... or are pulled by kernel
  GofLayout L;
  std::vector<IngestExtent> extents;
  bool trace = false;
};

// 1. measure: validation and what it learns of every frame; the per-frame capacity
int measure(GofBuild& b, uint64_t capacity_points) {
  vpcc_gof* g = b.g;
  g->shapes.resize(b.n);
  uint64_t cap = capacity_points;
  for (uint32_t i = 0; i < b.n; ++i) {
    const int st = validate_frame(&b.frames[i], &g->shapes[i]);
    if (st) return fail(b.ctx, st, "frame " + std::to_string(i) + ": " + vpcc_status_string(st));
    g->max_vb = std::max(g->max_vb, g->shapes[i].n_vblocks);
    if (capacity_points == 0) cap = std::max<uint64_t>(cap, vpcc_frame_capacity_bound(&b.frames[i]));
  }
  if (cap == 0) cap = 1;
  if (cap > 0xFFFFFFF0ull) return fail(b.ctx, VPCC_ERR_INVALID_ARG, "capacity_points exceeds 32 bits");
  g->capacity = cap;
  return VPCC_OK;
}

// 2. place: where the planes the gof keeps will lie in its planes blocks
void place(GofBuild& b) {
  b.L.f.assign(b.n, FrameOffsets{});
  b.own_planes = b.kind == VPCC_MEM_HOST || (b.flags & VPCC_GOF_COPY_PLANES) != 0;
  b.pinned = b.kind == VPCC_MEM_HOST && (b.flags & VPCC_GOF_ASYNC_UPLOAD) != 0;
  if (!b.own_planes) return;
  if (b.pinned && !getenv("VPCC_NO_EXTENT_INGEST")) {
    b.by_extent = classify_extents(b.frames, b.n, pinned_pieces, &b.L, &b.extents);
    (void)hipGetLastError();                                  // (pinned_pieces asks the runtime about memory it may not know)
    if (b.trace) fprintf(stderr, "[vpcc] ingest: %s (%zu stretches)\n", b.by_extent ? "by extent" : "the planes do not lie in page-locked stretches", b.extents.size());
  }
  // Plane ingest by kernel (k_ingest_planes) for page-locked host planes that are no stretches (vpcc_host_pin maps them for
  // the device): one launch instead of ten hipMemcpyAsync per frame.  VPCC_NO_PULL_INGEST=1: the copy engines.
  b.pull = b.pinned && !b.by_extent && !getenv("VPCC_NO_PULL_INGEST");
  if (!b.by_extent) {
    GofLayoutRequest rq{};
    rq.frames = b.frames; rq.n_frames = b.n; rq.pull_ingest = b.pull;
    place_planes(rq, &b.L);
  }
}

// The descriptor of frame i, given where the gof's memory lies.  With null bases the plane pointers are their offsets: the
// blocks are 2-MB aligned, so what the tile kernel asks of the planes' alignment can be answered before anything is allocated.
void describe_frame(const GofBuild& b, uint32_t i, char* arena, char* const block[2 * kGofParts], uint32_t* d_counts,
                    uint32_t* d_tickets, uint32_t* d_errors, uint64_t* d_scan, DevFrame* out) {
  const vpcc_frame_desc& F = b.frames[i];
  const FrameShape& S = b.g->shapes[i];
  const FrameOffsets& o = b.L.f[i];
  DevFrame& D = *out;
  std::memset(&D, 0, sizeof(D));
  char* const planes = block[2 * gof_part_of(i) + 0];
  char* const outputs = block[2 * gof_part_of(i) + 1];
  const bool tile = !b.g->general;
  D.vb_base = (const uint32_t*)(arena + o.vb_base);
  D.patch_items = tile ? (const TileItem*)(arena + o.patch_items) : nullptr;
  D.patches = (b.g->general || !b.g->plan_in_lds) ? (const DevPatch*)(arena + o.patches) : nullptr;
  D.vblocks = D.patches ? (VBlock*)(arena + o.vblocks) : nullptr;
  D.block_to_patch = (uint32_t*)(arena + o.b2p);
  D.vb_count = b.g->general ? (uint32_t*)(arena + o.vb_count) : nullptr;      // (the units' 64-bit status words)
  D.vb_offset = nullptr;
  D.out_xyz = (vpcc_point3*)(outputs + o.xyz);
  D.out_rgb = F.attribute_count ? (vpcc_color3*)(outputs + o.rgb) : nullptr;
  D.out_patch = (b.flags & VPCC_GOF_WANT_PATCH_INDEX) ? (uint16_t*)(outputs + o.pidx) : nullptr;
  D.n_points = d_counts + i;
  D.tiles = tile ? (TileItem*)(arena + o.items) : nullptr;
  D.n_tiles = 0;                                              // written by the planning kernel of every launch
  D.scan_state = d_scan + o.scan_word;
  D.ticket = reinterpret_cast<uint64_t*>(d_tickets + 64 * (size_t)i);
  D.error_flag = d_errors + i;
  D.width = F.width; D.height = F.height; D.R = F.occupancy_resolution; D.prec = F.occupancy_precision;
  D.prec_shift = 0;
  while ((1u << D.prec_shift) < D.prec && D.prec_shift < 31) ++D.prec_shift;
  D.bw = S.bw; D.bh = S.bh;
  D.n_patches = S.n_patches;
  D.n_vblocks = S.n_vblocks;
  D.map_count = F.map_count; D.absolute_d1 = F.absolute_d1 ? 1u : 0u; D.has_attr = F.attribute_count ? 1u : 0u;
  D.capacity = (uint32_t)b.g->capacity;
  D.occ_w = F.occupancy.width; D.occ_h = F.occupancy.height;
  if (!b.own_planes) {
    D.occ = F.occupancy.y; D.occ_stride = F.occupancy.stride;
    for (uint32_t m = 0; m < F.map_count; ++m) {
      D.geo[m] = F.geometry[m].y; D.geo_stride[m] = F.geometry[m].stride;
      if (F.attribute_count) {
        D.attr_y[m] = F.attribute[m].y; D.attr_u[m] = F.attribute[m].u; D.attr_v[m] = F.attribute[m].v;
        D.attr_stride[m] = F.attribute[m].stride; D.attr_cstride[m] = F.attribute[m].cstride;
      }
    }
  } else {
    // tight copies; chroma keeps its source stride: the reference indexes it as a flat array (v/2)*(width/2)+(u/2),
    // src/decoder.rs:977, which for odd widths runs across rows
    D.occ = (const uint8_t*)(planes + o.planes.occ); D.occ_stride = F.occupancy.width;
    for (uint32_t m = 0; m < F.map_count; ++m) {
      D.geo[m] = (const uint16_t*)(planes + o.planes.geo[m]); D.geo_stride[m] = F.geometry[m].width;
      if (F.attribute_count) {
        D.attr_y[m] = (const uint16_t*)(planes + o.planes.ay[m]);
        D.attr_u[m] = (const uint16_t*)(planes + o.planes.au[m]);
        D.attr_v[m] = (const uint16_t*)(planes + o.planes.av[m]);
        D.attr_stride[m] = F.attribute[m].width; D.attr_cstride[m] = F.attribute[m].cstride;
      }
    }
  }
  // The tile kernel loads both layers and the attribute planes unconditionally (branch-free loads keep
  // several items in flight): absent planes alias present ones; their samples are never used.
  if (F.map_count == 1) {
    D.geo[1] = D.geo[0]; D.geo_stride[1] = D.geo_stride[0];
    D.attr_y[1] = D.attr_y[0]; D.attr_u[1] = D.attr_u[0]; D.attr_v[1] = D.attr_v[0];
    D.attr_stride[1] = D.attr_stride[0]; D.attr_cstride[1] = D.attr_cstride[0];
  }
  if (!F.attribute_count)
    for (uint32_t m = 0; m < 2; ++m) {
      D.attr_y[m] = D.attr_u[m] = D.attr_v[m] = D.geo[m];
      D.attr_stride[m] = D.attr_cstride[m] = D.geo_stride[m];
    }
}

// 3. choose: the single-pass tile kernel needs every frame eligible (R = 16, Default/Swap patches) and its vector loads aligned on
// the planes as it will see them; and its store loop addresses a frame's positions with 32-bit byte offsets (6 bytes per point):
// frames that may hold more than 715 827 880 points — canvases beyond 18 900 x 18 900 with two maps, unless the caller gives a
// smaller bound — take the general sequence, whose indices are 64 bits wide (tools/exp_max_canvas.py: a 32768 x 32768 frame of
// 811 M points, 4.5 GiB of positions, equals the oracle).
void choose(GofBuild& b) {
  vpcc_gof* g = b.g;
  constexpr uint64_t kTilePathMaxPoints = 0xFFFFFFF0ull / sizeof(vpcc_point3);
  bool tiles_ok = !(b.flags & VPCC_GOF_FORCE_GENERAL) && g->capacity <= kTilePathMaxPoints;
  bool lds = true;
  for (uint32_t i = 0; i < b.n && tiles_ok; ++i) {
    const FrameShape& S = g->shapes[i];
    tiles_ok = S.tile_eligible;
    lds = lds && (uint64_t)S.bw * S.bh <= kPlanLdsBlocks && S.n_patches <= kPlanLdsPatches;
  }
  g->general = false;                                         // (describe_frame: the fields of the tile path)
  g->plan_in_lds = true;
  char* const nowhere[2 * kGofParts] = {nullptr, nullptr, nullptr, nullptr};
  for (uint32_t i = 0; i < b.n && tiles_ok; ++i) {
    DevFrame D;
    describe_frame(b, i, nullptr, nowhere, nullptr, nullptr, nullptr, nullptr, &D);
    tiles_ok = tile_planes_aligned(D);
  }
  g->general = !tiles_ok;
  g->plan_in_lds = tiles_ok && lds && !getenv("VPCC_NO_LDS_PLANNING");
}

// 5. allocate: the arena (from the context's cache when one fits), the big blocks, the page-locked staging buffer, the events
int allocate(GofBuild& b) {
  vpcc_ctx* ctx = b.ctx;
  vpcc_gof* g = b.g;
  g->arena_bytes = b.L.arena_bytes;
  for (size_t k = 0; k < ctx->arena_cache.size(); ++k) {          // a cached arena that fits
    auto& a = ctx->arena_cache[k];
    if (a.second >= b.L.arena_bytes && a.second <= b.L.arena_bytes + b.L.arena_bytes / 4) {
      g->arena = a.first;
      g->arena_bytes = a.second;
      ctx->arena_cache.erase(ctx->arena_cache.begin() + k);
      break;
    }
  }
  if (!g->arena && device_malloc(ctx, &g->arena, g->arena_bytes) != hipSuccess) {
    g->arena = nullptr;
    return fail(ctx, VPCC_ERR_DEVICE, "no device memory for a gof's arena of " + std::to_string(g->arena_bytes >> 20) + " MB");
  }
  for (int j = 0; j < 2 * kGofParts; ++j)
    if (b.L.block[j].total) {
      const int st = acquire_block(ctx, j / 2, b.L.block[j].total + 256, &g->block[j]);
      if (st) return st;
      if ((uintptr_t)g->block[j].ptr & 255u) return fail(ctx, VPCC_ERR_STATE, "a gof block that is not 256-byte aligned");
    }
  // descriptor staging: a page-locked buffer of the context's (kept for the next gof); the point counts come back into it
  for (size_t k = 0; k < ctx->stage_cache.size(); ++k)
    if (ctx->stage_cache[k].second >= b.L.stage_bytes) {
      g->stage = ctx->stage_cache[k].first;
      g->stage_bytes = ctx->stage_cache[k].second;
      ctx->stage_cache.erase(ctx->stage_cache.begin() + k);
      break;
    }
  if (!g->stage) {
    g->stage_bytes = b.L.stage_bytes + b.L.stage_bytes / 4;
    HIP_TRY(ctx, hipHostMalloc(&g->stage, g->stage_bytes, hipHostMallocDefault));
  }
  g->h_counts = (uint32_t*)((char*)g->stage + b.L.stage_counts);
  char* const base = (char*)g->arena;
  g->d_frames = (DevFrame*)(base + b.L.frames);
  g->d_counts = (uint32_t*)(base + b.L.counts);
  g->d_b2p = (uint32_t*)(base + b.L.b2p_begin);
  g->b2p_words = b.L.b2p_words;
  g->b2p_off.resize(b.n + 1);
  for (uint32_t i = 0; i < b.n; ++i) g->b2p_off[i] = (b.L.f[i].b2p - b.L.b2p_begin) / sizeof(uint32_t);
  g->b2p_off[b.n] = b.L.b2p_words;
  g->d_tickets = (uint32_t*)(base + b.L.tickets);
  g->d_errors = (uint32_t*)(base + b.L.errors);
  g->d_scan = (uint64_t*)(base + b.L.scan);
  g->ctrl_bytes = b.L.ctrl_bytes;
  HIP_TRY(ctx, hipEventCreateWithFlags(&g->upload_done, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&g->results_ready, hipEventDisableTiming));
  return VPCC_OK;
}

// 6. describe: what the host writes — frame descriptors, vb_base, item templates, affine patches — straight into the staging buffer
void describe(GofBuild& b) {
  vpcc_gof* g = b.g;
  char* const stage = (char*)g->stage;
  char* blocks[2 * kGofParts];
  for (int j = 0; j < 2 * kGofParts; ++j) blocks[j] = (char*)g->block[j].ptr;
  g->h_frames.resize(b.n);
  for (uint32_t i = 0; i < b.n; ++i) {
    const FrameOffsets& o = b.L.f[i];
    describe_frame(b, i, (char*)g->arena, blocks, g->d_counts, g->d_tickets, g->d_errors, g->d_scan, &g->h_frames[i]);
    const DevFrame& D = g->h_frames[i];
    write_frame_records(b.frames[i], (uint32_t*)(stage + o.vb_base), D.patch_items ? (TileItem*)(stage + o.patch_items) : nullptr,
                        D.patches ? (DevPatch*)(stage + o.patches) : nullptr);
  }
  std::memcpy(stage + b.L.frames, g->h_frames.data(), sizeof(DevFrame) * b.n);
}

// 7. enqueue: everything of this gof goes onto the copy stream, in the order it is needed — nothing waits for another stream
// (HIP maps streams onto a few hardware queues: the small copies of a set-up stream sat behind the 40 ms of the previous
// unit's planes although they were enqueued long before, and the copy engine then idled 10 ms between two units' planes).
int enqueue(GofBuild& b) {
  vpcc_ctx* ctx = b.ctx;
  vpcc_gof* g = b.g;
  hipStream_t s = ctx->copy_stream;
  char* const base = (char*)g->arena;
  HIP_TRY(ctx, hipMemcpyAsync(base, g->stage, b.L.host_end, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemsetAsync(g->d_counts, 0, sizeof(uint32_t) * b.n, s));
  HIP_TRY(ctx, hipMemsetAsync(base + b.L.ctrl_begin, 0, g->ctrl_bytes, s));
  // the virtual blocks of the general sequence (and of tile frames beyond the planning kernel's LDS): a function of the patch
  // tables alone, written once
  if (g->general || !g->plan_in_lds) {
    launch_plan_vblocks(g->d_frames, 0, b.n, g->max_vb, s);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (!b.own_planes) return VPCC_OK;
  const hipMemcpyKind dir = b.kind == VPCC_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  g->ingest_extents = 0;
  for (const IngestExtent& e : b.extents)
    for (const auto& pc : e.pieces) {                          // (one piece, unless the stretch crosses from one page-locked region into the next)
      HIP_TRY(ctx, hipMemcpyAsync((char*)g->block[2 * e.part + 0].ptr + e.dev + (size_t)(pc.first - e.lo), pc.first, pc.second,
                                  hipMemcpyHostToDevice, s));
      ++g->ingest_extents;
    }
  if (b.by_extent) return VPCC_OK;
  // One plane of the caller's into its place in the gof's planes block.  Tight planes in page-locked memory whose address is a
  // multiple of eight are left to the ingest kernel (place_planes moved the place by src mod 16, so that 16-byte pieces line
  // up on both sides); everything else goes through the copy engines.
  int st = VPCC_OK;
  auto ingest_plane = [&](const void* dst_c, const void* src, size_t elem, uint32_t width, uint32_t height, uint32_t stride) {
    char* dst = (char*)const_cast<void*>(dst_c);
    void* dev_src = nullptr;
    if (b.pull && stride == width && ((uintptr_t)src & 7u) == 0 &&
        hipHostGetDevicePointer(&dev_src, const_cast<void*>(src), 0) == hipSuccess && dev_src) {
      const size_t bytes = (size_t)width * height * elem;
      // (the first piece ends on a 16-byte boundary of the source: only a plane's first and last piece have bytes in front
      // of / behind their aligned body — single-byte reads over PCIe)
      for (size_t at = 0; at < bytes;) {
        const size_t len = std::min<size_t>(kIngestPieceBytes - (at ? 0u : ((uintptr_t)dev_src & 15u)), bytes - at);
        g->ingest.push_back(IngestPiece{(const char*)dev_src + at, dst + at, (uint32_t)len, 0u});
        at += len;
      }
      return;
    }
    (void)hipGetLastError();
    if (!st) st = copy_plane(ctx, dst, src, elem, width, height, stride, s, dir);
  };
  for (uint32_t i = 0; i < b.n; ++i) {
    const vpcc_frame_desc& F = b.frames[i];
    const DevFrame& D = g->h_frames[i];
    ingest_plane(D.occ, F.occupancy.y, 1, F.occupancy.width, F.occupancy.height, F.occupancy.stride);
    for (uint32_t m = 0; m < F.map_count; ++m) {
      const vpcc_image_u16& G = F.geometry[m];
      ingest_plane(D.geo[m], G.y, 2, G.width, G.height, G.stride);
      if (F.attribute_count) {
        const vpcc_image_u16& A = F.attribute[m];
        ingest_plane(D.attr_y[m], A.y, 2, A.width, A.height, A.stride);
        const size_t ce = chroma_elems(A);
        ingest_plane(D.attr_u[m], A.u, 2, (uint32_t)ce, 1, (uint32_t)ce);
        ingest_plane(D.attr_v[m], A.v, 2, (uint32_t)ce, 1, (uint32_t)ce);
      }
    }
  }
  if (st) return st;
  if (!g->ingest.empty()) {
    if (g->ingest.size() > b.L.ingest_bound) return fail(ctx, VPCC_ERR_STATE, "ingest piece list overflow");
    IngestPiece* d_pieces = (IngestPiece*)(base + b.L.ingest_pieces);
    HIP_TRY(ctx, hipMemcpyAsync(d_pieces, g->ingest.data(), sizeof(IngestPiece) * g->ingest.size(), hipMemcpyHostToDevice, s));
    launch_ingest_planes(d_pieces, (uint32_t)g->ingest.size(), s);
    HIP_TRY(ctx, hipGetLastError());
  }
  return VPCC_OK;
}

int gof_create_impl(vpcc_ctx* ctx, const vpcc_frame_desc* frames, uint32_t n_frames, vpcc_memory_kind kind,
                    uint64_t capacity_points, uint32_t gof_flags, vpcc_gof* g) {
  g->ctx = ctx;
  g->n_frames = n_frames;
  g->flags = gof_flags;
  GofBuild b{ctx, g, frames, n_frames, kind, gof_flags};
  b.trace = getenv("VPCC_RUNTIME_TRACE") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
  const auto t0 = now();
  int st = measure(b, capacity_points);
  if (st) return st;
  const auto t1 = now();
  place(b);
  choose(b);
  GofLayoutRequest rq{};
  rq.frames = frames; rq.shapes = g->shapes.data(); rq.n_frames = n_frames; rq.capacity = g->capacity;
  rq.want_patch_index = (gof_flags & VPCC_GOF_WANT_PATCH_INDEX) != 0;
  rq.tile_records = !g->general;
  rq.general_records = g->general || !g->plan_in_lds;
  rq.pull_ingest = b.pull;
  layout_gof(rq, &b.L);
  const auto t2 = now();
  st = allocate(b);
  if (st) return st;
  const auto t3 = now();
  describe(b);
  const auto t4 = now();
  st = enqueue(b);
  if (st) return st;
  HIP_TRY(ctx, hipEventRecord(g->upload_done, ctx->copy_stream));
  if (b.trace) {
    const auto t5 = now();
    fprintf(stderr, "[vpcc] gof of %u frames (%s): validation %.3f ms, placement + layout %.3f, allocations %.3f, descriptors %.3f, enqueue of %s %.3f "
                    "(%u extents, %zu pieces by kernel)   (@%.1f - %.1f)\n", n_frames, g->general ? "general sequence" : g->plan_in_lds ? "tile kernel" : "tile kernel, planning in global memory",
            ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), kind == VPCC_MEM_HOST ? "the upload" : b.own_planes ? "the copies" : "the descriptors", ms(t4, t5),
            g->ingest_extents, g->ingest.size(), ms(std::chrono::steady_clock::time_point{}, t0), ms(std::chrono::steady_clock::time_point{}, t5));
    for (int k = 0; k < 2 * kGofParts; ++k)
      if (g->block[k].ptr)
        fprintf(stderr, "[vpcc]   %s block of part %d: %p + %.2f GB%s\n", (k & 1) ? "output" : "planes", k / 2, g->block[k].ptr,
                g->block[k].bytes / 1073741824.0, g->block[k].pooled ? " (pool)" : "");
  }
  // the caller's planes must outlive the copies, so creation is synchronous unless the caller asked for overlapping ingest;
  // a gof that only borrows planes has nothing to wait for (its descriptors travel from the gof's own staging buffer)
  if (b.own_planes && !((gof_flags & VPCC_GOF_ASYNC_UPLOAD) && kind == VPCC_MEM_HOST)) HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
  return VPCC_OK;
}

}  // namespace

extern "C" int vpcc_gof_create(vpcc_ctx* ctx, const vpcc_frame_desc* frames, uint32_t n_frames,
                               vpcc_memory_kind planes, uint64_t capacity_points, uint32_t gof_flags,
                               vpcc_gof** out) {
  if (!ctx || !frames || !out || n_frames == 0) return VPCC_ERR_INVALID_ARG;
  if (planes != VPCC_MEM_HOST && planes != VPCC_MEM_DEVICE) return fail(ctx, VPCC_ERR_INVALID_ARG, "memory kind");
  *out = nullptr;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  vpcc_gof* g = new vpcc_gof();
  const int st = gof_create_impl(ctx, frames, n_frames, planes, capacity_points, gof_flags, g);
  if (st) {
    vpcc_gof_destroy(g);
    return st;
  }
  *out = g;
  return VPCC_OK;
}
