// vpcc_gof.hip — a gof after its creation: launches, point counts, downloads, kernel timings.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "vpcc_runtime.hpp"

using namespace vpcc;

// --------------------------------------------------------------------- gof
extern "C" void vpcc_gof_destroy(vpcc_gof* gof) {
  if (!gof) return;
  (void)hipSetDevice(gof->ctx->device);
  // Everything enqueued on THIS gof has to be over — its ingest and planning (upload_done), the last kernel launched on it
  // (results_ready), its asynchronous downloads — and nothing else: the context's streams carry the next units' ingest, and
  // a lane that waited here for the copy stream to drain (round 3 and the first half of round 4) left the link idle until it
  // had posted the unit after those (a 128-frame unit every 55 ms instead of every 41).
  if (gof->upload_done) (void)hipEventSynchronize(gof->upload_done);
  if (gof->launched && gof->results_ready) (void)hipEventSynchronize(gof->results_ready);
  for (hipEvent_t e : gof->download_done) if (e) (void)hipEventSynchronize(e);
  for (auto& l : gof->history)
    for (auto& t : l.k) {
      (void)hipEventDestroy(t.start);
      (void)hipEventDestroy(t.stop);
    }
  for (hipEvent_t e : gof->download_done) if (e) (void)hipEventDestroy(e);
  if (gof->upload_done) (void)hipEventDestroy(gof->upload_done);
  if (gof->results_ready) (void)hipEventDestroy(gof->results_ready);
  if (gof->smooth_grid) (void)hipFree(gof->smooth_grid);
  if (gof->smooth_keys) (void)hipFree(gof->smooth_keys);
  if (gof->smooth_moved) (void)hipFree(gof->smooth_moved);
  if (gof->arena) {                                   // all work on it is complete (streams synchronised above)
    auto& cache = gof->ctx->arena_cache;
    if (cache.size() < 4) cache.emplace_back(gof->arena, gof->arena_bytes);
    else (void)hipFree(gof->arena);
  }
  if (gof->stage) {
    auto& cache = gof->ctx->stage_cache;
    if (cache.size() < 4) cache.emplace_back(gof->stage, gof->stage_bytes);
    else (void)hipHostFree(gof->stage);
  }
  for (vpcc_ctx::Block& B : gof->block) release_block(gof->ctx, B);
  // (hipHostFree waits for the whole device — in a lane of the streaming Decoder: for the next units' ingest, 70 ms — so a gof
  // keeps its counts in its staging buffer, and the buffer goes back to the context's cache)
  delete gof;
}

Timer::Timer(vpcc_gof* g_, hipStream_t s_, bool new_launch) : g(g_), s(s_), on((g_->flags & VPCC_GOF_PROFILE) != 0) {
  if (on && new_launch) g->launch_is_timed = (g->reconstructs++ % g->profile_every) == 0;
  on = on && g->launch_is_timed;                         // a following vpcc_gof_smooth shares the decision
  if (!on) return;
  if (g->history.empty()) g->history.resize(kProfileRing);
  if (new_launch || g->launches_profiled == 0) {
    g->launches_profiled++;
    g->history[(g->launches_profiled - 1) % kProfileRing].n = 0;
  }
  slot = &g->history[(g->launches_profiled - 1) % kProfileRing];
}
void Timer::begin(const char* name) {
  if (!on) return;
  if (slot->n == slot->k.size()) {
    KernelTiming t{};
    (void)hipEventCreate(&t.start);
    (void)hipEventCreate(&t.stop);
    slot->k.push_back(t);
  }
  slot->k[slot->n].name = name;
  (void)hipEventRecord(slot->k[slot->n].start, s);
}
void Timer::end() {
  if (!on) return;
  (void)hipEventRecord(slot->k[slot->n].stop, s);
  slot->n++;
}

extern "C" int vpcc_gof_reconstruct(vpcc_gof* g, uint32_t first, uint32_t count, void* hip_stream) {
  if (!g) return VPCC_ERR_INVALID_ARG;
  vpcc_ctx* ctx = g->ctx;
  if (count == 0 || first >= g->n_frames || count > g->n_frames - first)
    return fail(ctx, VPCC_ERR_INVALID_ARG, "frame range");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  HIP_TRY(ctx, hipStreamWaitEvent(s, g->upload_done, 0));        // the planes' H2D copies (copy stream) come first
  // Launches on one gof are ordered: its ticket counters, look-back words and output arrays are reused by
  // every launch.  A launch on another stream than the previous one waits for that one's kernels.
  if (g->launched && g->last_stream != s) HIP_TRY(ctx, hipStreamWaitEvent(s, g->results_ready, 0));
  g->last_stream = s;
  g->counts_valid = false;
  Timer T(g, s, true);

  if (!g->general) {
    // Single-pass tile kernel.  First its work lists — generate_block_to_patch_from_occupancy_map_video, src/codec.rs:205-250 —
    // from the occupancy planes as they are NOW, on the launch stream: a gof that borrows the caller's planes may be launched
    // again after the caller has decoded new frames into them.
    uint32_t max_groups = 0, max_blocks = 0, max_patches = 0, max_vb = 0;
    for (uint32_t i = first; i < first + count; ++i) {
      const FrameShape& S = g->shapes[i];
      max_groups = std::max(max_groups, (uint32_t)((S.tile_bound + kTileItemsPerGroup - 1) / kTileItemsPerGroup));
      max_blocks = std::max(max_blocks, S.bw * S.bh);
      max_patches = std::max(max_patches, S.n_patches);
      max_vb = std::max(max_vb, S.n_vblocks);
    }
    if (g->plan_in_lds) {
      T.begin("k_plan_tiles");
      launch_plan_tiles(g->d_frames, first, count, plan_tiles_lds_launch_bytes(max_blocks, max_patches), false, s);
      T.end();
    } else {
      T.begin("k_plan_cover+items");
      launch_plan_tiles_global(g->d_frames, first, count, max_vb, g->d_b2p + g->b2p_off[first], g->b2p_off[first + count] - g->b2p_off[first], s);
      T.end();
    }
    // Nothing to clear: look-back words and ticket counters carry the launch generation (a counter of an earlier
    // launch is reset by the first workgroup that draws from it), and a frame's point count is rewritten by its last group
    // (a frame without tiles keeps the zero written at creation).
    if (!g->tile_map_valid || g->tile_map_first != first || g->tile_map_count != count) {
      std::vector<uint32_t> tiles(count);
      for (uint32_t i = 0; i < count; ++i) tiles[i] = g->shapes[first + i].tile_bound;   // (the exact counts are on the device: the planning kernel's)
      plan_tile_launch(tiles.data(), count, ctx->resident_tile_wgs_per_xcd, 3, g->tile_map);
      g->tile_map_first = first; g->tile_map_count = count; g->tile_map_valid = true;
    }
    g->generation = (g->generation % 0x3FFFFFFFu) + 1u;
    T.begin("k_recon_tiles");
    launch_tiles(g->d_frames, first, count, max_groups, g->generation, g->tile_map, ctx->resident_tile_wgs_per_xcd, s);
    T.end();
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(g->results_ready, s));
    g->launched = true;
    return VPCC_OK;
  }

  // general sequence: block ownership, then ONE pass over the virtual blocks (count, look-back, emit)
  const size_t b2p_len = g->b2p_off[first + count] - g->b2p_off[first];
  if (b2p_len) HIP_TRY(ctx, hipMemsetAsync(g->d_b2p + g->b2p_off[first], 0, b2p_len * sizeof(uint32_t), s));
  uint32_t max_vb = 0, max_units = 0, max_samples = 0;
  bool block_units = getenv("VPCC_GENERAL_ANY_FRAME") == nullptr;       // (tests: k_general on frames k_general_blocks would take)
  for (uint32_t i = first; i < first + count; ++i) {
    const DevFrame& D = g->h_frames[i];
    block_units = block_units && g->shapes[i].block_units;
    max_vb = std::max(max_vb, D.n_vblocks);
    max_units = std::max(max_units, general_units(D.R, D.n_vblocks));
    const uint32_t side = (D.R + D.prec - 1u) / D.prec + 1u; // samples under R pixels that start anywhere: at most ceil(R / precision) + 1
    max_samples = std::max(max_samples, side * side);
  }
  g->generation = (g->generation % 0x3FFFFFFFu) + 1u;        // tags the units' status words: nothing is cleared between launches
  T.begin("k_block_owner");
  launch_block_owner(g->d_frames, first, count, max_vb, max_samples, s);
  T.end();
  T.begin(block_units ? "k_general_blocks" : "k_general");
  launch_general(g->d_frames, first, count, max_units, g->generation, block_units, s);
  T.end();
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(g->results_ready, s));
  g->launched = true;
  return VPCC_OK;
}

extern "C" int vpcc_gof_sync(vpcc_gof* g) {
  if (!g) return VPCC_ERR_INVALID_ARG;
  HIP_TRY(g->ctx, hipSetDevice(g->ctx->device));
  if (g->launched) HIP_TRY(g->ctx, hipEventSynchronize(g->results_ready));
  return VPCC_OK;
}

namespace {
int fetch_counts(vpcc_gof* g) {
  if (!g->launched) return fail(g->ctx, VPCC_ERR_STATE, "no reconstruct issued");
  if (g->counts_valid) return VPCC_OK;
  hipStream_t s = g->ctx->d2h_stream;
  HIP_TRY(g->ctx, hipStreamWaitEvent(s, g->results_ready, 0));
  // The counts are PUSHED into their page-locked buffer by a kernel: as copies they go through a copy engine's queue, and in the
  // streaming Decoder that queue holds the 40 ms of the next unit's planes — every unit's point counts came back a unit late,
  // and with them its downloads and the creation of the unit after next (rocprofv3 --hip-runtime-trace: the lane sat in this
  // synchronisation from the end of one unit's ingest to the end of the next one's).
  void* dev_counts = nullptr;
  if (!getenv("VPCC_NO_PUSH_DOWNLOAD") && hipHostGetDevicePointer(&dev_counts, g->h_counts, 0) == hipSuccess && dev_counts) {
    IngestPiece pieces[3] = {};
    pieces[0] = IngestPiece{g->d_counts, dev_counts, (uint32_t)(sizeof(uint32_t) * g->n_frames), 0u};
    pieces[1] = IngestPiece{g->d_errors, (char*)dev_counts + sizeof(uint32_t) * g->n_frames, (uint32_t)(sizeof(uint32_t) * g->n_frames), 0u};
    launch_push_results(pieces, s);
    HIP_TRY(g->ctx, hipGetLastError());
  } else {
    (void)hipGetLastError();
    HIP_TRY(g->ctx, hipMemcpyAsync(g->h_counts, g->d_counts, sizeof(uint32_t) * g->n_frames, hipMemcpyDeviceToHost, s));
    HIP_TRY(g->ctx, hipMemcpyAsync(g->h_counts + g->n_frames, g->d_errors, sizeof(uint32_t) * g->n_frames,
                                   hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(g->ctx, hipStreamSynchronize(s));
  for (uint32_t i = 0; i < g->n_frames; ++i)
    if (g->h_counts[g->n_frames + i] & kErrorSpinLimit)
      return fail(g->ctx, VPCC_ERR_DEVICE, "look-back spin limit reached in frame " + std::to_string(i));
  for (uint32_t i = 0; i < g->n_frames; ++i)
    if (g->h_counts[g->n_frames + i] & kErrorSmoothCellOverflow)
      return fail(g->ctx, VPCC_ERR_UNSUPPORTED, "smoothing: more than 65 537 points of frame " + std::to_string(i) +
                  " in one grid cell (the cells' 32-bit sums may have overflowed; the frame's smoothed output is not the specification's)");
  g->counts_valid = true;
  return VPCC_OK;
}
}  // namespace

extern "C" int vpcc_gof_point_counts(vpcc_gof* g, uint32_t* counts_out) {
  if (!g || !counts_out) return VPCC_ERR_INVALID_ARG;
  HIP_TRY(g->ctx, hipSetDevice(g->ctx->device));
  const int st = fetch_counts(g);
  if (st) return st;
  std::memcpy(counts_out, g->h_counts, sizeof(uint32_t) * g->n_frames);
  return VPCC_OK;
}

extern "C" int vpcc_gof_block_to_patch(vpcc_gof* g, uint32_t frame, uint32_t* block_to_patch_out, uint32_t* work_items_out) {
  if (!g || frame >= g->n_frames) return VPCC_ERR_INVALID_ARG;
  vpcc_ctx* ctx = g->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->d2h_stream;
  if (!g->launched) return fail(ctx, VPCC_ERR_STATE, "no reconstruct issued: block_to_patch is built by every launch, from the planes as they are then");
  HIP_TRY(ctx, hipStreamWaitEvent(s, g->results_ready, 0));
  const size_t n = (size_t)g->shapes[frame].bw * g->shapes[frame].bh;
  if (!g->general && g->plan_in_lds) {
    // a launch keeps block_to_patch in the planning kernel's LDS (the tile kernel works from the items): planned once more
    // for this frame, from the planes as they are now, with the map written out (the items it rewrites are the same)
    launch_plan_tiles(g->d_frames, frame, 1, plan_tiles_lds_launch_bytes((uint32_t)n, g->shapes[frame].n_patches), true, s);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (block_to_patch_out && n)
    HIP_TRY(ctx, hipMemcpyAsync(block_to_patch_out, g->h_frames[frame].block_to_patch, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  DevFrame d{};
  if (work_items_out) HIP_TRY(ctx, hipMemcpyAsync(&d, g->d_frames + frame, sizeof d, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  if (work_items_out) *work_items_out = g->general ? 0u : d.n_tiles;
  // (the gof's next launch is ordered behind this: it shares the items the planning kernel rewrote)
  HIP_TRY(ctx, hipEventRecord(g->results_ready, s));
  g->last_stream = s;
  return VPCC_OK;
}

extern "C" int vpcc_gof_device_outputs(vpcc_gof* g, uint32_t frame, void** d_xyz, void** d_rgb, void** d_patch_index,
                                       void** d_count) {
  if (!g || frame >= g->n_frames) return VPCC_ERR_INVALID_ARG;
  const DevFrame& D = g->h_frames[frame];
  if (d_xyz) *d_xyz = D.out_xyz;
  if (d_rgb) *d_rgb = D.out_rgb;
  if (d_patch_index) *d_patch_index = D.out_patch;
  if (d_count) *d_count = D.n_points;
  return VPCC_OK;
}

extern "C" int vpcc_gof_frame_status(vpcc_gof* g, uint32_t frame) {
  if (!g || frame >= g->n_frames) return VPCC_ERR_INVALID_ARG;
  HIP_TRY(g->ctx, hipSetDevice(g->ctx->device));
  const int st = fetch_counts(g);
  if (st) return st;
  return g->h_counts[frame] > g->capacity ? VPCC_ERR_CAPACITY : VPCC_OK;
}

namespace {
// Enqueues the copies of one frame's result on the download stream (behind the gof's latest kernels).
int enqueue_download(vpcc_gof* g, uint32_t frame, vpcc_point3* xyz_out, vpcc_color3* rgb_out, uint16_t* patch_index_out,
                     size_t capacity, size_t* n_points, bool push_allowed) {
  vpcc_ctx* ctx = g->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int st = fetch_counts(g);
  if (st) return st;
  const size_t n = g->h_counts[frame];
  *n_points = n;
  if (n > g->capacity || n > capacity) return fail(ctx, VPCC_ERR_CAPACITY, "frame produced more points than capacity");
  const DevFrame& D = g->h_frames[frame];
  hipStream_t s = ctx->d2h_stream;
  HIP_TRY(ctx, hipStreamWaitEvent(s, g->results_ready, 0));   // the latest kernels on this gof (incl. smoothing)
  if (patch_index_out && !D.out_patch) return fail(ctx, VPCC_ERR_STATE, "gof was created without VPCC_GOF_WANT_PATCH_INDEX");
  if (n) {
    // Page-locked destinations (all of them: one launch) are written by a kernel — beside the ingest kernel of the next
    // unit a device-to-host hipMemcpyAsync crawls (k_push_results); anything else goes through the copy engine.
    struct Arr { void* dst; const void* src; size_t bytes; } arr[3] = {
        {xyz_out, D.out_xyz, n * sizeof(vpcc_point3)},
        {D.out_rgb ? (void*)rgb_out : nullptr, D.out_rgb, n * sizeof(vpcc_color3)},
        {patch_index_out, D.out_patch, n * sizeof(uint16_t)}};
    IngestPiece pieces[3] = {};
    bool push = push_allowed && !getenv("VPCC_NO_PUSH_DOWNLOAD");
    for (int a = 0; a < 3 && push; ++a) {
      if (!arr[a].dst) continue;
      void* dev_dst = nullptr;
      if (hipHostGetDevicePointer(&dev_dst, arr[a].dst, 0) != hipSuccess || !dev_dst) { (void)hipGetLastError(); push = false; break; }
      pieces[a] = IngestPiece{arr[a].src, dev_dst, (uint32_t)arr[a].bytes, (uint32_t)(arr[a].bytes >> 32)};
    }
    if (push) {
      launch_push_results(pieces, s);
      HIP_TRY(ctx, hipGetLastError());
    } else {
      for (int a = 0; a < 3; ++a)
        if (arr[a].dst) HIP_TRY(ctx, hipMemcpyAsync(arr[a].dst, arr[a].src, arr[a].bytes, hipMemcpyDeviceToHost, s));
    }
  }
  return VPCC_OK;
}
}  // namespace

extern "C" int vpcc_gof_download(vpcc_gof* g, uint32_t frame, vpcc_point3* xyz_out, vpcc_color3* rgb_out,
                                 uint16_t* patch_index_out, size_t capacity, size_t* n_points) {
  if (!g || frame >= g->n_frames || !n_points) return VPCC_ERR_INVALID_ARG;
  const int st = enqueue_download(g, frame, xyz_out, rgb_out, patch_index_out, capacity, n_points, false);
  if (st) return st;
  if (*n_points) HIP_TRY(g->ctx, hipStreamSynchronize(g->ctx->d2h_stream));
  return VPCC_OK;
}

extern "C" int vpcc_gof_download_async(vpcc_gof* g, uint32_t frame, vpcc_point3* xyz_out, vpcc_color3* rgb_out,
                                       uint16_t* patch_index_out, size_t capacity, size_t* n_points) {
  if (!g || frame >= g->n_frames || !n_points) return VPCC_ERR_INVALID_ARG;
  if (g->download_done.empty()) {                             // one event per frame, made at the first asynchronous download
    g->download_done.assign(g->n_frames, nullptr);
    for (hipEvent_t& e : g->download_done) HIP_TRY(g->ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const int st = enqueue_download(g, frame, xyz_out, rgb_out, patch_index_out, capacity, n_points, true);
  if (st) return st;
  HIP_TRY(g->ctx, hipEventRecord(g->download_done[frame], g->ctx->d2h_stream));
  return VPCC_OK;
}

extern "C" int vpcc_gof_download_wait(vpcc_gof* g, uint32_t frame) {
  if (!g || frame >= g->n_frames || g->download_done.empty()) return VPCC_ERR_INVALID_ARG;
  // (no hipSetDevice and no use of the context: this may be called from another thread than the one that drives it)
  return hipEventSynchronize(g->download_done[frame]) == hipSuccess ? VPCC_OK : VPCC_ERR_DEVICE;
}

extern "C" int vpcc_gof_kernel_times(vpcc_gof* g, const char** names_out, float* ms_out, int max) {
  if (!g || g->launches_profiled == 0) return 0;
  (void)hipSetDevice(g->ctx->device);
  // (this gof's last kernel — not its stream: in the streaming Decoder that stream already carries the NEXT unit's launch, which
  // waits for that unit's 40 ms of planes; the lane sat here a whole unit long and posted the unit after next that much too late)
  if (g->launched) (void)hipEventSynchronize(g->results_ready);
  const LaunchTimings& l = g->history[(g->launches_profiled - 1) % kProfileRing];
  int n = 0;
  for (uint32_t i = 0; i < l.n && n < max; ++i, ++n) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, l.k[i].start, l.k[i].stop);
    if (names_out) names_out[n] = l.k[i].name;
    if (ms_out) ms_out[n] = ms;
  }
  return n;
}

extern "C" int vpcc_gof_profile_interval(vpcc_gof* g, uint32_t every) {
  if (!g || every == 0) return VPCC_ERR_INVALID_ARG;
  g->profile_every = every;
  return VPCC_OK;
}

extern "C" int vpcc_gof_kernel_time_means(vpcc_gof* g, uint32_t last_n, const char** names_out, float* mean_ms_out,
                                          uint32_t* launches_out, int max) {
  if (launches_out) *launches_out = 0;
  if (!g || g->launches_profiled == 0 || max <= 0) return 0;
  (void)hipSetDevice(g->ctx->device);
  if (g->launched) (void)hipEventSynchronize(g->results_ready);
  const uint64_t have = std::min<uint64_t>(g->launches_profiled, kProfileRing);
  const uint64_t take = std::min<uint64_t>(last_n ? last_n : have, have);
  std::vector<const char*> names;
  std::vector<double> sums;
  for (uint64_t j = 0; j < take; ++j) {
    const LaunchTimings& l = g->history[(g->launches_profiled - 1 - j) % kProfileRing];
    for (uint32_t i = 0; i < l.n; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, l.k[i].start, l.k[i].stop) != hipSuccess) continue;
      size_t q = 0;
      while (q < names.size() && std::strcmp(names[q], l.k[i].name) != 0) ++q;
      if (q == names.size()) { names.push_back(l.k[i].name); sums.push_back(0.0); }
      sums[q] += ms;
    }
  }
  if (launches_out) *launches_out = (uint32_t)take;
  int n = 0;
  for (size_t q = 0; q < names.size() && n < max; ++q, ++n) {
    if (names_out) names_out[n] = names[q];
    if (mean_ms_out) mean_ms_out[n] = (float)(sums[q] / (double)take);
  }
  return n;
}

extern "C" int vpcc_gof_algorithmic_bytes(vpcc_gof* g, uint32_t frame, uint64_t* bytes_out) {
  if (!g || frame >= g->n_frames || !bytes_out) return VPCC_ERR_INVALID_ARG;
  HIP_TRY(g->ctx, hipSetDevice(g->ctx->device));
  const int st = fetch_counts(g);
  if (st) return st;
  *bytes_out = g->shapes[frame].plane_bytes + 9ull * g->h_counts[frame];
  return VPCC_OK;
}
