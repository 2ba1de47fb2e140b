// vpcc_gof_smooth.hip — vpcc_gof_smooth: grid-based geometry / colour smoothing of reconstructed frames (SURVEY §8 a12;
// kernels: vpcc_smooth.hip; specification: oracle/vpcc_smoothing_spec.h).
#include <algorithm>
#include <cstdlib>

#include "vpcc_runtime.hpp"

using namespace vpcc;

// ------------------------------------------------------------------ smoothing
extern "C" int vpcc_gof_smooth(vpcc_gof* g, uint32_t first, uint32_t count, const vpcc_smoothing_params* p,
                               void* hip_stream) {
  if (!g || !p) return VPCC_ERR_INVALID_ARG;
  vpcc_ctx* ctx = g->ctx;
  if (count == 0 || first >= g->n_frames || count > g->n_frames - first) return fail(ctx, VPCC_ERR_INVALID_ARG, "frame range");
  if (!(g->flags & VPCC_GOF_WANT_PATCH_INDEX)) return fail(ctx, VPCC_ERR_STATE, "smoothing needs VPCC_GOF_WANT_PATCH_INDEX");
  if (p->geometry_bitdepth_3d < 1 || p->geometry_bitdepth_3d > 16) return fail(ctx, VPCC_ERR_INVALID_ARG, "bit depth");
  if ((p->flags & VPCC_SMOOTH_GEOMETRY) && p->grid_size < 2) return fail(ctx, VPCC_ERR_INVALID_ARG, "grid size");
  if ((p->flags & VPCC_SMOOTH_COLOR) && p->color_grid_size < 2) return fail(ctx, VPCC_ERR_INVALID_ARG, "colour grid size");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!g->launched) return fail(ctx, VPCC_ERR_STATE, "no reconstruct issued");
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  if (g->last_stream != s) HIP_TRY(ctx, hipStreamWaitEvent(s, g->results_ready, 0));   // behind the reconstruction
  g->last_stream = s;
  g->counts_valid = false;                                     // (the filters may raise a frame's error flag)
  Timer T(g, s, false);
  // No host synchronisation: the kernels read every frame's point count from device memory; the launches are
  // sized for the capacity and surplus workgroups leave at once.
  const uint32_t max_points = (uint32_t)g->capacity;
  bool all_attr = true, any_attr = false;
  for (uint32_t i = first; i < first + count; ++i) {
    all_attr = all_attr && g->h_frames[i].has_attr != 0;
    any_attr = any_attr || g->h_frames[i].has_attr != 0;
  }
  const bool want_geo = (p->flags & VPCC_SMOOTH_GEOMETRY) != 0, want_col = (p->flags & VPCC_SMOOTH_COLOR) != 0 && any_attr;
  if (want_col && !all_attr) return fail(ctx, VPCC_ERR_INVALID_ARG, "colour smoothing on a frame without attribute");
  // Both filters over the same grid: ONE statistics / mark / clear sequence serves both (coordinate sums in the cells,
  // colour sums in a parallel array).  The colour filter's cells are those of the SMOOTHED positions: the few points
  // the geometry filter moves into another cell take their count, colour and patch sums with them
  // (k_smooth_moved_*), which keeps every sum what a second statistics pass would have produced.
  const bool both = want_geo && want_col && p->grid_size == p->color_grid_size;
  for (int pass = 0; pass < 2; ++pass) {
    const bool geo = pass == 0;
    if (!(geo ? want_geo : want_col) || (both && !geo)) continue;
    const uint32_t G = geo ? p->grid_size : p->color_grid_size;
    const uint32_t w = ((1u << p->geometry_bitdepth_3d) + G - 1) / G;
    const size_t cells = (size_t)w * w * w;
    if (cells >= (size_t(1) << 32)) return fail(ctx, VPCC_ERR_UNSUPPORTED, "smoothing grid of 2^32 cells or more (cell indices are 32 bits)");
    // Scratch: per frame slot a dense grid (the cell index of every point has its own allocation).  At most
    // ~16 GiB: a GOF whose grids need more is smoothed in chunks of frames.  The scratch is all-zero
    // between launches (k_smooth_clear restores what a launch touched), so it is cleared only when it is
    // (re)allocated or after a failed launch.
    SmoothGrid sg{};
    const size_t corners = (size_t)(w + 1) * (w + 1) * (w + 1);
    sg.flags_offset = align_up(cells * sizeof(SmoothCell), 256);
    sg.near_offset = align_up(sg.flags_offset + corners, 256);
    sg.color_offset = both ? align_up(sg.near_offset + cells, 256) : 0;
    sg.slot_bytes = align_up(both ? sg.color_offset + cells * sizeof(SmoothColorCell) : sg.near_offset + cells, 256);
    // per span of 1 024 points: a cell list (as many entries reserved, a few dozen used) and its length
    const size_t chunks = (g->capacity + kSmoothListSpan - 1) / kSmoothListSpan + 1;
    sg.list_stride = chunks * kSmoothListLen;
    sg.count_stride = align_up(chunks, 64);
    sg.painted_stride = chunks * (kSmoothListLen / 64);
    if (!g->smooth_keys)
      HIP_TRY(ctx, device_malloc(ctx, &g->smooth_keys, (sizeof(uint32_t) * (sg.list_stride + 2 * sg.count_stride) + sizeof(uint64_t) * sg.painted_stride) * g->n_frames));
    sg.painted_base = (uint64_t*)g->smooth_keys;
    sg.list_base = (uint32_t*)(sg.painted_base + sg.painted_stride * g->n_frames);
    sg.count_base = sg.list_base + sg.list_stride * g->n_frames;
    sg.flag_base = sg.count_base + sg.count_stride * g->n_frames;
    if (both) {
      sg.moved_stride = align_up((g->capacity + 63) / 64 + 4, 32);
      sg.oldkey_stride = align_up(g->capacity, 4);
      if (!g->smooth_moved)
        HIP_TRY(ctx, device_malloc(ctx, &g->smooth_moved, (sizeof(uint64_t) * 3 * sg.moved_stride + sizeof(uint32_t) * sg.oldkey_stride) * g->n_frames));
      sg.moved_base = (uint64_t*)g->smooth_moved;
      sg.moved_painted_base = sg.moved_base + sg.moved_stride * g->n_frames;
      sg.oldkey_base = (uint32_t*)(sg.moved_painted_base + 2 * sg.moved_stride * g->n_frames);
    }
    // (VPCC_SMOOTH_SCRATCH_LIMIT_MB: the limit in MB, for tests that want several chunks out of a small gof)
    const char* limit_env = getenv("VPCC_SMOOTH_SCRATCH_LIMIT_MB");
    const size_t scratch_limit = limit_env ? std::max<size_t>(1, (size_t)atoll(limit_env)) << 20 : size_t(16) << 30;
    uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(count, scratch_limit / sg.slot_bytes));
    // The grids of `chunk` frames at a time — of fewer when the device has not got that much left (other gofs' scratch, other
    // tenants): the chunk is halved while the allocation fails, the pools kept for the device's next context are given back
    // before the last attempt.  (VPCC_SMOOTH_ALLOC_FAIL_ABOVE_MB: allocations above that size "fail", for the test of this path.)
    const char* fail_env = getenv("VPCC_SMOOTH_ALLOC_FAIL_ABOVE_MB");
    const size_t fail_above = fail_env ? (size_t)atoll(fail_env) << 20 : ~size_t(0);
    for (bool pools_released = false;;) {
      const size_t need = sg.slot_bytes * chunk;
      if (g->smooth_bytes >= need) break;
      if (g->smooth_grid) HIP_TRY(ctx, hipFree(g->smooth_grid));
      g->smooth_grid = nullptr;
      g->smooth_bytes = 0;
      const hipError_t e = need > fail_above ? hipErrorOutOfMemory : hipMalloc(&g->smooth_grid, need);
      if (e == hipSuccess) {
        g->smooth_bytes = need;
        g->smooth_clean = false;
        break;
      }
      (void)hipGetLastError();
      g->smooth_grid = nullptr;
      if (chunk > 1) { chunk = (chunk + 1) / 2; continue; }
      if (!pools_released && release_kept_pools(ctx->device)) { pools_released = true; continue; }
      return fail(ctx, VPCC_ERR_DEVICE, "smoothing: no device memory for the grid of one frame (" + std::to_string(need >> 20) + " MB): " + hipGetErrorString(e));
    }
    if (!g->smooth_clean) {
      HIP_TRY(ctx, hipMemsetAsync(g->smooth_grid, 0, g->smooth_bytes, s));
      g->smooth_clean = true;
    }
    sg.base = (unsigned char*)g->smooth_grid;
    static const char* const kNames[3][3] = {
        {"k_smooth_stats<geometry>", "k_smooth_stats<color>", "k_smooth_stats<both>"},
        {"k_smooth_mark<geometry>", "k_smooth_mark<color>", "k_smooth_mark<both>"},
        {"k_smooth_clear<geometry>", "k_smooth_clear<color>", "k_smooth_clear<both>"}};
    const int tag = both ? 2 : geo ? 0 : 1;
    for (uint32_t c0 = first; c0 < first + count; c0 += chunk) {
      const uint32_t c = std::min(chunk, first + count - c0);
      g->smooth_clean = false;                              // until the clearing kernel of this chunk is enqueued
      // the moved-point bits are indexed by frame SLOT, and every chunk uses slots 0 .. c-1 again: zeroed per chunk (a bit
      // left by the previous chunk's frame would send k_smooth_moved_sums to a point this frame may not even have)
      if (both) HIP_TRY(ctx, hipMemsetAsync(sg.moved_base, 0, sizeof(uint64_t) * sg.moved_stride * c, s));
      T.begin(kNames[0][tag]);
      launch_smooth_stats(g->d_frames, c0, c, max_points, sg, w, G, both ? 2u : geo ? 0u : 1u, s);
      T.end();
      T.begin(kNames[1][tag]);
      launch_smooth_mark(g->d_frames, c0, c, max_points, sg, w, s);
      launch_smooth_spans(g->d_frames, c0, c, max_points, sg, s);     // (timed with the marking: which spans of points have anything to do)
      T.end();
      if (geo) {
        T.begin("k_smooth_apply_geometry");
        launch_smooth_apply_geometry(g->d_frames, c0, c, max_points, sg, w, G, p->threshold, both, s);
        T.end();
      }
      if (both) {
        T.begin("k_smooth_moved");
        launch_smooth_moved(g->d_frames, c0, c, max_points, sg, w, G, s);
        T.end();
      }
      if (!geo || both) {
        T.begin("k_smooth_apply_color");
        launch_smooth_apply_color(g->d_frames, c0, c, max_points, sg, w, G, p->color_threshold_smoothing,
                                  p->color_threshold_difference, both, s);
        T.end();
      }
      T.begin(kNames[2][tag]);
      launch_smooth_clear(g->d_frames, c0, c, max_points, sg, w, G, both, s);
      T.end();
      HIP_TRY(ctx, hipGetLastError());
      g->smooth_clean = true;
    }
  }
  HIP_TRY(ctx, hipEventRecord(g->results_ready, s));
  return VPCC_OK;
}
