// vpcc_runtime.hpp — objects and helpers shared by the files of the host runtime (internal to libvpcc_recon.so):
//   vpcc_runtime.hip      contexts, page-locked host memory, the one-shot seam replacements
//   vpcc_pool.hip         the context's pool ("two homes"): vpcc_ctx_reserve, blocks of a gof
//   vpcc_gof_create.hip   vpcc_gof_create: validate -> place planes -> lay out -> allocate -> describe -> enqueue
//   vpcc_gof.hip          launches, point counts, downloads, timings
//   vpcc_gof_smooth.hip   vpcc_gof_smooth
// One vpcc_ctx per GPU / worker thread; a vpcc_gof keeps a batch of independent atlas frames (reference: frames of a GOF
// are independent, src/decoder.rs:186, 403-407) resident in HBM and reconstructs them with batched launches.  There is no
// CPU fallback anywhere in these files: without a gfx950 device vpcc_ctx_create fails with VPCC_ERR_NO_DEVICE.
#pragma once

#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "vpcc_device.hpp"
#include "vpcc_host.hpp"

// ----------------------------------------------------------------- objects
struct vpcc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;        // kernels and D2H
  hipStream_t copy_stream = nullptr;   // everything of a new gof, in the order it is needed: descriptors, control words, H2D plane
                                       // ingest — overlaps the kernels of the previous GOF
  hipStream_t d2h_stream = nullptr;    // result downloads: wait for ONE gof's kernels only (results_ready), not for
                                       // whatever else has been queued behind them on the compute stream
  std::string last_error;
  uint32_t resident_tile_wgs_per_xcd = 128;   // workgroups of the tile kernel an XCD holds at a time (4 per CU)
  // Arenas of destroyed GOFs are kept for the next GOF of the same shape: hipMalloc/hipFree cost
  // milliseconds and hipFree synchronises the whole device, which would stall a GOF pipeline.
  std::vector<std::pair<void*, size_t>> arena_cache;
  std::vector<std::pair<void*, size_t>> stage_cache;   // ... and their page-locked descriptor staging buffers
  // The big blocks of a gof — the video planes it ingested and its output arrays — are kept in kParts parts by frame
  // (eight frames, one per XCD label, to part 0, the next eight to part 1, ...), and with a reserved pool
  // (vpcc_ctx_reserve) part p lies in home p: VRAM consists of KINDS of regions, 32 GB each, and a launch whose
  // traffic stays inside one kind is served 10 % slower than one that spreads it evenly over two (DESIGN.md 4.1,
  // "Two homes").  Which kind a piece of memory belongs to only a measurement tells: the pool is ONE allocation whose
  // GiB granules are classified once, when it is reserved, by timing the reconstruction kernel's output pattern
  // between granule 0 and each of them.
  static constexpr int kParts = vpcc::kGofParts;
  struct Block { void* ptr = nullptr; size_t bytes = 0; bool pooled = false; uint32_t run = 0; };
  struct Pool {
    static constexpr size_t kGranule = size_t(1) << 30;
    std::vector<void*> slabs;                            // the allocations (one; two when the first lay in one kind only)
    vpcc::PoolExtents space;                             // runs of one kind inside the slabs and the free extents (vpcc_host.cpp)
    vpcc_pool_info info{};
    bool reserved() const { return !slabs.empty(); }
  } pool;
  std::mutex pool_mutex;                                 // gofs of one context may be destroyed from another thread
  bool pool_pending = false;                             // a vpcc_ctx_reserve is under way
  std::vector<Block> block_cache;                        // big blocks of destroyed gofs that are allocations of their own
  std::vector<Block> lent;                               // vpcc_ctx_pool_alloc: blocks a producer of device planes holds
};

struct KernelTiming {
  const char* name;
  hipEvent_t start, stop;
};
// Profile mode keeps the event pairs of the last kProfileRing launches (one slot per vpcc_gof_reconstruct,
// a following vpcc_gof_smooth appends to the same slot), so that a caller can time a long back-to-back
// region and read the mean duration per kernel of exactly those launches afterwards.
struct LaunchTimings {
  std::vector<KernelTiming> k;   // event pairs are created once and reused when the ring wraps
  uint32_t n = 0;                // kernels timed in this launch
};
constexpr uint32_t kProfileRing = 512;

struct vpcc_gof {
  vpcc_ctx* ctx = nullptr;
  uint32_t n_frames = 0;
  uint32_t flags = 0;
  uint64_t capacity = 0;
  bool general = true;                 // general kernel sequence (vs single-pass fast path)
  bool plan_in_lds = false;            // tile path: every frame fits k_plan_tiles' LDS (else k_plan_cover / k_plan_items over the
                                       // virtual blocks k_plan_vblocks wrote when the gof was created)
  std::vector<vpcc::FrameShape> shapes;   // what validation learnt of every frame
  std::vector<vpcc::DevFrame> h_frames;   // host mirror of d_frames
  void* arena = nullptr;
  size_t arena_bytes = 0;
  vpcc_ctx::Block block[2 * vpcc_ctx::kParts];   // [2 * part]: ingested planes (gofs that own their planes), [2 * part + 1]: positions, colours, partition
  vpcc::DevFrame* d_frames = nullptr;
  uint32_t* d_counts = nullptr;        // n_frames contiguous point counters
  uint32_t* d_b2p = nullptr;           // all frames' block_to_patch, contiguous
  size_t b2p_words = 0;
  std::vector<size_t> b2p_off;         // per-frame offset (words) into d_b2p, n_frames + 1 entries
  // tile-kernel control words, one contiguous region: [tickets (one 256-B line per frame) | errors | scan states]
  uint32_t* d_tickets = nullptr;
  uint32_t* d_errors = nullptr;
  uint64_t* d_scan = nullptr;
  size_t ctrl_bytes = 0;
  uint32_t max_vb = 0;
  uint32_t* h_counts = nullptr;        // page-locked, inside the staging buffer: counts[n_frames] then errors[n_frames]
  bool counts_valid = false;
  bool launched = false;
  std::vector<vpcc::IngestPiece> ingest;   // plane ingest by kernel: the pieces (alive while their upload may read them)
  uint32_t ingest_extents = 0;         // plane ingest by extent: copies issued
  void* stage = nullptr;               // page-locked staging of the descriptors (returned to the context's cache)
  size_t stage_bytes = 0;
  std::vector<hipEvent_t> download_done;   // vpcc_gof_download_async: one per frame
  hipEvent_t upload_done = nullptr;
  hipEvent_t results_ready = nullptr;   // recorded behind the last kernel launched on this gof
  hipStream_t last_stream = nullptr;
  std::vector<LaunchTimings> history;   // profile mode: ring of kProfileRing launches
  uint64_t launches_profiled = 0;       // slot of the current launch = (launches_profiled - 1) % kProfileRing
  uint32_t profile_every = 1;           // profile mode: time every n-th reconstruct only (vpcc_gof_profile_interval)
  uint64_t reconstructs = 0;
  bool launch_is_timed = false;
  uint32_t generation = 0;             // launch counter of the tile kernel (tags look-back words)
  vpcc::TileLaunchMap tile_map;        // shares of the resident workgroups per frame, for the last (first, count) launched
  uint32_t tile_map_first = 0, tile_map_count = 0;
  bool tile_map_valid = false;
  void* smooth_grid = nullptr;         // smoothing scratch (on demand): dense cell grids + touched lists + list lengths
  size_t smooth_bytes = 0;
  bool smooth_clean = false;           // the scratch is all-zero (the invariant between launches)
  void* smooth_keys = nullptr;         // the cell lists of every chunk of 256 points of every frame, then their lengths
  void* smooth_moved = nullptr;        // both filters in one pass: which points moved (a bit each), and the cell each was counted in
};

namespace vpcc {

inline int fail(vpcc_ctx* ctx, int status, const std::string& msg) {
  if (ctx) ctx->last_error = msg;
  return status;
}

#define HIP_TRY(ctx, expr)                                                                       \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      return vpcc::fail((ctx), VPCC_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));    \
  } while (0)

// vpcc_runtime.hip: [lo, lo + bytes) as pieces that each lie inside one region page-locked through this library (an input
// page-locked chunk by chunk: a stretch of planes may cross from one chunk into the next) — or one piece when the caller
// page-locked it by its own means; false if some byte of it is in none.
bool pinned_pieces(const char* lo, size_t bytes, std::vector<std::pair<const char*, size_t>>* pieces);

// vpcc_pool.hip
void retire_pool(vpcc_ctx* ctx);                     // a context goes: its pool stays with the process when it is whole
bool release_kept_pools(int device);                 // out of memory somewhere: the kept pools go back to the driver
// hipMalloc that gives the context's cached blocks and the process's kept pools back to the driver before it gives up
hipError_t device_malloc(vpcc_ctx* ctx, void** out, size_t bytes);
int acquire_block(vpcc_ctx* ctx, int part, size_t bytes, vpcc_ctx::Block* out);
void release_block(vpcc_ctx* ctx, vpcc_ctx::Block& B);      // all work on the block is complete

// vpcc_gof.hip: event pairs around the kernels of a launch (profile mode)
struct Timer {
  vpcc_gof* g;
  hipStream_t s;
  bool on;
  LaunchTimings* slot = nullptr;
  // new_launch: a vpcc_gof_reconstruct opens the next ring slot; vpcc_gof_smooth appends to the current one
  Timer(vpcc_gof* g_, hipStream_t s_, bool new_launch);
  void begin(const char* name);
  void end();
};

}  // namespace vpcc
