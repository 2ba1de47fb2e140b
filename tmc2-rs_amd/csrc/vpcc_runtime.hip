// vpcc_runtime.hip — host runtime and C ABI (include/vpcc_recon.h) of libvpcc_recon.so.
//
// One vpcc_ctx per GPU / worker thread; a vpcc_gof keeps a batch of independent atlas frames
// (reference: frames of a GOF are independent, src/decoder.rs:186, 403-407) resident in HBM in
// ONE arena and reconstructs them with batched launches.  There is no CPU fallback anywhere in
// this file: without a gfx950 device vpcc_ctx_create fails with VPCC_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "vpcc_device.hpp"
#include "vpcc_host.hpp"

using namespace vpcc;

// ----------------------------------------------------------------- objects
struct vpcc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;        // kernels and D2H
  hipStream_t copy_stream = nullptr;   // H2D plane ingest: overlaps the kernels of the previous GOF
  hipStream_t setup_stream = nullptr;  // descriptors and work lists of a new gof (small copies from pageable host memory: the
                                       // caller's thread waits for each — on the copy stream that was a wait for the 40-ms
                                       // ingest of the gof before)
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
  static constexpr int kParts = 2;
  struct Block { void* ptr = nullptr; size_t bytes = 0; bool pooled = false; uint32_t run = 0; };
  struct Pool {
    static constexpr size_t kGranule = size_t(1) << 30;
    std::vector<void*> slabs;                            // the allocations (one; two when the first lay in one kind only)
    PoolExtents space;                                   // runs of one kind inside the slabs and the free extents (vpcc_host.cpp)
    vpcc_pool_info info{};
    bool reserved() const { return !slabs.empty(); }
  } pool;
  std::mutex pool_mutex;                                 // gofs of one context may be destroyed from another thread
  bool pool_pending = false;                             // a vpcc_ctx_reserve is under way
  std::vector<Block> block_cache;                        // big blocks of destroyed gofs that are allocations of their own
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
  std::vector<FramePlan> plans;        // host-side per-frame plan
  std::vector<DevFrame> h_frames;      // host mirror of d_frames
  void* arena = nullptr;
  size_t arena_bytes = 0;
  vpcc_ctx::Block block[2 * vpcc_ctx::kParts];   // [2 * part]: ingested planes (gofs that own their planes), [2 * part + 1]: positions, colours, partition
  DevFrame* d_frames = nullptr;
  uint32_t* d_counts = nullptr;        // n_frames contiguous point counters
  uint32_t* d_b2p = nullptr;           // all frames' block_to_patch, contiguous
  size_t b2p_words = 0;
  // tile-kernel control words, one contiguous region: [tickets (one 256-B line per frame) | errors | scan states]
  uint32_t* d_tickets = nullptr;
  uint32_t* d_errors = nullptr;
  uint64_t* d_scan = nullptr;
  std::vector<size_t> scan_off;        // per-frame offset (words) into d_scan, n_frames+1 entries (one word per group)
  size_t ctrl_bytes = 0;
  uint32_t max_vb = 0;
  uint32_t* h_counts = nullptr;        // pinned: counts[n_frames] then errors[n_frames]
  bool h_counts_in_stage = false;      // ... inside the descriptor staging buffer (not an allocation of its own)
  bool counts_valid = false;
  bool launched = false;
  std::vector<IngestPiece> ingest;     // plane ingest by kernel: the pieces (alive while their upload may read them)
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
  TileLaunchMap tile_map;              // shares of the resident workgroups per frame, for the last (first, count) launched
  uint32_t tile_map_first = 0, tile_map_count = 0;
  bool tile_map_valid = false;
  void* smooth_grid = nullptr;         // smoothing scratch (on demand): dense cell grids + touched lists + list lengths
  size_t smooth_bytes = 0;
  bool smooth_clean = false;           // the scratch is all-zero (the invariant between launches)
  void* smooth_keys = nullptr;         // the cell lists of every chunk of 256 points of every frame, then their lengths
  void* smooth_moved = nullptr;        // both filters in one pass: which points moved (a bit each), and the cell each was counted in
};

namespace {

int fail(vpcc_ctx* ctx, int status, const std::string& msg) {
  if (ctx) ctx->last_error = msg;
  return status;
}

#define HIP_TRY(ctx, expr)                                                                       \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      return fail((ctx), VPCC_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));    \
  } while (0)

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct ArenaLayout {
  size_t total = 0;
  size_t take(size_t bytes) {
    const size_t off = total;
    total = align_up(total + bytes, 256);
    return off;
  }
};

}  // namespace

// ------------------------------------------------------------------ basics
extern "C" int vpcc_abi_version(void) { return VPCC_ABI_VERSION; }

extern "C" const char* vpcc_status_string(int status) {
  switch (status) {
    case VPCC_OK: return "ok";
    case VPCC_ERR_INVALID_ARG: return "invalid argument";
    case VPCC_ERR_UNSUPPORTED: return "outside the supported envelope (reference: unimplemented!)";
    case VPCC_ERR_PATCH_OUT_OF_CANVAS: return "patch/pixel outside canvas or plane (reference: assert!)";
    case VPCC_ERR_SHORT_VIDEO: return "video shorter than the atlas (reference: unwrap on None)";
    case VPCC_ERR_CAPACITY: return "output capacity too small";
    case VPCC_ERR_DEVICE: return "HIP runtime error";
    case VPCC_ERR_NO_DEVICE: return "no usable gfx950 device (no CPU fallback)";
    case VPCC_ERR_STATE: return "call order violated";
    default: return "unknown status";
  }
}

namespace { void retire_pool(vpcc_ctx* ctx); }

extern "C" int vpcc_ctx_create(int device_id, vpcc_ctx** out) {
  if (!out) return VPCC_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return VPCC_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return VPCC_ERR_NO_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VPCC_ERR_NO_DEVICE;   // kernels exist for gfx950 only
  if (hipSetDevice(device_id) != hipSuccess) return VPCC_ERR_NO_DEVICE;
  vpcc_ctx* ctx = new vpcc_ctx();
  ctx->device = device_id;
  ctx->resident_tile_wgs_per_xcd = (uint32_t)std::max(1, prop.multiProcessorCount / 8) * 4u;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->setup_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return VPCC_ERR_DEVICE;
  }
  *out = ctx;
  return VPCC_OK;
}

extern "C" void vpcc_ctx_destroy(vpcc_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (auto& a : ctx->arena_cache) (void)hipFree(a.first);
  for (auto& a : ctx->stage_cache) (void)hipHostFree(a.first);
  retire_pool(ctx);
  for (auto& b : ctx->block_cache) (void)hipFree(b.ptr);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->setup_stream) (void)hipStreamDestroy(ctx->setup_stream);
  if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
  delete ctx;
}

extern "C" const char* vpcc_last_error(const vpcc_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

extern "C" void* vpcc_ctx_stream(const vpcc_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int vpcc_ctx_bind_thread(vpcc_ctx* ctx, int* node_out) {
  if (!ctx) return VPCC_ERR_INVALID_ARG;
  if (node_out) *node_out = -1;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, ctx->device) != hipSuccess) return VPCC_OK;
  for (char* c = bus; *c; ++c) *c = (char)tolower((unsigned char)*c);      // sysfs names are lower case
  const std::string dir = std::string("/sys/bus/pci/devices/") + bus;
  int node = -1;
  if (FILE* f = std::fopen((dir + "/numa_node").c_str(), "r")) {
    if (std::fscanf(f, "%d", &node) != 1) node = -1;
    std::fclose(f);
  }
  if (node < 0) return VPCC_OK;                        // single-socket machines and most VMs report -1
  cpu_set_t set;
  CPU_ZERO(&set);
  int n_cpus = 0;
  if (FILE* f = std::fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r")) {
    int a = 0, b = 0;
    for (;;) {                                         // "0-15,64-79"
      if (std::fscanf(f, "%d", &a) != 1) break;
      b = a;
      int ch = std::fgetc(f);
      if (ch == '-') { if (std::fscanf(f, "%d", &b) != 1) break; ch = std::fgetc(f); }
      for (int c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET(c, &set); ++n_cpus; }
      if (ch != ',') break;
    }
    std::fclose(f);
  }
  if (!n_cpus) return VPCC_OK;
  // only CPUs the process may use at all (cgroup / taskset restrictions stay in force)
  cpu_set_t allowed;
  if (sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
    cpu_set_t both;
    CPU_AND(&both, &set, &allowed);
    if (CPU_COUNT(&both) == 0) return VPCC_OK;          // the node's CPUs are not ours: leave the thread where it is
    set = both;
  }
  if (pthread_setaffinity_np(pthread_self(), sizeof set, &set) != 0)
    return fail(ctx, VPCC_ERR_DEVICE, "pthread_setaffinity_np failed");
  if (node_out) *node_out = node;
  return VPCC_OK;
}

namespace {
// The page-locked regions this library made (vpcc_host_pin, vpcc_host_alloc), for the whole process: a stretch of planes
// that lies inside one of them may be copied in one piece, what lies between the planes included (extent ingest).
std::mutex g_pinned_mutex;
std::vector<std::pair<const char*, size_t>> g_pinned;
void note_pinned(const void* p, size_t bytes) {
  std::lock_guard<std::mutex> lock(g_pinned_mutex);
  g_pinned.emplace_back((const char*)p, bytes);
}
void forget_pinned(const void* p) {
  std::lock_guard<std::mutex> lock(g_pinned_mutex);
  for (size_t k = 0; k < g_pinned.size(); ++k)
    if (g_pinned[k].first == (const char*)p) { g_pinned.erase(g_pinned.begin() + k); return; }
}
// [lo, lo + bytes) as pieces that each lie inside one page-locked region (an input page-locked chunk by chunk: a stretch of
// planes may cross from one chunk into the next); false if some byte of it is in none.
bool pinned_pieces(const char* lo, size_t bytes, std::vector<std::pair<const char*, size_t>>* pieces) {
  pieces->clear();
  {
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    const char* cur = lo;
    const char* const hi = lo + bytes;
    while (cur < hi) {
      const std::pair<const char*, size_t>* in = nullptr;
      for (const auto& r : g_pinned)
        if (cur >= r.first && cur < r.first + r.second) { in = &r; break; }
      if (!in) break;
      const char* end = std::min(hi, in->first + in->second);
      pieces->emplace_back(cur, (size_t)(end - cur));
      cur = end;
    }
    if (cur == hi) return true;
    pieces->clear();
  }
  // page-locked by the caller's own means: both ends map to the device, linearly, as parts of an allocation of one size
  void *d0 = nullptr, *d1 = nullptr, *b0 = nullptr, *b1 = nullptr;
  size_t s0 = 0, s1 = 0;
  const bool ok = hipHostGetDevicePointer(&d0, const_cast<char*>(lo), 0) == hipSuccess && d0 &&
                  hipHostGetDevicePointer(&d1, const_cast<char*>(lo + bytes - 1), 0) == hipSuccess && d1 &&
                  (char*)d1 - (char*)d0 == (ptrdiff_t)(bytes - 1) &&
                  hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&b0), &s0, (hipDeviceptr_t)d0) == hipSuccess &&
                  hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&b1), &s1, (hipDeviceptr_t)d1) == hipSuccess && s0 == s1 && s0 >= bytes;
  (void)hipGetLastError();
  if (ok) pieces->emplace_back(lo, bytes);
  return ok;
}
}  // namespace

extern "C" int vpcc_host_pin(vpcc_ctx* ctx, const void* ptr, size_t bytes) {
  if (!ctx || !ptr || !bytes) return VPCC_ERR_INVALID_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // Portable: every device (every vpcc_ctx of the process) may DMA from it, not only ctx's
  HIP_TRY(ctx, hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterPortable | hipHostRegisterMapped));   // mapped: the ingest kernel reads it in place
  note_pinned(ptr, bytes);
  return VPCC_OK;
}

extern "C" int vpcc_host_unpin(vpcc_ctx* ctx, const void* ptr) {
  if (!ctx || !ptr) return VPCC_ERR_INVALID_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  forget_pinned(ptr);
  HIP_TRY(ctx, hipHostUnregister(const_cast<void*>(ptr)));
  return VPCC_OK;
}

extern "C" int vpcc_host_alloc(vpcc_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out || !bytes) return VPCC_ERR_INVALID_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipHostMalloc(out, bytes, hipHostMallocPortable));     // usable as a DMA target by every device
  note_pinned(*out, bytes);
  return VPCC_OK;
}

extern "C" int vpcc_host_free(vpcc_ctx* ctx, void* ptr) {     // ctx may be NULL (context already destroyed)
  (void)ctx;
  if (!ptr) return VPCC_ERR_INVALID_ARG;
  // no hipSetDevice: page-locked host memory is freed from any thread with any current device, and this is
  // called from a consumer's thread when it drops a frame — it must not change that thread's device
  forget_pinned(ptr);
  return hipHostFree(ptr) == hipSuccess ? VPCC_OK : VPCC_ERR_DEVICE;
}

extern "C" int vpcc_frame_validate(const vpcc_frame_desc* frame) { return validate_frame(frame); }

extern "C" uint64_t vpcc_frame_capacity_bound(const vpcc_frame_desc* frame) {
  if (!frame) return 0;
  return (uint64_t)frame->map_count * frame->width * frame->height;
}

// ------------------------------------------------------------ pool ("two homes")
namespace {

// Pools of destroyed contexts stay with the process, by device: memory given back to the driver is wiped before it is
// handed out again (about 40 GB/s), and every allocation of the process waits for that — a second Decoder opened right
// behind the first would spend seconds in hipMalloc.  The next vpcc_ctx_reserve on the device takes a kept pool over as
// it is, classification included.
std::mutex g_kept_pools_mutex;
std::vector<std::pair<int, vpcc_ctx::Pool>> g_kept_pools;      // (device, pool); never freed: the process's memory

void retire_pool(vpcc_ctx* ctx) {
  std::lock_guard<std::mutex> lock(ctx->pool_mutex);
  vpcc_ctx::Pool& P = ctx->pool;
  if (!P.reserved()) return;
  if (P.space.in_use[0] == 0 && P.space.in_use[1] == 0) {          // every gof of the context is gone: the pool is whole
    std::lock_guard<std::mutex> keep(g_kept_pools_mutex);
    g_kept_pools.emplace_back(ctx->device, std::move(P));
  } else {
    for (void* q : P.slabs) (void)hipFree(q);
  }
  P = vpcc_ctx::Pool{};
}

// Out of memory somewhere: the pools kept for the device's next context go back to the driver.  Returns whether any did.
bool release_kept_pools(int device) {
  std::lock_guard<std::mutex> keep(g_kept_pools_mutex);
  bool any = false;
  for (size_t k = g_kept_pools.size(); k-- > 0;)
    if (g_kept_pools[k].first == device) {
      for (void* q : g_kept_pools[k].second.slabs) (void)hipFree(q);
      g_kept_pools.erase(g_kept_pools.begin() + (long)k);
      any = true;
    }
  return any;
}

// Blocks of destroyed gofs that did not come from the pool are kept for the next gof of the same size: hipMalloc /
// hipFree cost milliseconds and hipFree synchronises the whole device, which would stall a GOF pipeline.
constexpr size_t kBlockCacheEntries = 8;

// A block of `bytes` for part `part` of a gof: from the pool's home `part` (then from the other home), else an
// allocation of its own (from the context's cache of such blocks when one fits).
int acquire_block(vpcc_ctx* ctx, int part, size_t bytes, vpcc_ctx::Block* out) {
  bytes = (bytes + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
  std::lock_guard<std::mutex> lock(ctx->pool_mutex);
  if (ctx->pool.reserved()) {
    const int want = ctx->pool.info.kinds > 1 ? part % 2 : 0;
    for (int t = 0; t < 2; ++t) {
      char* p = nullptr;
      uint32_t run = 0;
      if (ctx->pool.space.take((want + t) % 2, bytes, &p, &run)) {
        *out = vpcc_ctx::Block{p, bytes, true, run};
        if (t) ctx->pool.info.other_home++;
        return VPCC_OK;
      }
    }
    ctx->pool.info.fallbacks++;
  }
  auto& cache = ctx->block_cache;
  for (size_t k = 0; k < cache.size(); ++k)
    if (cache[k].bytes >= bytes && cache[k].bytes <= bytes + bytes / 4) {
      *out = cache[k];
      cache.erase(cache.begin() + k);
      return VPCC_OK;
    }
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    for (auto& b : cache) (void)hipFree(b.ptr);          // make room and try once more
    cache.clear();
    if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();                             // (a failed call's error stays "last" until somebody asks)
      // ... and the pools kept for the device's next context
      if (!release_kept_pools(ctx->device) || hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        ctx->last_error = "no device memory for a gof block of " + std::to_string(bytes >> 20) + " MB";
        return VPCC_ERR_DEVICE;
      }
    }
  }
  *out = vpcc_ctx::Block{p, bytes, false, 0};
  return VPCC_OK;
}

void release_block(vpcc_ctx* ctx, vpcc_ctx::Block& B) {      // all work on the block is complete
  if (!B.ptr) return;
  std::lock_guard<std::mutex> lock(ctx->pool_mutex);
  if (B.pooled) {
    ctx->pool.space.give_back(B.run, (char*)B.ptr, B.bytes);
  } else if (ctx->block_cache.size() < kBlockCacheEntries) {
    ctx->block_cache.push_back(B);
  } else {
    (void)hipFree(B.ptr);
  }
  B = vpcc_ctx::Block{};
}

}  // namespace

extern "C" int vpcc_ctx_reserve(vpcc_ctx* ctx, uint64_t bytes, vpcc_pool_info* out) {
  // May run on a thread of its own beside the context's worker (the streaming Decoder does that): it works on a
  // stream and a Pool of its own, touches nothing of the context but its device id until the finished pool is handed
  // over under the pool mutex, and reports through its status only (not vpcc_last_error).
  if (!ctx) return VPCC_ERR_INVALID_ARG;
  {
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    if (ctx->pool.reserved() || ctx->pool_pending) return VPCC_ERR_STATE;
    ctx->pool_pending = true;
  }
  struct Pending { vpcc_ctx* c; ~Pending() { std::lock_guard<std::mutex> lock(c->pool_mutex); c->pool_pending = false; } } pending{ctx};
  constexpr size_t G = vpcc_ctx::Pool::kGranule;
  const size_t n = (size_t)((bytes + G - 1) / G);
  if (n < 2) return VPCC_ERR_INVALID_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return VPCC_ERR_DEVICE;
  const auto t0 = std::chrono::steady_clock::now();
  const bool trace = getenv("VPCC_RUNTIME_TRACE") != nullptr;
  {
    // a pool an earlier context of this process left behind on the device, if it is big enough (and not twice as big)
    std::lock_guard<std::mutex> keep(g_kept_pools_mutex);
    for (size_t k = 0; k < g_kept_pools.size(); ++k) {
      vpcc_ctx::Pool& K = g_kept_pools[k].second;
      if (g_kept_pools[k].first != ctx->device || K.info.bytes < n * G || K.info.bytes > 3 * n * G) continue;
      K.info.ms_spent = 0.f;
      K.info.reused = 1;
      K.info.other_home = K.info.fallbacks = 0;
      if (out) *out = K.info;
      if (trace) fprintf(stderr, "[vpcc] pool: took over the %llu-GiB pool an earlier context left on device %d\n",
                         (unsigned long long)(K.info.bytes >> 30), ctx->device);
      std::lock_guard<std::mutex> lock(ctx->pool_mutex);
      ctx->pool = std::move(K);
      g_kept_pools.erase(g_kept_pools.begin() + k);
      return VPCC_OK;
    }
  }
  vpcc_ctx::Pool P;
  void* base = nullptr;
  if (hipMalloc(&base, n * G) != hipSuccess) {
    (void)hipGetLastError();
    return VPCC_ERR_DEVICE;
  }
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipFree(base); return VPCC_ERR_DEVICE; }
  // Classification: the reconstruction kernel's OUTPUT pattern (thousands of waves, each writing its own run of
  // positions into one array and of colours into another) runs at 3.7 TB/s when both arrays lie in one kind of region
  // and at 5.3 TB/s when they lie in two (profiles/r03/pair_offset.txt, profiles/r04/pool.txt).  Positions in the
  // pool's granule 0, colours in granule j, for every j: the slow pairings are granule 0's kind.
  hipEvent_t a = nullptr, b = nullptr;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { (void)hipFree(base); (void)hipStreamDestroy(s); return VPCC_ERR_DEVICE; }
  constexpr uint32_t kItems = 333000;                     // 607 MB of positions + 304 MB of colours: one 128-frame launch
  unsigned char* const ref = (unsigned char*)base;
  auto probe = [&](unsigned char* other) -> float {        // GB/s
    launch_probe_outputs(ref, other, kItems, s);
    (void)hipEventRecord(a, s);
    for (int r = 0; r < 2; ++r) launch_probe_outputs(ref, other, kItems, s);
    (void)hipEventRecord(b, s);
    if (hipEventSynchronize(b) != hipSuccess) return 0.f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms > 0.f ? (float)((double)kItems * 2736.0 * 2.0 / (ms * 1e-3) / 1e9) : 0.f;
  };
  std::vector<float> rate(n, 0.f);
  for (size_t j = 1; j < n; ++j) rate[j] = probe(ref + j * G);
  {
    // a rate well away from both levels (somebody else used the GPU during that probe): measured again, the faster counts
    float lo0 = rate[1], hi0 = rate[1];
    for (size_t j = 1; j < n; ++j) { lo0 = std::min(lo0, rate[j]); hi0 = std::max(hi0, rate[j]); }
    for (size_t j = 1; j < n && hi0 > 1.15f * lo0; ++j)
      if (rate[j] > 1.06f * lo0 && rate[j] < 0.94f * hi0)
        for (int again = 0; again < 2; ++again) rate[j] = std::max(rate[j], probe(ref + j * G));
  }
  std::vector<float> sorted(rate.begin() + 1, rate.end());
  std::sort(sorted.begin(), sorted.end());
  const float lo = sorted.front(), hi = sorted.back(), med = sorted[sorted.size() / 2];
  // two levels in this slab: split between them; one level: it is granule 0's own (a pairing with another kind is faster)
  const float threshold = hi > 1.15f * lo ? 0.5f * (lo + hi) : 1.15f * med;
  auto classify = [&](std::vector<uint8_t>& kind, const std::vector<float>& r, size_t first) {
    for (size_t j = first; j < kind.size(); ++j) kind[j] = r[j] > threshold ? 1 : 0;
    for (size_t j = std::max<size_t>(first, 1); j + 1 < kind.size(); ++j)      // a lone granule between two of the other kind: a mis-measurement
      if (kind[j - 1] == kind[j + 1] && kind[j] != kind[j - 1]) kind[j] = kind[j - 1];
  };
  auto add_slab = [&](void* slab, const std::vector<uint8_t>& kind) {
    P.slabs.push_back(slab);
    for (size_t j = 0; j < kind.size();) {                    // runs of one kind -> free extents
      size_t e = j;
      while (e < kind.size() && kind[e] == kind[j]) ++e;
      P.space.add_run((char*)slab + j * G, (e - j) * G, (int)kind[j]);
      P.info.bytes_of_kind[kind[j]] += (e - j) * G;
      j = e;
    }
    P.info.bytes += kind.size() * G;
    P.info.granules += (uint32_t)kind.size();
  };
  auto kinds_string = [](const std::vector<uint8_t>& kind) { std::string t; for (uint8_t k : kind) t += k ? 'b' : 'a'; return t; };
  P.info = vpcc_pool_info{};
  std::vector<uint8_t> kind0(n, 0);
  classify(kind0, rate, 1);
  add_slab(base, kind0);
  P.info.probe_gbps_same = lo;
  P.info.probe_gbps_other = hi;
  if (trace) fprintf(stderr, "[vpcc] pool: %zu GiB classified: %s (probe %.0f .. %.0f GB/s)\n", n, kinds_string(kind0).c_str(), lo, hi);
  // A slab that lies in ONE kind (on some GPUs the first 60 GB of VRAM are alike), or all but a quarter of it: look further
  // away for more of the other — a spacer of 16 GiB nobody uses, then a candidate of half the pool's size, up to four times, while
  // at least a third of the device's memory stays free.  Spacers and rejected candidates are freed at the end (memory
  // given back is wiped by the driver before it is handed out again, and whoever allocates next waits for that).
  const int small = P.info.bytes_of_kind[1] < P.info.bytes_of_kind[0] ? 1 : 0;       // the kind the slab has less of
  if (P.info.bytes_of_kind[small] * 4 < n * G) {
    std::vector<void*> spare;
    const size_t m = std::max<size_t>(2, n / 2);
    for (int attempt = 0; attempt < 4; ++attempt) {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < total_b / 3 + (16 + m) * G) break;
      void* spacer = nullptr;
      void* cand = nullptr;
      if (hipMalloc(&spacer, 16 * G) != hipSuccess) { (void)hipGetLastError(); break; }
      spare.push_back(spacer);
      if (hipMalloc(&cand, m * G) != hipSuccess) { (void)hipGetLastError(); break; }
      std::vector<float> r(m, 0.f);
      for (size_t j = 0; j < m; ++j) r[j] = probe((unsigned char*)cand + j * G);
      std::vector<uint8_t> kind(m, 0);
      classify(kind, r, 0);
      size_t other = 0;
      for (uint8_t k : kind) other += k == small ? 1 : 0;
      if (trace) fprintf(stderr, "[vpcc] pool: candidate %d behind a 16-GiB spacer: %s\n", attempt, kinds_string(kind).c_str());
      if (other * 2 >= m) {
        add_slab(cand, kind);
        for (float x : r) P.info.probe_gbps_other = std::max(P.info.probe_gbps_other, x);
        break;
      }
      spare.push_back(cand);
    }
    for (void* q : spare) (void)hipFree(q);
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipStreamDestroy(s);
  P.info.kinds = P.info.bytes_of_kind[1] ? 2u : 1u;
  P.info.ms_spent = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (trace)
    fprintf(stderr, "[vpcc] pool reserved in %.1f ms: %llu GiB of granule 0's kind, %llu GiB of the other\n", P.info.ms_spent,
            (unsigned long long)(P.info.bytes_of_kind[0] >> 30), (unsigned long long)(P.info.bytes_of_kind[1] >> 30));
  if (out) *out = P.info;
  {
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    ctx->pool = std::move(P);
  }
  return VPCC_OK;
}

extern "C" int vpcc_ctx_pool_info(vpcc_ctx* ctx, vpcc_pool_info* out) {
  if (!ctx || !out) return VPCC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(ctx->pool_mutex);
  *out = ctx->pool.info;
  out->in_use[0] = ctx->pool.space.in_use[0];
  out->in_use[1] = ctx->pool.space.in_use[1];
  return VPCC_OK;
}

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
  // with a staging buffer keeps its counts in it, and the buffer goes back to the context's cache)
  if (gof->h_counts && !gof->h_counts_in_stage) (void)hipHostFree(gof->h_counts);
  delete gof;
}

namespace {

// Copies a (possibly strided) plane of the caller's — host memory, or device memory (VPCC_GOF_COPY_PLANES) — into a
// tight device plane.
int copy_plane(vpcc_ctx* ctx, void* dst, const void* src, size_t elem, uint32_t width, uint32_t height,
               uint32_t stride, hipStream_t s, hipMemcpyKind dir = hipMemcpyHostToDevice) {
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

int gof_create_impl(vpcc_ctx* ctx, const vpcc_frame_desc* frames, uint32_t n_frames, vpcc_memory_kind kind,
                    uint64_t capacity_points, uint32_t gof_flags, vpcc_gof* g) {
  g->ctx = ctx;
  g->n_frames = n_frames;
  g->flags = gof_flags;
  // the planes are copied into memory of the gof's own: host planes always, device planes on request
  const bool own_planes = kind == VPCC_MEM_HOST || (gof_flags & VPCC_GOF_COPY_PLANES) != 0;
  const hipMemcpyKind dir = kind == VPCC_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  g->plans.resize(n_frames);
  g->h_frames.resize(n_frames);

  // 1. validate + plan every frame on the host (patch table -> affine patches + virtual blocks)
  bool all_simple = true;
  uint64_t cap = capacity_points;
  for (uint32_t i = 0; i < n_frames; ++i) {
    const int st = validate_frame(&frames[i]);
    if (st) return fail(ctx, st, "frame " + std::to_string(i) + ": " + vpcc_status_string(st));
  }
  {
    // Frames are planned independently (0.2 ms each for S-longdress: ownership of 7 000 blocks, a 32-byte item per
    // owned block): a big batch is dealt to a few threads — in the streaming Decoder the lane's thread is the
    // bottleneck of a run once the kernels take 0.5 % of it.
    const auto t_plan = std::chrono::steady_clock::now();
    const uint32_t team = n_frames >= 16 ? std::min<uint32_t>(8u, std::max(1u, std::thread::hardware_concurrency() / 2u)) : 1u;
    auto work = [&](uint32_t t) { for (uint32_t i = t; i < n_frames; i += team) plan_frame(frames[i], &g->plans[i]); };
    std::vector<std::thread> helpers;
    for (uint32_t t = 1; t < team; ++t) helpers.emplace_back(work, t);
    work(0);
    for (std::thread& h : helpers) h.join();
    if (getenv("VPCC_RUNTIME_TRACE"))
      fprintf(stderr, "[vpcc] planned %u frames on %u thread(s) in %.1f ms\n", n_frames, team,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan).count());
  }
  for (uint32_t i = 0; i < n_frames; ++i) {
    all_simple = all_simple && g->plans[i].tile_eligible;
    g->max_vb = std::max(g->max_vb, (uint32_t)g->plans[i].vblocks.size());
    if (capacity_points == 0) cap = std::max<uint64_t>(cap, vpcc_frame_capacity_bound(&frames[i]));
  }
  if (cap == 0) cap = 1;
  if (cap > 0xFFFFFFF0ull) return fail(ctx, VPCC_ERR_INVALID_ARG, "capacity_points exceeds 32 bits");
  g->capacity = cap;
  (void)all_simple;

  // 2. arena layout
  ArenaLayout L;
  const size_t off_frames = L.take(sizeof(DevFrame) * n_frames);
  // what the HOST writes — frame descriptors, patches, virtual blocks, item templates — lies together at the arena's start:
  // with page-locked planes (VPCC_GOF_ASYNC_UPLOAD) it is put together in a page-locked staging buffer and arrives as ONE copy
  // at the head of the gof's ingest (below)
  struct HostOff { size_t patches, vblocks, patch_items; };
  std::vector<HostOff> hoffs(n_frames);
  for (uint32_t i = 0; i < n_frames; ++i) {
    const FramePlan& P = g->plans[i];
    hoffs[i].patches = L.take(sizeof(DevPatch) * std::max<size_t>(P.patches.size(), 1));
    hoffs[i].vblocks = L.take(sizeof(VBlock) * std::max<size_t>(P.vblocks.size(), 1));
    hoffs[i].patch_items = L.take(sizeof(TileItem) * std::max<size_t>(P.patch_items.size(), 1));
  }
  const size_t host_end = L.total;
  const size_t off_counts = L.take(sizeof(uint32_t) * n_frames);
  struct Off {
    size_t patches, vblocks, items, patch_items, b2p, vb_count, vb_offset, xyz, rgb, pidx, occ, geo[2], ay[2], au[2], av[2];
  };
  std::vector<Off> offs(n_frames);
  // control words of the single-pass path: one contiguous region, zeroed once at creation
  g->scan_off.assign(n_frames + 1, 0);
  for (uint32_t i = 0; i < n_frames; ++i)
    g->scan_off[i + 1] = g->scan_off[i] + (g->plans[i].tile_bound + kTileScanGranule - 1) / kTileScanGranule;
  const size_t ctrl_begin = L.total;
  const size_t off_tickets = L.total;
  L.total += 256 * (size_t)n_frames;              // one ticket per 256-B line: same-line atomics serialise
  const size_t off_errors = L.total;
  L.total += sizeof(uint32_t) * n_frames;
  L.total = align_up(L.total, 8);
  const size_t off_scan = L.total;
  L.total += sizeof(uint64_t) * std::max<size_t>(g->scan_off[n_frames], 1);
  g->ctrl_bytes = L.total - ctrl_begin;
  L.total = align_up(L.total, 256);
  // block_to_patch of all frames contiguous: one memset per reconstruct
  size_t b2p_begin = L.total;
  for (uint32_t i = 0; i < n_frames; ++i) {
    const FramePlan& P = g->plans[i];
    offs[i].b2p = L.total;
    L.total += sizeof(uint32_t) * (size_t)P.bw * P.bh;
  }
  g->b2p_words = (L.total - b2p_begin) / sizeof(uint32_t);
  L.total = align_up(L.total, 256);
  // The big blocks' own layouts: [2 * part] planes, [2 * part + 1] outputs, in kParts parts by frame — eight frames (one
  // per XCD label) to part 0, the next eight to part 1, and so on: every launch range moves the same bytes in both parts,
  // and with a reserved pool (vpcc_ctx_reserve) the parts lie in the two kinds of VRAM regions (DESIGN.md 4.1 "Two homes").
  ArenaLayout LB[2 * vpcc_ctx::kParts];
  auto part_of = [&](uint32_t i) { return (int)((i >> 3) % vpcc_ctx::kParts); };
  // kind 0 planes / 1 outputs
  auto takek = [&](uint32_t i, int kind, int, size_t bytes) { return LB[2 * part_of(i) + kind].take(bytes); };
  size_t ingest_bound = 0;
  // Plane ingest (page-locked host planes, VPCC_GOF_ASYNC_UPLOAD):
  //   * planes that lie next to each other in the caller's memory — the output of a decoder that allocates from one pool, a
  //     decoded-GOF container — keep that arrangement on the device and arrive as ONE copy per stretch (an EXTENT): with the
  //     copy engine moving 146-MB stretches the link runs at 57 GB/s host -> device AND 53 GB/s device -> host at the same
  //     time (tools/micro/zero_copy); a kernel that pulls the same bytes (zero-copy reads) gets 46 GB/s beside pushed
  //     results, and 1 280 copies of single planes 34 GB/s;
  //   * other tight planes are pulled by kernel (k_ingest_planes), strided ones go through the copy engine one by one.
  const bool pinned = kind == VPCC_MEM_HOST && (gof_flags & VPCC_GOF_ASYNC_UPLOAD);
  struct PlaneRef { const char* src; size_t bytes; size_t* slot; int part; };
  struct Extent { const char* lo; size_t bytes; size_t dev; int part; std::vector<std::pair<const char*, size_t>> pieces; };
  std::vector<Extent> extents;
  bool by_extent = false;
  if (own_planes && pinned && !getenv("VPCC_NO_EXTENT_INGEST")) {
    std::vector<PlaneRef> refs;
    bool tight = true;
    for (uint32_t i = 0; i < n_frames && tight; ++i) {
      const vpcc_frame_desc& F = frames[i];
      Off& o = offs[i];
      auto add = [&](const void* src, size_t bytes, size_t* slot) { refs.push_back(PlaneRef{(const char*)src, bytes, slot, part_of(i)}); };
      tight = F.occupancy.stride == F.occupancy.width;
      add(F.occupancy.y, (size_t)F.occupancy.width * F.occupancy.height, &o.occ);
      for (uint32_t m = 0; m < F.map_count; ++m) {
        tight = tight && F.geometry[m].stride == F.geometry[m].width;
        add(F.geometry[m].y, (size_t)F.geometry[m].width * F.geometry[m].height * 2, &o.geo[m]);
        if (F.attribute_count) {
          tight = tight && F.attribute[m].stride == F.attribute[m].width;
          add(F.attribute[m].y, (size_t)F.attribute[m].width * F.attribute[m].height * 2, &o.ay[m]);
          add(F.attribute[m].u, chroma_elems(F.attribute[m]) * 2, &o.au[m]);
          add(F.attribute[m].v, chroma_elems(F.attribute[m]) * 2, &o.av[m]);
        }
      }
    }
    if (tight && !refs.empty()) {
      std::stable_sort(refs.begin(), refs.end(), [](const PlaneRef& a, const PlaneRef& b) { return a.part != b.part ? a.part < b.part : a.src < b.src; });
      const size_t kGap = 256u << 10;                         // what may lie between two planes of an extent (patch tables, headers)
      std::vector<std::pair<size_t, size_t>> span;             // [first, last] plane of every extent
      for (size_t k = 0; k < refs.size(); ++k) {
        const char* end = extents.empty() ? nullptr : extents.back().lo + extents.back().bytes;
        if (!extents.empty() && extents.back().part == refs[k].part && refs[k].src <= end + kGap) {
          extents.back().bytes = std::max<size_t>(extents.back().bytes, (size_t)(refs[k].src + refs[k].bytes - extents.back().lo));
          span.back().second = k;
        } else {
          extents.push_back(Extent{refs[k].src, refs[k].bytes, 0, refs[k].part, {}});
          span.emplace_back(k, k);
        }
      }
      // worth it when stretches are long, and every stretch must be page-locked memory from end to end (what lies between its
      // planes is copied along)
      by_extent = extents.size() * 4 <= refs.size();
      if (getenv("VPCC_RUNTIME_TRACE")) fprintf(stderr, "[vpcc] ingest: %zu planes in %zu stretches\n", refs.size(), extents.size());
      for (size_t e = 0; e < extents.size() && by_extent; ++e) {
        by_extent = pinned_pieces(extents[e].lo, extents[e].bytes, &extents[e].pieces);
        if (!by_extent && getenv("VPCC_RUNTIME_TRACE"))
          fprintf(stderr, "[vpcc] ingest: stretch %zu (%zu bytes at %p) is not page-locked memory from end to end\n", e, extents[e].bytes,
                  (const void*)extents[e].lo);
      }
      (void)hipGetLastError();
      if (by_extent)
        for (size_t e = 0; e < extents.size(); ++e) {
          // the device copy lies where the host stretch lies modulo 256: every plane keeps its alignment
          const size_t shift = (uintptr_t)extents[e].lo & 255u;
          extents[e].dev = LB[2 * extents[e].part + 0].take(extents[e].bytes + 256) + shift;
          for (size_t k = span[e].first; k <= span[e].second; ++k) *refs[k].slot = extents[e].dev + (size_t)(refs[k].src - extents[e].lo);
        }
      else
        extents.clear();
    }
  }
  for (uint32_t i = 0; i < n_frames; ++i) {
    const vpcc_frame_desc& F = frames[i];
    const FramePlan& P = g->plans[i];
    Off& o = offs[i];
    o.patches = hoffs[i].patches;
    o.vblocks = hoffs[i].vblocks;
    o.items = L.take(sizeof(TileItem) * (((P.tile_bound + kTileItemsPerGroup - 1) / kTileItemsPerGroup) * kTileItemsPerGroup + kTileItemsPerGroup));
    o.patch_items = hoffs[i].patch_items;
    o.vb_count = L.take(sizeof(uint32_t) * std::max<size_t>(P.vblocks.size(), 1));
    o.vb_offset = L.take(sizeof(uint32_t) * std::max<size_t>(P.vblocks.size(), 1));
    o.xyz = takek(i, 1, 0, sizeof(vpcc_point3) * (cap + 4));                    // (output block: positions, colours, partition; + 4:
    o.rgb = F.attribute_count ? takek(i, 1, 1, sizeof(vpcc_color3) * (cap + 4)) : 0;   //  the smoothing kernels read whole quads of points,
    o.pidx = (gof_flags & VPCC_GOF_WANT_PATCH_INDEX) ? takek(i, 1, 0, sizeof(uint16_t) * (cap + 4)) : 0;   // so a quad that begins inside an array must end in memory)
    if (own_planes && !by_extent) {
      // (+ 16: a plane pulled by the ingest kernel starts 0 or 8 bytes behind its 256-byte boundary — where its source does modulo 16)
      auto plane = [&](int sub, size_t bytes) { ingest_bound += bytes / kIngestPieceBytes + 1; return takek(i, 0, sub, bytes + 16); };
      o.occ = plane(0, (size_t)F.occupancy.width * F.occupancy.height);
      for (uint32_t m = 0; m < F.map_count; ++m) {
        o.geo[m] = plane(0, (size_t)F.geometry[m].width * F.geometry[m].height * 2);
        if (F.attribute_count) {
          o.ay[m] = plane(1, (size_t)F.attribute[m].width * F.attribute[m].height * 2);
          o.au[m] = plane(1, chroma_elems(F.attribute[m]) * 2);
          o.av[m] = plane(1, chroma_elems(F.attribute[m]) * 2);
        }
      }
    }
  }
  // Plane ingest by kernel (k_ingest_planes) for page-locked host planes (VPCC_GOF_ASYNC_UPLOAD says they are; vpcc_host_pin
  // maps them for the device): one launch instead of ten hipMemcpyAsync per frame.  VPCC_NO_PULL_INGEST=1: the copy engines.
  const bool pull = pinned && !by_extent && !getenv("VPCC_NO_PULL_INGEST");
  const size_t off_ingest = pull ? L.take(sizeof(IngestPiece) * ingest_bound) : 0;
  g->arena_bytes = L.total;
  for (size_t k = 0; k < ctx->arena_cache.size(); ++k) {          // smallest cached arena that fits
    auto& a = ctx->arena_cache[k];
    if (a.second >= L.total && a.second <= L.total + L.total / 4) {
      g->arena = a.first;
      g->arena_bytes = a.second;
      ctx->arena_cache.erase(ctx->arena_cache.begin() + k);
      break;
    }
  }
  const auto t_alloc = std::chrono::steady_clock::now();
  if (!g->arena && hipMalloc(&g->arena, g->arena_bytes) != hipSuccess) {
    (void)hipGetLastError();
    g->arena = nullptr;
    if (!release_kept_pools(ctx->device) || hipMalloc(&g->arena, g->arena_bytes) != hipSuccess) {
      (void)hipGetLastError();
      g->arena = nullptr;
      return fail(ctx, VPCC_ERR_DEVICE, "no device memory for a gof's arena of " + std::to_string(g->arena_bytes >> 20) + " MB");
    }
  }
  for (int j = 0; j < 2 * vpcc_ctx::kParts; ++j)
    if (LB[j].total) {
      const int st = acquire_block(ctx, j / 2, LB[j].total + 256, &g->block[j]);
      if (st) return st;
    }
  char* base = (char*)g->arena;
  // descriptor staging: a page-locked buffer of the context's (kept for the next gof)
  const bool staged = pinned && !getenv("VPCC_NO_STAGED_DESCRIPTORS");
  const size_t stage_need = align_up(host_end, 256) + sizeof(uint32_t) * 2 * n_frames;
  if (staged) {
    for (size_t k = 0; k < ctx->stage_cache.size(); ++k)
      if (ctx->stage_cache[k].second >= stage_need) {
        g->stage = ctx->stage_cache[k].first;
        g->stage_bytes = ctx->stage_cache[k].second;
        ctx->stage_cache.erase(ctx->stage_cache.begin() + k);
        break;
      }
    if (!g->stage) {
      g->stage_bytes = stage_need + stage_need / 4;
      HIP_TRY(ctx, hipHostMalloc(&g->stage, g->stage_bytes, hipHostMallocDefault));
    }
    g->h_counts = (uint32_t*)((char*)g->stage + align_up(host_end, 256));      // the point counts come back into the same buffer
    g->h_counts_in_stage = true;
  } else {
    HIP_TRY(ctx, hipHostMalloc((void**)&g->h_counts, sizeof(uint32_t) * 2 * n_frames, hipHostMallocDefault));
  }
  char* const stage = (char*)g->stage;
  auto kb = [&](uint32_t i, int kind, int) { return (char*)g->block[2 * part_of(i) + kind].ptr; };
  g->d_frames = (DevFrame*)(base + off_frames);
  g->d_counts = (uint32_t*)(base + off_counts);
  g->d_b2p = (uint32_t*)(base + b2p_begin);
  g->d_tickets = (uint32_t*)(base + off_tickets);
  g->d_errors = (uint32_t*)(base + off_errors);
  g->d_scan = (uint64_t*)(base + off_scan);
  HIP_TRY(ctx, hipEventCreateWithFlags(&g->upload_done, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&g->results_ready, hipEventDisableTiming));

  const auto t_fill = std::chrono::steady_clock::now();
  // 3. fill descriptors and upload (plane ingest on the copy stream)
  hipStream_t s = ctx->copy_stream, sd = ctx->setup_stream;
  for (uint32_t i = 0; i < n_frames; ++i) {
    const vpcc_frame_desc& F = frames[i];
    const FramePlan& P = g->plans[i];
    const Off& o = offs[i];
    DevFrame& D = g->h_frames[i];
    std::memset(&D, 0, sizeof(D));
    D.patches = (const DevPatch*)(base + o.patches);
    D.vblocks = (const VBlock*)(base + o.vblocks);
    D.block_to_patch = (uint32_t*)(base + o.b2p);
    D.vb_count = (uint32_t*)(base + o.vb_count);
    D.vb_offset = (uint32_t*)(base + o.vb_offset);
    D.out_xyz = (vpcc_point3*)(kb(i, 1, 0) + o.xyz);
    D.out_rgb = F.attribute_count ? (vpcc_color3*)(kb(i, 1, 1) + o.rgb) : nullptr;
    D.out_patch = (gof_flags & VPCC_GOF_WANT_PATCH_INDEX) ? (uint16_t*)(kb(i, 1, 0) + o.pidx) : nullptr;
    D.n_points = g->d_counts + i;
    D.tiles = (TileItem*)(base + o.items);
    D.patch_items = P.tile_eligible ? (const TileItem*)(base + o.patch_items) : nullptr;
    D.n_tiles = 0;                                              // written by k_plan_items
    D.scan_state = g->d_scan + g->scan_off[i];
    D.ticket = reinterpret_cast<uint64_t*>(g->d_tickets + 64 * (size_t)i);
    D.error_flag = g->d_errors + i;
    D.width = F.width; D.height = F.height; D.R = F.occupancy_resolution; D.prec = F.occupancy_precision;
    D.prec_shift = 0;
    while ((1u << D.prec_shift) < D.prec && D.prec_shift < 31) ++D.prec_shift;
    D.bw = P.bw; D.bh = P.bh;
    D.n_patches = (uint32_t)P.patches.size();
    D.n_vblocks = (uint32_t)P.vblocks.size();
    D.map_count = F.map_count; D.absolute_d1 = F.absolute_d1 ? 1u : 0u; D.has_attr = F.attribute_count ? 1u : 0u;
    D.capacity = (uint32_t)cap;
    D.occ_w = F.occupancy.width; D.occ_h = F.occupancy.height;
    if (!own_planes) {
      D.occ = F.occupancy.y; D.occ_stride = F.occupancy.stride;
      for (uint32_t m = 0; m < F.map_count; ++m) {
        D.geo[m] = F.geometry[m].y; D.geo_stride[m] = F.geometry[m].stride;
        if (F.attribute_count) {
          D.attr_y[m] = F.attribute[m].y; D.attr_u[m] = F.attribute[m].u; D.attr_v[m] = F.attribute[m].v;
          D.attr_stride[m] = F.attribute[m].stride; D.attr_cstride[m] = F.attribute[m].cstride;
        }
      }
    } else {
      // One plane of the caller's into its place in the gof's planes block; returns where it lies there.  Tight planes in
      // page-locked memory whose address is a multiple of eight are left to the ingest kernel (the place is moved by
      // src mod 16, so that 16-byte pieces line up on both sides); everything else goes through the copy engines.
      int st = VPCC_OK;
      auto ingest_plane = [&](char* dst, const void* src, size_t elem, uint32_t width, uint32_t height, uint32_t stride) -> const void* {
        if (by_extent) return dst;                              // arrives with its extent (below)
        void* dev_src = nullptr;
        if (pull && stride == width && ((uintptr_t)src & 7u) == 0 &&
            hipHostGetDevicePointer(&dev_src, const_cast<void*>(src), 0) == hipSuccess && dev_src) {
          dst += (uintptr_t)src & 15u;
          const size_t bytes = (size_t)width * height * elem;
          // (the first piece ends on a 16-byte boundary of the source: only a plane's first and last piece have bytes in front
          // of / behind their aligned body — single-byte reads over PCIe)
          for (size_t at = 0; at < bytes;) {
            const size_t len = std::min<size_t>(kIngestPieceBytes - (at ? 0u : ((uintptr_t)dev_src & 15u)), bytes - at);
            g->ingest.push_back(IngestPiece{(const char*)dev_src + at, dst + at, (uint32_t)len, 0u});
            at += len;
          }
          return dst;
        }
        (void)hipGetLastError();
        if (!st) st = copy_plane(ctx, dst, src, elem, width, height, stride, s, dir);
        return dst;
      };
      D.occ = (const uint8_t*)ingest_plane(kb(i, 0, 0) + o.occ, F.occupancy.y, 1, F.occupancy.width, F.occupancy.height, F.occupancy.stride);
      D.occ_stride = F.occupancy.width;
      for (uint32_t m = 0; m < F.map_count; ++m) {
        const vpcc_image_u16& G = F.geometry[m];
        D.geo[m] = (const uint16_t*)ingest_plane(kb(i, 0, 0) + o.geo[m], G.y, 2, G.width, G.height, G.stride);
        D.geo_stride[m] = G.width;
        if (F.attribute_count) {
          const vpcc_image_u16& A = F.attribute[m];
          D.attr_y[m] = (const uint16_t*)ingest_plane(kb(i, 0, 1) + o.ay[m], A.y, 2, A.width, A.height, A.stride);
          // chroma keeps its source stride: the reference indexes it as a flat array
          // (v/2)*(width/2)+(u/2), src/decoder.rs:977, which for odd widths runs across rows
          const size_t ce = chroma_elems(A);
          D.attr_u[m] = (const uint16_t*)ingest_plane(kb(i, 0, 1) + o.au[m], A.u, 2, (uint32_t)ce, 1, (uint32_t)ce);
          D.attr_v[m] = (const uint16_t*)ingest_plane(kb(i, 0, 1) + o.av[m], A.v, 2, (uint32_t)ce, 1, (uint32_t)ce);
          D.attr_stride[m] = A.width; D.attr_cstride[m] = A.cstride;
        }
      }
      if (st) return st;
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
    if (staged) {
      if (!P.patches.empty()) std::memcpy(stage + o.patches, P.patches.data(), sizeof(DevPatch) * P.patches.size());
      if (!P.vblocks.empty()) std::memcpy(stage + o.vblocks, P.vblocks.data(), sizeof(VBlock) * P.vblocks.size());
      if (!P.patch_items.empty()) std::memcpy(stage + o.patch_items, P.patch_items.data(), sizeof(TileItem) * P.patch_items.size());
      continue;
    }
    if (!P.patches.empty())
      HIP_TRY(ctx, hipMemcpyAsync(base + o.patches, P.patches.data(), sizeof(DevPatch) * P.patches.size(),
                                  hipMemcpyHostToDevice, sd));
    if (!P.vblocks.empty())
      HIP_TRY(ctx, hipMemcpyAsync(base + o.vblocks, P.vblocks.data(), sizeof(VBlock) * P.vblocks.size(),
                                  hipMemcpyHostToDevice, sd));
    if (!P.patch_items.empty())
      HIP_TRY(ctx, hipMemcpyAsync(base + o.patch_items, P.patch_items.data(), sizeof(TileItem) * P.patch_items.size(),
                                  hipMemcpyHostToDevice, sd));
  }
  // the tile kernel needs every frame eligible and its vector loads aligned on the final pointers; and its store loop addresses a
  // frame's positions with 32-bit byte offsets (6 bytes per point): frames that may hold more than 715 827 880 points — canvases
  // beyond 18 900 x 18 900 with two maps, unless the caller gives a smaller bound — take the general sequence, whose indices are 64 bits
  // wide (tools/exp_max_canvas.py: a 32768 x 32768 frame of 811 M points, 4.5 GiB of positions, equals the oracle)
  constexpr uint64_t kTilePathMaxPoints = 0xFFFFFFF0ull / sizeof(vpcc_point3);
  bool tiles_ok = all_simple && !(gof_flags & VPCC_GOF_FORCE_GENERAL) && g->capacity <= kTilePathMaxPoints;
  for (uint32_t i = 0; i < n_frames && tiles_ok; ++i) tiles_ok = tile_planes_aligned(g->h_frames[i]);
  g->general = !tiles_ok;
  // Staged: everything of this gof goes onto the copy stream, in the order it is needed — nothing waits for another stream
  // (HIP maps streams onto a few hardware queues: the small copies of a set-up stream sat behind the 40 ms of the previous
  // unit's planes although they were enqueued long before, and the copy engine then idled 10 ms between two units' planes).
  hipStream_t const sm = staged ? s : sd;
  if (staged) {
    std::memcpy(stage + off_frames, g->h_frames.data(), sizeof(DevFrame) * n_frames);
    HIP_TRY(ctx, hipMemcpyAsync(base, stage, host_end, hipMemcpyHostToDevice, s));
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(g->d_frames, g->h_frames.data(), sizeof(DevFrame) * n_frames, hipMemcpyHostToDevice, sd));
  }
  HIP_TRY(ctx, hipMemsetAsync(g->d_counts, 0, sizeof(uint32_t) * n_frames, sm));
  HIP_TRY(ctx, hipMemsetAsync(base + ctrl_begin, 0, g->ctrl_bytes, sm));
  if (g->b2p_words) HIP_TRY(ctx, hipMemsetAsync(g->d_b2p, 0, g->b2p_words * sizeof(uint32_t), sm));
  IngestPiece* d_pieces = (IngestPiece*)(base + off_ingest);
  if (!g->ingest.empty()) {
    if (g->ingest.size() > ingest_bound) return fail(ctx, VPCC_ERR_STATE, "ingest piece list overflow");
    HIP_TRY(ctx, hipMemcpyAsync(d_pieces, g->ingest.data(), sizeof(IngestPiece) * g->ingest.size(), hipMemcpyHostToDevice, sm));
  }
  // the planes follow on the copy stream (plane copies of the fallback path are queued there already; nothing of them
  // depends on the descriptors), behind the set-up: upload_done then stands for both
  if (!staged) {
    HIP_TRY(ctx, hipEventRecord(g->upload_done, sd));
    HIP_TRY(ctx, hipStreamWaitEvent(s, g->upload_done, 0));
  }
  if (!g->ingest.empty()) {
    launch_ingest_planes(d_pieces, (uint32_t)g->ingest.size(), s);
    HIP_TRY(ctx, hipGetLastError());
  }
  g->ingest_extents = 0;
  for (const Extent& e : extents)
    for (const auto& pc : e.pieces) {                          // (one piece, unless the stretch crosses from one page-locked region into the next)
      HIP_TRY(ctx, hipMemcpyAsync((char*)g->block[2 * e.part + 0].ptr + e.dev + (size_t)(pc.first - e.lo), pc.first, pc.second,
                                  hipMemcpyHostToDevice, s));
      ++g->ingest_extents;
    }
  // the tile kernel's work lists, from the occupancy planes where they now lie (src/codec.rs:205-250 on the device)
  if (!g->general) {
    launch_plan_tiles(g->d_frames, 0, n_frames, g->max_vb, s);
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipEventRecord(g->upload_done, s));
  if (getenv("VPCC_RUNTIME_TRACE")) {
    const auto t_end = std::chrono::steady_clock::now();
    fprintf(stderr, "[vpcc] gof of %u frames: allocations %.1f ms, descriptors + %s of the planes %.1f ms (%u extents, %zu pieces by kernel)   (@%.1f - %.1f)\n", n_frames,
            std::chrono::duration<double, std::milli>(t_fill - t_alloc).count(), kind == VPCC_MEM_HOST ? "upload enqueue" : "binding",
            std::chrono::duration<double, std::milli>(t_end - t_fill).count(), g->ingest_extents, g->ingest.size(),
            std::chrono::duration<double, std::milli>(t_alloc.time_since_epoch()).count(), std::chrono::duration<double, std::milli>(t_end.time_since_epoch()).count());
    for (int k = 0; k < 2 * vpcc_ctx::kParts; ++k)
      if (g->block[k].ptr)
        fprintf(stderr, "[vpcc]   %s block of part %d: %p + %.2f GB%s\n", (k & 1) ? "output" : "planes", k / 2, g->block[k].ptr,
                g->block[k].bytes / 1073741824.0, g->block[k].pooled ? " (pool)" : "");
  }
  // descriptor staging (plans, h_frames) lives in the gof; the caller's planes must outlive the copies,
  // so creation is synchronous unless the caller asked for overlapping ingest
  if (!(gof_flags & VPCC_GOF_ASYNC_UPLOAD) || kind != VPCC_MEM_HOST) HIP_TRY(ctx, hipStreamSynchronize(s));
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

namespace {

struct Timer {
  vpcc_gof* g;
  hipStream_t s;
  bool on;
  LaunchTimings* slot = nullptr;
  // new_launch: a vpcc_gof_reconstruct opens the next ring slot; vpcc_gof_smooth appends to the current one
  Timer(vpcc_gof* g_, hipStream_t s_, bool new_launch) : g(g_), s(s_), on((g_->flags & VPCC_GOF_PROFILE) != 0) {
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
  void begin(const char* name) {
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
  void end() {
    if (!on) return;
    (void)hipEventRecord(slot->k[slot->n].stop, s);
    slot->n++;
  }
};

}  // namespace

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
    // single-pass tile kernel: ONE kernel, nothing to prepare
    uint32_t max_groups = 0;
    for (uint32_t i = first; i < first + count; ++i)
      max_groups = std::max(max_groups, (uint32_t)((g->plans[i].tile_bound + kTileItemsPerGroup - 1) / kTileItemsPerGroup));
    // Nothing to clear: look-back words and ticket counters carry the launch generation (a counter of an earlier
    // launch is reset by the first workgroup that draws from it), and a frame's point count is rewritten by its last group
    // (a frame without tiles keeps the zero written at creation).
    if (!g->tile_map_valid || g->tile_map_first != first || g->tile_map_count != count) {
      std::vector<uint32_t> tiles(count);
      for (uint32_t i = 0; i < count; ++i) tiles[i] = g->plans[first + i].tile_bound;   // (the exact counts are on the device: k_plan_items)
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

  // general sequence: owner -> count -> scan -> emit
  size_t b2p_first = 0, b2p_len = 0;
  for (uint32_t i = 0; i < first + count; ++i) {
    const size_t n = (size_t)g->plans[i].bw * g->plans[i].bh;
    if (i < first) b2p_first += n; else b2p_len += n;
  }
  if (b2p_len) HIP_TRY(ctx, hipMemsetAsync(g->d_b2p + b2p_first, 0, b2p_len * sizeof(uint32_t), s));
  uint32_t max_vb = 0;
  for (uint32_t i = first; i < first + count; ++i) max_vb = std::max(max_vb, (uint32_t)g->plans[i].vblocks.size());
  T.begin("k_block_owner");
  launch_block_owner(g->d_frames, first, count, max_vb, s);
  T.end();
  T.begin("k_count");
  launch_count(g->d_frames, first, count, max_vb, s);
  T.end();
  T.begin("k_scan");
  launch_scan(g->d_frames, first, count, s);
  T.end();
  T.begin("k_emit");
  launch_emit(g->d_frames, first, count, max_vb, s);
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
  HIP_TRY(ctx, hipStreamWaitEvent(s, g->upload_done, 0));              // the planning kernels of vpcc_gof_create
  if (g->launched) HIP_TRY(ctx, hipStreamWaitEvent(s, g->results_ready, 0));   // (general sequence: written by every launch)
  const size_t n = (size_t)g->plans[frame].bw * g->plans[frame].bh;
  if (block_to_patch_out && n)
    HIP_TRY(ctx, hipMemcpyAsync(block_to_patch_out, g->h_frames[frame].block_to_patch, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  DevFrame d{};
  if (work_items_out) HIP_TRY(ctx, hipMemcpyAsync(&d, g->d_frames + frame, sizeof d, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  if (work_items_out) *work_items_out = g->general ? 0u : d.n_tiles;
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
  *bytes_out = g->plans[frame].plane_bytes + 9ull * g->h_counts[frame];
  return VPCC_OK;
}

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
      HIP_TRY(ctx, hipMalloc(&g->smooth_keys, (sizeof(uint32_t) * (sg.list_stride + 2 * sg.count_stride) + sizeof(uint64_t) * sg.painted_stride) * g->n_frames));
    sg.painted_base = (uint64_t*)g->smooth_keys;
    sg.list_base = (uint32_t*)(sg.painted_base + sg.painted_stride * g->n_frames);
    sg.count_base = sg.list_base + sg.list_stride * g->n_frames;
    sg.flag_base = sg.count_base + sg.count_stride * g->n_frames;
    if (both) {
      sg.moved_stride = align_up((g->capacity + 63) / 64 + 4, 32);
      sg.oldkey_stride = align_up(g->capacity, 4);
      if (!g->smooth_moved)
        HIP_TRY(ctx, hipMalloc(&g->smooth_moved, (sizeof(uint64_t) * 3 * sg.moved_stride + sizeof(uint32_t) * sg.oldkey_stride) * g->n_frames));
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

// ------------------------------------------------ one-shot seam replacements
namespace {
struct OneShot {
  vpcc_gof* g = nullptr;
  ~OneShot() { vpcc_gof_destroy(g); }
};
}  // namespace

extern "C" int vpcc_generate_block_to_patch(vpcc_ctx* ctx, const vpcc_frame_desc* frame, vpcc_memory_kind planes,
                                            uint32_t* block_to_patch_out) {
  if (!ctx || !frame || !block_to_patch_out) return VPCC_ERR_INVALID_ARG;
  OneShot o;
  int st = vpcc_gof_create(ctx, frame, 1, planes, 1, VPCC_GOF_FORCE_GENERAL, &o.g);
  if (st) return st;
  vpcc_gof* g = o.g;
  hipStream_t s = ctx->stream;
  const size_t n = (size_t)g->plans[0].bw * g->plans[0].bh;
  if (n == 0) return VPCC_OK;
  HIP_TRY(ctx, hipMemsetAsync(g->d_b2p, 0, n * sizeof(uint32_t), s));
  launch_block_owner(g->d_frames, 0, 1, (uint32_t)g->plans[0].vblocks.size(), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(block_to_patch_out, g->d_b2p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return VPCC_OK;
}

extern "C" int vpcc_upsample_occupancy(vpcc_ctx* ctx, const vpcc_frame_desc* frame, vpcc_memory_kind planes,
                                       uint8_t* occupancy_map_out) {
  if (!ctx || !frame || !occupancy_map_out) return VPCC_ERR_INVALID_ARG;
  OneShot o;
  int st = vpcc_gof_create(ctx, frame, 1, planes, 1, VPCC_GOF_FORCE_GENERAL, &o.g);
  if (st) return st;
  hipStream_t s = ctx->stream;
  const size_t n = (size_t)frame->width * frame->height;
  uint8_t* d_out = nullptr;
  HIP_TRY(ctx, hipMalloc((void**)&d_out, n));
  launch_upsample_occupancy(o.g->d_frames, 0, d_out, frame->width, frame->height, s);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(occupancy_map_out, d_out, n, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(d_out);
  if (e != hipSuccess) return fail(ctx, VPCC_ERR_DEVICE, hipGetErrorString(e));
  return VPCC_OK;
}

extern "C" int vpcc_reconstruct_frame(vpcc_ctx* ctx, const vpcc_frame_desc* frame, vpcc_memory_kind planes,
                                      vpcc_point3* xyz_out, vpcc_color3* rgb_out, uint16_t* patch_index_out,
                                      size_t capacity, size_t* n_points) {
  if (!ctx || !frame || !n_points) return VPCC_ERR_INVALID_ARG;
  OneShot o;
  const uint64_t cap = std::min<uint64_t>(capacity ? capacity : 1, std::max<uint64_t>(vpcc_frame_capacity_bound(frame), 1));
  int st = vpcc_gof_create(ctx, frame, 1, planes, cap, patch_index_out ? VPCC_GOF_WANT_PATCH_INDEX : 0u, &o.g);
  if (st) return st;
  st = vpcc_gof_reconstruct(o.g, 0, 1, nullptr);
  if (st) return st;
  return vpcc_gof_download(o.g, 0, xyz_out, rgb_out, patch_index_out, capacity, n_points);
}
