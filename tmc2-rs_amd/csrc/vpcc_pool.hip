// vpcc_pool.hip — the context's pool ("two homes"): vpcc_ctx_reserve and the big blocks of a gof.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "vpcc_runtime.hpp"

using namespace vpcc;

// ------------------------------------------------------------ pool ("two homes")
namespace {

// Pools of destroyed contexts stay with the process, by device: memory given back to the driver is wiped before it is
// handed out again (about 40 GB/s), and every allocation of the process waits for that — a second Decoder opened right
// behind the first would spend seconds in hipMalloc.  The next vpcc_ctx_reserve on the device takes a kept pool over as
// it is, classification included.
std::mutex g_kept_pools_mutex;
std::vector<std::pair<int, vpcc_ctx::Pool>> g_kept_pools;      // (device, pool); never freed: the process's memory

}  // namespace

void vpcc::retire_pool(vpcc_ctx* ctx) {
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
bool vpcc::release_kept_pools(int device) {
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

extern "C" int vpcc_release_kept_pools(int device) { return release_kept_pools(device) ? 1 : 0; }

// hipMalloc that makes room before it gives up: the context's cached blocks of destroyed gofs, then the pools kept for the
// device's next context.  (A failed call's error stays "last" until somebody asks: asked here.)
hipError_t vpcc::device_malloc(vpcc_ctx* ctx, void** out, size_t bytes) {
  hipError_t e = hipMalloc(out, bytes);
  if (e == hipSuccess) return e;
  (void)hipGetLastError();
  bool freed = false;
  {
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    for (auto& b : ctx->block_cache) { (void)hipFree(b.ptr); freed = true; }
    ctx->block_cache.clear();
  }
  if (freed && (e = hipMalloc(out, bytes)) == hipSuccess) return e;
  (void)hipGetLastError();
  if (release_kept_pools(ctx->device) && (e = hipMalloc(out, bytes)) == hipSuccess) return e;
  (void)hipGetLastError();
  *out = nullptr;
  return e;
}

// Blocks of destroyed gofs that did not come from the pool are kept for the next gof of the same size: hipMalloc /
// hipFree cost milliseconds and hipFree synchronises the whole device, which would stall a GOF pipeline.
namespace { constexpr size_t kBlockCacheEntries = 8; }

// A block of `bytes` for part `part` of a gof: from the pool's home `part` (then from the other home), else an
// allocation of its own (from the context's cache of such blocks when one fits).
int vpcc::acquire_block(vpcc_ctx* ctx, int part, size_t bytes, vpcc_ctx::Block* out) {
  bytes = (bytes + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
  {
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
  }
  // (outside the pool mutex: device_malloc takes it, and the mutex of the kept pools, when it has to make room)
  void* p = nullptr;
  if (device_malloc(ctx, &p, bytes) != hipSuccess) {
    ctx->last_error = "no device memory for a gof block of " + std::to_string(bytes >> 20) + " MB";
    return VPCC_ERR_DEVICE;
  }
  *out = vpcc_ctx::Block{p, bytes, false, 0};
  return VPCC_OK;
}

void vpcc::release_block(vpcc_ctx* ctx, vpcc_ctx::Block& B) {      // all work on the block is complete
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

extern "C" int vpcc_ctx_pool_alloc(vpcc_ctx* ctx, int home, size_t bytes, void** out) {
  if (!ctx || !out || !bytes || home < 0) return VPCC_ERR_INVALID_ARG;
  *out = nullptr;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  vpcc_ctx::Block B;
  const int st = acquire_block(ctx, home, bytes, &B);
  if (st) return st;
  std::lock_guard<std::mutex> lock(ctx->pool_mutex);
  ctx->lent.push_back(B);
  *out = B.ptr;
  return VPCC_OK;
}

extern "C" int vpcc_ctx_pool_free(vpcc_ctx* ctx, void* ptr) {
  if (!ctx || !ptr) return VPCC_ERR_INVALID_ARG;
  vpcc_ctx::Block B;
  {
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    size_t k = 0;
    while (k < ctx->lent.size() && ctx->lent[k].ptr != ptr) ++k;
    if (k == ctx->lent.size()) return fail(ctx, VPCC_ERR_INVALID_ARG, "not a pointer of vpcc_ctx_pool_alloc");
    B = ctx->lent[k];
    ctx->lent.erase(ctx->lent.begin() + (long)k);
  }
  release_block(ctx, B);
  return VPCC_OK;
}

extern "C" int vpcc_ctx_reserve(vpcc_ctx* ctx, uint64_t bytes, vpcc_pool_info* out) {
  return vpcc_ctx_reserve_within(ctx, bytes, 0.f, out);
}

extern "C" int vpcc_ctx_reserve_within(vpcc_ctx* ctx, uint64_t bytes, float budget_ms, vpcc_pool_info* out) {
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
    // a pool an earlier context of this process left behind on the device, if it is big enough (and not three times as big);
    // kept pools of another size go back to the driver instead of lying beside the new one.  The kept pool leaves the list
    // under the list's mutex and enters the context under the context's: never both (acquire_block takes them in the other
    // order when it has to make room).
    vpcc_ctx::Pool taken;
    bool have = false;
    {
      std::lock_guard<std::mutex> keep(g_kept_pools_mutex);
      for (size_t k = g_kept_pools.size(); k-- > 0;) {
        if (g_kept_pools[k].first != ctx->device) continue;
        vpcc_ctx::Pool& K = g_kept_pools[k].second;
        const bool fits = K.info.bytes >= n * G && K.info.bytes <= 3 * n * G;
        if (fits && have) continue;                                  // (another lane of the same device may want it)
        if (fits) { taken = std::move(K); have = true; }
        else for (void* q : K.slabs) (void)hipFree(q);
        g_kept_pools.erase(g_kept_pools.begin() + (long)k);
      }
    }
    if (have) {
      taken.info.ms_spent = 0.f;
      taken.info.reused = 1;
      taken.info.other_home = taken.info.fallbacks = 0;
      if (out) *out = taken.info;
      if (trace) fprintf(stderr, "[vpcc] pool: took over the %llu-GiB pool an earlier context left on device %d\n",
                         (unsigned long long)(taken.info.bytes >> 30), ctx->device);
      std::lock_guard<std::mutex> lock(ctx->pool_mutex);
      ctx->pool = std::move(taken);
      return VPCC_OK;
    }
  }
  vpcc_ctx::Pool P;
  void* base = nullptr;
  if (device_malloc(ctx, &base, n * G) != hipSuccess) return VPCC_ERR_DEVICE;
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
  auto spent_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  const bool in_budget = budget_ms <= 0.f || spent_ms() < budget_ms;
  if (P.info.bytes_of_kind[small] * 4 < n * G && !in_budget && trace)
    fprintf(stderr, "[vpcc] pool: %.0f ms spent of %.0f: no search for the other kind, the homes are one\n", spent_ms(), budget_ms);
  if (P.info.bytes_of_kind[small] * 4 < n * G && in_budget) {
    std::vector<void*> spare;
    const size_t m = std::max<size_t>(2, n / 2);
    for (int attempt = 0; attempt < 4 && (budget_ms <= 0.f || spent_ms() < budget_ms); ++attempt) {
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
