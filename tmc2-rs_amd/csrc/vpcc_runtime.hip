// vpcc_runtime.hip — contexts, page-locked host memory and the one-shot seam replacements of the C ABI
// (include/vpcc_recon.h).  The other files of the host runtime: vpcc_runtime.hpp.
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>

#include "vpcc_runtime.hpp"

using namespace vpcc;

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
      hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return VPCC_ERR_DEVICE;
  }
  // the code objects of the per-frame path are loaded now (the runtime loads a translation unit's at its first launch:
  // 8 ms in front of a cold process's first gof), asynchronously on the copy stream
  launch_warm_kernels(ctx->copy_stream);
  launch_warm_tiles(ctx->stream);
  (void)hipGetLastError();
  *out = ctx;
  return VPCC_OK;
}

extern "C" void vpcc_ctx_destroy(vpcc_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (auto& a : ctx->arena_cache) (void)hipFree(a.first);
  for (auto& a : ctx->stage_cache) (void)hipHostFree(a.first);
  for (auto& b : ctx->lent) release_block(ctx, b);          // (what a producer still holds goes with the context)
  ctx->lent.clear();
  retire_pool(ctx);
  for (auto& b : ctx->block_cache) (void)hipFree(b.ptr);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
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
}  // namespace
// [lo, lo + bytes) as pieces that each lie inside one page-locked region (an input page-locked chunk by chunk: a stretch of
// planes may cross from one chunk into the next); false if some byte of it is in none.
bool vpcc::pinned_pieces(const char* lo, size_t bytes, std::vector<std::pair<const char*, size_t>>* pieces) {
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
  const size_t n = (size_t)g->shapes[0].bw * g->shapes[0].bh;
  if (n == 0) return VPCC_OK;
  HIP_TRY(ctx, hipMemsetAsync(g->d_b2p, 0, n * sizeof(uint32_t), s));
  launch_block_owner(g->d_frames, 0, 1, g->shapes[0].n_vblocks, 0xFFFFFFFFu, s);
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
  HIP_TRY(ctx, device_malloc(ctx, (void**)&d_out, n));
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
