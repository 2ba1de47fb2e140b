// micro-benchmark: how fast does a KERNEL pull page-locked host memory over PCIe into HBM (zero-copy reads), against the
// copy engines (hipMemcpyAsync)?  Decides how the Decoder ingests planes: 1 280 hipMemcpyAsync per 128-frame unit (the
// enqueue alone blocks the lane's thread for 46-57 ms) or one kernel.
//   zero_copy [registered]     registered: malloc + hipHostRegister(Portable | Mapped) instead of hipHostMalloc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <sys/mman.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define G __attribute__((address_space(1)))
__global__ __launch_bounds__(256) void k_pull(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, int unroll) {
  // every workgroup walks its own contiguous stretch, 4 KB per step and wave: plain coalesced 16-B loads
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t b = (size_t)blockIdx.x * per, e = b + per < n16 ? b + per : n16;
  for (size_t i = b + threadIdx.x; i < e; i += 256 * 4) {
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * 256 < e) v[k] = __builtin_nontemporal_load((const G u32x4*)(src + i + k * 256));
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * 256 < e) __builtin_nontemporal_store(v[k], (G u32x4*)(dst + i + k * 256));
  }
}
int main(int argc, char** argv) {
  const size_t bytes = 1ull << 30;
  const bool reg = argc > 1 && !strcmp(argv[1], "registered");
  void* h = nullptr; void* hd = nullptr; void* d = nullptr;
  if (reg) {
    h = aligned_alloc(2 << 20, bytes);
    if (argc > 2 && !strcmp(argv[2], "nohuge")) madvise(h, bytes, MADV_NOHUGEPAGE);      // 4-KB pages
    if (argc > 2 && !strcmp(argv[2], "huge")) madvise(h, bytes, MADV_HUGEPAGE);
    memset(h, 1, bytes);
    if (hipHostRegister(h, bytes, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) { printf("register failed\n"); return 1; }
  } else {
    if (hipHostMalloc(&h, bytes, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { printf("hostmalloc failed\n"); return 1; }
    memset(h, 1, bytes);
  }
  if (hipHostGetDevicePointer(&hd, h, 0) != hipSuccess) { printf("no device pointer\n"); return 1; }
  hipMalloc(&d, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float ms;
  hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, 0);
  hipEventRecord(a); for (int r = 0; r < 3; ++r) hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, 0); hipEventRecord(b); hipEventSynchronize(b);
  hipEventElapsedTime(&ms, a, b); printf("%s: hipMemcpyAsync 1 GiB            %.1f GB/s\n", reg ? "registered" : "hipHostMalloc", 3 * bytes / ms / 1e6);
  // 1 280 copies of 0.8 MB (a 128-frame unit's planes): enqueue time and rate
  { const size_t piece = bytes / 1280 & ~255ull; auto t0 = std::chrono::steady_clock::now();
    hipEventRecord(a); for (int k = 0; k < 1280; ++k) hipMemcpyAsync((char*)d + k * piece, (char*)h + k * piece, piece, hipMemcpyHostToDevice, 0); hipEventRecord(b);
    double enq = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    printf("%s: 1 280 hipMemcpyAsync of %.2f MB   %.1f GB/s, enqueue %.1f ms\n", reg ? "registered" : "hipHostMalloc", piece / 1e6, 1280 * piece / ms / 1e6, enq); }
  for (int wgs : {64, 128, 256, 512, 1024, 2048}) {
    hipLaunchKernelGGL(k_pull, dim3(wgs), dim3(256), 0, 0, (const u32x4*)hd, (u32x4*)d, bytes / 16, 4);
    hipEventRecord(a); for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_pull, dim3(wgs), dim3(256), 0, 0, (const u32x4*)hd, (u32x4*)d, bytes / 16, 4); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); printf("%s: kernel pull, %4d workgroups      %.1f GB/s\n", reg ? "registered" : "hipHostMalloc", wgs, 3 * bytes / ms / 1e6);
  }
  // duplex: the kernel pulls 1 GiB from the host while the copy engine (or a second kernel) sends 0.4 GiB back, in 7-MB copies
  {
    void* h2 = nullptr; void* d2 = nullptr; void* h2d = nullptr;
    hipHostMalloc(&h2, bytes, hipHostMallocPortable | hipHostMallocMapped); hipMalloc(&d2, bytes); hipHostGetDevicePointer(&h2d, h2, 0);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t a2, b2; hipEventCreate(&a2); hipEventCreate(&b2);
    const size_t piece = 7ull << 20, back = 58;     // 58 x 7 MB = 0.4 GiB
    for (int mode = 0; mode < 2; ++mode)
      for (int wgs : {16, 32, 64, 128}) {
        hipDeviceSynchronize();
        hipEventRecord(a, s1); for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k_pull, dim3(wgs), dim3(256), 0, s1, (const u32x4*)hd, (u32x4*)d, bytes / 16, 4); hipEventRecord(b, s1);
        hipEventRecord(a2, s2);
        for (int r = 0; r < 2; ++r)
          for (size_t k = 0; k < back; ++k) {
            if (mode == 0) hipMemcpyAsync((char*)h2 + k * piece, (char*)d2 + k * piece, piece, hipMemcpyDeviceToHost, s2);
            else hipLaunchKernelGGL(k_pull, dim3(16), dim3(256), 0, s2, (const u32x4*)((char*)d2 + k * piece), (u32x4*)((char*)h2d + k * piece), piece / 16, 4);
          }
        hipEventRecord(b2, s2);
        hipEventSynchronize(b); hipEventSynchronize(b2);
        float m1, m2; hipEventElapsedTime(&m1, a, b); hipEventElapsedTime(&m2, a2, b2);
        printf("duplex, pull with %3d workgroups + %s back: H2D %.1f GB/s (%.1f ms), D2H %.1f GB/s (%.1f ms)\n", wgs,
               mode ? "16-workgroup push kernels" : "hipMemcpyAsync D2H      ", 2 * bytes / m1 / 1e6, m1, 2 * back * piece / m2 / 1e6, m2);
      }
  }
  // duplex with the COPY ENGINE pulling: 2 x 7 copies of 146 MB (the planes of eight frames, contiguous in the container) host -> device
  // while 0.8 GiB of results go back in 7-MB copies (copy engine) or by push kernels
  {
    void* h2 = nullptr; void* d2 = nullptr; void* h2d = nullptr;
    hipHostMalloc(&h2, bytes, hipHostMallocPortable | hipHostMallocMapped); hipMalloc(&d2, bytes); hipHostGetDevicePointer(&h2d, h2, 0);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t a2, b2; hipEventCreate(&a2); hipEventCreate(&b2);
    const size_t piece = 7ull << 20, back = 58, big = 146ull << 20, nbig = bytes / big;
    for (int mode = 0; mode < 3; ++mode) {
      hipDeviceSynchronize();
      hipEventRecord(a, s1);
      for (int r = 0; r < 2; ++r) for (size_t k = 0; k < nbig; ++k) hipMemcpyAsync((char*)d + k * big, (char*)h + k * big, big, hipMemcpyHostToDevice, s1);
      hipEventRecord(b, s1);
      hipEventRecord(a2, s2);
      if (mode < 2)
        for (int r = 0; r < 4; ++r)
          for (size_t k = 0; k < back; ++k) {
            if (mode == 0) hipMemcpyAsync((char*)h2 + k * piece, (char*)d2 + k * piece, piece, hipMemcpyDeviceToHost, s2);
            else hipLaunchKernelGGL(k_pull, dim3(16), dim3(256), 0, s2, (const u32x4*)((char*)d2 + k * piece), (u32x4*)((char*)h2d + k * piece), piece / 16, 4);
          }
      hipEventRecord(b2, s2);
      hipEventSynchronize(b); hipEventSynchronize(b2);
      float m1, m2; hipEventElapsedTime(&m1, a, b); hipEventElapsedTime(&m2, a2, b2);
      printf("duplex, %zu copies of 146 MB host -> device + %s back: H2D %.1f GB/s (%.1f ms), D2H %.1f GB/s (%.1f ms)\n", 2 * nbig,
             mode == 0 ? "hipMemcpyAsync D2H (7 MB)" : mode == 1 ? "16-workgroup push kernels" : "nothing", 2 * nbig * big / m1 / 1e6, m1,
             mode < 2 ? 4 * back * piece / m2 / 1e6 : 0.0, m2);
    }
  }
  return 0;
}