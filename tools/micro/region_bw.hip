// micro-benchmark: does streaming bandwidth depend on WHERE in VRAM a buffer lies?  Allocates N buffers of 4 GiB one
// after the other (kept alive, so that they cover consecutive regions) and copies the first half of each to its
// second half with 16-B accesses; prints GB/s (read + written) per buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
// reads `n` uint4 `reps` times (a window that fits the 256 MB Infinity Cache): the re-read rate of the region
__global__ __launch_bounds__(256) void kr(const uint4* __restrict__ src, unsigned* __restrict__ out, size_t n, int reps) {
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) { const uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  }
  if (acc == 0x12345u) out[0] = acc;
}
// one 16-B read per 4 KB page, pages visited in a scattered order: bound by address translation, not by bandwidth
__global__ __launch_bounds__(256) void kp(const unsigned char* __restrict__ src, unsigned* __restrict__ out, size_t pages) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  unsigned acc = 0;
  for (; i < pages; i += stride) {
    const size_t pg = (i * 2654435761ull) % pages;
    const uint4 v = *(const uint4*)(src + pg * 4096 + (i & 255) * 16);
    acc += v.x ^ v.y;
  }
  if (acc == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) dst[i] = src[i];
}
int main(int argc, char** argv) {
  const int nbuf = argc > 1 ? atoi(argv[1]) : 24;
  const size_t bytes = 4ull << 30, half = bytes / 2, n = half / 16;
  std::vector<unsigned char*> bufs;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < nbuf; ++i) {
    unsigned char* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) break;
    bufs.push_back(p);
    hipMemset(p, 1, half);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, 8192, 256, 0, 0, (const uint4*)p, (uint4*)(p + half), n);
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k, 8192, 256, 0, 0, (const uint4*)p, (uint4*)(p + half), n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const size_t win = (128ull << 20) / 16;            // a 128 MB window, read 8 times per launch
    hipLaunchKernelGGL(kr, 2048, 256, 0, 0, (const uint4*)p, (unsigned*)(p + half), win, 8);
    hipEventRecord(a);
    hipLaunchKernelGGL(kr, 2048, 256, 0, 0, (const uint4*)p, (unsigned*)(p + half), win, 8);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms2; hipEventElapsedTime(&ms2, a, b);
    const size_t pages = half / 4096;
    hipLaunchKernelGGL(kp, 2048, 256, 0, 0, p, (unsigned*)(p + half), pages);
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(kp, 2048, 256, 0, 0, p, (unsigned*)(p + half), pages);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms3; hipEventElapsedTime(&ms3, a, b); ms3 /= 5;
    printf("   scattered page reads %.1f G pages/s  |  ", pages / ms3 / 1e6);
    printf("buffer %2d at +%3zu GiB (%p): copy %.0f GB/s   re-read of a 128 MB window %.0f GB/s\n", i, (size_t)i * 4, (void*)p,
           2.0 * half / ms / 1e6, 8.0 * (128ull << 20) / ms2 / 1e6);
  }
  return 0;
}
