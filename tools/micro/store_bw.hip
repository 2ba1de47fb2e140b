// micro-benchmark: throughput of byte-misaligned wide global stores on gfx950.
// Each lane writes a contiguous piece (12 B as dwordx3, or 16 B as dwordx4) at piece stride,
// the whole stream shifted by `off` bytes from a 256-B aligned base.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct __attribute__((packed)) P3 { unsigned a, b, c; };
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned char* base, size_t npieces, int off) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < npieces; i += stride) {
    if (MODE == 3) {
      P3 v{(unsigned)i, 1u, 2u};
      __builtin_memcpy(base + off + 12 * i, &v, 12);
    } else if (MODE == 4) {
      uint4 v = make_uint4((unsigned)i, 1u, 2u, 3u);
      __builtin_memcpy(base + off + 16 * i, &v, 16);
    } else if (MODE == 2) {   // 6 B as three shorts (what the first kernels did)
      unsigned short* p = (unsigned short*)(base + (off & ~1) + 6 * i);
      p[0] = (unsigned short)i; p[1] = 1; p[2] = 2;
    } else {                  // 3 B as bytes
      unsigned char* p = base + off + 3 * i;
      p[0] = (unsigned char)i; p[1] = 1; p[2] = 2;
    }
  }
}
template <int MODE> void run(unsigned char* d, size_t bytes, int piece, int off) {
  size_t np = (bytes - 64) / piece;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, 4096, 256, 0, 0, d, np, off);
  hipEventRecord(a);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<MODE>, 4096, 256, 0, 0, d, np, off);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("mode %d piece %2d off %2d : %7.1f GB/s\n", MODE, piece, off, 5.0 * np * piece / (ms * 1e-3) / 1e9);
}
int main() {
  size_t bytes = 1ull << 30; unsigned char* d; hipMalloc(&d, bytes);
  for (int off : {0, 2, 4, 6, 8, 14}) run<3>(d, bytes, 12, off);
  for (int off : {1, 3, 5}) run<3>(d, bytes, 12, off);
  for (int off : {0, 2, 6, 8, 1}) run<4>(d, bytes, 16, off);
  run<2>(d, bytes, 6, 0); run<1>(d, bytes, 3, 0);
  return 0;
}
