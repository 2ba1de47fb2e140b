// micro-benchmark: a map of write bandwidth over ONE large allocation, slice by slice (whole-line 16-B stores from
// 1024 workgroups, each slice written 4 times): at which granularity do fast and slow regions alternate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void kw(uint4* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) { uint4 v = {(unsigned)i, 1u, 2u, 3u}; dst[i] = v; }
}
int main(int argc, char** argv) {
  const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 48) << 30, slice = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 20;
  unsigned char* p = nullptr;
  if (hipMalloc(&p, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  printf("slices of %zu MB over %zu GB (GB/s): ", slice >> 20, total >> 30);
  for (size_t off = 0; off + slice <= total; off += slice) {
    hipLaunchKernelGGL(kw, 1024, 256, 0, 0, (uint4*)(p + off), slice / 16);
    hipEventRecord(a);
    for (int w = 0; w < 4; ++w) hipLaunchKernelGGL(kw, 1024, 256, 0, 0, (uint4*)(p + off), slice / 16);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 4;
    printf("%.0f ", slice / ms / 1e6);
  }
  printf("\n");
  return 0;
}
