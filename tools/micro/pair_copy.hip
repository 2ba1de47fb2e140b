// micro-benchmark: does a plain COPY care where its source and its destination lie relative to each other?
// (pair_offset.hip: two concurrently WRITTEN arrays run at 3.85 TB/s inside one 32-GB chunk and 5.2 TB/s across two.)
// One 48-GB allocation; 1 GiB copied from base + a to base + b by a grid-stride kernel (16 B per lane and trip).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define G __attribute__((address_space(1)))
__global__ __launch_bounds__(256) void copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull)
    __builtin_nontemporal_store(((const G u32x4*)src)[i], (G u32x4*)dst + i);
}
static float rate(unsigned char* s, unsigned char* d, size_t bytes, hipEvent_t a, hipEvent_t b) {
  hipLaunchKernelGGL(copy, 4096, 256, 0, 0, (const u32x4*)s, (u32x4*)d, bytes / 16);
  hipEventRecord(a);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(copy, 4096, 256, 0, 0, (const u32x4*)s, (u32x4*)d, bytes / 16);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return (float)(4.0 * 2.0 * bytes / (ms * 1e6));
}
int main() {
  const size_t GB = 1ull << 30;
  unsigned char* base;
  if (hipMalloc(&base, 48 * GB) != hipSuccess) { printf("no 48 GB\n"); return 1; }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  printf("copy of 1 GiB, source at 0, destination at b GB (GB/s read + written):");
  for (int g = 1; g <= 44; ++g) printf(" %.0f", rate(base, base + g * GB, GB, a, b));
  printf("\nsource at 40 GB, destination at b GB:");
  for (int g = 0; g <= 38; ++g) printf(" %.0f", rate(base + 40 * GB, base + g * GB, GB, a, b));
  printf("\n");
  return 0;
}
