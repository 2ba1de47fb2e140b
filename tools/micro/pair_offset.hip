// micro-benchmark: does the rate of the tile kernel's OUTPUT pattern (4 096 waves writing 1.8-KB runs of positions and
// 0.9-KB runs of colours) depend on where the two arrays lie RELATIVE to each other, or on where each lies?  Both arrays
// are carved out of ONE 48-GB allocation: xyz at base + a, rgb at base + b.
//   sweep 1: a = 0, b = 1 .. 40 GB in 1-GB steps          (relative offset, coarse)
//   sweep 2: a = 0, b = 1 GB + 0 .. 64 MB in 4-MB steps    (relative offset, fine)
//   sweep 3: a = k GB, b = a + 1 GB, k = 0 .. 40           (both move together)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
#define G __attribute__((address_space(1)))
__global__ __launch_bounds__(256) void k(unsigned char* xyz, unsigned char* rgb, int items) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int groups = (items + 15) / 16;
  for (int g = blockIdx.x; g < groups; g += gridDim.x)
    for (int t = 0; t < 4; ++t) {
      const int i = g * 16 + t * 4 + wave;
      if (i >= items) continue;
      unsigned char* px = xyz + (size_t)i * 1824;
      unsigned char* pc = rgb + (size_t)i * 912;
      const int shift = (int)(((size_t)i * 304) & 127);
      for (int kk = 2 * lane; kk < 304 + shift; kk += 128) {
        const int k = kk - shift;
        if (k < 0 || k + 1 >= 304) continue;
        u32x3 v = {(unsigned)k, (unsigned)lane, (unsigned)i};
        __builtin_nontemporal_store(v, (G u32x3*)(px + k * 6));
        if (!(lane & 1) && k + 3 < 304) __builtin_nontemporal_store(v, (G u32x3*)(pc + k * 3));
      }
    }
}
static float run(unsigned char* xyz, unsigned char* rgb, int items, hipEvent_t a, hipEvent_t b) {
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, 1024, 256, 0, 0, xyz, rgb, items);
  hipEventRecord(a);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k, 1024, 256, 0, 0, xyz, rgb, items);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return (float)((double)items * 2736 / (ms / 5) / 1e6);
}
int main() {
  const int items = 333000;
  const size_t GB = 1ull << 30;
  unsigned char* base;
  if (hipMalloc(&base, 48 * GB) != hipSuccess) { printf("no 48 GB\n"); return 1; }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  printf("sweep 1 (xyz at 0, rgb at b GB):");
  for (int g = 1; g <= 40; ++g) printf(" %.0f", run(base, base + g * GB, items, a, b));
  printf("\nsweep 2 (xyz at 0, rgb at 1 GB + m MB, m = 0, 4, .. 64):");
  for (int m = 0; m <= 64; m += 4) printf(" %.0f", run(base, base + GB + ((size_t)m << 20), items, a, b));
  printf("\nsweep 3 (xyz at k GB, rgb 1 GB behind it):");
  for (int g = 0; g <= 40; ++g) printf(" %.0f", run(base + g * GB, base + (g + 1) * GB, items, a, b));
  printf("\nsweep 4 (xyz alone at k GB; rgb always at 47 GB):");
  for (int g = 0; g <= 40; ++g) printf(" %.0f", run(base + g * GB, base + 47 * GB, items, a, b));
  printf("\n");
  return 0;
}
