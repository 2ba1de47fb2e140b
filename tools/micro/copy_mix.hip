// micro-benchmark: what a pure streaming kernel reaches on MI355X with the tile kernel's memory-side traffic mix
// (S-longdress x 32: ~290 MB of 128-byte line reads and ~260 MB of writes per launch).  Reads `rd_mb` MB with
// 16-byte loads, writes the first `wr_mb` MB of it back (plain or non-temporal stores); no arithmetic, perfectly
// coalesced, every byte used.  The ratio bytes / time is the practical ceiling for that mix.
// usage: copy_mix [rd_mb=290] [wr_mb=260] [reps=50]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const u4* __restrict__ src, u4* __restrict__ dst, size_t n_rd, size_t n_wr,
                                              unsigned* sink) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  unsigned acc = 0;
  for (size_t i0 = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i0 < n_rd; i0 += stride) {
    u4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = i0 + 256u * j < n_rd ? src[i0 + 256u * j] : u4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t i = i0 + 256u * j;
      if (i < n_wr) {
        if (NT) __builtin_nontemporal_store(v[j], dst + i);
        else dst[i] = v[j];
      } else acc += v[j].x ^ v[j].w;
    }
  }
  if (acc == 0x12345u) *sink = acc;
}
int main(int argc, char** argv) {
  const size_t rd = (size_t)(argc > 1 ? atof(argv[1]) : 290) * 1000000, wr = (size_t)(argc > 2 ? atof(argv[2]) : 260) * 1000000;
  const int reps = argc > 3 ? atoi(argv[3]) : 50;
  // rotate over enough distinct buffers that nothing is served from the 256 MB memory-side cache
  const int nbuf = 4;
  u4 *src[nbuf], *dst[nbuf];
  unsigned* sink;
  for (int b = 0; b < nbuf; ++b) {
    if (hipMalloc(&src[b], rd) != hipSuccess || hipMalloc(&dst[b], wr) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src[b], 1 + b, rd);
  }
  hipMalloc(&sink, 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int nt = 0; nt < 2; ++nt)
    for (int wgs : {1024, 2048, 4096, 8192, 16384}) {
      for (int w = 0; w < 3; ++w)
        hipLaunchKernelGGL(nt ? k_copy<true> : k_copy<false>, dim3(wgs), dim3(256), 0, 0, src[w % nbuf], dst[w % nbuf], rd / 16, wr / 16, sink);
      hipEventRecord(a, 0);
      for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(nt ? k_copy<true> : k_copy<false>, dim3(wgs), dim3(256), 0, 0, src[r % nbuf], dst[r % nbuf], rd / 16, wr / 16, sink);
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      ms /= reps;
      printf("read %zu MB + write %zu MB, %s stores, %5d workgroups: %.4f ms per launch = %.2f TB/s\n", rd / 1000000, wr / 1000000,
             nt ? "non-temporal" : "plain", wgs, ms, (rd + wr) / ms / 1e9);
    }
  return 0;
}
