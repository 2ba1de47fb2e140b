// micro-benchmark: how fast does the dispatcher place workgroups shaped like k_general_blocks' (256 threads, 64 VGPRs, 11 KB of
// LDS), and how full does the chip stay when each lives a few microseconds?  The general sequence launches ~100 000 of them per
// 128 S-longdress frames, and its counters show a third of the wave slots occupied on average.
//   dispatch_rate            grid (790, 128); workgroups that (a) end at once, (b) stay for 2, 4, 6, 8 us (s_memrealtime)
//                            — and the same work as a PERSISTENT grid of 2 048 workgroups that loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ void stay(unsigned ticks) {   // s_memrealtime: 100 MHz
  if (!ticks) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}
template <int kLds>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_wg(unsigned ticks, unsigned* out) {
  __shared__ unsigned lds[kLds / 4];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  stay(ticks);
  if (lds[(threadIdx.x + 1) & 255] == 0xFFFFFFFFu) out[0] = 1;   // (keeps the LDS)
}
template <int kLds>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_loop(unsigned ticks, unsigned groups, unsigned* out) {
  __shared__ unsigned lds[kLds / 4];
  for (unsigned g = blockIdx.x; g < groups; g += gridDim.x) {
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    stay(ticks);
    if (lds[(threadIdx.x + 1) & 255] == 0xFFFFFFFFu) out[0] = 1;
    __syncthreads();
  }
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const dim3 grid(790, 128);
  const unsigned groups = grid.x * grid.y;
  for (unsigned us : {0u, 2u, 4u, 6u, 8u}) {
    float ms;
    for (int lds : {1024, 11428}) {
      auto run = [&](bool loop) {
        for (int r = 0; r < 4; ++r) {
          if (r == 1) (void)hipEventRecord(a);
          if (!loop) { if (lds == 1024) hipLaunchKernelGGL(k_wg<1024>, grid, dim3(256), 0, 0, us * 100, d); else hipLaunchKernelGGL(k_wg<11428>, grid, dim3(256), 0, 0, us * 100, d); }
          else { if (lds == 1024) hipLaunchKernelGGL(k_loop<1024>, dim3(2048), dim3(256), 0, 0, us * 100, groups, d); else hipLaunchKernelGGL(k_loop<11428>, dim3(2048), dim3(256), 0, 0, us * 100, groups, d); }
        }
        (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);
        return ms / 3;
      };
      const float one = run(false), loop = run(true);
      printf("stay %u us, %5d B LDS: %u workgroups of their own %.3f ms (%.0f per us; full chip would take %.3f)   2 048 looping %.3f ms\n", us, lds, groups,
             one, groups / one / 1e3, us ? groups * (us * 1e-3) / 2048 : 0.0, loop);
    }
  }
  return 0;
}
