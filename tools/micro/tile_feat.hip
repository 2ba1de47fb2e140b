// micro-benchmark: start from the tile-read pattern that reaches ~5 TB/s (one 16x16 tile per wave, 16 adjacent
// tiles per 1024-thread workgroup, contiguous 2.3 KB store per wave) and add the real kernel's structural
// features one by one (bit flags) to see which one costs throughput.
//   1 ticket atomic + barrier at workgroup start        2 dependent descriptor load before the plane loads
//   4 64 KB of LDS per workgroup + barrier after loads   8 a second barrier (count) + third barrier (look-back slot)
//  16 stores as unaligned dwordx3 on even lanes (12 B) + dwordx3 on every 4th lane (like the kernel)
//  32 variable amount of work per wave (0..8 store iterations, mean ~4.8)   64 ~600 dummy VALU instrs per wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct U3 { unsigned a, b, c; };
__global__ __launch_bounds__(1024) void k(const unsigned short* __restrict__ base, size_t plane_elems, int planes,
                                          int W, int H, int frames, int feat, unsigned char* __restrict__ out,
                                          unsigned* __restrict__ tickets, const unsigned* __restrict__ desc) {
  __shared__ unsigned s_g;
  extern __shared__ unsigned char lds[];
  const int bw = W / 16, tiles = bw * (H / 16);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int frame = blockIdx.x % frames;
  unsigned grp = blockIdx.x / frames;
  if (feat & 1) {
    if (threadIdx.x == 0) s_g = atomicAdd(&tickets[frame * 64], 1u);
    __syncthreads();
    grp = s_g;
  }
  long t = (long)grp * 16 + wave;
  if (t >= tiles) return;
  if (feat & 2) t = desc[(size_t)frame * tiles + t];                // dependent load (identity table)
  const unsigned short* fb = base + (size_t)frame * planes * plane_elems;
  const int bx = t % bw, by = t / bw;
  unsigned acc = 0;
  for (int p = 0; p < planes; ++p) {
    const size_t idx = (size_t)(by * 16 + (lane >> 2)) * W + bx * 16 + 4 * (lane & 3);
    const uint2 v = *reinterpret_cast<const uint2*>(fb + (size_t)p * plane_elems + idx);
    acc += v.x ^ (v.y >> 3);
  }
  if (feat & 4) { ((unsigned*)lds)[threadIdx.x] = acc; __syncthreads(); acc += ((unsigned*)lds)[threadIdx.x ^ 64]; }
  if (feat & 8) { __syncthreads(); if (wave == 0) ((unsigned*)lds)[lane] = acc; __syncthreads(); }
  if (feat & 64) { for (int i = 0; i < 300; ++i) acc = acc * 1664525u + 1013904223u; }
  int iters = 9;                                                    // 9 x 256 B = 2304 B per wave
  if (feat & 32) iters = (int)((t * 2654435761u >> 7) % 17);        // 0..16 half-iterations -> mean 8
  unsigned char* o = out + ((size_t)frame * tiles + t) * 4608;
  if (feat & 16) {
    // 1 iteration = 64 points: xyz 384 B via dwordx3 on even lanes, rgb 192 B via dwordx3 on every 4th lane
    const int it2 = (iters * 256) / 576;                            // same bytes: 576 B per iteration
    for (int i = 0; i < it2; ++i) {
      if (!(lane & 1)) { U3 v{acc, acc + i, 7u}; __builtin_memcpy(o + 2 + i * 384 + lane * 6, &v, 12); }
      if (!(lane & 3)) { U3 v{acc, acc ^ i, 9u}; __builtin_memcpy(o + 3073 + i * 192 + lane * 3, &v, 12); }
    }
  } else {
    for (int i = 0; i < iters; ++i) ((unsigned*)o)[i * 64 + lane] = acc + i;
  }
}
int main() {
  const int W = 1280, H = 1408, frames = 32, planes = 5;
  const size_t pe = (size_t)W * H; const int tiles = (W / 16) * (H / 16);
  unsigned short* d; hipMalloc(&d, pe * 2 * planes * frames); hipMemset(d, 1, pe * 2 * planes * frames);
  unsigned char* out; hipMalloc(&out, (size_t)tiles * frames * 4608 + 64);
  unsigned* tk; hipMalloc(&tk, 256 * frames);
  unsigned* desc; hipMalloc(&desc, (size_t)tiles * frames * 4);
  { unsigned* h = (unsigned*)malloc((size_t)tiles * frames * 4); for (int f = 0; f < frames; ++f) for (int i = 0; i < tiles; ++i) h[(size_t)f * tiles + i] = i;
    hipMemcpy(desc, h, (size_t)tiles * frames * 4, hipMemcpyHostToDevice); free(h); }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int groups = tiles / 16;
  int feats[] = {0, 1, 2, 4, 8, 12, 16, 32, 64, 3, 15, 31, 63, 127};
  for (int feat : feats) {
    const size_t lds = (feat & 12) ? 65536 : 0;
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      hipMemsetAsync(tk, 0, 256 * frames, 0);
      hipEventRecord(a);
      hipLaunchKernelGGL(k, groups * frames, 1024, lds, 0, d, pe, planes, W, H, frames, feat, out, tk, desc);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (rep > 0 && ms < best) best = ms;
    }
    const double rd = (double)tiles * frames * planes * 512, wr = (double)tiles * frames * 2304 * ((feat & 32) ? 8.0 / 9 : 1.0);
    printf("feat %3d: %.3f ms  (read %.0f + write %.0f = %.0f GB/s)\n", feat, best, rd / best / 1e6, wr / best / 1e6, (rd + wr) / best / 1e6);
  }
  return 0;
}
