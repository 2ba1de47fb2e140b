// micro-benchmark: what does the memory system deliver for the tile kernel's access pattern?
//   mode 0: each wave reads one 16x16 u16 tile (lane: 8 B at row l>>2, col 4*(l&3)) from P planes of a
//           W x H raster (row stride W*2 bytes), tiles in raster block order, frames interleaved like the kernel
//   mode 1: each wave reads the same number of bytes, but 512 contiguous bytes per plane (linear)
//   mode 2: like 0 but a wave takes a 64x4-pixel strip instead (lane: 8 B, 16 lanes per row): full 128-B lines
// plus optionally a contiguous 2.3 KB store per wave (like the kernel's output).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(1024) void k(const unsigned short* __restrict__ base, size_t plane_elems, int planes,
                                          int W, int H, int frames, int mode, unsigned* __restrict__ out, int do_store,
                                          float fill) {
  const int bw = W / 16, bh = H / 16, tiles = bw * bh;
  const int wave_in_block = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * 16 + wave_in_block;          // global wave index
  int frame = gw % frames;                                        // frames interleaved
  long t = gw / frames;
  if (mode >= 3) {                                                // like the kernel: a workgroup = 16 consecutive tiles of one frame
    frame = blockIdx.x % frames;
    t = (long)(blockIdx.x / frames) * 16 + wave_in_block;
  }
  if (t >= tiles) return;
  if ((unsigned)((t * 2654435761u) >> 8 & 1023) > (unsigned)(fill * 1024)) return;   // skip tiles like unowned blocks
  const unsigned short* fb = base + (size_t)frame * planes * plane_elems;
  unsigned acc = 0;
  for (int p = 0; p < planes; ++p) {
    const unsigned short* pl = fb + (size_t)p * plane_elems;
    size_t idx;
    if (mode == 4) {   // workgroup-cooperative strips: wave w reads rows 4w'.. of a 64-px strip made of tiles 4*(w>>2)..+3
      const long tq = (t & ~3L) ;                                    // first tile of this wave's quad
      const int bx = tq % bw, by = tq / bw;
      idx = (size_t)(by * 16 + 4 * (wave_in_block & 3) + (lane >> 4)) * W + bx * 16 + 4 * (lane & 15);
    } else if (mode == 0 || mode == 3) { const int bx = t % bw, by = t / bw; idx = (size_t)(by * 16 + (lane >> 2)) * W + bx * 16 + 4 * (lane & 3); }
    else if (mode == 1) { idx = (size_t)t * 256 + 4 * lane; }
    else { const int sx = t % (W / 64), sy = t / (W / 64); idx = (size_t)(sy * 4 + (lane >> 4)) * W + sx * 64 + 4 * (lane & 15); }
    const uint2 v = *reinterpret_cast<const uint2*>(pl + idx);
    acc += v.x ^ v.y;
  }
  if (do_store) {
    unsigned* o = out + (size_t)gw * 576;                          // 2304 B per wave
    for (int i = 0; i < 9; ++i) o[i * 64 + lane] = acc + i;
  } else if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const int W = 1280, H = 1408, frames = 32, planes = 5;           // 5 full-res planes ~ geo x2, attrY x2, chroma
  const size_t pe = (size_t)W * H;
  unsigned short* d; hipMalloc(&d, pe * 2 * planes * frames);
  hipMemset(d, 1, pe * 2 * planes * frames);
  unsigned* out; hipMalloc(&out, (size_t)7040 * frames * 2304 + 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const long waves = 7040L * frames;
  for (float fill : {1.0f, 0.37f})
  for (int st = 0; st < 2; ++st)
    for (int mode = 0; mode < 5; ++mode) {
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, (waves + 15) / 16, 1024, 0, 0, d, pe, planes, W, H, frames, mode, out, st, fill);
      hipEventRecord(a);
      for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k, (waves + 15) / 16, 1024, 0, 0, d, pe, planes, W, H, frames, mode, out, st, fill);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
      const double rd = (double)waves * fill * planes * 512, wr = st ? (double)waves * fill * 2304 : 0;
      printf("fill %.2f mode %d store %d: %.3f ms  read %.0f GB/s  write %.0f GB/s  total %.0f GB/s\n", fill, mode, st, ms,
             rd / ms / 1e6, wr / ms / 1e6, (rd + wr) / ms / 1e6);
    }
  return 0;
}
