// micro-benchmark: what do the L2's memory-side read counters report when a kernel uses only PART of each
// 128-byte line?  One 16x16 u16 tile per wave (8 B per lane, 32 B per row — the tile kernel's access
// pattern) over 5 raster planes of 1280x1408 x 32 frames (0.58 GB, >> the 256 MB Infinity Cache), but only the
// tiles with (bx % 4) < K are read: K = 1 uses 32 B of every 128-B line, K = 2 the first 64 B, K = 4 all of it.
// Run under  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
// and compare  32*n32 + 64*n64 + 128*n128  with the requested bytes and with lines x 128 (tools/attribution.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k_sparse(const unsigned short* __restrict__ base, size_t plane_elems, int planes,
                                                int W, int H, int K, int rowsel, unsigned* __restrict__ out) {
  const int bw = W / 16, tiles = bw * (H / 16);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int frame = blockIdx.y;
  const int t = blockIdx.x * 4 + wave;
  if (t >= tiles) return;
  const int bx = t % bw, by = t / bw;
  if ((bx & 3) >= K) return;
  if (rowsel && ((lane >> 2) & 1)) return;            // rowsel: only even rows of the tile
  const unsigned short* fb = base + (size_t)frame * planes * plane_elems;
  unsigned acc = 0;
  for (int p = 0; p < planes; ++p) {
    const size_t idx = (size_t)(by * 16 + (lane >> 2)) * W + bx * 16 + 4 * (lane & 3);
    const uint2 v = *reinterpret_cast<const uint2*>(fb + (size_t)p * plane_elems + idx);
    acc += v.x ^ (v.y >> 3);
  }
  if (acc == 0x12345678u) out[t] = acc;               // keeps the loads alive, never true on 0x0101 fill
}
int main() {
  const int W = 1280, H = 1408, frames = 32, planes = 5;
  const size_t pe = (size_t)W * H; const int tiles = (W / 16) * (H / 16);
  unsigned short* d; hipMalloc(&d, pe * 2 * planes * frames); hipMemset(d, 1, pe * 2 * planes * frames);
  unsigned* out; hipMalloc(&out, (size_t)tiles * 4);
  unsigned char* flush; const size_t fl = 600u << 20; hipMalloc(&flush, fl);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int cfg[][2] = {{4, 0}, {2, 0}, {1, 0}, {3, 0}, {4, 1}, {1, 1}};
  for (auto& c : cfg) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemsetAsync(flush, rep, fl, 0);              // evict the planes from the Infinity Cache between launches
      hipEventRecord(a);
      hipLaunchKernelGGL(k_sparse, dim3((tiles + 3) / 4, frames), 256, 0, 0, d, pe, planes, W, H, c[0], c[1], out);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double req = (double)tiles / 4 * c[0] * frames * planes * 512 / (c[1] ? 2 : 1);
      const double lines = (double)tiles / 4 * frames * planes * 16 / (c[1] ? 2 : 1) * 128;
      printf("K %d rowsel %d rep %d: %.3f ms requested %.0f B, distinct 128-B lines %.0f B\n", c[0], c[1], rep, ms, req, lines);
    }
  }
  return 0;
}
