// micro-benchmark: does a BRICKED cell grid (4 x 4 x 4 cells = 2 KB contiguous) serve the smoothing passes' scattered 32-byte cell
// accesses faster than the linear (z, y, x) one?  Real cell lists (tools/micro/bin/cell_lists.{hdr,bin}: 8 S-longdress frames, the cells
// of every span of 1 024 points in order of first appearance), 128 frame slots of w^3 = 128^3 cells of 32 B + colour cells of 16 B, as
// in k_smooth_clear (stores only) and k_smooth_mark (one 32-byte read per entry).  The lists: tools/micro/gen_cell_lists.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned remap(unsigned key, int bricked) {
  if (!bricked) return key;
  const unsigned cx = key & 127u, cy = (key >> 7) & 127u, cz = key >> 14;
  return ((((cz >> 2) * 32u + (cy >> 2)) * 32u + (cx >> 2)) << 6) | (cx & 3u) | ((cy & 3u) << 2) | ((cz & 3u) << 4);
}
__global__ __launch_bounds__(256) void k_clear(const unsigned* keys, const unsigned* first, const unsigned* count, char* cells, char* ccells, size_t slot_cells, int bricked) {
  const unsigned f = blockIdx.y, src = f & 7u;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= count[src]) return;
  const unsigned k = remap(keys[first[src] + i], bricked);
  u32x4* c = (u32x4*)(cells + ((size_t)f * slot_cells + k) * 32u);
  c[0] = u32x4{0, 0, 0, 0}; c[1] = u32x4{0, 0, 0, 0};
  *(u32x4*)(ccells + ((size_t)f * slot_cells + k) * 16u) = u32x4{0, 0, 0, 0};
}
__global__ __launch_bounds__(256) void k_mark(const unsigned* keys, const unsigned* first, const unsigned* count, const char* cells, size_t slot_cells, int bricked, unsigned* out) {
  const unsigned f = blockIdx.y, src = f & 7u;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= count[src]) return;
  const unsigned k = remap(keys[first[src] + i], bricked);
  const u32x4* c = (const u32x4*)(cells + ((size_t)f * slot_cells + k) * 32u);
  const u32x4 a = c[0], b = c[1];
  if ((a.x | b.w) == 0x12345678u) out[0] = 1;      // never
}
int main() {
  FILE* h = fopen("tools/micro/bin/cell_lists.hdr", "rb"); FILE* b = fopen("tools/micro/bin/cell_lists.bin", "rb");
  if (!h || !b) { printf("no cell lists\n"); return 1; }
  unsigned nf; fread(&nf, 4, 1, h); std::vector<unsigned> cnt(nf), first(nf); fread(cnt.data(), 4, nf, h);
  unsigned tot = 0, mx = 0; for (unsigned i = 0; i < nf; ++i) { first[i] = tot; tot += cnt[i]; mx = cnt[i] > mx ? cnt[i] : mx; }
  std::vector<unsigned> keys(tot); fread(keys.data(), 4, tot, b);
  unsigned *dk, *df, *dc, *dout; hipMalloc(&dk, 4 * tot); hipMalloc(&df, 4 * nf); hipMalloc(&dc, 4 * nf); hipMalloc(&dout, 4);
  hipMemcpy(dk, keys.data(), 4 * tot, hipMemcpyHostToDevice); hipMemcpy(df, first.data(), 4 * nf, hipMemcpyHostToDevice); hipMemcpy(dc, cnt.data(), 4 * nf, hipMemcpyHostToDevice);
  const size_t slot_cells = 128ull * 128 * 128, frames = 128;
  char *cells, *ccells; hipMalloc(&cells, frames * slot_cells * 32); hipMalloc(&ccells, frames * slot_cells * 16);
  hipMemset(cells, 0, frames * slot_cells * 32); hipMemset(ccells, 0, frames * slot_cells * 16);
  hipEvent_t a, e; hipEventCreate(&a); hipEventCreate(&e);
  const dim3 grid((mx + 255) / 256, frames);
  for (int bricked = 0; bricked < 2; ++bricked)
    for (int which = 0; which < 2; ++which) {
      float best = 1e9f;
      for (int r = 0; r < 6; ++r) {
        hipEventRecord(a);
        if (which == 0) hipLaunchKernelGGL(k_clear, grid, dim3(256), 0, 0, dk, df, dc, cells, ccells, slot_cells, bricked);
        else hipLaunchKernelGGL(k_mark, grid, dim3(256), 0, 0, dk, df, dc, cells, slot_cells, bricked, dout);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, a, e); if (r && ms < best) best = ms;
      }
      printf("%s grid, %s: %.3f ms per 128 frames (%u entries per frame)\n", bricked ? "bricked" : "linear ", which ? "read 32 B per entry (mark) " : "zero 48 B per entry (clear)", best, tot / nf);
    }
  return 0;
}
