// micro-benchmark: a kernel pulls only SOME 128-byte lines of page-locked host memory over PCIe (zero-copy reads) — the lines of
// raster planes that owned and occupied blocks touch (0.57 of a S-longdress frame's plane bytes) — against pulling everything.
// Layout: rows of 2 560 bytes (a 1280-pixel luma row = 20 lines); a CELL is one line column over 16 rows (what a 16x16 block row
// needs, shared by four blocks); each cell is needed with probability p, in runs along a row (patches are several cells wide).
//   sparse_pull [p = 0.57]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <sys/mman.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define G __attribute__((address_space(1)))
constexpr size_t kRow = 2560, kLinesPerRow = kRow / 128;
// a wave per listed cell: lane = (row & 7, 16-byte piece): two trips cover the cell's 16 rows
__global__ __launch_bounds__(256) void k_pull_cells(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                    const unsigned* __restrict__ cells, unsigned n_cells) {
  const unsigned w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  for (unsigned c = w; c < n_cells; c += gridDim.x * 4u) {
    const unsigned cell = cells[c], group = cell / kLinesPerRow, col = cell % kLinesPerRow;
    const size_t at = ((size_t)group * 16u + (lane >> 3)) * kRow + (size_t)col * 128u + (lane & 7u) * 16u;
    const u32x4 a = __builtin_nontemporal_load((const G u32x4*)(src + at));
    const u32x4 b = __builtin_nontemporal_load((const G u32x4*)(src + at + 8 * kRow));
    __builtin_nontemporal_store(a, (G u32x4*)(dst + at));
    __builtin_nontemporal_store(b, (G u32x4*)(dst + at + 8 * kRow));
  }
}
__global__ __launch_bounds__(256) void k_push(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(src[i], (G u32x4*)(dst + i));
}
int main(int argc, char** argv) {
  const double p = argc > 1 ? atof(argv[1]) : 0.57;
  const size_t bytes = 1ull << 30, groups = bytes / (16 * kRow), n_all = groups * kLinesPerRow;
  void* h = aligned_alloc(2 << 20, bytes);
  madvise(h, bytes, MADV_HUGEPAGE);
  memset(h, 1, bytes);
  if (hipHostRegister(h, bytes, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) { printf("register failed\n"); return 1; }
  void *hd = nullptr, *d = nullptr;
  hipHostGetDevicePointer(&hd, h, 0);
  hipMalloc(&d, bytes);
  // results going back meanwhile: 0.4 GiB device -> page-locked host, by kernel
  void *h2 = nullptr, *h2d = nullptr, *d2 = nullptr;
  hipHostMalloc(&h2, bytes, hipHostMallocPortable | hipHostMallocMapped); hipHostGetDevicePointer(&h2d, h2, 0); hipMalloc(&d2, bytes);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipEvent_t a, b, c, e; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c); hipEventCreate(&e);
  srand(7);
  for (int variant = 0; variant < 3; ++variant) {
    // 0: every cell; 1: cells needed with probability p in runs of 1-6 along a row; 2: the same cells, each row group's cells adjacent in the list
    std::vector<unsigned> cells;
    if (variant == 0) { for (unsigned k = 0; k < n_all; ++k) cells.push_back(k); }
    else {
      for (size_t g = 0; g < groups; ++g)
        for (unsigned col = 0; col < kLinesPerRow;) {
          const unsigned run = 1 + rand() % 6;
          const bool need = rand() / (double)RAND_MAX < p;
          for (unsigned k = 0; k < run && col < kLinesPerRow; ++k, ++col) if (need) cells.push_back((unsigned)(g * kLinesPerRow + col));
        }
      if (variant == 2) continue;
    }
    unsigned* dc = nullptr; hipMalloc(&dc, cells.size() * 4); hipMemcpy(dc, cells.data(), cells.size() * 4, hipMemcpyHostToDevice);
    const double useful = (double)cells.size() * 16 * 128;
    for (int duplex = 0; duplex < 2; ++duplex)
      for (int wgs : {256, 1024, 4096}) {
        float ms = 0, ms2 = 0;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(a, s1);
          for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_pull_cells, dim3(wgs), dim3(256), 0, s1, (const unsigned char*)hd, (unsigned char*)d, dc, (unsigned)cells.size());
          hipEventRecord(b, s1);
          if (duplex) { hipEventRecord(c, s2); for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_push, dim3(256), dim3(256), 0, s2, (const u32x4*)d2, (u32x4*)h2d, (bytes * 2 / 5) / 16); hipEventRecord(e, s2); }
          hipDeviceSynchronize();
          hipEventElapsedTime(&ms, a, b);
          if (duplex) hipEventElapsedTime(&ms2, c, e);
        }
        printf("%s, %4d workgroups%s: %.1f GB/s of needed bytes (%.0f MB of %.0f)", variant ? "needed cells only" : "every cell       ", wgs,
               duplex ? ", results going back" : "                    ", 3 * useful / ms / 1e6, useful / 1e6, bytes / 1e6);
        if (duplex) printf("; back %.1f GB/s", 3 * (bytes * 2 / 5) / ms2 / 1e6);
        printf("\n");
      }
    hipFree(dc);
  }
  return 0;
}
