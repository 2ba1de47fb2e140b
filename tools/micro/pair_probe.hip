// micro-benchmark: can a SHORT two-stream write probe tell whether two separate allocations lie in the same kind of
// memory region (pair_offset.hip: two arrays written concurrently run at 3.85 TB/s inside one region and at 5.2 TB/s
// across two)?  One 256-MB reference block R, 24 candidate blocks of 1.1 GB (alive together), the probe writes
// `mb` MB into R and the same amount at the START, the MIDDLE and the END of each candidate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define G __attribute__((address_space(1)))
// every wave writes 2-KB runs alternately into both blocks: run r of the launch at offset r * 2 KB of each
__global__ __launch_bounds__(256) void probe(unsigned char* a, unsigned char* b, unsigned runs) {
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (unsigned r = blockIdx.x * 4 + wave; r < runs; r += gridDim.x * 4) {
    const size_t off = (size_t)r * 2048 + 16 * lane;
    const u32x4 v = {r, lane, 0u, 1u};
    __builtin_nontemporal_store(v, (G u32x4*)(a + off));
    __builtin_nontemporal_store(v, (G u32x4*)(a + off + 1024));
    __builtin_nontemporal_store(v, (G u32x4*)(b + off));
    __builtin_nontemporal_store(v, (G u32x4*)(b + off + 1024));
  }
}
static float rate(unsigned char* a, unsigned char* b, size_t bytes, hipEvent_t e0, hipEvent_t e1) {
  const unsigned runs = (unsigned)(bytes / 2048);
  hipLaunchKernelGGL(probe, 1024, 256, 0, 0, a, b, runs);
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe, 1024, 256, 0, 0, a, b, runs);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (float)(3.0 * 2.0 * bytes / (ms * 1e6));
}
int main(int argc, char** argv) {
  const size_t MB = 1 << 20, cand_bytes = 1100 * MB;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned char* R; hipMalloc(&R, 256 * MB);
  std::vector<unsigned char*> c(24);
  for (auto& p : c) hipMalloc(&p, cand_bytes);
  for (size_t mb : {32, 64, 128, 256}) {
    printf("probe %3zu MB: start ", mb);
    for (auto p : c) printf(" %4.0f", rate(R, p, mb * MB, e0, e1));
    printf("\n              middle");
    for (auto p : c) printf(" %4.0f", rate(R, p + 512 * MB, mb * MB, e0, e1));
    printf("\n              end   ");
    for (auto p : c) printf(" %4.0f", rate(R, p + cand_bytes - mb * MB, mb * MB, e0, e1));
    printf("\n");
  }
  return 0;
}
