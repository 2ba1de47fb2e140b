// micro-benchmark: how fast does the memory system take the tile kernel's OUTPUT, by store instruction shape?
// 4096 resident waves (1024 workgroups x 4) each write "items": a contiguous run of 304 points = 1824 B of positions
// and 912 B of colours, item i of the launch at i * 1824 / i * 912 (runs of neighbouring waves are adjacent, as in the
// kernel), items dealt to waves the way the kernel's groups are (16 consecutive items per workgroup and turn).
//   shape 0: positions dwordx3 per lane (12 B, 768 B per instruction), colours dwordx3 on even lanes (the kernel today)
//   shape 1: positions and colours dwordx4 per lane (16 B, 1 KB per instruction, 16-B aligned)
//   shape 2: positions dwordx2 per lane, colours dword per lane
//   shape 3: positions dword + short per lane (one point per lane), colours short + byte
//   shape 4: as 0, but every instruction's range starts on a 128-B line (what the aligned lane <-> point map gives)
//   shape 5: as 1, but every item starts on a 128-B line and covers whole lines only (1920 B + 1024 B per item): no line is
//            shared between two waves
// nt = non-temporal stores.  Reports GB/s of stored bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define G __attribute__((address_space(1)))
template <class T, bool NT> __device__ __forceinline__ void st(unsigned char* p, T v) {
  if (NT) __builtin_nontemporal_store(v, (G T*)p); else *(G T*)p = v;
}
template <int SHAPE, bool NT>
__global__ __launch_bounds__(256) void k(unsigned char* xyz, unsigned char* rgb, int items) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int groups = (items + 15) / 16;
  for (int g = blockIdx.x; g < groups; g += gridDim.x)
    for (int t = 0; t < 4; ++t) {
      const int i = g * 16 + t * 4 + wave;
      if (i >= items) continue;
      unsigned char* px = xyz + (size_t)i * 1824;
      unsigned char* pc = rgb + (size_t)i * 912;
      if (SHAPE == 0 || SHAPE == 4) {
        const int shift = SHAPE == 4 ? (int)(((size_t)i * 304) & 127) : 0;   // points before the item in its 128-point block
        for (int kk = 2 * lane; kk < 304 + shift; kk += 128) {
          const int k = kk - shift;
          if (k < 0 || k + 1 >= 304) continue;
          u32x3 v = {(unsigned)k, (unsigned)lane, (unsigned)i};
          st<u32x3, NT>(px + k * 6, v);
          if (!(lane & 1) && k + 3 < 304) st<u32x3, NT>(pc + k * 3, v);
        }
      } else if (SHAPE == 1) {
        for (int o = 16 * lane; o < 1824; o += 1024) { u32x4 v = {(unsigned)o, 1u, 2u, 3u}; st<u32x4, NT>(px + o, v); }
        for (int o = 16 * lane; o < 912; o += 1024) { u32x4 v = {(unsigned)o, 1u, 2u, 3u}; st<u32x4, NT>(pc + o, v); }
      } else if (SHAPE == 5) {
        unsigned char* qx = xyz + (size_t)i * 1920;
        unsigned char* qc = rgb + (size_t)i * 1024;
        for (int o = 16 * lane; o < 1920; o += 1024) { u32x4 v = {(unsigned)o, 1u, 2u, 3u}; st<u32x4, NT>(qx + o, v); }
        { u32x4 v = {(unsigned)lane, 1u, 2u, 3u}; st<u32x4, NT>(qc + 16 * lane, v); }
      } else if (SHAPE == 2) {
        for (int o = 8 * lane; o < 1824; o += 512) { u32x2 v = {(unsigned)o, 1u}; st<u32x2, NT>(px + o, v); }
        for (int o = 4 * lane; o < 912; o += 256) st<unsigned, NT>(pc + o, (unsigned)o);
      } else {
        for (int k = lane; k < 304; k += 64) {
          st<unsigned, NT>(px + k * 6, (unsigned)k); st<unsigned short, NT>(px + k * 6 + 4, (unsigned short)k);
          st<unsigned short, NT>(pc + k * 3, (unsigned short)k); st<unsigned char, NT>(pc + k * 3 + 2, (unsigned char)k);
        }
      }
    }
}
static bool g_compact = false;
static int g_wgs = 1024;
template <int SHAPE, bool NT> void run(unsigned char* xyz, unsigned char* rgb, int items, hipEvent_t a, hipEvent_t b) {
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<SHAPE, NT>), g_wgs, 256, 0, 0, xyz, rgb, items);
  hipEventRecord(a);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<SHAPE, NT>), g_wgs, 256, 0, 0, xyz, rgb, items);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("shape %d %s: %.3f ms  %.0f GB/s   ", SHAPE, NT ? "nt   " : "plain", ms, (double)items * (SHAPE == 5 ? 2944 : 2736) / ms / 1e6);
  if (!g_compact) printf("\n");
}
int main(int argc, char** argv) {
  const int items = 333000;                       // one 128-frame S-longdress launch: 911 MB of output
  unsigned char *xyz, *rgb;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  if (argc > 1) {                                 // "regions N": the kernel's shape on N output buffers one after the other in VRAM
    const int n = atoi(argv[1]);
    for (int i = 0; i < n; ++i) {
      g_compact = true;
      hipMalloc(&xyz, (size_t)items * 1920 + 4096); hipMalloc(&rgb, (size_t)items * 1024 + 4096);
      unsigned char* pad; hipMalloc(&pad, 3ull << 30);      // (kept: the next pair lies 4 GB further on)
      printf("region %2d: ", i);
      if (argc > 2) {                               // "regions N wgs": shape 4 nt by the number of workgroups (concurrent write streams)
        for (int wg : {128, 256, 512, 1024, 2048}) { g_wgs = wg; printf("[%d wgs] ", wg); run<4, true>(xyz, rgb, items, a, b); }
        printf("\n");
        continue;
      }
      run<4, true>(xyz, rgb, items, a, b); run<4, false>(xyz, rgb, items, a, b);
      run<1, true>(xyz, rgb, items, a, b); run<5, true>(xyz, rgb, items, a, b); run<5, false>(xyz, rgb, items, a, b); printf("\n");
    }
    return 0;
  }
  hipMalloc(&xyz, (size_t)items * 1824 + 4096); hipMalloc(&rgb, (size_t)items * 912 + 4096);
  run<0, true>(xyz, rgb, items, a, b); run<4, true>(xyz, rgb, items, a, b); run<1, true>(xyz, rgb, items, a, b);
  run<2, true>(xyz, rgb, items, a, b); run<3, true>(xyz, rgb, items, a, b);
  run<0, false>(xyz, rgb, items, a, b); run<4, false>(xyz, rgb, items, a, b); run<1, false>(xyz, rgb, items, a, b);
  return 0;
}
