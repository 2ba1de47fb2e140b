// micro-test: are byte-misaligned dword / dwordx4 global stores and loads handled by gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(unsigned char* base, int off) {
  unsigned* p = (unsigned*)(base + off + 16 * threadIdx.x);   // misaligned by `off`
  uint4 v = make_uint4(threadIdx.x, 0x11111111u, 0x22222222u, 0x33333333u);
  __builtin_nontemporal_store(v.x, p);  // plain dword
  *(uint4*)(base + 4096 + off + 16 * threadIdx.x) = v;  // dwordx4 misaligned
}
int main() {
  unsigned char* d; hipMalloc(&d, 16384); 
  for (int off = 0; off < 4; ++off) {
    hipMemset(d, 0xEE, 16384);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d, off);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned char> h(16384); hipMemcpy(h.data(), d, 16384, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) { unsigned v; memcpy(&v, &h[off + 16*t], 4); if (v != (unsigned)t) bad++; 
      unsigned w[4]; memcpy(w, &h[4096 + off + 16*t], 16); if (w[0]!=(unsigned)t||w[1]!=0x11111111u||w[3]!=0x33333333u) bad++; }
    printf("off %d: err=%s bad=%d\n", off, hipGetErrorString(e), bad);
  }
  return 0;
}
