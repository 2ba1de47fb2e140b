// micro-benchmark: does the copy engine pull page-locked host memory it has never read before as fast as memory it has?
// The Decoder's input is read ONCE (a 10-GB container); tools/micro/zero_copy copies the same GiB again and again.
//   cold_copy [GiB = 8] [piece MB = 146]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>
int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? atoi(argv[1]) : 8, piece = (size_t)(argc > 2 ? atoi(argv[2]) : 146) << 20;
  const size_t bytes = gib << 30;
  std::vector<unsigned char> v(bytes, 1);                       // like the Decoder's input: a std::vector filled by the caller
  auto t0 = std::chrono::steady_clock::now();
  if (hipHostRegister(v.data(), bytes, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) { printf("register failed\n"); return 1; }
  printf("page-locking %zu GiB: %.0f ms\n", gib, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  void* d = nullptr; hipMalloc(&d, 2 * piece);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int pass = 0; pass < 3; ++pass) {
    hipEventRecord(a, s);
    size_t k = 0;
    for (size_t at = 0; at + piece <= bytes; at += piece, ++k) hipMemcpyAsync((char*)d + (k & 1) * piece, v.data() + at, piece, hipMemcpyHostToDevice, s);
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("pass %d (%s): %zu copies of %zu MB: %.1f GB/s\n", pass, pass ? "read before" : "never read by the device", k, piece >> 20, k * piece / ms / 1e6);
  }
  hipHostUnregister(v.data());
  return 0;
}
