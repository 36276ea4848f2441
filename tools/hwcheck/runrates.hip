// How much do partially contiguous accesses cost?  Lanes read/write 4-byte elements in RUNS of L consecutive
// elements at random bases (L = 1 .. 64).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); exit(1);} } while (0)
template <int MODE>  // 0 store, 1 load
__global__ void k(const uint32_t* __restrict__ idx, uint32_t* __restrict__ base, uint32_t* __restrict__ sink, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
  uint32_t a[4], acc = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = (i + u * blockDim.x < n) ? idx[i + u * blockDim.x] : 0xFFFFFFFFu;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (a[u] == 0xFFFFFFFFu) continue;
    if (MODE == 0) base[a[u]] = a[u]; else acc += base[a[u]];
  }
  if (MODE == 1 && acc == 0x12345u) sink[0] = acc;
}
template <typename F> float timeit(F f, int iters = 20) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters * 1e3f;
}
int main() {
  const size_t n = 4800000, region = 4800000;  // elements
  uint32_t *didx, *A, *sink; CK(hipMalloc(&didx, n * 4)); CK(hipMalloc(&A, region * 4 + 1024)); CK(hipMalloc(&sink, 64));
  std::vector<uint32_t> h(n);
  for (int L : {1, 2, 4, 8, 14, 16, 32, 64}) {
    for (int aligned = 0; aligned < 2; ++aligned) {
      srand(7);
      for (size_t i = 0; i < n; i += L) {
        size_t b = (size_t)rand() * 977 % (region - 64);
        if (aligned) b = b / L * L;
        for (int j = 0; j < L && i + j < n; ++j) h[i + j] = (uint32_t)(b + j);
      }
      CK(hipMemcpy(didx, h.data(), n * 4, hipMemcpyHostToDevice));
      unsigned g = (unsigned)((n + 1023) / 1024);
      float ts = timeit([&] { k<0><<<g, 256>>>(didx, A, sink, n); });
      float tl = timeit([&] { k<1><<<g, 256>>>(didx, A, sink, n); });
      printf("run %2d %s: store %.1f us (%.0f G elem/s) | load %.1f us (%.0f G elem/s)\n", L, aligned ? "aligned  " : "unaligned", ts, n / ts * 1e-3, tl, n / tl * 1e-3);
    }
  }
  return 0;
}
