// Price of 4-byte-aligned (not 16-byte-aligned) 16-byte accesses: random point slots of 20 B (xyzw + e) vs
// 16-byte aligned quads.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); exit(1);} } while (0)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int MODE>
__global__ void k(const uint32_t* __restrict__ idx, float* __restrict__ base, float* __restrict__ sink, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
  uint32_t a[4]; float acc = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = (i + u * blockDim.x < n) ? idx[i + u * blockDim.x] : 0xFFFFFFFFu;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (a[u] == 0xFFFFFFFFu) continue;
    if (MODE == 0) { float* d = base + (size_t)a[u] * 5; *reinterpret_cast<f4u*>(d) = f4u{1, 2, 3, 4}; }            // unaligned 16 B store
    if (MODE == 1) { float* d = base + (size_t)a[u] * 8; *reinterpret_cast<float4*>(d) = make_float4(1, 2, 3, 4); }  // aligned 16 B store
    if (MODE == 2) { float* d = base + (size_t)a[u] * 5; *reinterpret_cast<f4u*>(d) = f4u{1, 2, 3, 4}; d[4] = 5.f; } // 20 B = unaligned 16 + 4
    if (MODE == 3) { float* d = base + (size_t)a[u] * 8; *reinterpret_cast<float4*>(d) = make_float4(1, 2, 3, 4); *reinterpret_cast<float4*>(d + 4) = make_float4(5, 0, 0, 0); }  // 2 aligned
    if (MODE == 4) { const float* d = base + (size_t)a[u] * 5; f4u v = *reinterpret_cast<const f4u*>(d); acc += v.x + v.w; }     // unaligned 16 B load
    if (MODE == 5) { const float* d = base + (size_t)a[u] * 8; float4 v = *reinterpret_cast<const float4*>(d); acc += v.x + v.w; } // aligned 16 B load
    if (MODE == 6) { float* d = base + (size_t)a[u] * 5; d[0] = 1; d[1] = 2; d[2] = 3; d[3] = 4; d[4] = 5; }  // five dwords
  }
  if (acc == 123.456f) sink[0] = acc;
}
template <typename F> float timeit(F f, int iters = 20) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters * 1e3f;
}
int main() {
  const size_t n = 2150000, slots = 2400000;
  uint32_t* didx; float *A, *sink; CK(hipMalloc(&didx, n * 4)); CK(hipMalloc(&A, slots * 32 + 1024)); CK(hipMalloc(&sink, 64));
  std::vector<uint32_t> h(n);
  const char* names[] = {"unaligned 16B store", "aligned 16B store", "20B (unaligned 16+4) store", "2x aligned 16B store", "unaligned 16B load", "aligned 16B load", "5 dword stores"};
  for (int pat = 0; pat < 2; ++pat) {
    srand(5);
    // pat 0: random slots over the whole array; pat 1: runs of 4 consecutive slots (points of one cell) at random bases
    for (size_t i = 0; i < n; ++i) h[i] = pat == 0 ? (uint32_t)((size_t)rand() * 1103 % slots) : (i % 4 ? h[i - 1] + 1 : (uint32_t)((size_t)rand() * 1103 % (slots - 4)));
    if (pat == 1) { // spread the runs' members over different lanes: shuffle within blocks of 4096
      for (size_t b = 0; b + 4096 <= n; b += 4096) for (int j = 4095; j > 0; --j) std::swap(h[b + j], h[b + rand() % (j + 1)]);
    }
    CK(hipMemcpy(didx, h.data(), n * 4, hipMemcpyHostToDevice));
    unsigned g = (unsigned)((n + 1023) / 1024);
    float t[7];
    t[0] = timeit([&] { k<0><<<g, 256>>>(didx, A, sink, n); }); t[1] = timeit([&] { k<1><<<g, 256>>>(didx, A, sink, n); });
    t[2] = timeit([&] { k<2><<<g, 256>>>(didx, A, sink, n); }); t[3] = timeit([&] { k<3><<<g, 256>>>(didx, A, sink, n); });
    t[4] = timeit([&] { k<4><<<g, 256>>>(didx, A, sink, n); }); t[5] = timeit([&] { k<5><<<g, 256>>>(didx, A, sink, n); });
    t[6] = timeit([&] { k<6><<<g, 256>>>(didx, A, sink, n); });
    printf("pattern %d (%zu accesses):\n", pat, n);
    for (int m = 0; m < 7; ++m) printf("  %-28s %.1f us (%.0f G/s)\n", names[m], t[m], n / t[m] * 1e-3);
  }
  return 0;
}
