// Machine constants the voxelizer's design rests on (MI355X): streaming ceilings and the price of scattered
// small accesses (memory TRANSACTIONS per second, not bytes).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); exit(1);} } while (0)

__global__ void k_fill(float4* __restrict__ dst, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u, i += blockDim.x) if (i < n4) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
  float4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = (i + u * blockDim.x < n4) ? src[i + u * blockDim.x] : make_float4(0, 0, 0, 0);
#pragma unroll
  for (int u = 0; u < 4; ++u) if (i + u * blockDim.x < n4) dst[i + u * blockDim.x] = v[u];
}
__global__ void k_read(const float4* __restrict__ src, float* __restrict__ sink, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) if (i + u * blockDim.x < n4) { float4 v = src[i + u * blockDim.x]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) sink[0] = acc;
}
// each lane: ILP independent scattered accesses.  idx[] precomputed (coalesced read).
template <int MODE>  // 0: 4-B store, 1: 4-B load, 2: 16-B + 4-B store at 20-B slots, 3: 1-B store, 4: 8-B store
__global__ void k_scatter(const uint32_t* __restrict__ idx, char* __restrict__ base, float* __restrict__ sink, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
  uint32_t a[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = (i + u * blockDim.x < n) ? idx[i + u * blockDim.x] : 0xFFFFFFFFu;
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (a[u] == 0xFFFFFFFFu) continue;
    if (MODE == 0) reinterpret_cast<uint32_t*>(base)[a[u]] = a[u];
    if (MODE == 1) acc += reinterpret_cast<float*>(base)[a[u]];
    if (MODE == 2) { float* d = reinterpret_cast<float*>(base) + (size_t)a[u] * 5;
      typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
      *reinterpret_cast<f4u*>(d) = f4u{1.f, 2.f, 3.f, 4.f}; d[4] = 5.f; }
    if (MODE == 3) reinterpret_cast<unsigned char*>(base)[a[u]] = (unsigned char)a[u];
    if (MODE == 4) reinterpret_cast<uint2*>(base)[a[u]] = make_uint2(a[u], 1u);
  }
  if (MODE == 1 && acc == 123.456f) sink[0] = acc;
}

template <typename F> float timeit(F f, int iters = 20) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters * 1e3f;  // us
}

int main() {
  const size_t MB = 1 << 20;
  char *A, *B; float* sink;
  CK(hipMalloc(&A, 512 * MB)); CK(hipMalloc(&B, 512 * MB)); CK(hipMalloc(&sink, 256));
  CK(hipMemset(A, 0, 512 * MB)); CK(hipMemset(B, 0, 512 * MB));
  for (size_t bytes : {96 * MB, 200 * MB, 400 * MB}) {
    size_t n4 = bytes / 16; unsigned g = (unsigned)((n4 + 1023) / 1024);
    float tf = timeit([&] { k_fill<<<g, 256>>>((float4*)A, n4); });
    float tr = timeit([&] { k_read<<<g, 256>>>((const float4*)A, sink, n4); });
    float tc = timeit([&] { k_copy<<<g, 256>>>((const float4*)A, (float4*)B, n4); });
    float tm = timeit([&] { CK(hipMemsetAsync(A, 0, bytes, 0)); });
    printf("%4zu MB: fill %.1f us %.2f TB/s | read %.1f us %.2f TB/s | copy %.1f us %.2f TB/s (r+w) | hipMemset %.1f us %.2f TB/s\n",
           bytes / MB, tf, bytes / tf * 1e-6, tr, bytes / tr * 1e-6, tc, 2.0 * bytes / tc * 1e-6, tm, bytes / tm * 1e-6);
  }
  // scattered accesses: n accesses; window = region (in elements) each group of 4096 consecutive accesses falls in
  const size_t n = 4800000;
  std::vector<uint32_t> h(n);
  uint32_t* didx; CK(hipMalloc(&didx, n * 4));
  struct Case { const char* name; size_t window; size_t total; };  // element counts
  Case cases[] = {{"window 4096 elems (tile slice)", 4096, n}, {"window 262144 elems (1 MB table) per 300k", 262144, 0},
                  {"window 300000 elems per 300k (frame)", 300000, 1}};
  for (auto& c : cases) {
    srand(3);
    for (size_t i = 0; i < n; ++i) {
      size_t blk = c.total == n ? i / 4096 : i / 300000;
      size_t off = c.total == n ? blk * 4096 : blk * c.window;
      h[i] = (uint32_t)(off + (size_t)rand() % c.window);
    }
    CK(hipMemcpy(didx, h.data(), n * 4, hipMemcpyHostToDevice));
    unsigned g = (unsigned)((n + 1023) / 1024);
    float t0 = timeit([&] { k_scatter<0><<<g, 256>>>(didx, A, sink, n); });
    float t1 = timeit([&] { k_scatter<1><<<g, 256>>>(didx, A, sink, n); });
    float t2 = timeit([&] { k_scatter<2><<<g, 256>>>(didx, A, sink, n); });
    float t3 = timeit([&] { k_scatter<3><<<g, 256>>>(didx, A, sink, n); });
    float t4 = timeit([&] { k_scatter<4><<<g, 256>>>(didx, A, sink, n); });
    printf("%s, %zu accesses: 4B store %.1f us (%.0f G/s) | 4B load %.1f us (%.0f G/s) | 20B store %.1f us (%.0f G/s) | 1B store %.1f us (%.0f G/s) | 8B store %.1f us (%.0f G/s)\n",
           c.name, n, t0, n / t0 * 1e-3, t1, n / t1 * 1e-3, t2, n / t2 * 1e-3, t3, n / t3 * 1e-3, t4, n / t4 * 1e-3);
  }
  // launch overhead: back-to-back trivial kernels
  float te = timeit([&] { for (int i = 0; i < 5; ++i) k_fill<<<256, 256>>>((float4*)A, 1024); });
  printf("5 dependent trivial kernels: %.1f us per sequence (%.2f us each)\n", te, te / 5);
  return 0;
}
