// Batched exclusive prefix sum over int32 (one independent scan per frame, grid.y = frame).
//
// Three launches: tile sums -> scan of tile sums (one workgroup per frame) -> rescan + offset.
// A tile is 256 threads x 16 consecutive ints per thread (64 B per lane: four dwordx4 loads).
// `Load` maps the stored int to the scanned value (identity, or "is non-negative" for flag arrays);
// `Epi` is called once per element with (frame, index, loaded value, exclusive prefix) so that the
// consumer of the prefix can be fused into the last pass instead of re-reading it.
#pragma once
#include "common.hpp"

namespace pd3 {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;
constexpr int kScanTile = kScanThreads * kScanItems;  // 4096

struct LoadIdentity {
  __device__ __forceinline__ int operator()(int v) const { return v; }
};
struct LoadNonNegative {
  __device__ __forceinline__ int operator()(int v) const { return v >= 0 ? 1 : 0; }
};
struct EpiNone {
  __device__ __forceinline__ void operator()(int, int64_t, int, int, int) const {}
};

static inline int scan_num_tiles(int64_t n) { return (int)ceil_div(n, kScanTile); }

template <typename Load>
__global__ __launch_bounds__(kScanThreads) void scan_reduce_kernel(const int* __restrict__ in,
                                                                   int64_t stride, int64_t n,
                                                                   int* __restrict__ partial,
                                                                   Load load) {
  __shared__ int smem[kScanThreads / kWave + 1];
  const int frame = blockIdx.y;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  const int* src = in + frame * stride;
  int sum = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = base + j;
    if (i < n) sum += load(src[i]);
  }
  int total;
  (void)block_exclusive_scan<kScanThreads>(sum, smem, total);
  if (threadIdx.x == 0) partial[(int64_t)frame * gridDim.x + blockIdx.x] = total;
}

// One workgroup of 1024 threads per frame scans the tile sums in place; totals[frame] = grand total.
static __global__ __launch_bounds__(1024) void scan_partials_kernel(int* __restrict__ partial, int ntiles,
                                                             int* __restrict__ totals) {
  __shared__ int smem[1024 / kWave + 1];
  int* p = partial + (int64_t)blockIdx.x * ntiles;
  const int per = (int)ceil_div(ntiles, 1024);
  const int lo = threadIdx.x * per;
  int sum = 0;
  for (int j = 0; j < per; ++j)
    if (lo + j < ntiles) sum += p[lo + j];
  int total;
  int run = block_exclusive_scan<1024>(sum, smem, total);
  for (int j = 0; j < per; ++j)
    if (lo + j < ntiles) {
      const int v = p[lo + j];
      p[lo + j] = run;
      run += v;
    }
  if (threadIdx.x == 0 && totals) totals[blockIdx.x] = total;
}

template <typename Load, typename Epi>
__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(const int* __restrict__ in,
                                                                  int64_t stride, int64_t n,
                                                                  const int* __restrict__ partial,
                                                                  int* __restrict__ out, Load load,
                                                                  Epi epi) {
  __shared__ int smem[kScanThreads / kWave + 1];
  const int frame = blockIdx.y;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  const int* src = in + frame * stride;
  int v[kScanItems];
  int sum = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = base + j;
    v[j] = (i < n) ? load(src[i]) : 0;
    sum += v[j];
  }
  int total;
  int run = block_exclusive_scan<kScanThreads>(sum, smem, total) +
            partial[(int64_t)frame * gridDim.x + blockIdx.x];
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = base + j;
    if (i < n) {
      if (out) out[frame * stride + i] = run;
      epi(frame, i, v[j], run, 0);
    }
    run += v[j];
  }
}

// Enqueue the three launches.  `partial` needs batch * scan_num_tiles(n) ints; `totals` (nullable)
// batch ints.  `out` may alias `in` or be null when only the epilogue consumes the prefix.
template <typename Load, typename Epi>
static inline void enqueue_exclusive_scan(const int* in, int64_t stride, int64_t n, int batch,
                                          int* partial, int* totals, int* out, Load load, Epi epi,
                                          hipStream_t s) {
  const int nt = scan_num_tiles(n);
  dim3 grid(nt, batch);
  scan_reduce_kernel<Load><<<grid, kScanThreads, 0, s>>>(in, stride, n, partial, load);
  scan_partials_kernel<<<batch, 1024, 0, s>>>(partial, nt, totals);
  scan_apply_kernel<Load, Epi><<<grid, kScanThreads, 0, s>>>(in, stride, n, partial, out, load, epi);
}

}  // namespace pd3
