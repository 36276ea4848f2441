// Batched stable LSD radix sort of (uint32 key, uint32 value) pairs, one independent sort per frame.
//
// Used wherever the hot path needs an ORDER-PRESERVING grouping: hard_voxelize (group points by cell,
// keeping input order inside a cell = the reference's sequential scan order), centerpoint_postprocess
// (stable descending score order) and the bev_pool index prep.
//
// Per pass and frame:  tile histogram -> exclusive scan over [digit][tile] -> stable scatter.
// The stable in-tile rank is computed without sorting: every wave ORs its lane bit into an LDS
// bitmask table indexed by digit; rank = popcount of lower lanes (+ lower waves, + earlier rounds).
// Digits are up to 10 bits wide so that a 19-bit pillar key needs two passes.
#pragma once
#include "common.hpp"
#include "scan.hpp"

namespace pd3 {

constexpr int kRsThreads = 256;
constexpr int kRsRounds = 8;
constexpr int kRsTile = kRsThreads * kRsRounds;  // 2048 keys per workgroup
constexpr int kRsMaxBins = 1024;
constexpr int kRsWaves = kRsThreads / kWave;

struct RadixPlan {
  int passes;
  int digit_bits;
  int bins;
  int tiles;  // per frame
};

static inline RadixPlan radix_plan(uint32_t max_key, int64_t n) {
  int bits = 1;
  while (bits < 32 && (max_key >> bits) != 0) ++bits;
  RadixPlan p;
  p.passes = (bits + 9) / 10;
  p.digit_bits = (bits + p.passes - 1) / p.passes;
  p.bins = 1 << p.digit_bits;
  p.tiles = (int)ceil_div(n, kRsTile);
  return p;
}

// ints of scratch per frame for the [digit][tile] table
static inline size_t radix_hist_ints(const RadixPlan& p) { return (size_t)p.bins * p.tiles; }

static __global__ __launch_bounds__(kRsThreads) void rs_hist_kernel(const uint32_t* __restrict__ keys,
                                                             int64_t stride, int64_t n, int shift,
                                                             int bins, int tiles,
                                                             int* __restrict__ hist) {
  __shared__ int h[kRsMaxBins];
  const int frame = blockIdx.y, tile = blockIdx.x;
  for (int d = threadIdx.x; d < bins; d += kRsThreads) h[d] = 0;
  __syncthreads();
  const uint32_t* k = keys + frame * stride;
#pragma unroll
  for (int r = 0; r < kRsRounds; ++r) {
    const int64_t i = (int64_t)tile * kRsTile + r * kRsThreads + threadIdx.x;
    if (i < n) atomicAdd(&h[(k[i] >> shift) & (bins - 1)], 1);
  }
  __syncthreads();
  int* dst = hist + (int64_t)frame * bins * tiles;
  for (int d = threadIdx.x; d < bins; d += kRsThreads) dst[(int64_t)d * tiles + tile] = h[d];
}

// vals_in == nullptr means "value = position" (first pass of an argsort).
static __global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int64_t stride, int64_t n,
    int shift, int bins, int tiles, const int* __restrict__ offs) {
  __shared__ unsigned long long mask[kRsWaves][kRsMaxBins];
  __shared__ int run[kRsMaxBins];
  const int frame = blockIdx.y, tile = blockIdx.x;
  const int lane = lane_id(), wave = wave_id();
  const int* o = offs + (int64_t)frame * bins * tiles;
  for (int d = threadIdx.x; d < bins; d += kRsThreads) {
    run[d] = o[(int64_t)d * tiles + tile];
#pragma unroll
    for (int w = 0; w < kRsWaves; ++w) mask[w][d] = 0ull;
  }
  __syncthreads();
  const uint32_t* kin = keys_in + frame * stride;
  const uint32_t* vin = vals_in ? vals_in + frame * stride : nullptr;
  uint32_t* kout = keys_out + frame * stride;
  uint32_t* vout = vals_out + frame * stride;
  const unsigned long long below_me = (1ull << lane) - 1ull;
  for (int r = 0; r < kRsRounds; ++r) {
    const int64_t i = (int64_t)tile * kRsTile + r * kRsThreads + threadIdx.x;
    const bool valid = i < n;
    uint32_t key = 0, val = 0;
    int dig = 0;
    if (valid) {
      key = kin[i];
      val = vin ? vin[i] : (uint32_t)i;
      dig = (key >> shift) & (bins - 1);
      atomicOr(&mask[wave][dig], 1ull << lane);
    }
    __syncthreads();
    int rank = 0, total = 0, pos = 0;
    if (valid) {
#pragma unroll
      for (int w = 0; w < kRsWaves; ++w) {
        const unsigned long long m = mask[w][dig];
        const int c = __popcll(m);
        total += c;
        if (w < wave) rank += c;
        if (w == wave) rank += __popcll(m & below_me);
      }
      pos = run[dig] + rank;
    }
    __syncthreads();
    if (valid) {
      if (rank == 0) {  // exactly one lane per distinct digit of this round
        run[dig] += total;
#pragma unroll
        for (int w = 0; w < kRsWaves; ++w) mask[w][dig] = 0ull;
      }
      kout[pos] = key;
      vout[pos] = val;
    }
    __syncthreads();
  }
}

// Sorts n pairs per frame.  keys_a/vals_a hold the input (vals_a may be null on entry semantics:
// pass `identity_vals = true` to sort positions).  Ping-pongs between (a) and (b); returns 0 if the
// sorted result ends in (a), 1 if in (b).  hist: batch * radix_hist_ints ints, partial: batch *
// scan_num_tiles(radix_hist_ints) ints.
static inline int enqueue_radix_sort(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b,
                                     uint32_t* vals_b, int64_t stride, int64_t n, int batch,
                                     const RadixPlan& p, bool identity_vals, int* hist,
                                     int* partial, hipStream_t s) {
  dim3 grid(p.tiles, batch);
  const int64_t hist_n = (int64_t)radix_hist_ints(p);
  int cur = 0;
  for (int pass = 0; pass < p.passes; ++pass) {
    const uint32_t* kin = cur ? keys_b : keys_a;
    const uint32_t* vin = cur ? vals_b : vals_a;
    uint32_t* kout = cur ? keys_a : keys_b;
    uint32_t* vout = cur ? vals_a : vals_b;
    const int shift = pass * p.digit_bits;
    rs_hist_kernel<<<grid, kRsThreads, 0, s>>>(kin, stride, n, shift, p.bins, p.tiles, hist);
    enqueue_exclusive_scan(hist, hist_n, hist_n, batch, partial, nullptr, hist, LoadIdentity{},
                           EpiNone{}, s);
    rs_scatter_kernel<<<grid, kRsThreads, 0, s>>>(kin, (pass == 0 && identity_vals) ? nullptr : vin,
                                                  kout, vout, stride, n, shift, p.bins, p.tiles,
                                                  hist);
    cur ^= 1;
  }
  return cur;
}

}  // namespace pd3
