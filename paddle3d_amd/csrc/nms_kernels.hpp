// Device-side greedy NMS shared by iou3d_nms.hip and postprocess.hip.
// (reference: suppression bit matrix nms_kernel iou3d_nms_kernel.cu:310-363, host sweep
//  iou3d_nms.cpp:119-137 -- here both stay on the device, so no mask D2H copy and no host loop.)
//
// Batched over "sets" (grid.z / grid.x): set s has counts[s] boxes (device-side count, <= cap) stored
// at boxes + s * cap * 7, already in score order.  mask layout [set][cap][cb_cap] uint64.
#pragma once
#include "common.hpp"
#include "iou3d_geom.hpp"

namespace pd3 {

constexpr int kNmsMaxWords = 1024;  // sweep supports up to 65536 boxes per set

// grid (cb_cap, cb_cap, sets), 64 threads (one wave): tile (row block, col block) of the bit matrix.
// Only tiles with col >= row are needed by the sweep; the others are skipped.
template <bool NORMAL>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes,
                                                      const int* __restrict__ counts, int n_fixed,
                                                      int cap, int cb_cap, float thresh,
                                                      unsigned long long* __restrict__ mask) {
  const int set = blockIdx.z;
  const int n = counts ? min(counts[set], cap) : n_fixed;
  const int row_blk = blockIdx.y, col_blk = blockIdx.x;
  if (col_blk < row_blk) return;
  if (row_blk * 64 >= n || col_blk * 64 >= n) return;
  const float* bx = boxes + (int64_t)set * cap * 7;
  const int col_size = min(n - col_blk * 64, 64);
  const int row_size = min(n - row_blk * 64, 64);
  const int lane = threadIdx.x;

  __shared__ BoxPre col_pre[64];
  __shared__ float col_raw[64 * 7];
  if (lane < col_size) {
    const float* b = bx + (int64_t)(col_blk * 64 + lane) * 7;
    if (NORMAL) {
#pragma unroll
      for (int k = 0; k < 7; ++k) col_raw[lane * 7 + k] = b[k];
    } else {
      col_pre[lane] = box_prepare(b);
    }
  }
  __syncthreads();
  if (lane < row_size) {
    const int row = row_blk * 64 + lane;
    const float* b = bx + (int64_t)row * 7;
    unsigned long long bits = 0ull;
    const int start = (row_blk == col_blk) ? lane + 1 : 0;
    if (NORMAL) {
      float me[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) me[k] = b[k];
      for (int i = start; i < col_size; ++i)
        if (iou_normal(me, col_raw + i * 7) > thresh) bits |= 1ull << i;
    } else {
      const BoxPre me = box_prepare(b);
      for (int i = start; i < col_size; ++i)
        if (iou_bev(me, col_pre[i]) > thresh) bits |= 1ull << i;
    }
    mask[((int64_t)set * cap + row) * cb_cap + col_blk] = bits;
  }
}

// One wave per set.  keep [set][cap] receives kept indices in order; num_keep[set] their number.
static __global__ __launch_bounds__(64) void nms_sweep_kernel(const unsigned long long* __restrict__ mask,
                                                       const int* __restrict__ counts, int n_fixed,
                                                       int cap, int cb_cap,
                                                       int32_t* __restrict__ keep,
                                                       int32_t* __restrict__ num_keep) {
  __shared__ unsigned long long remv[kNmsMaxWords];
  const int set = blockIdx.x;
  const int n = counts ? min(counts[set], cap) : n_fixed;
  const int lane = threadIdx.x;
  const int cbs = (n + 63) / 64;
  const unsigned long long* m = mask + (int64_t)set * cap * cb_cap;
  int32_t* kp = keep + (int64_t)set * cap;
  for (int j = lane; j < cbs; j += 64) remv[j] = 0ull;
  __syncthreads();
  int kept_total = 0;
  for (int nb = 0; nb < cbs; ++nb) {
    const int rows = min(n - nb * 64, 64);
    // diagonal word of each row of this block
    unsigned long long diag = 0ull;
    if (lane < rows) diag = m[(int64_t)(nb * 64 + lane) * cb_cap + nb];
    unsigned long long cur = remv[nb];  // uniform
    unsigned long long keepbits = 0ull;
    for (int t = 0; t < rows; ++t) {
      const unsigned lo = __builtin_amdgcn_readlane((unsigned)(diag & 0xffffffffull), t);
      const unsigned hi = __builtin_amdgcn_readlane((unsigned)(diag >> 32), t);
      if (!((cur >> t) & 1ull)) {
        keepbits |= 1ull << t;
        cur |= ((unsigned long long)hi << 32) | lo;
      }
    }
    // append kept rows in order
    if (lane < rows && ((keepbits >> lane) & 1ull))
      kp[kept_total + __popcll(keepbits & ((1ull << lane) - 1ull))] = nb * 64 + lane;
    kept_total += __popcll(keepbits);
    // OR the kept rows' words into the later column blocks
    for (int j = nb + 1 + lane; j < cbs; j += 64) {
      unsigned long long acc = remv[j];
      unsigned long long kb = keepbits;
      while (kb) {
        const int t = __ffsll((long long)kb) - 1;
        kb &= kb - 1ull;
        acc |= m[(int64_t)(nb * 64 + t) * cb_cap + j];
      }
      remv[j] = acc;
    }
    __syncthreads();
  }
  if (lane == 0) num_keep[set] = kept_total;
}

}  // namespace pd3
