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
//
// Two phases per tile.  (1) every lane = one row walks the tile's columns with the cheap exact early-out
// (disjoint circumscribed circles => the reference's overlap is exactly 0) and pushes the surviving
// (row, col) pairs into an LDS list with a wave-aggregated append.  (2) the list is processed one pair per
// lane: the expensive polygon clip only runs for the ~1-2 % of pairs that can overlap, with all lanes busy,
// instead of diverging inside a 64-step column loop.  Same per-pair arithmetic as before => same bits.
// `pre` (optional): the boxes' BoxPre records, prepared once per box by the kernel that lays out the NMS boxes (two
// fp64 sin / cos pairs each) instead of once per tile that touches the box -- a set of 1000 boxes has 136 tiles, every
// box would be prepared 17 times.
template <bool NORMAL>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes,
                                                      const int* __restrict__ counts, int n_fixed,
                                                      int cap, int cb_cap, float thresh,
                                                      unsigned long long* __restrict__ mask,
                                                      const BoxPre* __restrict__ pre = nullptr) {
  const int set = blockIdx.z;
  const int n = counts ? min(counts[set], cap) : n_fixed;
  const int row_blk = blockIdx.y, col_blk = blockIdx.x;
  if (col_blk < row_blk) return;
  if (row_blk * 64 >= n || col_blk * 64 >= n) return;
  const float* bx = boxes + (int64_t)set * cap * 7;
  const int col_size = min(n - col_blk * 64, 64);
  const int row_size = min(n - row_blk * 64, 64);
  const int lane = threadIdx.x;

  // LDS per single-wave workgroup decides how many of them a CU holds (the kernel is a serial candidate walk per
  // wave: occupancy is its throughput): the arrays of the other variant are not declared, and the candidate list is
  // built and consumed per half of the column tile (2048 entries instead of 4096): 12.8 KB, 12 workgroups per CU
  __shared__ unsigned long long bits_s[64];
  bits_s[lane] = 0ull;
  if constexpr (NORMAL) {
    __shared__ float col_raw[64 * 7];
    if (lane < col_size) {
      const float* b = bx + (int64_t)(col_blk * 64 + lane) * 7;
#pragma unroll
      for (int k = 0; k < 7; ++k) col_raw[lane * 7 + k] = b[k];
    }
    __syncthreads();
    if (lane < row_size) {
      const float* b = bx + (int64_t)(row_blk * 64 + lane) * 7;
      float me[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) me[k] = b[k];
      unsigned long long bits = 0ull;
      const int start = (row_blk == col_blk) ? lane + 1 : 0;
      for (int i = start; i < col_size; ++i)
        if (iou_normal(me, col_raw + i * 7) > thresh) bits |= 1ull << i;
      mask[((int64_t)set * cap + row_blk * 64 + lane) * cb_cap + col_blk] = bits;
    }
  } else {
    __shared__ BoxPre col_pre[64];
    __shared__ BoxPre row_pre[64];
    __shared__ unsigned short pairs[64 * 32];
    if (lane < col_size)
      col_pre[lane] = pre ? pre[(int64_t)set * cap + col_blk * 64 + lane]
                          : box_prepare(bx + (int64_t)(col_blk * 64 + lane) * 7);
    if (lane < row_size)
      row_pre[lane] = pre ? pre[(int64_t)set * cap + row_blk * 64 + lane]
                          : box_prepare(bx + (int64_t)(row_blk * 64 + lane) * 7);
    __syncthreads();
    const bool live = lane < row_size;
    const float mx = live ? row_pre[lane].cx : 0.f, my = live ? row_pre[lane].cy : 0.f;
    const float mr = live ? row_pre[lane].rad : 0.f;
    const int start = (row_blk == col_blk) ? lane + 1 : 0;
    for (int c0 = 0; c0 < col_size; c0 += 32) {
      // phase 1: candidate pairs of this half of the columns
      int npairs = 0;
      const int c1 = min(col_size, c0 + 32);
      for (int i = c0; i < c1; ++i) {
        bool cand = false;
        if (live && i >= start) {
          const float dx = mx - col_pre[i].cx, dy = my - col_pre[i].cy, r = mr + col_pre[i].rad + 0.25f;
          cand = !(dx * dx + dy * dy > r * r);  // the same test box_overlap starts with
        }
        const unsigned long long m = __ballot(cand);
        if (cand) pairs[npairs + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)((lane << 6) | i);
        npairs += __popcll(m);
      }
      __syncthreads();
      // phase 2: one candidate pair per lane
      for (int p = lane; p < npairs; p += 64) {
        const int r = pairs[p] >> 6, i = pairs[p] & 63;
        if (iou_bev(row_pre[r], col_pre[i]) > thresh) atomicOr(&bits_s[r], 1ull << i);
      }
      __syncthreads();
    }
    if (lane < row_size) mask[((int64_t)set * cap + row_blk * 64 + lane) * cb_cap + col_blk] = bits_s[lane];
  }
}

// One workgroup per set.  keep [set][cap] receives kept indices in order; num_keep[set] their number.
// The greedy sweep is inherently serial over the boxes; what can be removed is the memory latency: for
// sets of up to kNmsLdsBoxes boxes the needed half of the bit matrix is copied into LDS by all 256 threads
// first (one bulk round trip), then wave 0 sweeps 64 boxes per step entirely out of LDS.
constexpr int kNmsLdsBoxes = 1024;
constexpr int kNmsLdsWords = kNmsLdsBoxes / 64;

static __global__ __launch_bounds__(256) void nms_sweep_kernel(const unsigned long long* __restrict__ mask,
                                                               const int* __restrict__ counts, int n_fixed,
                                                               int cap, int cb_cap,
                                                               int32_t* __restrict__ keep,
                                                               int32_t* __restrict__ num_keep) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long nms_smem[];
  unsigned long long* remv = nms_smem;               // [kNmsMaxWords]
  unsigned long long* mlds = nms_smem + kNmsMaxWords;  // [n][cbs] when the set fits
  const int set = blockIdx.x;
  const int n = counts ? min(counts[set], cap) : n_fixed;
  const int lane = threadIdx.x & 63;
  const int cbs = (n + 63) / 64;
  const unsigned long long* m = mask + (int64_t)set * cap * cb_cap;
  int32_t* kp = keep + (int64_t)set * cap;
  const bool in_lds = n <= kNmsLdsBoxes;
  for (int j = threadIdx.x; j < cbs; j += blockDim.x) remv[j] = 0ull;
  if (in_lds) {
    if (cbs == cb_cap) {  // full-width rows: one linear copy, 16 bytes per load where the set's block is aligned
      const int total = n * cbs;
      if ((reinterpret_cast<uintptr_t>(m) & 15u) == 0) {
        const ulonglong2* m2 = reinterpret_cast<const ulonglong2*>(m);
        ulonglong2* l2 = reinterpret_cast<ulonglong2*>(mlds);  // mlds = nms_smem + 1024 words: 16-byte aligned
#pragma unroll 4
        for (int e = threadIdx.x; e < (total >> 1); e += blockDim.x) l2[e] = m2[e];
        if ((total & 1) && threadIdx.x == 0) mlds[total - 1] = m[total - 1];
      } else {
#pragma unroll 4
        for (int e = threadIdx.x; e < total; e += blockDim.x) mlds[e] = m[e];
      }
    } else {
      for (int e = threadIdx.x; e < n * cbs; e += blockDim.x) {
        const int i = e / cbs, j = e - i * cbs;
        if (j >= (i >> 6)) mlds[e] = m[(int64_t)i * cb_cap + j];  // upper triangle only
      }
    }
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  int kept_total = 0;
  for (int nb = 0; nb < cbs; ++nb) {
    const int rows = min(n - nb * 64, 64);
    // diagonal word of each row of this block
    unsigned long long diag = 0ull;
    if (lane < rows)
      diag = in_lds ? mlds[(nb * 64 + lane) * cbs + nb] : m[(int64_t)(nb * 64 + lane) * cb_cap + nb];
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
    if (in_lds) {
      // cbs <= 16 here: lane = (row group g, word jj); every lane ORs 16 of the block's 64 rows (independent LDS loads,
      // pipelined; rows that were not kept contribute 0), the four groups are combined across the wave
      const int g = lane >> 4, j = nb + 1 + (lane & 15);
      unsigned long long acc = 0ull;
      if (j < cbs) {
#pragma unroll 8
        for (int t = g * 16; t < min(g * 16 + 16, rows); ++t) {
          const unsigned long long v = mlds[(nb * 64 + t) * cbs + j];
          acc |= ((keepbits >> t) & 1ull) ? v : 0ull;
        }
      }
      unsigned lo = (unsigned)(acc & 0xffffffffull), hi = (unsigned)(acc >> 32);
      lo |= (unsigned)__shfl_xor((int)lo, 16, 64);
      hi |= (unsigned)__shfl_xor((int)hi, 16, 64);
      lo |= (unsigned)__shfl_xor((int)lo, 32, 64);
      hi |= (unsigned)__shfl_xor((int)hi, 32, 64);
      if (g == 0 && j < cbs) remv[j] |= ((unsigned long long)hi << 32) | lo;
    }
    for (int j = nb + 1 + lane; !in_lds && j < cbs; j += 64) {
      unsigned long long acc = remv[j];
      {
        unsigned long long kb = keepbits;
        while (kb) {
          const int t = __ffsll((long long)kb) - 1;
          kb &= kb - 1ull;
          acc |= m[(int64_t)(nb * 64 + t) * cb_cap + j];
        }
      }
      remv[j] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) num_keep[set] = kept_total;
}

// dynamic LDS bytes for nms_sweep_kernel given the capacity of a set.  The kernel takes the LDS path for every
// set of <= kNmsLdsBoxes boxes whatever the capacity (a launch with nms_pre_max_size = 4096 still has sets of a
// few hundred candidates), so the matrix area is sized for min(cap, kNmsLdsBoxes) boxes.
static inline size_t nms_sweep_lds(int cap) {
  const int c = cap < kNmsLdsBoxes ? cap : kNmsLdsBoxes;
  return (size_t)kNmsMaxWords * 8 + (size_t)c * ((c + 63) / 64) * 8;
}

}  // namespace pd3
