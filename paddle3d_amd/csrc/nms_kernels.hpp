// Device-side greedy NMS shared by iou3d_nms.hip and postprocess.hip.
// (reference: suppression bit matrix nms_kernel iou3d_nms_kernel.cu:310-363, host sweep
//  iou3d_nms.cpp:119-137 -- here both stay on the device, so no mask D2H copy and no host loop.)
//
// Batched over "sets" (grid.z / grid.x): set s has counts[s] boxes (device-side count, <= cap) stored
// at boxes + s * cap * 7, already in score order.  mask layout [set][cap][cb_cap] uint64.
#pragma once
#include "common.hpp"
#include "iou3d_geom.hpp"

#include <algorithm>

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
  // built and consumed per half of the column tile (2048 entries instead of 4096): 25 KB with the vertex lists of
  // box_overlap, six workgroups per CU
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
    __shared__ float poly_s[kPolyWaveFloats];
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
        if (iou_bev(row_pre[r], col_pre[i], poly_s + lane) > thresh) atomicOr(&bits_s[r], 1ull << i);
      }
      __syncthreads();
    }
    if (lane < row_size) mask[((int64_t)set * cap + row_blk * 64 + lane) * cb_cap + col_blk] = bits_s[lane];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Pooled form of the rotated-box bit matrix for the batched callers (all sets of a call, BoxPre records prepared).
// A tile of the matrix holds 4096 pairs of which 10-20 pass the circle test on typical inputs, so the tile form
// above spends its polygon clip -- a few thousand instructions -- on a wave with a quarter of its lanes alive, 136
// times per set.  Here the cheap test and the clip are separate launches:
//   nms_cand_kernel   one wave per tile: lane = row keeps the tile's candidates as a 64-bit word (columns arrive
//                     by v_readlane), zeroes the tile's words of the matrix and appends its candidate pairs
//                     (row, column) to the set's pool: one atomic per tile on the SET's counter (one counter for
//                     the whole call made 13 k same-address atomics of them: 10 ns each, 136 us);
//   nms_pairs_kernel  one lane per pooled pair, full waves: clip, compare with the threshold, OR the bit into the
//                     matrix (bits are independent, so the order of the pool does not matter: same matrix).
// A tile whose pairs do not fit what is left of its set's pool is put on the set's list of unpooled tiles instead,
// and the pair kernel walks those tiles pair by pair; the pool holds 32 candidates per box.
constexpr uint32_t kNmsPoolHole = 0xffffffffu;  // row = col = 65535: not a pair (col > row in every real entry)
constexpr int kNmsPoolPerBox = 32;
constexpr int kNmsPairWgs = 8;  // workgroups of the pair kernel per set
constexpr int kNmsCtrStride = 64;  // ints between two counters (256 bytes)

struct NmsPool {
  float4* xyr;       // [sets][cap]: centre x, y and circumscribed radius of every box (BoxPre's, packed: the candidate
                     // pass reads 16 contiguous bytes per box instead of three floats out of a 64-byte record)
  uint32_t* pairs;   // [sets][per_set]: row << 16 | col
  int* counts;       // [sets] pairs appended, then [sets] unpooled tiles, one counter per kNmsCtrStride ints (counters of
                     // different sets in one cache line serialise in one L2 atomic unit): zeroed before every call
  uint32_t* tiles;   // [sets][cb * cb]: row block << 16 | column block of the unpooled tiles
  int per_set;
};

static inline int nms_pool_per_set(int cap) {
  return (int)std::min<int64_t>(std::max<int64_t>((int64_t)cap * kNmsPoolPerBox, 4096), (int64_t)1 << 24);
}

static __global__ __launch_bounds__(64) void nms_cand_kernel(const int* __restrict__ counts, int sets, int cap,
                                                             int cb_cap, unsigned long long* __restrict__ mask,
                                                             NmsPool pl) {
  // grid.x walks the upper triangle of the cb_cap x cb_cap tiles row by row (a square grid started as many waves
  // again only to end them)
  const int set = blockIdx.y;
  const int n = min(counts[set], cap);
  int row_blk = 0, col_blk = blockIdx.x;
  for (int len = cb_cap; col_blk >= len; --len) {  // row r holds cb_cap - r tiles
    col_blk -= len;
    ++row_blk;
  }
  col_blk += row_blk;
  if (row_blk * 64 >= n || col_blk * 64 >= n) return;
  const int lane = threadIdx.x;
  const float4* ps = pl.xyr + (int64_t)set * cap;
  const int row = row_blk * 64 + lane, col0 = col_blk * 64;
  const int col_size = min(n - col0, 64);
  const bool live = row < n;
  const float4 me = live ? ps[row] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float mx = me.x, my = me.y, mr = me.z;
  // a column's centre and radius are the same for all lanes: read through the scalar cache (wave-uniform index into a
  // read-only array -> s_load), they are SGPR operands of the test and cost no vector instruction (round 5 moved them
  // lane to lane with three v_readlane per column: a quarter of the loop).  All 64 columns are tested; the columns
  // past the set's last box and, in a diagonal tile, the columns up to the row itself are masked off afterwards
  unsigned long long bits = 0ull;
#pragma unroll 1
  for (int g8 = 0; g8 < 64; g8 += 8) {  // eight columns per trip: their 32 scalars fit the SGPR file without spills
    unsigned byte = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 cc = ps[min(col0 + g8 + j, cap - 1)];
      const float dx = mx - cc.x, dy = my - cc.y, r = mr + cc.z + 0.25f;
      const bool cand = !(dx * dx + dy * dy > r * r);  // the same test box_overlap starts with
      byte |= cand ? 1u << j : 0u;
    }
    bits |= (unsigned long long)byte << g8;
  }
  if (col_size < 64) bits &= (1ull << col_size) - 1ull;
  if (row_blk == col_blk) bits &= lane < 63 ? ~0ull << (lane + 1) : 0ull;
  if (!live) bits = 0ull;
  if (live && row_blk == col_blk) {  // the diagonal tile clears its rows of the matrix (contiguous words per lane)
    unsigned long long* mrow = mask + ((int64_t)set * cap + row) * cb_cap;
    for (int k = 0; k < cb_cap; ++k) mrow[k] = 0ull;
  }
  const int mine = __popcll(bits);
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  const int total = __shfl(incl, 63, 64);
  if (total == 0) return;
  int base = 0;
  if (lane == 0) base = atomicAdd(&pl.counts[set * kNmsCtrStride], total);
  base = __shfl(base, 0, 64);
  uint32_t* pool = pl.pairs + (int64_t)set * pl.per_set;
  int pos = base + incl - mine;
  if (base + total <= pl.per_set) {
    const uint32_t head = (uint32_t)row << 16;
    while (bits) {
      const int i = __builtin_ctzll(bits);
      bits &= bits - 1;
      pool[pos++] = head | (uint32_t)(col0 + i);
    }
  } else {
    // what was reserved inside the pool stays without pairs: mark it (the pair kernel skips holes) and hand the
    // tile over as a whole
    for (int k = 0; k < mine; ++k, ++pos)
      if (pos < pl.per_set) pool[pos] = kNmsPoolHole;
    if (lane == 0) {
      const int k = atomicAdd(&pl.counts[(sets + set) * kNmsCtrStride], 1);
      pl.tiles[(int64_t)set * cb_cap * cb_cap + k] = ((uint32_t)row_blk << 16) | (uint32_t)col_blk;
    }
  }
}

// grid (kNmsPairWgs, sets)
static __global__ __launch_bounds__(256) void nms_pairs_kernel(const BoxPre* __restrict__ pre,
                                                               const int* __restrict__ counts, int sets, int cap,
                                                               int cb_cap, float thresh,
                                                               unsigned long long* __restrict__ mask, NmsPool pl) {
  __shared__ float poly_s[4 * kPolyWaveFloats];
  float* st = poly_s + wave_id() * kPolyWaveFloats + lane_id();
  const int set = blockIdx.y;
  const BoxPre* ps = pre + (int64_t)set * cap;
  unsigned long long* ms = mask + (int64_t)set * cap * cb_cap;
  const int total = min(pl.counts[set * kNmsCtrStride], pl.per_set);
  const uint32_t* pool = pl.pairs + (int64_t)set * pl.per_set;
  const int stride = gridDim.x * 256, first = blockIdx.x * 256 + threadIdx.x;
  for (int p = first; p < total; p += stride) {
    const uint32_t e = pool[p];
    if (e == kNmsPoolHole) continue;
    const int row = (int)(e >> 16), col = (int)(e & 0xffffu);
    const BoxPre a = ps[row], b = ps[col];
    if (iou_bev(a, b, st) > thresh) atomicOr(&ms[(int64_t)row * cb_cap + (col >> 6)], 1ull << (col & 63));
  }
  // tiles that did not fit the pool: every pair of the tile through the same circle test, then the clip
  const int ntiles = pl.counts[(sets + set) * kNmsCtrStride];
  if (ntiles == 0) return;
  const int n = min(counts[set], cap);
  for (int t = 0; t < ntiles; ++t) {
    const uint32_t tile = pl.tiles[(int64_t)set * cb_cap * cb_cap + t];
    const int r0 = (int)(tile >> 16) * 64, c0 = (int)(tile & 0xffffu) * 64;
    for (int idx = first; idx < 4096; idx += stride) {
      const int row = r0 + (idx >> 6), col = c0 + (idx & 63);
      if (row >= n || col >= n || col <= row) continue;
      const BoxPre a = ps[row], b = ps[col];
      const float dx = a.cx - b.cx, dy = a.cy - b.cy, r = a.rad + b.rad + 0.25f;
      if (dx * dx + dy * dy > r * r) continue;
      if (iou_bev(a, b, st) > thresh) atomicOr(&ms[(int64_t)row * cb_cap + (col >> 6)], 1ull << (col & 63));
    }
  }
}

// The rotated-box bit matrix of `sets` sets (counts on the device): candidate pass, then the pooled pairs.
static inline void nms_enqueue_mask_pooled(const BoxPre* pre, const int* counts, int sets, int cap, int cb, float thresh,
                                           unsigned long long* mask, const NmsPool& pl, hipStream_t s) {
  nms_cand_kernel<<<dim3(cb * (cb + 1) / 2, sets), 64, 0, s>>>(counts, sets, cap, cb, mask, pl);
  nms_pairs_kernel<<<dim3(kNmsPairWgs, sets), 256, 0, s>>>(pre, counts, sets, cap, cb, thresh, mask, pl);
}

// One workgroup per set.  keep [set][cap] receives kept indices in order; num_keep[set] their number.
// The greedy sweep is inherently serial over the boxes; what can be removed is the memory latency: for
// sets of up to kNmsLdsBoxes boxes the needed half of the bit matrix is copied into LDS by all 256 threads
// first (one bulk round trip), then wave 0 sweeps 64 boxes per step entirely out of LDS.
constexpr int kNmsSweepThreads = 1024;  // all of them copy the bit matrix (one bulk round trip), wave 0 then sweeps
constexpr int kNmsLdsBoxes = 1024;
constexpr int kNmsLdsWords = kNmsLdsBoxes / 64;

static __global__ __launch_bounds__(kNmsSweepThreads) void nms_sweep_kernel(const unsigned long long* __restrict__ mask,
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
    // The serial resolve of the diagonal word, on the SCALAR unit with constant lane indices (round 5: the rolled loop
    // with a runtime trip count cost ~80 cycles per row).  One wave alone on its CU issues an instruction every four or
    // five cycles, so what counts is the NUMBER of instructions per row (round 6, cycle stamps: 53 cycles per row = nine
    // instructions).  Row t only has bits above t (upper triangle), so (a) bit t of `cur` is final once row t - 1 is
    // done: the kept rows are simply ~cur at the end, no per-row bookkeeping; (b) rows 0..31 decide on the low word
    // alone and rows 32..63 on the high word alone (their low words are zero): 32-bit tests and ORs, the high words of
    // the kept rows 0..31 collected on the side.  Seven instructions per row for the first half, four for the second.
    // Rows past `rows` carry a zero word and are masked out of keepbits afterwards.
    const unsigned long long cur0 = remv[nb];  // uniform
    // (the builtin returns int: through `unsigned` first, or the low word's bit 31 sign-extends into the high word)
    unsigned cur_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(cur0 >> 32));
    unsigned cur_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(cur0 & 0xffffffffull));
    const unsigned dlo = (unsigned)(diag & 0xffffffffull), dhi = (unsigned)(diag >> 32);
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      const unsigned lo = __builtin_amdgcn_readlane(dlo, t);
      const unsigned hi = __builtin_amdgcn_readlane(dhi, t);
      const bool keep = !((cur_lo >> t) & 1u);
      cur_lo |= keep ? lo : 0u;
      cur_hi |= keep ? hi : 0u;
    }
#pragma unroll
    for (int t = 32; t < 64; ++t) {
      const unsigned hi = __builtin_amdgcn_readlane(dhi, t);
      const bool keep = !((cur_hi >> (t - 32)) & 1u);
      cur_hi |= keep ? hi : 0u;
    }
    unsigned long long keepbits = ~(((unsigned long long)cur_hi << 32) | (unsigned long long)cur_lo);
    keepbits &= rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
    // append kept rows in order
    if (lane < rows && ((keepbits >> lane) & 1ull))
      kp[kept_total + __popcll(keepbits & ((1ull << lane) - 1ull))] = nb * 64 + lane;
    kept_total += __popcll(keepbits);
    // OR the kept rows' words into the later column blocks
    if (in_lds) {
      // cbs <= 16 here: lane = (row group g, word jj); every lane ORs 16 of the block's 64 rows (independent LDS loads,
      // pipelined; rows that were not kept -- or lie past the set's last box: their keep bit is zero, and the LDS area
      // is sized for whole blocks of 64 rows -- contribute 0: the row's keep bit as an all-ones / all-zeros word, one
      // and-or per half), the four groups are combined across the wave
      const int g = lane >> 4, j = nb + 1 + (lane & 15);
      unsigned lo = 0u, hi = 0u;
      if (j < cbs) {
        const int k16 = (int)((keepbits >> (g * 16)) & 0xffffull);
        const unsigned long long* src = mlds + (nb * 64 + g * 16) * cbs + j;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const unsigned long long v = src[t * cbs];
          const unsigned m = (unsigned)-((k16 >> t) & 1);
          lo |= (unsigned)(v & 0xffffffffull) & m;
          hi |= (unsigned)(v >> 32) & m;
        }
      }
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
  const int cb = (c + 63) / 64;
  return (size_t)kNmsMaxWords * 8 + (size_t)cb * 64 * cb * 8;  // whole blocks of 64 rows (the sweep's OR step reads them all)
}

}  // namespace pd3
