// centerpoint_postprocess for gfx950: all tasks of a frame in one launch sequence, no host round trip.
// (reference: paddle3d/ops/centerpoint_postprocess/postprocess.cu:104-280 postprocess_gpu, decode_kernel
//  :32-80, private NMS iou3d_nms_kernel.cu:274-352.)
//
// The reference runs ~20 Paddle/CUDA launches and two blocking host syncs PER TASK (masked_select's
// numel, the NMS mask copy + host sweep).  Here grid.y / grid.z indexes the task and every count stays
// on the device:
//   1. score              sigmoid->max/argmax, range/score mask on the raw reg / height values, sort key (all cells):
//                         key = 0x3F800000 - bits(score) for masked-in cells (descending score, ties in cell order =
//                         masked_select order + stable argsort), 0x3FFFFFFF otherwise
//   2. top-K selection    the first min(selected, nms_pre_max_size) cells of the stable key order
//   3. decode             those cells: box (exp(dim), atan2), score, class, and the box remapped for the NMS
//                         (dx<->dy, -rot - pi/2) with its per-box records
//      1-3 are ONE kernel, cp_topk_kernel (one workgroup per set: keys computed into LDS, radix select + ordered
//      compaction + bitonic sort of <= 1024 pairs, decode), or, with selection = 1 / maps beyond 16 k cells / a pre-NMS
//      cap beyond 1024, cp_score_kernel + a full stable radix sort of all keys + cp_nms_boxes_kernel -- identical results
//   4. nms_cand_kernel + nms_pairs_kernel + nms_sweep_kernel (nms_kernels.hpp), counts read on the device
//   5. cp_output_kernel   concatenates the tasks' kept rows (or the reference's fake row) in task order
// Work is tiny (4.6 MB read per nuScenes frame); the op is launch-latency bound, which is why the
// launch count (5 for all tasks and frames of a batch) and the absence of syncs are what matter.
#include "../../include/paddle3d_amd.h"
#include "bf16x3.hpp"
#include "common.hpp"
#include "nms_kernels.hpp"
#include "radix_sort.hpp"

#include <algorithm>
#include <cstring>

namespace pd3 {

constexpr int kMaxTasks = 16;
constexpr uint32_t kKeyOut = 0x3FFFFFFFu;   // sorts after every selected cell
constexpr uint32_t kKeyOne = 0x3F800000u;   // bits of 1.0f

struct CpHeads {
  const float* hm[kMaxTasks];
  const float* reg[kMaxTasks];
  const float* height[kMaxTasks];
  const float* dim[kMaxTasks];
  const float* vel[kMaxTasks];
  const float* rot[kMaxTasks];
  int ncls[kMaxTasks];
  int label_offset[kMaxTasks];
  int64_t batch_stride;  // elements between two frames of EVERY head tensor (views into one map); 0 = contiguous
};

struct CpCfg {
  int hw, feat_w, dims, with_velocity, num_tasks;
  float down_ratio, vx, vy, pc_x, pc_y;
  float r[6];  // post_center_range
  float score_threshold;
};

__device__ __forceinline__ float exp_rn(float x) { return lm::expf(x); }  // glibc's bits (libm_exact.hpp)

// postprocess.cu:145-149  sigmoid, then max / argmax over the class axis (first maximum wins)
__device__ __forceinline__ float cp_best_class(const CpHeads& h, const CpCfg& c, int t, int frame, int i, int& arg) {
  const int64_t bs = h.batch_stride;
  const float* hm = h.hm[t] + (int64_t)frame * (bs ? bs : (int64_t)h.ncls[t] * c.hw);
  float best = 0.f;
  arg = 0;
  for (int k = 0; k < h.ncls[t]; ++k) {
    const float s = 1.0f / (1.0f + exp_rn(-hm[(int64_t)k * c.hw + i]));
    if (k == 0 || s > best) {
      best = s;
      arg = k;
    }
  }
  return best;
}

// Score, the mask of postprocess.cu:72-77 on the RAW reg / height values and the sort key of one cell (kKeyOut: masked
// out).  Nothing else is computed for the 16 k cells of a set: a cell's box is only ever read if the cell is among the
// nms_pre_max_size best of its set, and those are decoded by cp_decode_row -- a few hundred to a thousand cells per set
// (the box, three exp and an atan2 each, and its 36-byte row were 90 MB of writes per 16 frames).
__device__ __forceinline__ uint32_t cp_cell_key(const CpHeads& h, const CpCfg& c, int t, int frame, int i) {
  const int64_t bs = h.batch_stride;
  const float* regp = h.reg[t] + (int64_t)frame * (bs ? bs : (int64_t)2 * c.hw);
  const float* heip = h.height[t] + (int64_t)frame * (bs ? bs : (int64_t)c.hw);
  int arg;
  const float best = cp_best_class(h, c, t, frame, i, arg);
  const float x = regp[i], y = regp[i + c.hw], z = heip[i];
  const bool m = best > c.score_threshold && x <= c.r[3] && y <= c.r[4] && z <= c.r[5] &&
                 x >= c.r[0] && y >= c.r[1] && z >= c.r[2];
  if (!m) return kKeyOut;
  const uint32_t bits = __float_as_uint(best);
  return bits <= kKeyOne ? kKeyOne - bits : 0u;
}

// The full-sort selection's first pass: the keys of all cells and the number of masked-in cells per set.
__global__ __launch_bounds__(256) void cp_score_kernel(CpHeads h, CpCfg c, uint32_t* __restrict__ keys,
                                                       int* __restrict__ counts) {
  const int set = blockIdx.y;  // frame * num_tasks + task
  const int t = set % c.num_tasks, frame = set / c.num_tasks;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int selected = 0;
  if (i < c.hw) {
    const uint32_t key = cp_cell_key(h, c, t, frame, i);
    selected = key != kKeyOut;
    keys[(int64_t)set * c.hw + i] = key;
  }
  // block count of selected cells -> counts[t]
  const unsigned long long ball = __ballot(selected);
  __shared__ int wsum[4];
  if (lane_id() == 0) wsum[wave_id()] = __popcll(ball);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (s) atomicAdd(&counts[set], s);
  }
}

struct CpRows {  // the decoded candidates of every set, [set][rank]
  float *boxes, *scores;
  int* labels;
  float* nms_boxes;
  BoxPre* pre;
  float4* xyr;
};

// The head values of one cell a box is made of.
struct CpCell {
  float x, y, z, d0, d1, d2, r0, r1, v0, v1;
};

__device__ __forceinline__ CpCell cp_gather_cell(const CpHeads& h, const CpCfg& c, int t, int frame, int i) {
  const int64_t bs = h.batch_stride;
  const float* regp = h.reg[t] + (int64_t)frame * (bs ? bs : (int64_t)2 * c.hw);
  const float* heip = h.height[t] + (int64_t)frame * (bs ? bs : (int64_t)c.hw);
  const float* dimp = h.dim[t] + (int64_t)frame * (bs ? bs : (int64_t)3 * c.hw);
  const float* velp = h.vel[t] + (int64_t)frame * (bs ? bs : (int64_t)2 * c.hw);
  const float* rotp = h.rot[t] + (int64_t)frame * (bs ? bs : (int64_t)2 * c.hw);
  CpCell v;
  v.x = regp[i], v.y = regp[i + c.hw], v.z = heip[i];
  v.d0 = dimp[i], v.d1 = dimp[i + c.hw], v.d2 = dimp[i + 2 * c.hw];
  v.r0 = rotp[i], v.r1 = rotp[i + c.hw];
  v.v0 = v.v1 = 0.f;
  if (c.with_velocity) {
    v.v0 = velp[i];
    v.v1 = velp[i + c.hw];
  }
  return v;
}

// Row r of a set = its r-th best cell i, whose head values, score and class the caller has: decode_kernel :41-70 for
// the cell (the rows the operator can return: box, score, class), and iou3d_nms_kernel.cu:294-308's remap of the box
// into NMS layout.  exp3(a, b, c, out) = exp_rn of three values.
template <class Exp3>
__device__ __forceinline__ void cp_decode_row(const CpCfg& c, int set, int r, int i, int cap, const CpRows& o,
                                              const CpCell& v, float best, int arg, Exp3 exp3) {
  const int xs = i % c.feat_w, ys = i / c.feat_w;
  const float x = v.x, y = v.y, z = v.z, d0 = v.d0, d1 = v.d1, d2 = v.d2, r0 = v.r0, r1 = v.r1, v0 = v.v0, v1 = v.v1;
  float ed[3];
  exp3(d0, d1, d2, ed);  // :151 exp(dim)
  // every value is kept in a register: read back through the row pointers, each one is a store -> load round trip
  // (the rows may alias as far as the compiler knows), and six of them in a row were most of this function's time
  const float cx = (x + xs) * c.down_ratio * c.vx + c.pc_x;
  const float cy = (y + ys) * c.down_ratio * c.vy + c.pc_y;
  const float ang = atan2_rn(r0, r1);
  float* bx = o.boxes + ((int64_t)set * cap + r) * c.dims;
  bx[0] = cx;
  bx[1] = cy;
  bx[2] = z;
  bx[3] = ed[0];
  bx[4] = ed[1];
  bx[5] = ed[2];
  if (c.with_velocity) {
    bx[6] = v0;
    bx[7] = v1;
    bx[8] = ang;
  } else {
    bx[6] = ang;
  }
  o.scores[(int64_t)set * cap + r] = best;
  o.labels[(int64_t)set * cap + r] = arg;
  const float nb[7] = {cx, cy, z, ed[1], ed[0], ed[2], (float)(-(double)ang - 3.141592653589793 / 2)};
  float* q = o.nms_boxes + ((int64_t)set * cap + r) * 7;
#pragma unroll
  for (int k = 0; k < 7; ++k) q[k] = nb[k];
  const BoxPre bp = box_prepare(nb);  // what the suppression matrix needs of this box
  o.pre[(int64_t)set * cap + r] = bp;
  o.xyr[(int64_t)set * cap + r] = make_float4(bp.cx, bp.cy, bp.rad, 0.f);
}

// Top-K selection instead of a full sort of the score keys: only the first min(count, nms_pre_max_size) cells of
// the stable ascending-key order are ever used (postprocess.cu:176-206 sorts the masked scores and slices
// [:nms_pre_max_size]).  One workgroup per set, everything in LDS, from the head maps to the decoded candidates: the
// keys of the set's hw cells are COMPUTED into LDS (no key array in memory, no count atomics); a radix SELECT (10-bit
// LDS histograms over the bits a selected key can have) finds the exact cut-off key and how many cells with that key
// still fit; the selected cells are compacted in cell order -- their head values are requested at this point -- and
// sorted as (key, cell) pairs by a bitonic network: the same total order a stable key sort gives; the fetched values
// move to their rank through LDS and thread r decodes the r-th cell (cp_decode_row).  Writes counts[set], clears the
// set's NMS counters and writes the rows [set][0 .. K).  Every phase was timed by returning early after it
// (DESIGN_HISTORY 4.4): the kernel is a chain of latencies, and each comment below names the one it removes.
constexpr int kTopkThreads = 1024;  // one workgroup per set and nothing else on its CU: the kernel is a chain of
                                    // dependent passes, so its time is its latency -- 16 waves shorten every pass
constexpr int kTopkMaxHw = 16384;   // keys held in LDS
constexpr int kTopkMaxK = 1024;     // bitonic list
constexpr int kTopkBatch = 4;       // cells of a thread whose keys are computed side by side
constexpr int kTopkCopies = 8;      // histogram replicas (lane & 7): scores crowd into a handful of exponent bins, and
                                    // LDS atomics of one wave on one address run one lane at a time

// Where cell i's key sits in LDS: one word of padding per 16 cells, so that the compaction's threads (16 consecutive
// cells each at 16 k cells) read 64 different banks instead of 4.
__device__ __forceinline__ int kidx(int i) { return i + (i >> 4); }

__global__ __launch_bounds__(kTopkThreads) void cp_topk_kernel(CpHeads h, CpCfg c, int* __restrict__ counts,
                                                               int* __restrict__ pool_counts, int cap, CpRows rows, int key_bits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char topk_smem[];
  const int hw = c.hw;
  uint32_t* ks = reinterpret_cast<uint32_t*>(topk_smem);                       // [hw]
  unsigned long long* list = reinterpret_cast<unsigned long long*>(ks + ((max(hw + hw / 16 + 1, 12 * kTopkMaxK) + 1) & ~1));  // [kTopkMaxK]
  int* hist = reinterpret_cast<int*>(list + kTopkMaxK);                        // [kTopkCopies][1024]
  int* scr = hist + kTopkCopies * 1024;                                                      // [32]: scan scratch, [30], [31] broadcast
  uint64_t* etab = reinterpret_cast<uint64_t*>(scr + 32);                                    // [32]: expf's table
  unsigned char* cls = reinterpret_cast<unsigned char*>(etab + 32);                          // [hw]: best class of a cell
  const int set = blockIdx.x, sets = gridDim.x;
  if (threadIdx.x < 32) etab[threadIdx.x] = lm::exp2f_tab((int)threadIdx.x);
  __syncthreads();
  // ---- keys of all cells (cp_cell_key, kTopkBatch cells of a thread at a time) ---------------------------------
  // A cell is a chain of dependent reads (head values -> expf's table -> key); one cell after the other, 16 such
  // chains per thread were half of this kernel's time.  Here the head values of a batch are fetched together, the
  // polynomial part of expf runs straight-line on all of them (table reads from LDS, all in flight), and the
  // arguments expf treats specially (|x| >= 88, NaN) take lm::expf afterwards.
  int selected = 0;
  {
    const int t = set % c.num_tasks, frame = set / c.num_tasks, ncls = h.ncls[t];
    const int64_t bs = h.batch_stride;
    const float* hmp = h.hm[t] + (int64_t)frame * (bs ? bs : (int64_t)ncls * hw);
    const float* regp = h.reg[t] + (int64_t)frame * (bs ? bs : (int64_t)2 * hw);
    const float* heip = h.height[t] + (int64_t)frame * (bs ? bs : (int64_t)hw);
    const auto tab = [&](int i) { return etab[i]; };
    // software pipeline: the reads of batch b + 1 (reg, height, the first class map) and of class k + 1 are issued
    // before batch b / class k is computed
    int ii_n[kTopkBatch];
    float x_n[kTopkBatch], y_n[kTopkBatch], z_n[kTopkBatch], v_n[kTopkBatch];
    const auto fetch = [&](int base) {
#pragma unroll
      for (int j = 0; j < kTopkBatch; ++j) {
        ii_n[j] = min(base + j * kTopkThreads + (int)threadIdx.x, hw - 1);  // past the end: the last cell again, dropped below
        x_n[j] = regp[ii_n[j]];
        y_n[j] = regp[ii_n[j] + hw];
        z_n[j] = heip[ii_n[j]];
        v_n[j] = hmp[ii_n[j]];
      }
    };
    fetch(0);
    for (int base = 0; base < hw; base += kTopkBatch * kTopkThreads) {
      int ii[kTopkBatch], arg[kTopkBatch];
      float best[kTopkBatch], v[kTopkBatch];
      unsigned in_range = 0;  // bit j: cell j passes the mask on reg / height
#pragma unroll
      for (int j = 0; j < kTopkBatch; ++j) {
        ii[j] = ii_n[j];
        v[j] = v_n[j];
        if (x_n[j] <= c.r[3] && y_n[j] <= c.r[4] && z_n[j] <= c.r[5] && x_n[j] >= c.r[0] && y_n[j] >= c.r[1] &&
            z_n[j] >= c.r[2])
          in_range |= 1u << j;
        best[j] = 0.f;
        arg[j] = 0;
      }
      if (base + kTopkBatch * kTopkThreads < hw) fetch(base + kTopkBatch * kTopkThreads);
      for (int k = 0; k < ncls; ++k) {
        float e[kTopkBatch], vk[kTopkBatch];
#pragma unroll
        for (int j = 0; j < kTopkBatch; ++j) vk[j] = -v[j];
        if (k + 1 < ncls) {
#pragma unroll
          for (int j = 0; j < kTopkBatch; ++j) v[j] = hmp[(int64_t)(k + 1) * hw + ii[j]];
        }
#pragma unroll
        for (int j = 0; j < kTopkBatch; ++j) e[j] = lm::expf_main(vk[j], tab);
#pragma unroll
        for (int j = 0; j < kTopkBatch; ++j) {
          if (lm::expf_is_special(vk[j])) e[j] = exp_rn(vk[j]);
          const float sg = 1.0f / (1.0f + e[j]);
          if (k == 0 || sg > best[j]) {  // cp_best_class
            best[j] = sg;
            arg[j] = k;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < kTopkBatch; ++j) {
        const int i = base + j * kTopkThreads + (int)threadIdx.x;
        const bool m = best[j] > c.score_threshold && ((in_range >> j) & 1u);
        const uint32_t bits = __float_as_uint(best[j]);
        const uint32_t key = m ? (bits <= kKeyOne ? kKeyOne - bits : 0u) : kKeyOut;
        if (i < hw) {
          ks[kidx(i)] = key;
          cls[i] = (unsigned char)arg[j];
          selected += m ? 1 : 0;
        }
      }
    }
  }
  for (int i = threadIdx.x; i < kTopkMaxK; i += kTopkThreads) list[i] = ~0ull;
  int count;
  block_exclusive_scan<kTopkThreads>(selected, scr, count);  // (its barriers also publish ks / list)
  if (threadIdx.x == 0) {
    counts[set] = count;
    // the two counters the suppression-matrix kernels append through (nms_kernels.hpp NmsPool): no set-up memset
    pool_counts[set * kNmsCtrStride] = 0;
    pool_counts[(sets + set) * kNmsCtrStride] = 0;
  }
  const int K = min(count, cap);
  if (K <= 0) return;
  // ---- cut-off key kc and the number r of cells with key == kc that are taken (0: take every key < kc) -----
  // Radix select over the key_bits bits a masked-in key can have (scores above the threshold: 25 bits at 0.1 -- the
  // bits above are zero, and a digit taken from them put all cells into 27 bins), ten bits at a time from the top;
  // it stops as soon as the bin holding the cut-off is needed whole, which after two digits (16 k cells over a million
  // bins) it nearly always is.
  uint32_t kc = kKeyOut;
  int r = 0;
  if (count > K) {
    uint32_t prefix = 0;  // decided high bits
    int need = K;         // rank of the cut-off inside the still-undecided set (1-based)
    int hi = key_bits;    // bits [0, hi) are undecided
    while (hi > 0) {
      const int w = min(hi, 10), shift = hi - w;
      for (int i = threadIdx.x; i < kTopkCopies * 1024; i += kTopkThreads) hist[i] = 0;
      __syncthreads();
      for (int i = threadIdx.x; i < hw; i += kTopkThreads) {
        const uint32_t k = ks[kidx(i)];
        if ((k >> hi) == prefix)
          atomicAdd(&hist[(threadIdx.x & (kTopkCopies - 1)) * 1024 + ((k >> shift) & ((1u << w) - 1u))], 1);
      }
      __syncthreads();
      // thread t owns bins kBpt t .. kBpt t + kBpt - 1: exclusive prefix over bins, find the bin holding rank `need`
      constexpr int kBpt = 1024 / kTopkThreads;
      const int b0 = threadIdx.x * kBpt;
      int hh[kBpt], hsum = 0;
#pragma unroll
      for (int j = 0; j < kBpt; ++j) {
        hh[j] = 0;
#pragma unroll
        for (int c = 0; c < kTopkCopies; ++c) hh[j] += hist[c * 1024 + b0 + j];
        hsum += hh[j];
      }
      int total;
      const int base = block_exclusive_scan<kTopkThreads>(hsum, scr, total);
      int cum = base;
#pragma unroll
      for (int j = 0; j < kBpt; ++j) {
        if (need > cum && need <= cum + hh[j]) {  // exactly one (thread, j) satisfies this
          scr[29] = hh[j];
          scr[30] = b0 + j;
          scr[31] = need - cum;
        }
        cum += hh[j];
      }
      __syncthreads();
      const int in_bin = scr[29];
      prefix = (prefix << w) | (uint32_t)scr[30];
      need = scr[31];
      hi = shift;
      __syncthreads();
      if (need == in_bin) {  // the whole bin is taken: every key below the next prefix, none at it
        prefix = (prefix + 1u) << hi;
        need = 0;
        break;
      }
    }
    kc = prefix;
    r = need;
  }
  // ---- compaction in cell order: thread t owns the contiguous cells [t*ept, (t+1)*ept) ------------------------
  const int ept = (hw + kTopkThreads - 1) / kTopkThreads;
  const int c0 = threadIdx.x * ept, c1 = min(c0 + ept, hw);
  int nless = 0, neq = 0;
  for (int i = c0; i < c1; ++i) {
    const uint32_t k = ks[kidx(i)];
    nless += k < kc ? 1 : 0;
    neq += k == kc ? 1 : 0;
  }
  int tot_less, tot_eq;
  int pos_less = block_exclusive_scan<kTopkThreads>(nless, scr, tot_less);
  int pos_eq = block_exclusive_scan<kTopkThreads>(neq, scr, tot_eq);
  // entry = key : cell : position in this (cell-ordered) list -- the order of (key, cell) with the way back to the position
  for (int i = c0; i < c1; ++i) {
    const uint32_t k = ks[kidx(i)];
    if (k < kc) {
      list[pos_less] = ((unsigned long long)k << 32) | (uint32_t)(i << 10) | (uint32_t)pos_less;
      ++pos_less;
    } else if (k == kc) {
      if (pos_eq < r) list[tot_less + pos_eq] = ((unsigned long long)k << 32) | (uint32_t)(i << 10) | (uint32_t)(tot_less + pos_eq);
      ++pos_eq;
    }
  }
  __syncthreads();
  // ---- the head values of the selected cells: fetched NOW, by thread p for the p-th cell in cell order ---------
  // Ten reads of four bytes from ten planes per cell are the slowest thing the decode does (17 of its 22 us when it
  // came after the sort).  Issued here they are in flight during the sort, whose barriers (px_lds_barrier) wait for
  // LDS only; neighbouring threads hold neighbouring cells, which share cache lines where selections cluster.
  const int p = threadIdx.x;
  const int t_of = set % c.num_tasks, frame_of = set / c.num_tasks;
  int my_cell = 0;
  float my_best = 0.f;
  CpCell my{};
  if (p < K) {
    const unsigned long long ent = list[p];
    my_cell = (int)((uint32_t)ent >> 10);
    my_best = __uint_as_float(kKeyOne - (uint32_t)(ent >> 32));  // the score is in the key (a sigmoid is <= 1)
    my = cp_gather_cell(h, c, t_of, frame_of, my_cell);
  }
  const int my_cls = cls[my_cell];
  // ---- bitonic sort of the entries, padded with ~0: one compare-exchange per thread and step ---------------------
  // Thread t of wave w works on entries [128 w, 128 w + 128) at every stride up to 64 -- entries no other wave touches
  // at such a step -- so those steps follow each other in the wave's own LDS order; a barrier is only needed around the
  // steps with a stride of 128 and more: 10 of them at 1024 entries instead of one after each of the 55 steps.
  {
    int n2 = 64;
    while (n2 < K) n2 <<= 1;  // uniform
    const int t = threadIdx.x;
    for (int size = 2; size <= n2; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        if (t < (n2 >> 1)) {
          const int lo = 2 * t - (t & (stride - 1));
          const int hi = lo + stride;
          const bool up = (lo & size) == 0;
          const unsigned long long a = list[lo], b = list[hi];
          if ((a > b) == up) {
            list[lo] = b;
            list[hi] = a;
          }
        }
        if (stride >= 2 * kWave || (stride == 1 && size >= 2 * kWave))
          px_lds_barrier();
        else
          asm volatile("" ::: "memory");
      }
    }
    px_lds_barrier();
  }
  // ---- position -> rank, then the fetched values travel to their rank through LDS (the keys' 64 KB are free) ------
  unsigned short* rank_of = reinterpret_cast<unsigned short*>(hist);  // [kTopkMaxK]
  float* vals = reinterpret_cast<float*>(ks);                          // [12][kTopkMaxK]
  if (p < K) rank_of[(uint32_t)list[p] & 1023u] = (unsigned short)p;
  px_lds_barrier();
  if (p < K) {
    const int rk = rank_of[p];
    const float row[10] = {my.x, my.y, my.z, my.d0, my.d1, my.d2, my.r0, my.r1, my.v0, my.v1};
#pragma unroll
    for (int k = 0; k < 10; ++k) vals[k * kTopkMaxK + rk] = row[k];
    vals[10 * kTopkMaxK + rk] = my_best;
    vals[11 * kTopkMaxK + rk] = __int_as_float((my_cell << 8) | my_cls);
  }
  px_lds_barrier();
  // ---- decode: thread r takes the r-th row --------------------------------------------------------------------------
  if (p < K) {
    CpCell v;
    v.x = vals[p], v.y = vals[kTopkMaxK + p], v.z = vals[2 * kTopkMaxK + p];
    v.d0 = vals[3 * kTopkMaxK + p], v.d1 = vals[4 * kTopkMaxK + p], v.d2 = vals[5 * kTopkMaxK + p];
    v.r0 = vals[6 * kTopkMaxK + p], v.r1 = vals[7 * kTopkMaxK + p];
    v.v0 = vals[8 * kTopkMaxK + p], v.v1 = vals[9 * kTopkMaxK + p];
    const float best = vals[10 * kTopkMaxK + p];
    const int cc = __float_as_int(vals[11 * kTopkMaxK + p]);
    const auto tab = [&](int i) { return etab[i]; };
    cp_decode_row(c, set, p, cc >> 8, cap, rows, v, best, cc & 255, [&](float a, float b, float d, float* e) {
      e[0] = lm::expf_main(a, tab);
      e[1] = lm::expf_main(b, tab);
      e[2] = lm::expf_main(d, tab);
      if (lm::expf_is_special(a)) e[0] = exp_rn(a);
      if (lm::expf_is_special(b)) e[1] = exp_rn(b);
      if (lm::expf_is_special(d)) e[2] = exp_rn(d);
    });
  }
}

// LDS of cp_topk_kernel: keys (later the 12 x 1024 values on their way to rank order), list, histograms, scan scratch,
// expf's table, classes
static inline size_t cp_topk_keys_words(int hw) { return std::max((size_t)hw + (size_t)hw / 16 + 1, (size_t)12 * kTopkMaxK); }
static inline size_t cp_topk_lds(int hw) {
  return cp_topk_keys_words(hw) * 4 + 8 + (size_t)kTopkMaxK * 8 + (size_t)kTopkCopies * 1024 * 4 + 32 * 4 + 32 * 8 + (size_t)hw;
}

// The full-sort selection's last pass: the nms_pre_max_size best cells of every set, in sorted order.
__global__ __launch_bounds__(256) void cp_nms_boxes_kernel(CpHeads h, CpCfg c, const uint32_t* __restrict__ sidx,
                                                           const int* __restrict__ counts, int cap, CpRows rows) {
  const int set = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = min(counts[set], cap);
  if (r >= n) return;
  const int i = (int)sidx[(int64_t)set * c.hw + r];
  int arg;
  const float best = cp_best_class(h, c, set % c.num_tasks, set / c.num_tasks, i, arg);
  const CpCell v = cp_gather_cell(h, c, set % c.num_tasks, set / c.num_tasks, i);
  cp_decode_row(c, set, r, i, cap, rows, v, best, arg, [](float a, float b, float d, float* e) {
    e[0] = exp_rn(a);
    e[1] = exp_rn(b);
    e[2] = exp_rn(d);
  });
}

// Concatenate the tasks' rows in task order (postprocess.cu:247-278).  One workgroup per (task, frame): the row its
// task starts at is the sum of the earlier tasks' row counts (a few scalar loads), so the tasks copy side by side; the
// last task's workgroup also clears the rows behind the end and writes the frame's count.
__global__ __launch_bounds__(256) void cp_output_kernel(
    const float* __restrict__ boxes, const float* __restrict__ scores,
    const int* __restrict__ labels,
    const int* __restrict__ counts, const int32_t* __restrict__ keep,
    const int32_t* __restrict__ nkeep, CpHeads h, int num_tasks, int hw, int dims, int cap, int pre_max,
    int post_max, float* __restrict__ out_boxes, float* __restrict__ out_scores,
    int64_t* __restrict__ out_labels, int32_t* __restrict__ out_count, float* __restrict__ out_records,
    int max_per_img) {
  const int task = blockIdx.x, frame = blockIdx.y;
  const int rows_cap = num_tasks * max(post_max, 1);
  out_boxes += (int64_t)frame * rows_cap * dims;
  out_scores += (int64_t)frame * rows_cap;
  out_labels += (int64_t)frame * rows_cap;
  // optional second form of the same rows: the fixed-shape record of the multi-GPU result hand-off
  // (paddle3d_amd/dist.py: [max_per_img, 11] = box (zero padded to 9 values), score, label), written here instead
  // of by a handful of tensor-library kernels after the fact
  constexpr int kRec = 11;
  float* rec = out_records ? out_records + (int64_t)frame * max_per_img * kRec : nullptr;
  // rows of a task: the fake row of :190-201 when nothing passed the mask, otherwise the kept boxes up to
  // nms_post_max_size -- none with a zero pre-NMS cap (num_bboxes_for_nms = min(count, nms_pre_max_size), :212-216)
  auto rows_of = [&](int tk) {
    const int st = frame * num_tasks + tk;
    return counts[st] <= 0 ? 1 : (pre_max > 0 ? min(nkeep[st], post_max) : 0);
  };
  int offset = 0;
  for (int tk = 0; tk < task; ++tk) offset += rows_of(tk);
  const int t = frame * num_tasks + task;  // set index
  const int rows = rows_of(task);
  if (counts[t] <= 0) {
    if ((int)threadIdx.x < dims) out_boxes[(int64_t)offset * dims + threadIdx.x] = 0.f;
    if (threadIdx.x == 0) {
      out_scores[offset] = -1.f;
      out_labels[offset] = 0;
    }
    if (rec && offset < max_per_img && (int)threadIdx.x < kRec)
      rec[(int64_t)offset * kRec + threadIdx.x] = threadIdx.x == 9 ? -1.f : 0.f;
  } else {
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
      const int pos = keep[(int64_t)t * cap + r];  // index into the sorted order = row of the decoded candidates
      const float* bx = boxes + ((int64_t)t * cap + pos) * dims;
      for (int k = 0; k < dims; ++k) out_boxes[(int64_t)(offset + r) * dims + k] = bx[k];
      const float sc = scores[(int64_t)t * cap + pos];
      const int lb = labels[(int64_t)t * cap + pos] + h.label_offset[task];
      out_scores[offset + r] = sc;
      out_labels[offset + r] = (int64_t)lb;
      if (rec && offset + r < max_per_img) {
        float* q = rec + (int64_t)(offset + r) * kRec;
        for (int k = 0; k < 9; ++k) q[k] = k < dims ? bx[k] : 0.f;
        q[9] = sc;
        q[10] = (float)lb;
      }
    }
  }
  if (task != num_tasks - 1) return;
  offset += rows;
  // rows behind the last one read zero (the outputs need no clearing by the caller)
  for (int r = offset + (int)threadIdx.x; r < rows_cap; r += blockDim.x) {
    for (int k = 0; k < dims; ++k) out_boxes[(int64_t)r * dims + k] = 0.f;
    out_scores[r] = 0.f;
    out_labels[r] = 0;
  }
  if (rec)
    for (int e = min(offset, max_per_img) * kRec + (int)threadIdx.x; e < max_per_img * kRec; e += blockDim.x) rec[e] = 0.f;
  if (threadIdx.x == 0) out_count[frame] = offset;
}

struct CpWorkspace {
  float *boxes, *scores, *nms_boxes;
  int *labels, *counts, *hist, *partial;
  uint32_t *keys_a, *vals_a, *keys_b, *vals_b;
  unsigned long long* mask;
  BoxPre* pre;
  NmsPool pool;
  int32_t *keep, *nkeep;
  size_t bytes;
};

static CpWorkspace cp_carve(void* base, int tasks, int hw, int pre_max, const RadixPlan& plan) {
  Carver c(base);
  CpWorkspace w;
  const size_t th = (size_t)tasks * hw;
  const int cap = std::max(pre_max, 1);
  const size_t cb = ((size_t)cap + 63) / 64;
  w.boxes = c.take<float>((size_t)tasks * cap * 9);  // the decoded candidates: [set][rank]
  w.scores = c.take<float>((size_t)tasks * cap);
  w.labels = c.take<int>((size_t)tasks * cap);
  w.counts = c.take<int>((size_t)tasks);
  w.pool.counts = c.take<int>((size_t)tasks * 2 * kNmsCtrStride);  // directly behind `counts`: one memset clears both
  w.keys_a = c.take<uint32_t>(th);
  w.vals_a = c.take<uint32_t>(th);
  w.keys_b = c.take<uint32_t>(th);
  w.vals_b = c.take<uint32_t>(th);
  w.hist = c.take<int>((size_t)tasks * radix_hist_ints(plan));
  w.partial = c.take<int>((size_t)tasks * scan_num_tiles((int64_t)radix_hist_ints(plan)));
  w.nms_boxes = c.take<float>((size_t)tasks * cap * 7);
  w.mask = c.take<unsigned long long>((size_t)tasks * cap * cb);
  w.pre = c.take<BoxPre>((size_t)tasks * cap);
  w.pool.xyr = c.take<float4>((size_t)tasks * cap);
  w.pool.per_set = nms_pool_per_set(cap);
  w.pool.pairs = c.take<uint32_t>((size_t)tasks * w.pool.per_set);
  w.pool.tiles = c.take<uint32_t>((size_t)tasks * cb * cb);
  w.keep = c.take<int32_t>((size_t)tasks * cap);
  w.nkeep = c.take<int32_t>((size_t)tasks);
  w.bytes = c.off;
  return w;
}

}  // namespace pd3

using namespace pd3;

extern "C" size_t pd3_centerpoint_postprocess_workspace(int batch, int num_tasks, int feat_h,
                                                        int feat_w, int nms_pre_max_size,
                                                        int nms_post_max_size) {
  (void)nms_post_max_size;
  if (batch <= 0 || num_tasks <= 0 || feat_h <= 0 || feat_w <= 0) return 0;
  const int hw = feat_h * feat_w;
  return cp_carve(nullptr, batch * num_tasks, hw, nms_pre_max_size, radix_plan(kKeyOut, hw)).bytes;
}

static int cp_postprocess_impl(
    int64_t head_batch_stride, int selection,
    const float* const* hm, const float* const* reg, const float* const* height,
    const float* const* dim, const float* const* vel, const float* const* rot, int batch,
    int num_tasks, const int* hm_channels, int feat_h, int feat_w, const float* voxel_size,
    const float* point_cloud_range, const float* post_center_range, const int* label_offsets,
    int down_ratio, float score_threshold, float nms_iou_threshold, int nms_pre_max_size,
    int nms_post_max_size, int with_velocity, float* out_bboxes, float* out_scores,
    int64_t* out_labels, int32_t* out_count, void* workspace, size_t workspace_bytes,
    void* stream, float* out_records = nullptr, int max_per_img = 0) {
  if (!hm || !reg || !height || !dim || !vel || !rot || !hm_channels || !label_offsets ||
      !voxel_size || !point_cloud_range || !post_center_range || !out_bboxes || !out_scores ||
      !out_labels || !out_count || !workspace)
    return PD3_EINVAL;
  if (batch <= 0 || num_tasks <= 0 || num_tasks > kMaxTasks || feat_h <= 0 || feat_w <= 0 ||
      nms_pre_max_size < 0 || nms_post_max_size < 0)
    return PD3_EINVAL;
  const int hw = feat_h * feat_w;
  const int cap = std::max(nms_pre_max_size, 1);
  const int cb = (cap + 63) / 64;
  if (cb > kNmsMaxWords) return PD3_EUNSUPPORTED;
  const RadixPlan plan = radix_plan(kKeyOut, hw);
  const int sets = batch * num_tasks;
  CpWorkspace w = cp_carve(workspace, sets, hw, nms_pre_max_size, plan);
  if (workspace_bytes < w.bytes) return PD3_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);

  CpHeads h;
  h.batch_stride = head_batch_stride;
  for (int t = 0; t < num_tasks; ++t) {
    h.hm[t] = hm[t];
    h.reg[t] = reg[t];
    h.height[t] = height[t];
    h.dim[t] = dim[t];
    h.vel[t] = vel[t];
    h.rot[t] = rot[t];
    h.ncls[t] = hm_channels[t];
    h.label_offset[t] = label_offsets[t];
    if (!hm[t] || !reg[t] || !height[t] || !dim[t] || !vel[t] || !rot[t] || hm_channels[t] <= 0)
      return PD3_EINVAL;
  }
  CpCfg c;
  c.hw = hw;
  c.feat_w = feat_w;
  c.num_tasks = num_tasks;
  c.with_velocity = with_velocity ? 1 : 0;
  c.dims = with_velocity ? 9 : 7;
  c.down_ratio = (float)down_ratio;  // int attr received as float, postprocess.cu:35,85
  c.vx = voxel_size[0];
  c.vy = voxel_size[1];
  c.pc_x = point_cloud_range[0];
  c.pc_y = point_cloud_range[1];
  for (int k = 0; k < 6; ++k) c.r[k] = post_center_range[k];
  c.score_threshold = score_threshold;

  hipError_t e;
  if (selection < 0 || selection > 1) return PD3_EINVAL;
  const CpRows rows{w.boxes, w.scores, w.labels, w.nms_boxes, w.pre, w.pool.xyr};
  bool byte_classes = true;
  for (int t = 0; t < num_tasks; ++t) byte_classes = byte_classes && hm_channels[t] <= 256;
  if (hw <= kTopkMaxHw && cap <= kTopkMaxK && byte_classes && selection == 0) {
    const size_t lds = cp_topk_lds(hw);
    e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(cp_topk_kernel), (int)cp_topk_lds(kTopkMaxHw));
    if (e != hipSuccess) return (int)e;
    // bits a masked-in key can have: key = bits(1.0f) - bits(score) with score in (threshold, 1]
    uint32_t thr_bits;
    memcpy(&thr_bits, &score_threshold, 4);
    const uint32_t top = (score_threshold > 0.f && thr_bits < kKeyOne) ? kKeyOne - thr_bits : kKeyOne;
    int key_bits = 1;
    while (key_bits < 30 && (top >> key_bits) != 0) ++key_bits;
    cp_topk_kernel<<<sets, kTopkThreads, lds, s>>>(h, c, w.counts, w.pool.counts, cap, rows, key_bits);
  } else {
    e = hipMemsetAsync(w.counts, 0, (size_t)((char*)(w.pool.counts + (size_t)sets * 2 * kNmsCtrStride) - (char*)w.counts), s);
    if (e != hipSuccess) return (int)e;
    dim3 dgrid((hw + 255) / 256, sets);
    cp_score_kernel<<<dgrid, 256, 0, s>>>(h, c, w.keys_a, w.counts);
    const int where = enqueue_radix_sort(w.keys_a, w.vals_a, w.keys_b, w.vals_b, hw, hw, sets,
                                         plan, /*identity_vals=*/true, w.hist, w.partial, s);
    dim3 bgrid((cap + 255) / 256, sets);
    cp_nms_boxes_kernel<<<bgrid, 256, 0, s>>>(h, c, where ? w.vals_b : w.vals_a, w.counts, cap, rows);
  }
  nms_enqueue_mask_pooled(w.pre, w.counts, sets, cap, cb, nms_iou_threshold, w.mask, w.pool, s);
  {
    const size_t lds = nms_sweep_lds(cap);
    if (lds > 48 * 1024) {
      e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(nms_sweep_kernel), (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    nms_sweep_kernel<<<sets, kNmsSweepThreads, lds, s>>>(w.mask, w.counts, 0, cap, cb, w.keep, w.nkeep);
  }
  cp_output_kernel<<<dim3(num_tasks, batch), 256, 0, s>>>(w.boxes, w.scores, w.labels, w.counts, w.keep, w.nkeep, h,
                                     num_tasks, hw, c.dims, cap, nms_pre_max_size, nms_post_max_size, out_bboxes,
                                     out_scores, out_labels, out_count, out_records, max_per_img);
  return launch_status();
}

extern "C" int pd3_centerpoint_postprocess(
    const float* const* hm, const float* const* reg, const float* const* height,
    const float* const* dim, const float* const* vel, const float* const* rot, int batch,
    int num_tasks, const int* hm_channels, int feat_h, int feat_w, const float* voxel_size,
    const float* point_cloud_range, const float* post_center_range, const int* label_offsets,
    int down_ratio, float score_threshold, float nms_iou_threshold, int nms_pre_max_size,
    int nms_post_max_size, int with_velocity, float* out_bboxes, float* out_scores,
    int64_t* out_labels, int32_t* out_count, void* workspace, size_t workspace_bytes,
    void* stream) {
  return cp_postprocess_impl(0, 0, hm, reg, height, dim, vel, rot, batch, num_tasks, hm_channels, feat_h, feat_w,
                             voxel_size, point_cloud_range, post_center_range, label_offsets, down_ratio,
                             score_threshold, nms_iou_threshold, nms_pre_max_size, nms_post_max_size,
                             with_velocity, out_bboxes, out_scores, out_labels, out_count, workspace,
                             workspace_bytes, stream);
}

extern "C" int pd3_centerpoint_postprocess_strided(
    const float* const* hm, const float* const* reg, const float* const* height,
    const float* const* dim, const float* const* vel, const float* const* rot, int64_t head_batch_stride,
    int batch, int num_tasks, const int* hm_channels, int feat_h, int feat_w, const float* voxel_size,
    const float* point_cloud_range, const float* post_center_range, const int* label_offsets,
    int down_ratio, float score_threshold, float nms_iou_threshold, int nms_pre_max_size,
    int nms_post_max_size, int with_velocity, float* out_bboxes, float* out_scores,
    int64_t* out_labels, int32_t* out_count, void* workspace, size_t workspace_bytes,
    void* stream, int selection) {
  if (head_batch_stride <= 0) return PD3_EINVAL;
  return cp_postprocess_impl(head_batch_stride, selection, hm, reg, height, dim, vel, rot, batch, num_tasks, hm_channels,
                             feat_h, feat_w, voxel_size, point_cloud_range, post_center_range, label_offsets,
                             down_ratio, score_threshold, nms_iou_threshold, nms_pre_max_size,
                             nms_post_max_size, with_velocity, out_bboxes, out_scores, out_labels, out_count,
                             workspace, workspace_bytes, stream);
}

extern "C" int pd3_centerpoint_postprocess_records(
    const float* const* hm, const float* const* reg, const float* const* height,
    const float* const* dim, const float* const* vel, const float* const* rot, int64_t head_batch_stride,
    int batch, int num_tasks, const int* hm_channels, int feat_h, int feat_w, const float* voxel_size,
    const float* point_cloud_range, const float* post_center_range, const int* label_offsets,
    int down_ratio, float score_threshold, float nms_iou_threshold, int nms_pre_max_size,
    int nms_post_max_size, int with_velocity, float* out_bboxes, float* out_scores,
    int64_t* out_labels, int32_t* out_count, float* out_records, int max_per_img, void* workspace,
    size_t workspace_bytes, void* stream) {
  if (head_batch_stride <= 0 || !out_records || max_per_img <= 0) return PD3_EINVAL;
  return cp_postprocess_impl(head_batch_stride, 0, hm, reg, height, dim, vel, rot, batch, num_tasks, hm_channels,
                             feat_h, feat_w, voxel_size, point_cloud_range, post_center_range, label_offsets,
                             down_ratio, score_threshold, nms_iou_threshold, nms_pre_max_size,
                             nms_post_max_size, with_velocity, out_bboxes, out_scores, out_labels, out_count,
                             workspace, workspace_bytes, stream, out_records, max_per_img);
}
