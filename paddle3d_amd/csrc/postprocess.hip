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
#include "common.hpp"
#include "nms_kernels.hpp"
#include "radix_sort.hpp"

#include <algorithm>

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

// Row r of a set = its r-th best cell i: decode_kernel :41-70 for the cell (the rows the operator can return: box,
// score, class), and iou3d_nms_kernel.cu:294-308's remap of the box into NMS layout.
__device__ __forceinline__ void cp_decode_row(const CpHeads& h, const CpCfg& c, int set, int r, int i, int cap,
                                              const CpRows& o) {
  const int t = set % c.num_tasks, frame = set / c.num_tasks;
  const int64_t bs = h.batch_stride;
  const float* regp = h.reg[t] + (int64_t)frame * (bs ? bs : (int64_t)2 * c.hw);
  const float* heip = h.height[t] + (int64_t)frame * (bs ? bs : (int64_t)c.hw);
  const float* dimp = h.dim[t] + (int64_t)frame * (bs ? bs : (int64_t)3 * c.hw);
  const float* velp = h.vel[t] + (int64_t)frame * (bs ? bs : (int64_t)2 * c.hw);
  const float* rotp = h.rot[t] + (int64_t)frame * (bs ? bs : (int64_t)2 * c.hw);
  int arg;
  const float best = cp_best_class(h, c, t, frame, i, arg);
  const int xs = i % c.feat_w, ys = i / c.feat_w;
  const float x = regp[i], y = regp[i + c.hw], z = heip[i];
  float* bx = o.boxes + ((int64_t)set * cap + r) * c.dims;
  bx[0] = (x + xs) * c.down_ratio * c.vx + c.pc_x;
  bx[1] = (y + ys) * c.down_ratio * c.vy + c.pc_y;
  bx[2] = z;
  bx[3] = exp_rn(dimp[i]);  // :151 exp(dim)
  bx[4] = exp_rn(dimp[i + c.hw]);
  bx[5] = exp_rn(dimp[i + 2 * c.hw]);
  const float ang = atan2_rn(rotp[i], rotp[i + c.hw]);
  if (c.with_velocity) {
    bx[6] = velp[i];
    bx[7] = velp[i + c.hw];
    bx[8] = ang;
  } else {
    bx[6] = ang;
  }
  o.scores[(int64_t)set * cap + r] = best;
  o.labels[(int64_t)set * cap + r] = arg;
  float* q = o.nms_boxes + ((int64_t)set * cap + r) * 7;
  q[0] = bx[0];
  q[1] = bx[1];
  q[2] = bx[2];
  q[3] = bx[4];
  q[4] = bx[3];
  q[5] = bx[5];
  q[6] = (float)(-(double)ang - 3.141592653589793 / 2);
  const float nb[7] = {q[0], q[1], q[2], q[3], q[4], q[5], q[6]};
  const BoxPre bp = box_prepare(nb);  // what the suppression matrix needs of this box
  o.pre[(int64_t)set * cap + r] = bp;
  o.xyr[(int64_t)set * cap + r] = make_float4(bp.cx, bp.cy, bp.rad, 0.f);
}

// Top-K selection instead of a full sort of the score keys: only the first min(count, nms_pre_max_size) cells of
// the stable ascending-key order are ever used (postprocess.cu:176-206 sorts the masked scores and slices
// [:nms_pre_max_size]).  One workgroup per set, everything in LDS, from the head maps to the decoded candidates: the
// keys of the set's hw cells are COMPUTED into LDS (16 cells per thread; no key array in memory, no count atomics); a
// 3-pass radix SELECT (10-bit LDS histograms) finds the exact cut-off key and how many cells with that key still fit;
// the selected cells are compacted in cell order and sorted as (key, cell) pairs by a bitonic network -- the same total
// order a stable key sort gives; thread r then decodes the r-th cell (cp_decode_row).  Writes counts[set], clears the
// set's NMS counters and writes the rows [set][0 .. K).
constexpr int kTopkThreads = 1024;  // one workgroup per set and nothing else on its CU: the kernel is a chain of
                                    // dependent passes, so its time is its latency -- 16 waves shorten every pass
constexpr int kTopkMaxHw = 16384;   // keys held in LDS
constexpr int kTopkMaxK = 1024;     // bitonic list
constexpr int kTopkBatch = 8;       // cells of a thread whose keys are computed side by side
constexpr int kTopkCopies = 8;      // histogram replicas (lane & 7): scores crowd into a handful of exponent bins, and
                                    // LDS atomics of one wave on one address run one lane at a time

__global__ __launch_bounds__(kTopkThreads) void cp_topk_kernel(CpHeads h, CpCfg c, int* __restrict__ counts,
                                                               int* __restrict__ pool_counts, int cap, CpRows rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char topk_smem[];
  const int hw = c.hw;
  uint32_t* ks = reinterpret_cast<uint32_t*>(topk_smem);                       // [hw]
  unsigned long long* list = reinterpret_cast<unsigned long long*>(ks + hw);   // [kTopkMaxK]
  int* hist = reinterpret_cast<int*>(list + kTopkMaxK);                        // [kTopkCopies][1024]
  int* scr = hist + kTopkCopies * 1024;                                                      // [32]: scan scratch, [30], [31] broadcast
  uint64_t* etab = reinterpret_cast<uint64_t*>(scr + 32);                                    // [32]: expf's table
  const int set = blockIdx.x;
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
    for (int base = 0; base < hw; base += kTopkBatch * kTopkThreads) {
      int ii[kTopkBatch];
      float x[kTopkBatch], y[kTopkBatch], z[kTopkBatch], best[kTopkBatch];
#pragma unroll
      for (int j = 0; j < kTopkBatch; ++j) {
        ii[j] = min(base + j * kTopkThreads + (int)threadIdx.x, hw - 1);  // past the end: the last cell again, dropped below
        x[j] = regp[ii[j]];
        y[j] = regp[ii[j] + hw];
        z[j] = heip[ii[j]];
        best[j] = 0.f;
      }
      for (int k = 0; k < ncls; ++k) {
        float v[kTopkBatch], e[kTopkBatch];
#pragma unroll
        for (int j = 0; j < kTopkBatch; ++j) v[j] = -hmp[(int64_t)k * hw + ii[j]];
#pragma unroll
        for (int j = 0; j < kTopkBatch; ++j) e[j] = lm::expf_main(v[j], tab);
#pragma unroll
        for (int j = 0; j < kTopkBatch; ++j) {
          if (lm::expf_is_special(v[j])) e[j] = exp_rn(v[j]);
          const float sg = 1.0f / (1.0f + e[j]);
          if (k == 0 || sg > best[j]) best[j] = sg;  // cp_best_class
        }
      }
#pragma unroll
      for (int j = 0; j < kTopkBatch; ++j) {
        const int i = base + j * kTopkThreads + (int)threadIdx.x;
        const bool m = best[j] > c.score_threshold && x[j] <= c.r[3] && y[j] <= c.r[4] && z[j] <= c.r[5] &&
                       x[j] >= c.r[0] && y[j] >= c.r[1] && z[j] >= c.r[2];
        const uint32_t bits = __float_as_uint(best[j]);
        const uint32_t key = m ? (bits <= kKeyOne ? kKeyOne - bits : 0u) : kKeyOut;
        if (i < hw) {
          ks[i] = key;
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
    pool_counts[((int)gridDim.x + set) * kNmsCtrStride] = 0;
  }
  const int K = min(count, cap);
  if (K <= 0) return;
  // ---- cut-off key kc and the number r of cells with key == kc that are taken (0: take every key < kc) -----
  uint32_t kc = kKeyOut;
  int r = 0;
  if (count > K) {
    uint32_t prefix = 0;  // decided high bits
    int need = K;         // rank of the cut-off inside the still-undecided set (1-based)
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = 20 - 10 * pass;
      for (int i = threadIdx.x; i < kTopkCopies * 1024; i += kTopkThreads) hist[i] = 0;
      __syncthreads();
      for (int i = threadIdx.x; i < hw; i += kTopkThreads) {
        const uint32_t k = ks[i];
        if (pass == 0 || (k >> (shift + 10)) == prefix)
          atomicAdd(&hist[(threadIdx.x & (kTopkCopies - 1)) * 1024 + ((k >> shift) & 1023u)], 1);
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
          scr[30] = b0 + j;
          scr[31] = need - cum;
        }
        cum += hh[j];
      }
      __syncthreads();
      prefix = (prefix << 10) | (uint32_t)scr[30];
      need = scr[31];
      __syncthreads();
    }
    kc = prefix;
    r = need;
  }
  // ---- compaction in cell order: thread t owns the contiguous cells [t*ept, (t+1)*ept) ------------------------
  const int ept = (hw + kTopkThreads - 1) / kTopkThreads;
  const int c0 = threadIdx.x * ept, c1 = min(c0 + ept, hw);
  int nless = 0, neq = 0;
  for (int i = c0; i < c1; ++i) {
    const uint32_t k = ks[i];
    nless += k < kc ? 1 : 0;
    neq += k == kc ? 1 : 0;
  }
  int tot_less, tot_eq;
  int pos_less = block_exclusive_scan<kTopkThreads>(nless, scr, tot_less);
  int pos_eq = block_exclusive_scan<kTopkThreads>(neq, scr, tot_eq);
  for (int i = c0; i < c1; ++i) {
    const uint32_t k = ks[i];
    if (k < kc) {
      list[pos_less++] = ((unsigned long long)k << 32) | (uint32_t)i;
    } else if (k == kc) {
      if (pos_eq < r) list[tot_less + pos_eq] = ((unsigned long long)k << 32) | (uint32_t)i;
      ++pos_eq;
    }
  }
  __syncthreads();
  // ---- bitonic sort of the (key, cell) pairs, padded with ~0 -------------------------------------------------
  int n2 = 64;
  while (n2 < K) n2 <<= 1;  // uniform
  for (int size = 2; size <= n2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (n2 >> 1); t += kTopkThreads) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = list[lo], b = list[hi];
        if ((a > b) == up) {
          list[lo] = b;
          list[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int r = threadIdx.x; r < K; r += kTopkThreads) cp_decode_row(h, c, set, r, (int)(list[r] & 0xffffffffull), cap, rows);
}

static inline size_t cp_topk_lds(int hw) {
  return (size_t)hw * 4 + (size_t)kTopkMaxK * 8 + (size_t)kTopkCopies * 1024 * 4 + 32 * 4 + 32 * 8;
}

// The full-sort selection's last pass: the nms_pre_max_size best cells of every set, in sorted order.
__global__ __launch_bounds__(256) void cp_nms_boxes_kernel(CpHeads h, CpCfg c, const uint32_t* __restrict__ sidx,
                                                           const int* __restrict__ counts, int cap, CpRows rows) {
  const int set = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = min(counts[set], cap);
  if (r >= n) return;
  cp_decode_row(h, c, set, r, (int)sidx[(int64_t)set * c.hw + r], cap, rows);
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
  if (hw <= kTopkMaxHw && cap <= kTopkMaxK && selection == 0) {
    const size_t lds = cp_topk_lds(hw);
    e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(cp_topk_kernel), (int)cp_topk_lds(kTopkMaxHw));
    if (e != hipSuccess) return (int)e;
    cp_topk_kernel<<<sets, kTopkThreads, lds, s>>>(h, c, w.counts, w.pool.counts, cap, rows);
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
