// PointPillars SSD head post-processing for gfx950: anchor mask, box decoding, score / range filter, rotated
// NMS and the result rows of a whole batch in one launch sequence, every count on the device.
// (reference: paddle3d/models/detection/pointpillars/pointpillars_head.py:86-196 post_process /
//  _single_post_process / _box_not_empty, anchors_generator.py:103-121 + :191-210 generate_anchors_mask,
//  pointpillars_coder.py:126-148 second_box_decode_paddle, models/layers/layer_libs.py:210-249 rotate_nms_pcdet.)
//
// The reference runs this per frame in Python: a scatter_nd_add + two cumsums + four gather_nd for the anchor
// mask, three boolean-mask selections (each a host sync on the selected count), an argsort, the NMS op with its
// mask copy + host sweep, and three index_selects.  Here:
//   1. ssd_occupancy_kernel      pillar coordinates -> per-frame occupancy counts [B, ny, nx] (int32)
//   2. ssd_integral_rows/cols    summed-area table in place (integers: the reference's fp32 cumsums are exact)
//   3. ssd_decode_kernel         per anchor: area test, sigmoid -> max / argmax, direction bit, box decode, score
//                                and centre-range test, heading flip, bottom -> object centre, sort key
//   4. top-K selection           key = bits(1.0f) - bits(score) for kept anchors (descending score, ties in anchor
//                                order = boolean-mask order + stable argsort), 0x3FFFFFFF otherwise; ssd_topk_kernel
//                                (radix select + ordered compaction + bitonic sort of <= 1024 pairs), or with
//                                selection = 1 a full stable radix sort of all keys -- identical results
//   5. ssd_nms_boxes_kernel      top min(kept, nms_pre_max_size) boxes in the NMS kernel's layout
//                                (x, y, z, l, w, h, -theta - pi/2)
//   6. nms_cand_kernel + nms_pairs_kernel + nms_sweep_kernel (nms_kernels.hpp)
//   7. ssd_output_kernel         the kept rows, object centre -> bottom centre again (both roundings kept)
// The head maps are read where the 1x1 convolutions wrote them (NCHW, channel = anchor * width + component): the
// reference's transpose + reshape to [B, A, width] is index arithmetic here.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"
#include "nms_kernels.hpp"
#include "radix_sort.hpp"

#include <algorithm>

namespace pd3 {

constexpr uint32_t kSsdKeyOut = 0x3FFFFFFFu;  // sorts after every kept anchor
constexpr uint32_t kSsdKeyOne = 0x3F800000u;  // bits of 1.0f

struct SsdCfg {
  int fh, fw, hw;          // head map
  int apl;                 // anchors per location
  int ncls;                // classes scored per anchor
  int cls_width, cls_skip; // channels per anchor in the class group; leading channels skipped (background)
  int cls0, box0, dir0;    // first channel of each group in the map (dir0 < 0: no direction classifier)
  int64_t batch_stride;    // elements between two frames of the map
  int num_anchors;         // hw * apl
  int nx, ny;              // pillar grid
  float area_threshold, score_threshold;
  int limit;               // centre range test on / off
  float lim[6];
};

__device__ __forceinline__ float ssd_exp(float x) { return (float)exp((double)x); }

__global__ __launch_bounds__(256) void ssd_occupancy_kernel(const int32_t* __restrict__ coors, int64_t m, int batch,
                                                            int ny, int nx, int* __restrict__ occ) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int b = coors[i * 4], y = coors[i * 4 + 2], x = coors[i * 4 + 3];
  if (b < 0 || b >= batch || (unsigned)y >= (unsigned)ny || (unsigned)x >= (unsigned)nx) return;  // padding rows
  atomicAdd(&occ[((int64_t)b * ny + y) * nx + x], 1);  // scatter_nd_add of ones (anchors_generator.py:191-195)
}

// grid (ny, batch), one wave: inclusive prefix along x
__global__ __launch_bounds__(64) void ssd_integral_rows_kernel(int* __restrict__ occ, int ny, int nx) {
  int* row = occ + ((int64_t)blockIdx.y * ny + blockIdx.x) * nx;
  int carry = 0;
  for (int x0 = 0; x0 < nx; x0 += kWave) {
    const int x = x0 + lane_id();
    const int v = x < nx ? row[x] : 0;
    const int inc = wave_inclusive_scan(v) + carry;
    if (x < nx) row[x] = inc;
    carry = __shfl(inc, kWave - 1, kWave);
  }
}

// thread per (x, frame): inclusive prefix along y, eight rows of loads in flight
__global__ __launch_bounds__(256) void ssd_integral_cols_kernel(int* __restrict__ occ, int ny, int nx) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nx) return;
  int* col = occ + (int64_t)blockIdx.y * ny * nx + x;
  int run = 0;
  int y = 0;
  for (; y + 8 <= ny; y += 8) {
    int v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = col[(int64_t)(y + j) * nx];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      run += v[j];
      col[(int64_t)(y + j) * nx] = run;
    }
  }
  for (; y < ny; ++y) {
    run += col[(int64_t)y * nx];
    col[(int64_t)y * nx] = run;
  }
}

// grid (ceil(hw / 256), apl, batch): thread = one anchor; neighbouring lanes read neighbouring map pixels
__global__ __launch_bounds__(256) void ssd_decode_kernel(const float* __restrict__ maps, SsdCfg c,
                                                         const float* __restrict__ anchors,
                                                         const int32_t* __restrict__ anchors_bv,
                                                         const int* __restrict__ integral,
                                                         float* __restrict__ boxes, float* __restrict__ scores,
                                                         int* __restrict__ labels, uint32_t* __restrict__ keys,
                                                         int* __restrict__ counts) {
  const int frame = blockIdx.z, j = blockIdx.y;
  const int loc = blockIdx.x * blockDim.x + threadIdx.x;
  int selected = 0;
  if (loc < c.hw) {
    const int a = loc * c.apl + j;  // index in the reference's [B, A, .] views (head.py:85-96 transpose + reshape)
    const int64_t o = (int64_t)frame * c.num_anchors + a;
    uint32_t key = kSsdKeyOut;
    // anchors_generator.py:103-121, :197-210: ID - IB - IC + IA on the summed occupancy, corners as the reference
    // takes them (D = (ymax, xmax), A = (ymin, xmin), B = (ymax, xmin), C = (ymin, xmax))
    const int* I = integral + (int64_t)frame * c.ny * c.nx;
    const int x0 = min(max(anchors_bv[a * 4 + 0], 0), c.nx - 1), y0 = min(max(anchors_bv[a * 4 + 1], 0), c.ny - 1);
    const int x1 = min(max(anchors_bv[a * 4 + 2], 0), c.nx - 1), y1 = min(max(anchors_bv[a * 4 + 3], 0), c.ny - 1);
    const int area = I[(int64_t)y1 * c.nx + x1] - I[(int64_t)y1 * c.nx + x0] - I[(int64_t)y0 * c.nx + x1] +
                     I[(int64_t)y0 * c.nx + x0];
    if ((float)area > c.area_threshold) {
      const float* m = maps + (int64_t)frame * c.batch_stride + loc;
      // head.py:145-150 sigmoid, max / argmax over the classes (first maximum wins)
      float best = 0.f;
      int arg = 0;
      for (int k = 0; k < c.ncls; ++k) {
        const float s = 1.0f / (1.0f + ssd_exp(-m[(int64_t)(c.cls0 + j * c.cls_width + c.cls_skip + k) * c.hw]));
        if (k == 0 || s > best) {
          best = s;
          arg = k;
        }
      }
      // pointpillars_coder.py:126-148
      const float* an = anchors + (int64_t)a * 7;
      const float xa = an[0], ya = an[1], za = an[2], wa = an[3], la = an[4], ha = an[5], ra = an[6];
      const float* e = m + (int64_t)(c.box0 + j * 7) * c.hw;
      const float diag = sqrtf(la * la + wa * wa);
      const float xg = e[0] * diag + xa;
      const float yg = e[(int64_t)c.hw] * diag + ya;
      const float zg = e[(int64_t)2 * c.hw] * ha + za;
      const float wg = ssd_exp(e[(int64_t)3 * c.hw]) * wa;
      const float lg = ssd_exp(e[(int64_t)4 * c.hw]) * la;
      const float hg = ssd_exp(e[(int64_t)5 * c.hw]) * ha;
      float rg = e[(int64_t)6 * c.hw] + ra;
      // head.py:156-161  score >= threshold, centre inside the limit range (both ends inclusive)
      bool kept = best >= c.score_threshold;
      if (c.limit)
        kept = kept && xg >= c.lim[0] && yg >= c.lim[1] && zg >= c.lim[2] && xg <= c.lim[3] && yg <= c.lim[4] &&
               zg <= c.lim[5];
      if (kept) {
        if (c.dir0 >= 0) {  // head.py:152-154 argmax of the two direction logits, :180-183 heading flip
          const float d0 = m[(int64_t)(c.dir0 + j * 2) * c.hw], d1 = m[(int64_t)(c.dir0 + j * 2 + 1) * c.hw];
          const bool dl = d1 > d0;
          if ((rg > 0.f) != dl) rg = rg + 3.14159265358979323846f;
        }
        float* bx = boxes + o * 7;
        bx[0] = xg;
        bx[1] = yg;
        bx[2] = zg + hg * 0.5f;  // :185 bottom centre -> object centre
        bx[3] = wg;
        bx[4] = lg;
        bx[5] = hg;
        bx[6] = rg;
        scores[o] = best;
        labels[o] = arg;
        const uint32_t bits = __float_as_uint(best);
        key = bits <= kSsdKeyOne ? kSsdKeyOne - bits : 0u;
        selected = 1;
      }
    }
    keys[o] = key;
  }
  const unsigned long long ball = __ballot(selected);
  __shared__ int wsum[4];
  if (lane_id() == 0) wsum[wave_id()] = __popcll(ball);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (s) atomicAdd(&counts[frame], s);
  }
}

// Top-K selection in place of the full sort: only the first min(kept, nms_pre_max_size) anchors of the stable
// ascending-key order are ever used.  One 1024-thread workgroup per frame, keys read from global memory (428 KB per
// KITTI frame: L2 resident): a 3-pass radix SELECT (10-bit LDS histograms) finds the exact cut-off key kc and how
// many anchors with key == kc still fit; the selected anchors are compacted IN ANCHOR ORDER (wave w owns a
// contiguous sixteenth of the anchors, ranks by ballot) and sorted as (key, anchor) pairs by a bitonic network --
// the same total order a stable key sort gives (layer_libs.py:231-233 argsort + [:pre_max_size]).
constexpr int kSsdTopkThreads = 1024;
constexpr int kSsdTopkMaxK = 1024;

__global__ __launch_bounds__(kSsdTopkThreads) void ssd_topk_kernel(const uint32_t* __restrict__ keys,
                                                                   const int* __restrict__ counts, int n, int cap,
                                                                   uint32_t* __restrict__ sidx) {
  __shared__ unsigned long long list[kSsdTopkMaxK];
  __shared__ int hist[1024];
  __shared__ int scr[kSsdTopkThreads / kWave + 2];
  __shared__ int pick[2];
  __shared__ int wless[kSsdTopkThreads / kWave], weq[kSsdTopkThreads / kWave];
  const int frame = blockIdx.x;
  const int count = counts[frame];
  const int K = min(count, cap);
  if (K <= 0) return;
  const uint32_t* kg = keys + (int64_t)frame * n;
  const int t = threadIdx.x, lane = lane_id(), wave = wave_id();
  list[t] = ~0ull;
  uint32_t kc = kSsdKeyOut;  // take every key < kc ...
  int r = 0;                 // ... and the first r anchors with key == kc
  if (count > K) {
    uint32_t prefix = 0;
    int need = K;
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = 20 - 10 * pass;
      hist[t] = 0;
      __syncthreads();
      auto tally = [&](uint32_t k) {
        if (k != kSsdKeyOut && (pass == 0 || (k >> (shift + 10)) == prefix)) atomicAdd(&hist[(k >> shift) & 1023u], 1);
      };
      if ((n & 3) == 0) {  // frames start 16-byte aligned then: four keys per load, four loads in flight
#pragma unroll 4
        for (int i = t * 4; i < n; i += kSsdTopkThreads * 4) {
          const uint4 kv = *reinterpret_cast<const uint4*>(kg + i);
          tally(kv.x);
          tally(kv.y);
          tally(kv.z);
          tally(kv.w);
        }
      } else {
#pragma unroll 8
        for (int i = t; i < n; i += kSsdTopkThreads) tally(kg[i]);
      }
      __syncthreads();
      const int h = hist[t];
      int total;
      const int base = block_exclusive_scan<kSsdTopkThreads>(h, scr, total);
      if (need > base && need <= base + h) {  // exactly one thread
        pick[0] = t;
        pick[1] = need - base;
      }
      __syncthreads();
      prefix = (prefix << 10) | (uint32_t)pick[0];
      need = pick[1];
      __syncthreads();
    }
    kc = prefix;
    r = need;
  }
  // order-preserving compaction: wave w owns anchors [w * per, (w + 1) * per)
  const int per = (int)ceil_div(ceil_div((int64_t)n, kSsdTopkThreads / kWave), kWave) * kWave;
  const int i0 = wave * per, i1 = min(i0 + per, n);
  int nless = 0, neq = 0;
#pragma unroll 4
  for (int i = i0 + lane; i < i1; i += kWave) {
    const uint32_t k = kg[i];
    nless += k < kc ? 1 : 0;
    neq += k == kc ? 1 : 0;
  }
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    nless += __shfl_xor(nless, d, kWave);
    neq += __shfl_xor(neq, d, kWave);
  }
  if (lane == 0) {
    wless[wave] = nless;
    weq[wave] = neq;
  }
  __syncthreads();
  int pos_less = 0, pos_eq = 0, tot_less = 0;
  for (int w = 0; w < kSsdTopkThreads / kWave; ++w) {
    if (w < wave) {
      pos_less += wless[w];
      pos_eq += weq[w];
    }
    tot_less += wless[w];
  }
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int ib = i0; ib < i1; ib += 4 * kWave) {  // uniform trip count per wave: the ballots see the whole wave
    uint32_t kq[4];  // four keys per lane in flight, then their four ordered appends
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = ib + q * kWave + lane;
      kq[q] = i < i1 ? kg[i] : kSsdKeyOut;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = ib + q * kWave + lane;
      const uint32_t k = kq[q];
      const bool isl = k < kc, ise = k == kc && kc != kSsdKeyOut;
      const unsigned long long bl = __ballot(isl), be = __ballot(ise);
      if (isl) list[pos_less + __popcll(bl & below)] = ((unsigned long long)k << 32) | (uint32_t)i;
      const int slot = pos_eq + __popcll(be & below);
      if (ise && slot < r) list[tot_less + slot] = ((unsigned long long)k << 32) | (uint32_t)i;
      pos_less += __popcll(bl);
      pos_eq += __popcll(be);
    }
  }
  __syncthreads();
  int n2 = 64;
  while (n2 < K) n2 <<= 1;  // uniform
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
      // thread t of wave w works on entries [128 w, 128 w + 128) at every stride up to 64: those steps follow each
      // other in the wave's own LDS order, a barrier is only needed around the strides of 128 and more
      if (stride >= 2 * kWave || (stride == 1 && size >= 2 * kWave))
        __syncthreads();
      else
        asm volatile("" ::: "memory");
    }
  }
  __syncthreads();
  if (t < K) sidx[(int64_t)frame * n + t] = (uint32_t)(list[t] & 0xffffffffull);
}

// layer_libs.py:222-229: columns (x, y, z, l, w, h, theta) and theta -> -theta - pi/2 in fp32
__global__ __launch_bounds__(256) void ssd_nms_boxes_kernel(const float* __restrict__ boxes,
                                                            const uint32_t* __restrict__ sidx,
                                                            const int* __restrict__ counts, int num_anchors,
                                                            int cap, float* __restrict__ nms_boxes,
                                                            BoxPre* __restrict__ pre, float4* __restrict__ xyr) {
  const int frame = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = min(counts[frame], cap);
  if (r >= n) return;
  const uint32_t a = sidx[(int64_t)frame * num_anchors + r];
  const float* bx = boxes + ((int64_t)frame * num_anchors + a) * 7;
  float* o = nms_boxes + ((int64_t)frame * cap + r) * 7;
  // (registers first: read back through `o`, every value is a store -> load round trip -- the rows may alias as far
  // as the compiler knows)
  const float nb[7] = {bx[0], bx[1], bx[2], bx[4], bx[3], bx[5], (-bx[6]) - 1.57079632679489661923f};
#pragma unroll
  for (int k = 0; k < 7; ++k) o[k] = nb[k];
  const BoxPre bp = box_prepare(nb);  // what the suppression matrix needs of this box
  pre[(int64_t)frame * cap + r] = bp;
  xyr[(int64_t)frame * cap + r] = make_float4(bp.cx, bp.cy, bp.rad, 0.f);
}

// grid (batch): the kept rows in NMS order; a frame without detections gets the reference's `_box_empty` row
// (zeros, score -1, label -1) in row 0 and count 0
__global__ __launch_bounds__(256) void ssd_output_kernel(const float* __restrict__ boxes,
                                                         const float* __restrict__ scores,
                                                         const int* __restrict__ labels,
                                                         const uint32_t* __restrict__ sidx,
                                                         const int* __restrict__ counts,
                                                         const int32_t* __restrict__ keep,
                                                         const int32_t* __restrict__ nkeep, int num_anchors, int cap,
                                                         int pre_max, int post_max, float* __restrict__ out_boxes,
                                                         float* __restrict__ out_scores,
                                                         int64_t* __restrict__ out_labels,
                                                         int32_t* __restrict__ out_count) {
  const int frame = blockIdx.x;
  const int rows_cap = max(post_max, 1);
  out_boxes += (int64_t)frame * rows_cap * 7;
  out_scores += (int64_t)frame * rows_cap;
  out_labels += (int64_t)frame * rows_cap;
  const int rows = (counts[frame] > 0 && pre_max > 0) ? min(nkeep[frame], post_max) : 0;
  if (rows <= 0) {
    if (threadIdx.x < 7) out_boxes[threadIdx.x] = 0.f;
    if (threadIdx.x == 0) {
      out_scores[0] = -1.f;
      out_labels[0] = -1;
      out_count[frame] = 0;
    }
    return;
  }
  for (int r = threadIdx.x; r < rows; r += blockDim.x) {
    const int pos = keep[(int64_t)frame * cap + r];
    const uint32_t a = sidx[(int64_t)frame * num_anchors + pos];
    const float* bx = boxes + ((int64_t)frame * num_anchors + a) * 7;
    float* o = out_boxes + (int64_t)r * 7;
    o[0] = bx[0];
    o[1] = bx[1];
    o[2] = bx[2] - bx[5] * 0.5f;  // head.py:196 object centre -> bottom centre
    o[3] = bx[3];
    o[4] = bx[4];
    o[5] = bx[5];
    o[6] = bx[6];
    out_scores[r] = scores[(int64_t)frame * num_anchors + a];
    out_labels[r] = labels[(int64_t)frame * num_anchors + a];
  }
  if (threadIdx.x == 0) out_count[frame] = rows;
}

struct SsdWorkspace {
  int *occ, *counts, *labels, *hist, *partial;
  float *boxes, *scores, *nms_boxes;
  uint32_t *keys_a, *vals_a, *keys_b, *vals_b;
  unsigned long long* mask;
  BoxPre* pre;
  NmsPool pool;
  int32_t *keep, *nkeep;
  size_t bytes;
};

static SsdWorkspace ssd_carve(void* base, int batch, int64_t num_anchors, int nx, int ny, int pre_max,
                              const RadixPlan& plan) {
  Carver c(base);
  SsdWorkspace w;
  const size_t ba = (size_t)batch * num_anchors;
  const int cap = std::max(pre_max, 1);
  const size_t cb = ((size_t)cap + 63) / 64;
  w.occ = c.take<int>((size_t)batch * ny * nx);
  w.counts = c.take<int>((size_t)batch);
  w.pool.counts = c.take<int>((size_t)batch * 2 * kNmsCtrStride);  // inside the span the set-up memset clears
  w.boxes = c.take<float>(ba * 7);
  w.scores = c.take<float>(ba);
  w.labels = c.take<int>(ba);
  w.keys_a = c.take<uint32_t>(ba);
  w.vals_a = c.take<uint32_t>(ba);
  w.keys_b = c.take<uint32_t>(ba);
  w.vals_b = c.take<uint32_t>(ba);
  w.hist = c.take<int>((size_t)batch * radix_hist_ints(plan));
  w.partial = c.take<int>((size_t)batch * scan_num_tiles((int64_t)radix_hist_ints(plan)));
  w.nms_boxes = c.take<float>((size_t)batch * cap * 7);
  w.mask = c.take<unsigned long long>((size_t)batch * cap * cb);
  w.pre = c.take<BoxPre>((size_t)batch * cap);
  w.pool.xyr = c.take<float4>((size_t)batch * cap);
  w.pool.per_set = nms_pool_per_set(cap);
  w.pool.pairs = c.take<uint32_t>((size_t)batch * w.pool.per_set);
  w.pool.tiles = c.take<uint32_t>((size_t)batch * cb * cb);
  w.keep = c.take<int32_t>((size_t)batch * cap);
  w.nkeep = c.take<int32_t>((size_t)batch);
  w.bytes = c.off;
  return w;
}

}  // namespace pd3

using namespace pd3;

extern "C" size_t pd3_ssd_postprocess_workspace(int batch, int feat_h, int feat_w, int anchors_per_loc, int grid_x,
                                                int grid_y, int nms_pre_max_size) {
  if (batch <= 0 || feat_h <= 0 || feat_w <= 0 || anchors_per_loc <= 0 || grid_x <= 0 || grid_y <= 0 ||
      nms_pre_max_size < 0)
    return 0;
  const int64_t a = (int64_t)feat_h * feat_w * anchors_per_loc;
  return ssd_carve(nullptr, batch, a, grid_x, grid_y, nms_pre_max_size, radix_plan(kSsdKeyOut, a)).bytes;
}

extern "C" int pd3_ssd_postprocess(const float* head_map, int64_t batch_stride, int cls_channel0, int box_channel0,
                                   int dir_channel0, int batch, int feat_h, int feat_w, int anchors_per_loc,
                                   int num_classes, int encode_background_as_zeros, const float* anchors,
                                   const int32_t* anchors_bv, const int32_t* coors, int64_t num_coors, int grid_x,
                                   int grid_y, float anchor_area_threshold, float score_threshold,
                                   const float* center_limit_range, float nms_iou_threshold, int nms_pre_max_size,
                                   int nms_post_max_size, float* out_boxes, float* out_scores, int64_t* out_labels,
                                   int32_t* out_count, void* workspace, size_t workspace_bytes, void* stream,
                                   int selection) {
  if (selection < 0 || selection > 1) return PD3_EINVAL;
  if (!head_map || !anchors || !anchors_bv || (!coors && num_coors > 0) || !out_boxes || !out_scores ||
      !out_labels || !out_count || !workspace)
    return PD3_EINVAL;
  if (batch <= 0 || feat_h <= 0 || feat_w <= 0 || anchors_per_loc <= 0 || num_classes <= 0 || grid_x <= 0 ||
      grid_y <= 0 || num_coors < 0 || nms_pre_max_size < 0 || nms_post_max_size < 0 || cls_channel0 < 0 ||
      box_channel0 < 0)
    return PD3_EINVAL;
  const int64_t hw = (int64_t)feat_h * feat_w, a = hw * anchors_per_loc;
  if (a >= (int64_t)1 << 30 || (int64_t)batch * a >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  if (batch_stride < hw) return PD3_EINVAL;
  const int cap = std::max(nms_pre_max_size, 1);
  const int cb = (cap + 63) / 64;
  if (cb > kNmsMaxWords || batch > 65535 || anchors_per_loc > 65535) return PD3_EUNSUPPORTED;
  const RadixPlan plan = radix_plan(kSsdKeyOut, a);
  SsdWorkspace w = ssd_carve(workspace, batch, a, grid_x, grid_y, nms_pre_max_size, plan);
  if (workspace_bytes < w.bytes) return PD3_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);

  SsdCfg c;
  c.fh = feat_h;
  c.fw = feat_w;
  c.hw = (int)hw;
  c.apl = anchors_per_loc;
  c.ncls = num_classes;
  c.cls_skip = encode_background_as_zeros ? 0 : 1;  // head.py:145-148: the background logit is dropped
  c.cls_width = num_classes + c.cls_skip;
  c.cls0 = cls_channel0;
  c.box0 = box_channel0;
  c.dir0 = dir_channel0;
  c.batch_stride = batch_stride;
  c.num_anchors = (int)a;
  c.nx = grid_x;
  c.ny = grid_y;
  c.area_threshold = anchor_area_threshold;
  c.score_threshold = score_threshold;
  c.limit = center_limit_range ? 1 : 0;
  for (int k = 0; k < 6; ++k) c.lim[k] = center_limit_range ? center_limit_range[k] : 0.f;

  // occ and counts are neighbours in the workspace: one memset covers both
  hipError_t e = hipMemsetAsync(w.occ, 0, (size_t)((char*)w.boxes - (char*)w.occ), s);
  if (e != hipSuccess) return (int)e;
  if (num_coors > 0)
    ssd_occupancy_kernel<<<(unsigned)ceil_div(num_coors, 256), 256, 0, s>>>(coors, num_coors, batch, grid_y, grid_x,
                                                                            w.occ);
  ssd_integral_rows_kernel<<<dim3(grid_y, batch), 64, 0, s>>>(w.occ, grid_y, grid_x);
  ssd_integral_cols_kernel<<<dim3((unsigned)ceil_div(grid_x, 256), batch), 256, 0, s>>>(w.occ, grid_y, grid_x);
  ssd_decode_kernel<<<dim3((unsigned)ceil_div(hw, 256), anchors_per_loc, batch), 256, 0, s>>>(
      head_map, c, anchors, anchors_bv, w.occ, w.boxes, w.scores, w.labels, w.keys_a, w.counts);
  const uint32_t* sidx;
  if (selection == 0 && cap <= kSsdTopkMaxK) {
    ssd_topk_kernel<<<batch, kSsdTopkThreads, 0, s>>>(w.keys_a, w.counts, (int)a, cap, w.vals_a);
    sidx = w.vals_a;
  } else {  // the reference's own selection: a full stable sort of all anchors' keys
    const int where = enqueue_radix_sort(w.keys_a, w.vals_a, w.keys_b, w.vals_b, a, a, batch, plan,
                                         /*identity_vals=*/true, w.hist, w.partial, s);
    sidx = where ? w.vals_b : w.vals_a;
  }
  ssd_nms_boxes_kernel<<<dim3((cap + 255) / 256, batch), 256, 0, s>>>(w.boxes, sidx, w.counts, (int)a, cap,
                                                                      w.nms_boxes, w.pre, w.pool.xyr);
  nms_enqueue_mask_pooled(w.pre, w.counts, batch, cap, cb, nms_iou_threshold, w.mask, w.pool, s);
  const size_t lds = nms_sweep_lds(cap);
  if (lds > 48 * 1024) {
    e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(nms_sweep_kernel), (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  nms_sweep_kernel<<<batch, kNmsSweepThreads, lds, s>>>(w.mask, w.counts, 0, cap, cb, w.keep, w.nkeep);
  ssd_output_kernel<<<batch, 256, 0, s>>>(w.boxes, w.scores, w.labels, sidx, w.counts, w.keep, w.nkeep, (int)a, cap,
                                          nms_pre_max_size, nms_post_max_size, out_boxes, out_scores, out_labels,
                                          out_count);
  return launch_status();
}
