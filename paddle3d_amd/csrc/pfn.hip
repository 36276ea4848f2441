// Pillar feature net (eval mode) and VoxelMean for gfx950.
// (reference: paddle3d/models/voxel_encoders/pillar_encoder.py:156-210 PillarFeatureNet.forward,
//  :81-105 PFNLayer.forward; voxel_encoder.py:30-40 get_paddings_indicator, :44-57 VoxelMean.)
//
// One wave per pillar, lane = output channel.  The reference materialises [M, P, 64] intermediates
// (154 MB per nuScenes frame) between a dozen elementwise kernels; here a pillar's rows never leave
// LDS: decorate -> Linear/BN/ReLU -> max over the rows -> (concat) -> Linear/BN/ReLU -> max, and only
// the [M, C] result is written.  The max over the P rows is the wavefront segmented reduce: every
// lane owns one channel and folds the rows of ITS pillar, so no cross-lane traffic is needed.
// Padded rows (k >= num_points) are all identical after the reference's mask multiply (a zero input
// row -> relu(shift)), so ONE pad row is evaluated and included in the max, as the reference's max over
// all P rows does.  BatchNorm arrives folded (scale, shift).  fmaf is explicit (library builds with
// -ffp-contract=off); results match the torch fp32 statement within 1e-4 (tests assert 1e-3 abs).
#include "../../include/paddle3d_amd.h"
#include "common.hpp"

#include <algorithm>

namespace pd3 {

// LDS written by some lanes of a wave and read by others: DS ops of one wave execute in order, so only
// the compiler must be kept from reordering across this point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Same ordering point without the fence: a wavefront-scope fence also drains vmcnt, i.e. it would wait for the
// next pillar's prefetch loads at every LDS hand-off.  The LDS queue of a wave is in order, so a compiler-level
// barrier is all the streaming kernels need.
__device__ __forceinline__ void wave_lds_order() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

constexpr int kPfnMaxWaves = 8;
constexpr int kPfnRowChunk = 8;

struct PfnArgs {
  const float* voxels;
  const int32_t* num_points;
  const int32_t* coors;
  int64_t m;
  int p, d;
  float vx, vy, vz, x_off, y_off, z_off;
  int center_dims;  // 2: PillarFeatureNet (x, y); 3: HardVFE (x, y, z)
  const float *w1, *scale1, *shift1;
  int c1;
  const float *w2, *scale2, *shift2;
  int c2;
  float* out;
  int in_dim, in_pad;
  // indexed form (pd3_pillar_feature_net_indexed): the pillars' points are read from the point cloud itself through
  // the voxelizer's index (pd3_hard_voxelize_index), no padded [M, P, D] tensor exists
  const float* points = nullptr;     // [frames, n_pts, D]
  int64_t n_pts = 0;
  const uint2* span = nullptr;       // [M] (start in the frame's list, stored points)
  const uint32_t* plist = nullptr;   // [frames, list_stride] point indices, a voxel's points consecutive
  int64_t list_stride = 0;
  int vper = 0;                      // pillars per frame (a multiple of the chunk: a chunk never straddles frames)
};

__global__ __launch_bounds__(kPfnMaxWaves * 64) void pfn_kernel(PfnArgs a) {
  const int kPfnWaves = blockDim.x / kWave;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = lane_id(), wave = wave_id();
  const bool two = a.w2 != nullptr;
  float* W1 = smem;                                   // [in_pad][c1], rows >= in_dim are zero
  float* W2 = W1 + a.in_pad * a.c1;                   // [2*c1][c2]
  float* wave_base = W2 + (two ? 2 * a.c1 * a.c2 : 0);
  const int xs_sz = a.p * a.in_pad, y1_sz = a.p * a.c1;
  float* xs = wave_base + wave * (xs_sz + y1_sz);     // [rows][in_pad]
  float* y1s = xs + xs_sz;                            // [rows][c1]

  for (int i = threadIdx.x; i < a.in_pad * a.c1; i += blockDim.x)
    W1[i] = (i / a.c1 < a.in_dim) ? a.w1[i] : 0.f;
  if (two)
    for (int i = threadIdx.x; i < 2 * a.c1 * a.c2; i += blockDim.x) W2[i] = a.w2[i];
  __syncthreads();

  const float sc1 = lane < a.c1 ? a.scale1[lane] : 0.f, sh1 = lane < a.c1 ? a.shift1[lane] : 0.f;
  const float sc2 = (two && lane < a.c2) ? a.scale2[lane] : 0.f;
  const float sh2 = (two && lane < a.c2) ? a.shift2[lane] : 0.f;
  const int out_c = two ? a.c2 : a.c1;

  const int64_t stride = (int64_t)gridDim.x * kPfnWaves;
  for (int64_t pil = (int64_t)blockIdx.x * kPfnWaves + wave; pil < a.m; pil += stride) {
    const int np_raw = a.num_points[pil];
    if (np_raw <= 0) {  // padding row of a fixed-shape [B*V] batch: no pillar here
      if (lane < out_c) a.out[pil * out_c + lane] = 0.f;
      continue;
    }
    const int np = min(np_raw, a.p);
    const int rows = np + (np < a.p ? 1 : 0);  // + one representative padded row
    const float* vox = a.voxels + pil * a.p * a.d;
    // ---- decorate (pillar_encoder.py:166-199) ------------------------------------------------
    // cluster mean over the stored points; the reference divides by num_points without epsilon.
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int k = 0; k < np; ++k) {  // uniform loads (broadcast), P <= a few dozen
      sx += vox[k * a.d + 0];
      sy += vox[k * a.d + 1];
      sz += vox[k * a.d + 2];
    }
    const float cnt = (float)np_raw;
    const float mx = sx / cnt, my = sy / cnt, mz = sz / cnt;
    const float pcx = (float)a.coors[pil * 4 + 3] * a.vx + a.x_off;
    const float pcy = (float)a.coors[pil * 4 + 2] * a.vy + a.y_off;
    const float pcz = (float)a.coors[pil * 4 + 1] * a.vz + a.z_off;  // HardVFE only (voxel_encoder.py:262-264)
    for (int e = lane; e < rows * a.in_pad; e += 64) {
      const int k = e / a.in_pad, i = e - k * a.in_pad;
      float v = 0.f;
      if (k < np && i < a.in_dim) {
        if (i < a.d) v = vox[k * a.d + i];
        else if (i < a.d + 3) {
          const int ax = i - a.d;
          v = vox[k * a.d + ax] - (ax == 0 ? mx : (ax == 1 ? my : mz));
        } else {
          const int ax = i - a.d - 3;
          v = vox[k * a.d + ax] - (ax == 0 ? pcx : (ax == 1 ? pcy : pcz));
        }
      }
      xs[e] = v;
    }
    wave_lds_sync();  // xs visible to the whole wave
    // ---- layer 1: Linear(no bias) -> BN -> ReLU, max over rows (PFNLayer :81-105) ----------------
    float m1 = -INFINITY;
    for (int r0 = 0; r0 < rows; r0 += kPfnRowChunk) {
      float acc[kPfnRowChunk];
#pragma unroll
      for (int r = 0; r < kPfnRowChunk; ++r) acc[r] = 0.f;
      if (lane < a.c1) {
        for (int i4 = 0; i4 < a.in_pad; i4 += 4) {
          const float w0 = W1[(i4 + 0) * a.c1 + lane], w1v = W1[(i4 + 1) * a.c1 + lane];
          const float w2v = W1[(i4 + 2) * a.c1 + lane], w3 = W1[(i4 + 3) * a.c1 + lane];
#pragma unroll
          for (int r = 0; r < kPfnRowChunk; ++r) {
            if (r0 + r < rows) {
              const float4 x = *reinterpret_cast<const float4*>(xs + (r0 + r) * a.in_pad + i4);
              acc[r] = fmaf(x.x, w0, acc[r]);
              acc[r] = fmaf(x.y, w1v, acc[r]);
              acc[r] = fmaf(x.z, w2v, acc[r]);
              acc[r] = fmaf(x.w, w3, acc[r]);
            }
          }
        }
#pragma unroll
        for (int r = 0; r < kPfnRowChunk; ++r) {
          if (r0 + r < rows) {
            const float y = fmaxf(fmaf(acc[r], sc1, sh1), 0.f);
            m1 = fmaxf(m1, y);
            if (two) y1s[(r0 + r) * a.c1 + lane] = y;
          }
        }
      }
    }
    if (!two) {
      if (lane < out_c) a.out[pil * out_c + lane] = m1;
      wave_lds_sync();
      continue;
    }
    // ---- layer 2 on [y1 | max(y1)] (the concat of :100-104) -------------------------------------
    // every lane needs all c1 maxima: park them in xs, whose rows are dead after layer 1
    float* mx_store = xs;
    wave_lds_sync();
    if (lane < a.c1) mx_store[lane] = m1;
    wave_lds_sync();
    float base = 0.f;  // row-independent half of the dot product: sum_i m1[i] * W2[c1 + i][c]
    if (lane < a.c2)
      for (int i = 0; i < a.c1; ++i) base = fmaf(mx_store[i], W2[(a.c1 + i) * a.c2 + lane], base);
    float m2 = -INFINITY;
    for (int r0 = 0; r0 < rows; r0 += kPfnRowChunk) {
      float acc[kPfnRowChunk];
#pragma unroll
      for (int r = 0; r < kPfnRowChunk; ++r) acc[r] = base;
      if (lane < a.c2) {
        for (int i4 = 0; i4 < a.c1; i4 += 4) {
          const float w0 = W2[(i4 + 0) * a.c2 + lane], w1v = W2[(i4 + 1) * a.c2 + lane];
          const float w2v = W2[(i4 + 2) * a.c2 + lane], w3 = W2[(i4 + 3) * a.c2 + lane];
#pragma unroll
          for (int r = 0; r < kPfnRowChunk; ++r) {
            if (r0 + r < rows) {
              const float4 x = *reinterpret_cast<const float4*>(y1s + (r0 + r) * a.c1 + i4);
              acc[r] = fmaf(x.x, w0, acc[r]);
              acc[r] = fmaf(x.y, w1v, acc[r]);
              acc[r] = fmaf(x.z, w2v, acc[r]);
              acc[r] = fmaf(x.w, w3, acc[r]);
            }
          }
        }
#pragma unroll
        for (int r = 0; r < kPfnRowChunk; ++r)
          if (r0 + r < rows) m2 = fmaxf(m2, fmaxf(fmaf(acc[r], sc2, sh2), 0.f));
      }
    }
    if (lane < a.c2) a.out[pil * a.c2 + lane] = m2;
    wave_lds_sync();  // next pillar overwrites xs / y1s
  }
}

// ---------------------------------------------------------------------------------------------------
// Fast path for the single-layer, 64-channel PillarFeatureNet of PointPillars / CenterPoint-Pillars
// (pillar_encoder.py:156-210 with feat_channels (64,)): lane = output channel with its column of W in
// registers; a wave streams pillars: the pillar's <= 128 raw floats arrive with one coalesced load per lane
// (prefetched one pillar ahead), are parked in a wave-private LDS line and read back as broadcasts, and
// only the real points are evaluated (+ the one representative padded row, see the header comment).  The
// arithmetic is the same fmaf chain, in the same order, as pfn_kernel.
template <int D, int CD>
__global__ __launch_bounds__(256) void pfn_single64_kernel(PfnArgs a) {
  constexpr int IN = D + 3 + CD;
  __shared__ float line[4][128];
  const int lane = lane_id(), wave = wave_id();
  float w[IN];
#pragma unroll
  for (int i = 0; i < IN; ++i) w[i] = a.w1[i * 64 + lane];
  const float sc = a.scale1[lane], sh = a.shift1[lane];
  const float ypad = fmaxf(sh, 0.f);  // a zero input row: relu(bn(0))
  const int pd = a.p * D;             // <= 128 floats per pillar
  float* ln = line[wave];
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t pil = (int64_t)blockIdx.x * 4 + wave;
  if (pil >= a.m) return;
  // prefetch registers for the pillar about to be processed
  float v0 = 0.f, v1 = 0.f;
  int npn = a.num_points[pil];
  int c1 = a.coors[pil * 4 + 1], c2 = a.coors[pil * 4 + 2], c3 = a.coors[pil * 4 + 3];
  if (lane < pd) v0 = a.voxels[pil * pd + lane];
  if (lane + 64 < pd) v1 = a.voxels[pil * pd + 64 + lane];
  for (; pil < a.m; pil += stride) {
    const int np_raw = npn;
    const float pcx = (float)c3 * a.vx + a.x_off, pcy = (float)c2 * a.vy + a.y_off;
    const float pcz = (float)c1 * a.vz + a.z_off;
    ln[lane] = v0;
    ln[64 + lane] = v1;
    wave_lds_order();
    const int64_t nxt = pil + stride;
    if (nxt < a.m) {  // next pillar's loads fly while this one is evaluated
      npn = a.num_points[nxt];
      c1 = a.coors[nxt * 4 + 1];
      c2 = a.coors[nxt * 4 + 2];
      c3 = a.coors[nxt * 4 + 3];
      v0 = lane < pd ? a.voxels[nxt * pd + lane] : 0.f;
      v1 = lane + 64 < pd ? a.voxels[nxt * pd + 64 + lane] : 0.f;
    }
    if (np_raw <= 0) {  // padding row of a fixed-shape [B*V] batch: no pillar here
      a.out[pil * 64 + lane] = 0.f;
      wave_lds_order();
      continue;
    }
    const int np = min(np_raw, a.p);
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int k = 0; k < np; ++k) {
      sx += ln[k * D + 0];
      sy += ln[k * D + 1];
      sz += ln[k * D + 2];
    }
    const float cnt = (float)np_raw;
    const float mx = sx / cnt, my = sy / cnt, mz = sz / cnt;
    float m1 = np < a.p ? ypad : -INFINITY;
    for (int k = 0; k < np; ++k) {
      float f[IN];
#pragma unroll
      for (int i = 0; i < D; ++i) f[i] = ln[k * D + i];
      f[D + 0] = f[0] - mx;
      f[D + 1] = f[1] - my;
      f[D + 2] = f[2] - mz;
      f[D + 3] = f[0] - pcx;
      f[D + 4] = f[1] - pcy;
      if (CD == 3) f[D + 3 + CD - 1] = f[2] - pcz;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < IN; ++i) acc = fmaf(f[i], w[i], acc);
      m1 = fmaxf(m1, fmaxf(fmaf(acc, sc, sh), 0.f));
    }
    a.out[pil * 64 + lane] = m1;
    wave_lds_order();  // the next trip overwrites the line
  }
}

// ---------------------------------------------------------------------------------------------------
// Fast path for the two-layer PillarFeatureNet of CenterPoint-Pillars (feat_channels (64, 64): PFNLayer 1 is
// Linear(in, 32), layer 2 Linear([y1 | max y1] = 64, 64); pillar_encoder.py:81-105, :156-210) on the fp32
// matrix cores.  One wave per pillar, streaming pillars; the pillar's rows (its real points + the one
// representative padded row) form one or two 16-row MFMA blocks:
//   layer 1:  Y1[16 x 32] = X[16 x 12] W1       3 k-steps x 2 column blocks of v_mfma_f32_16x16x4_f32
//   layer 2:  Y2[16 x 64] = Y1 W2[0:32] + base   8 k-steps x 4 column blocks, accumulator preset to
//             base = max_rows(Y1) W2[32:64] (the row-independent half of the concat, VALU, lane = channel)
// Both weight matrices live in registers as MFMA B operands for the whole kernel.  Lane (row, k) builds its
// three decorated features of layer 1's A operand in registers from the wave-private copy of the pillar; Y1
// passes through a wave-private LDS tile (row stride 34 floats: conflict-free A-operand reads); rows past the
// pillar's last one repeat point 0, so no row masks are needed in the two max reductions.  The only global
// traffic is the pillar's raw floats (one coalesced load, prefetched a pillar ahead) and its 64 outputs.
// The same kernel serves HardVFE's two VFE layers (voxel_encoder.py:142-283: Linear(10, 64), then
// Linear([y1 | max y1] = 128, 64), P = 64, D = 4): template C1 = first layer's width (32 / 64), MAXP = rows of a
// pillar (32 / 64), LINE = floats of a pillar's raw copy (128 / 256).
typedef float pfn_f32x4 __attribute__((ext_vector_type(4)));

template <int D, int CD, int C1, int MAXP, int LINE>
__global__ __launch_bounds__(256, C1 == 32 ? 4 : 2) void pfn_two_mfma_kernel(PfnArgs a) {
  constexpr int IN = D + 3 + CD;
  static_assert(IN <= 12, "layer-1 K is padded to 12");
  constexpr int NB1 = C1 / 16;   // layer-1 column blocks
  constexpr int KS2 = C1 / 4;    // layer-2 k-steps over the point-wise half of the concat
  constexpr int YS = C1 + 2;     // y1s row stride: conflict-free A-operand reads
  constexpr int NV = LINE / 64;  // raw floats per lane
  constexpr int kWaveFloats = LINE + MAXP * YS + C1 + 64 + 4 * 64;  // line, y1s, m1, base, part
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = lane_id(), wave = wave_id();
  const int r16 = lane & 15, g = lane >> 4;
  float* ln = smem + wave * kWaveFloats;    // raw pillar, [k][D]
  float* y1s = ln + LINE;                   // [MAXP rows][YS]
  float* m1s = y1s + MAXP * YS;             // [C1]
  float* bases = m1s + C1;                  // [64]
  float* part = bases + 64;                 // [4][64]
  // ---- weights and folded BatchNorm in registers ------------------------------------------------
  // W2's second half (the weights of the max-pooled concat half) in registers for the wide variant; for C1 = 32 it
  // lives in LDS, shared by the four waves: the kernel is a chain of dependent LDS / MFMA steps per pillar, and the 32
  // registers decide between three and four waves per SIMD to hide it behind
  constexpr bool kW2bLds = C1 == 32;
  float* w2bs = smem + 4 * kWaveFloats;  // [C1][64] (kW2bLds)
  float w1r[3][NB1], w2a[KS2][4], w2b[kW2bLds ? 1 : C1];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int cb = 0; cb < NB1; ++cb) {
      const int k = ks * 4 + g;
      w1r[ks][cb] = k < IN ? a.w1[k * C1 + cb * 16 + r16] : 0.f;
    }
#pragma unroll
  for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) w2a[ks][cb] = a.w2[(ks * 4 + g) * 64 + cb * 16 + r16];
  if (kW2bLds) {
    for (int i = threadIdx.x; i < C1 * 64; i += 256) w2bs[i] = a.w2[C1 * 64 + i];
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < (kW2bLds ? 1 : C1); ++i) w2b[i] = a.w2[(C1 + i) * 64 + lane];
  }
  float sc1[NB1], sh1[NB1], sc2[4], sh2[4];
#pragma unroll
  for (int cb = 0; cb < NB1; ++cb) {
    sc1[cb] = a.scale1[cb * 16 + r16];
    sh1[cb] = a.shift1[cb * 16 + r16];
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    sc2[cb] = a.scale2[cb * 16 + r16];
    sh2[cb] = a.shift2[cb * 16 + r16];
  }
  // this lane's A-operand features: k-step ks carries decorated feature i = 4 ks + g of point lane & 15
  int fsrc[3], fkind[3];  // source coordinate; 0 raw, 1 minus cluster mean, 2 minus pillar centre, 3 padding (zero)
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int i = ks * 4 + g;
    fkind[ks] = i >= IN ? 3 : (i < D ? 0 : (i < D + 3 ? 1 : 2));
    fsrc[ks] = i >= IN ? 0 : (i < D ? i : (i < D + 3 ? i - D : i - D - 3));
  }
  const int pd = a.p * D;  // <= LINE floats per pillar
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t pil = (int64_t)blockIdx.x * 4 + wave;
  if (pil >= a.m) return;
  float vreg[NV];
  int npn = a.num_points[pil];
  int c1 = a.coors[pil * 4 + 1], c2 = a.coors[pil * 4 + 2], c3 = a.coors[pil * 4 + 3];
#pragma unroll
  for (int q = 0; q < NV; ++q) vreg[q] = lane + 64 * q < pd ? a.voxels[pil * pd + 64 * q + lane] : 0.f;
  for (; pil < a.m; pil += stride) {
    const int np_raw = npn;
    float pc[3];
    pc[0] = (float)c3 * a.vx + a.x_off;
    pc[1] = (float)c2 * a.vy + a.y_off;
    pc[2] = (float)c1 * a.vz + a.z_off;
#pragma unroll
    for (int q = 0; q < NV; ++q) ln[64 * q + lane] = vreg[q];
    wave_lds_order();
    const int64_t nxt = pil + stride;
    if (nxt < a.m) {  // next pillar's loads fly while this one is evaluated
      npn = a.num_points[nxt];
      c1 = a.coors[nxt * 4 + 1];
      c2 = a.coors[nxt * 4 + 2];
      c3 = a.coors[nxt * 4 + 3];
#pragma unroll
      for (int q = 0; q < NV; ++q) vreg[q] = lane + 64 * q < pd ? a.voxels[nxt * pd + 64 * q + lane] : 0.f;
    }
    if (np_raw <= 0) {  // padding row of a fixed-shape [B*V] batch: no pillar here
      a.out[pil * 64 + lane] = 0.f;
      wave_lds_order();
      continue;
    }
    const int np = min(np_raw, a.p);
    const int rows = np + (np < a.p ? 1 : 0);  // + one representative padded (all-zero) row
    const int nblk = (rows + 15) >> 4;
    // ---- decorate (pillar_encoder.py:166-199) straight into the layer-1 A operand ------------------
    float mean[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < np; ++k) {
      mean[0] += ln[k * D + 0];
      mean[1] += ln[k * D + 1];
      mean[2] += ln[k * D + 2];
    }
    const float cnt = (float)np_raw;
    mean[0] = mean[0] / cnt;
    mean[1] = mean[1] / cnt;
    mean[2] = mean[2] / cnt;
    float sub[3];  // what this lane's three features subtract: 0, a cluster mean or a pillar centre
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const float mv = fsrc[ks] == 0 ? mean[0] : (fsrc[ks] == 1 ? mean[1] : mean[2]);
      const float pv = fsrc[ks] == 0 ? pc[0] : (fsrc[ks] == 1 ? pc[1] : pc[2]);
      sub[ks] = fkind[ks] == 1 ? mv : (fkind[ks] == 2 ? pv : 0.f);
    }
    // ---- layer 1 on the matrix cores; Y1 = relu(bn1(X W1)) -> y1s ---------------------------------
    // rows >= `rows` repeat point 0, so every row of a block is a legitimate member of the max below
    float pm1[NB1];
#pragma unroll
    for (int cb = 0; cb < NB1; ++cb) pm1[cb] = -INFINITY;
    for (int b = 0; b < nblk; ++b) {
      const int k = b * 16 + r16;
      const bool real = k < np, zero_row = (k == np) && (np < a.p);
      const int kk = real ? k : 0;
      pfn_f32x4 acc1[NB1];
#pragma unroll
      for (int cb = 0; cb < NB1; ++cb) acc1[cb] = (pfn_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        float av = ln[kk * D + fsrc[ks]] - sub[ks];
        av = (zero_row || fkind[ks] == 3) ? 0.f : av;
#pragma unroll
        for (int cb = 0; cb < NB1; ++cb)
          acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w1r[ks][cb], acc1[cb], 0, 0, 0);
      }
#pragma unroll
      for (int cb = 0; cb < NB1; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y = fmaxf(fmaf(acc1[cb][r], sc1[cb], sh1[cb]), 0.f);
          y1s[(b * 16 + 4 * g + r) * YS + cb * 16 + r16] = y;
          pm1[cb] = fmaxf(pm1[cb], y);
        }
    }
    // ---- max over the rows of Y1 (the concat half of PFNLayer :100-104), base = m1 W2[C1:2 C1] ------
#pragma unroll
    for (int cb = 0; cb < NB1; ++cb) part[g * 64 + cb * 16 + r16] = pm1[cb];
    wave_lds_order();
    if (lane < C1)
      m1s[lane] = fmaxf(fmaxf(part[lane], part[64 + lane]), fmaxf(part[128 + lane], part[192 + lane]));
    wave_lds_order();
    float base = 0.f;
#pragma unroll
    for (int i4 = 0; i4 < C1; i4 += 4) {
      const pfn_f32x4 mv = *reinterpret_cast<const pfn_f32x4*>(m1s + i4);
      if (kW2bLds) {
        base = fmaf(mv[0], w2bs[(i4 + 0) * 64 + lane], base);
        base = fmaf(mv[1], w2bs[(i4 + 1) * 64 + lane], base);
        base = fmaf(mv[2], w2bs[(i4 + 2) * 64 + lane], base);
        base = fmaf(mv[3], w2bs[(i4 + 3) * 64 + lane], base);
      } else {
        base = fmaf(mv[0], w2b[kW2bLds ? 0 : i4 + 0], base);
        base = fmaf(mv[1], w2b[kW2bLds ? 0 : i4 + 1], base);
        base = fmaf(mv[2], w2b[kW2bLds ? 0 : i4 + 2], base);
        base = fmaf(mv[3], w2b[kW2bLds ? 0 : i4 + 3], base);
      }
    }
    bases[lane] = base;
    wave_lds_order();
    // ---- layer 2 on the matrix cores, accumulators preset to the base; max over the rows --------------
    float pm[4], bs[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      pm[cb] = -INFINITY;
      bs[cb] = bases[cb * 16 + r16];
    }
    for (int b = 0; b < nblk; ++b) {
      pfn_f32x4 acc2[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc2[cb] = (pfn_f32x4){bs[cb], bs[cb], bs[cb], bs[cb]};
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        const float av = y1s[(b * 16 + r16) * YS + ks * 4 + g];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w2a[ks][cb], acc2[cb], 0, 0, 0);
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pm[cb] = fmaxf(pm[cb], fmaxf(fmaf(acc2[cb][r], sc2[cb], sh2[cb]), 0.f));
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) part[g * 64 + cb * 16 + r16] = pm[cb];
    wave_lds_order();
    a.out[pil * 64 + lane] = fmaxf(fmaxf(part[lane], part[64 + lane]), fmaxf(part[128 + lane], part[192 + lane]));
    wave_lds_order();  // the next trip overwrites the tile
  }
}

// ---------------------------------------------------------------------------------------------------
// Packed form of the two-layer fast path (C1 = 32, C2 = 64).  The form above spends one wave and one or two
// 16-row MFMA blocks on every pillar, whatever its fill level; a nuScenes sweep has 4.5 points per pillar
// on average (half of the pillars hold one), so most of the rows it multiplies are padding and, worse, every
// pillar pays the full stage -> mean -> layer 1 -> max -> base -> layer 2 -> max chain of dependent LDS / MFMA
// steps.  Here a wave takes a CHUNK of 8 consecutive pillars, packs their stored points back to back into 16-row
// blocks, and runs both layers block by block:
//   * a pillar's maximum over the rows of layer 2 commutes with everything that follows the GEMM: with
//     t_r = y1_r W2[0:32], the output is max_r relu(bn2(t_r + base)) = relu(bn2(ext_r t_r + base)) where
//     ext is the maximum for a non-negative folded scale and the minimum for a negative one (fp32 add, fma
//     and max are monotone, so this is exact, not approximate).  The sign is folded into the columns of the
//     B operand, so the walk below only takes maxima.  base = max_r(y1_r) W2[32:64] therefore is needed once
//     per pillar, AFTER its last row, and a block can run layer 2 right behind layer 1 without knowing the
//     pillar's maximum yet: no row tile survives a block, and pillars may straddle blocks;
//   * the padded rows of a pillar that is not full are all the same row whatever the pillar (zero input ->
//     y1 = relu(shift1), t = relu(shift1) W2[0:32]): two per-kernel constant vectors that the running maxima
//     of such a pillar START from, so no padded row is ever multiplied;
//   * the segmented maxima are taken by lane = channel walking the 16 rows of the block's two LDS tiles; the
//     rows that end a pillar are a wave-uniform 16-bit mask (one ballot), so the walk is scalar control flow.
//     A finished pillar parks its two maxima in the LDS slot of its own raw points (dead by then);
//   * after the chunk's last block, base for all 8 pillars is ONE more 16-row MFMA block (rows = pillars),
//     and BatchNorm 2 + ReLU are applied once per pillar and channel.
// Per chunk (8 pillars, 36 points on average) that is 2.8 blocks x 38 + 32 MFMAs where the per-pillar form issues
// 8.9 x 38, and one latency chain per chunk instead of eight.
constexpr int kPkPillars = 8;

// v_max_f32 as it is: fmaxf() first canonicalises both operands (two more v_max per call) because it cannot
// know that they are not signalling NaNs; the walk below is nothing but maxima.
__device__ __forceinline__ float pk_max(float x, float y) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// C1 = width of the first layer (32: PillarFeatureNet of the pillar models; 64: HardVFE of the BEVFusion LiDAR stream,
// voxel_encoder.py:142-283 -- the same two-layer algebra with Linear(10, 64), Linear([64 | 64], 64) and up to 64 points
// per pillar), PC = pillars per chunk (8; 4 for HardVFE, whose pillars are 256 floats each: two workgroups per CU).
// W2G: W2[C1:2 C1] (the B operand of the `base` block, 16 KB for C1 = 64) is read from global memory / L1 instead of
// being staged in LDS, which is what lets eight-pillar chunks of 8 KB of raw floats keep two workgroups on a CU.
// (Tried with it and dropped: not loading the 64-float pieces behind a pillar's last stored point -- 22 of a HardVFE
// pillar's 256 floats are stored on average -- with num_points travelling two chunks ahead; every form of the
// conditional loads made the register allocator spill 32-219 registers at the 256 this kernel may use.)
// IDX (pd3_pillar_feature_net_indexed, the model path of CenterPoint-Pillars): the chunk's stored points come straight
// from the point cloud through the voxelizer's index -- lane = point slot (pillar, k) of the chunk: the pillars' (start,
// count) words travel two chunks ahead, the index-list entries one chunk ahead, the points' D floats are requested
// before a chunk's blocks are multiplied and parked in the same LDS layout the copy of a padded tensor would have
// produced at the top of the next chunk.  The padded tensor (78 % zeros, 199 MB per 16 frames written by the row
// writer and 180 MB read back here) never exists; everything after the staging is the same code: same bytes out.
template <int D, int CD, int NV, int C1 = 32, int PC = kPkPillars, bool W2G = false, bool IDX = false, int NSL = 4>
__global__ __launch_bounds__(256, C1 == 32 ? 3 : 2) void pfn_packed_kernel(PfnArgs a) {
  constexpr int IN = D + 3 + CD;
  static_assert(IN <= 12, "layer-1 K is padded to 12");
  static_assert((C1 == 32 || C1 == 64) && (PC == 8 || PC == 4), "shapes the walk and the finishing step are written for");
  constexpr int CB1 = C1 / 16;          // 16-column blocks of layer 1
  constexpr int KS2 = C1 / 4;           // K steps of layer 2 (both halves of the concat)
  constexpr int LPP = 64 / PC;          // lanes per pillar in the cluster-mean step
  constexpr int REP = 16 / PC;          // copies of the chunk's pillars among the 16 rows of the `base` block
  constexpr int CBF = 4 / REP;          // column blocks a lane group finishes
  // row strides of the two LDS tiles: conflict-free for the A-operand reads (lane = (row, k group)) and the stores
  constexpr int YS = C1 == 32 ? 34 : 68, AS = 68;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = lane_id(), wave = wave_id();
  const int r16 = lane & 15, g = lane >> 4;
  const int pd = a.p * D;  // PC * pd <= NV * 64 (dispatch)
  const int rcap = (PC * a.p + 15) & ~15;
  const bool park_in_ln = pd >= C1 + 64;  // a pillar's C1 + 64 maxima fit the slot of its raw points
  const int ln_floats = NV * 64;
  const int wave_floats = ln_floats + 16 * YS + 16 * AS + PC * 8 + rcap + 16 + (park_in_ln ? 0 : PC * (C1 + 64));
  float* ln = smem + wave * wave_floats;            // the chunk's raw pillars, [pillar][k][D]
  float* y1T = ln + ln_floats;                      // [16 rows][YS]: layer-1 output of the current block
  float* accT = y1T + 16 * YS;                      // [16 rows][AS]: y1 W2[0:32] (sign-folded) of the block
  float* sub = accT + 16 * AS;                      // [PC][8]: cluster mean xyz, pillar centre xyz, 0, count
  int* rinfo = reinterpret_cast<int*>(sub + PC * 8);  // [rcap]: pillar | k << 4 | last row << 13
  int* ends = rinfo + rcap;                         // [PC] inclusive row prefix, [PC] stored points
  float* park = park_in_ln ? ln : reinterpret_cast<float*>(ends + 16);  // [PC][pst]: max y1 (C1), max t (64)
  const int pst = park_in_ln ? pd : C1 + 64;
  float* w2bs = smem + 4 * wave_floats;             // [C1][64]: W2[C1:2 C1], shared by the four waves
  // ---- weights and folded BatchNorm in registers ------------------------------------------------
  float w1r[3][CB1], w2a[KS2][4], sc1[CB1], sh1[CB1], sc2m[CBF], sh2m[CBF];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int cb = 0; cb < CB1; ++cb) {
      const int k = ks * 4 + g;
      w1r[ks][cb] = k < IN ? a.w1[k * C1 + cb * 16 + r16] : 0.f;
    }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const float sg = a.scale2[cb * 16 + r16] < 0.f ? -1.f : 1.f;
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) w2a[ks][cb] = a.w2[(ks * 4 + g) * 64 + cb * 16 + r16] * sg;
  }
  if (!W2G) {
    for (int i = threadIdx.x; i < C1 * 64; i += 256) w2bs[i] = a.w2[C1 * 64 + i];
    __syncthreads();
  }
  const float* w2b = W2G ? a.w2 + C1 * 64 : w2bs;
#pragma unroll
  for (int cb = 0; cb < CB1; ++cb) {
    sc1[cb] = a.scale1[cb * 16 + r16];
    sh1[cb] = a.shift1[cb * 16 + r16];
  }
  // the finishing step: the 16 rows of the `base` block hold the chunk's PC pillars REP times; the lanes of copy
  // `fin` (PC = 8: row groups 0, 1 -> copy 0, groups 2, 3 -> copy 1; PC = 4: group g -> copy g) take column blocks
  // fin * CBF .. fin * CBF + CBF - 1
  const int fin = (4 * g) / PC;
#pragma unroll
  for (int cb = 0; cb < CBF; ++cb) {
    sc2m[cb] = a.scale2[(fin * CBF + cb) * 16 + r16];
    sh2m[cb] = a.shift2[(fin * CBF + cb) * 16 + r16];
  }
  // the padded row (lane = channel): y1 = relu(bn1(0)), t = y1 W2[0:C1] with the column's sign folded in
  const float y1pad = fmaxf(a.shift1[lane & (C1 - 1)], 0.f);
  float tpad = 0.f;
  {
    const float sg = a.scale2[lane] < 0.f ? -1.f : 1.f;
    for (int c = 0; c < C1; ++c) tpad = fmaf(fmaxf(a.shift1[c], 0.f), a.w2[c * 64 + lane] * sg, tpad);
  }
  // this lane's A-operand features: k-step ks carries decorated feature i = 4 ks + g of the row lane & 15
  int fsrc[3], fsub[3];  // source coordinate; slot of `sub` that is subtracted (6 = the zero slot)
  bool fzero[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int i = ks * 4 + g;
    fzero[ks] = i >= IN;
    fsrc[ks] = i >= IN ? 0 : (i < D ? i : (i < D + 3 ? i - D : i - D - 3));
    fsub[ks] = (i >= IN || i < D) ? 6 : (i < D + 3 ? i - D : 3 + (i - D - 3));
  }
  const int64_t nchunks = (a.m + PC - 1) / PC;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t ch = (int64_t)blockIdx.x * 4 + wave;
  if (ch >= nchunks) return;
  float vreg[IDX ? 1 : NV];
  int npn = 0, c1 = 0, c2 = 0, c3 = 0;
  // ---- indexed staging (IDX) --------------------------------------------------------------------
  // NSL = point slots per lane: PC * p <= 64 NSL (3 for the 20-point pillars of the nuScenes models)
  float pf[IDX ? NSL : 1][D];                // the D floats of this lane's slots, chunk `ch`
  uint32_t sw1 = 0, yx1 = 0, sw2 = 0, yx2 = 0;  // (start | count << 24), (x | y << 16) of pillar `lane`: chunk + 1, + 2
  int z1 = 0, z2 = 0;
  const float inv_p = 1.0f / (float)a.p, inv_vper = IDX ? 1.0f / (float)a.vper : 0.f;
  const int nslots = PC * a.p;
  // frame of a chunk (a chunk never straddles frames; exact in fp32 for pillar ids below 2^22, the dispatcher's bound)
  auto frame_of = [&](int64_t c) { return (int64_t)(int)(((float)(c * PC) + 0.5f) * inv_vper); };
  auto slot_of = [&](int j, int& q, int& k) {  // slot lane + 64 j = (pillar q of the chunk, point k)
    const int sl = lane + 64 * j;
    q = (int)(((float)sl + 0.5f) * inv_p);
    k = sl - q * a.p;
  };
  auto load_span = [&](int64_t c, uint32_t& w, uint32_t& yx, int& z) {
    const int64_t q0 = c * PC;
    const int cntp = (int)(a.m - q0 < PC ? a.m - q0 : PC);
    const int ql = min(lane, cntp - 1);
    const uint2 sp = a.span[q0 + ql];
    const int32_t* co = a.coors + (q0 + ql) * 4;
    z = co[1];
    yx = (uint32_t)co[3] | ((uint32_t)co[2] << 16);
    w = lane < cntp ? ((sp.x & 0xFFFFFFu) | (min(sp.y, (uint32_t)a.p) << 24)) : 0u;
  };
  uint32_t pidx[IDX ? NSL : 1];
  auto issue_idx = [&](int64_t c, uint32_t w) {
    const uint32_t* lst = a.plist + frame_of(c) * a.list_stride;
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
      int q, k;
      slot_of(j, q, k);
      const uint32_t wq = (uint32_t)__shfl((int)w, min(q, PC - 1), kWave);
      const bool lv = lane + 64 * j < nslots && k < (int)(wq >> 24);
      pidx[IDX ? j : 0] = lv ? lst[(wq & 0xFFFFFFu) + (uint32_t)k] : 0xFFFFFFFFu;
    }
  };
  auto issue_pts = [&](int64_t c) {
    const float* pb = a.points + frame_of(c) * a.n_pts * D;
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
      const uint32_t ix = pidx[IDX ? j : 0];
      if (ix != 0xFFFFFFFFu) {
        const float* src = pb + (int64_t)ix * D;
#pragma unroll
        for (int e = 0; e < D; ++e) pf[IDX ? j : 0][e] = src[e];
      }
    }
  };
  auto fetch = [&](int64_t c) {
    const int64_t q0 = c * PC;
    const int cntp = (int)(a.m - q0 < PC ? a.m - q0 : PC);
    const int lim = cntp * pd;
    const float* src = a.voxels + q0 * pd;
#pragma unroll
    for (int q = 0; q < (IDX ? 1 : NV); ++q) vreg[q] = src[min(lane + 64 * q, lim - 1)];  // clamped, not predicated: no branches;
    const int ql = min(lane, cntp - 1);                                         // floats past `lim` belong to no pillar
    npn = a.num_points[q0 + ql];
    c1 = a.coors[(q0 + ql) * 4 + 1];
    c2 = a.coors[(q0 + ql) * 4 + 2];
    c3 = a.coors[(q0 + ql) * 4 + 3];
    npn = lane < cntp ? npn : 0;
  };
  if (IDX) {
    load_span(ch, sw1, yx1, z1);
    issue_idx(ch, sw1);
    issue_pts(ch);
    if (ch + stride < nchunks) load_span(ch + stride, sw2, yx2, z2);
  } else {
    fetch(ch);
  }
  for (; ch < nchunks; ch += stride) {
    const int64_t p0 = ch * PC;
    if (IDX) {
      // this chunk's words (sw1 ..) and points (pf) arrived during the previous chunk: park the stored points where
      // the copy of a padded tensor would have put them ([pillar][k][D]; slots without a point keep stale floats that
      // nothing reads: the means and the row layout stop at a pillar's count)
#pragma unroll
      for (int j = 0; j < NSL; ++j) {
        int q, k;
        slot_of(j, q, k);
        const uint32_t wq = (uint32_t)__shfl((int)sw1, min(q, PC - 1), kWave);
        if (lane + 64 * j < nslots && k < (int)(wq >> 24)) {
#pragma unroll
          for (int e = 0; e < D; ++e) ln[q * pd + k * D + e] = pf[IDX ? j : 0][e];
        }
      }
      npn = (int)(sw1 >> 24);
      c1 = z1;
      c2 = (int)(yx1 >> 16);
      c3 = (int)(yx1 & 0xFFFFu);
      sw1 = sw2;
      yx1 = yx2;
      z1 = z2;
    } else {
#pragma unroll
      for (int q = 0; q < (IDX ? 1 : NV); ++q) ln[64 * q + lane] = vreg[q];
    }
    const int np_raw = npn;
    const float pcx = (float)c3 * a.vx + a.x_off, pcy = (float)c2 * a.vy + a.y_off;
    const float pcz = (float)c1 * a.vz + a.z_off;  // HardVFE-style centres only (CD == 3)
    wave_lds_order();
    if (IDX) {
      if (ch + stride < nchunks) issue_idx(ch + stride, sw1);           // (sw1 now holds the next chunk's words)
      if (ch + 2 * stride < nchunks) load_span(ch + 2 * stride, sw2, yx2, z2);
    } else if (ch + stride < nchunks) {
      fetch(ch + stride);  // the next chunk's loads fly while this one is evaluated
    }
    // ---- the chunk's row layout: lane q < PC is pillar p0 + q -------------------------------------
    const int np = np_raw > 0 ? min(np_raw, a.p) : 0;  // a padding row of a fixed-shape batch has no rows
    int end = np;
#pragma unroll
    for (int dlt = 1; dlt < PC; dlt <<= 1) {
      const int t = __shfl_up(end, dlt, PC);
      if ((lane & (PC - 1)) >= dlt) end += t;
    }
    const int R = __builtin_amdgcn_readlane(end, PC - 1);
    const unsigned live0 = (unsigned)__ballot(lane < PC && np > 0);
    const unsigned notfull = (unsigned)__ballot(lane < PC && np < a.p);
    unsigned long long empty = __ballot(lane < PC && np == 0 && p0 + lane < a.m);
    if (lane < PC) {
      ends[lane] = end;
      ends[PC + lane] = np;
      sub[lane * 8 + 3] = pcx;
      sub[lane * 8 + 4] = pcy;
      sub[lane * 8 + 5] = pcz;
      sub[lane * 8 + 6] = 0.f;
      sub[lane * 8 + 7] = (float)np_raw;
    }
    while (empty) {
      const int q = __builtin_ctzll(empty);
      empty &= empty - 1;
      a.out[(p0 + q) * 64 + lane] = 0.f;
    }
    wave_lds_order();
    // cluster means (pillar_encoder.py:166-176; the reference divides by num_points without epsilon): LPP lanes
    // per pillar sum every LPP-th stored point, a butterfly adds the partial sums
    {
      const int q = lane / LPP, j = lane % LPP;
      const int npl = ends[PC + q];
      const float* src = ln + q * pd;
      float sx = 0.f, sy = 0.f, sz = 0.f;
      for (int k = j; k < npl; k += LPP) {
        sx += src[k * D + 0];
        sy += src[k * D + 1];
        sz += src[k * D + 2];
      }
#pragma unroll
      for (int dlt = 1; dlt < LPP; dlt <<= 1) {
        sx += __shfl_xor(sx, dlt, LPP);
        sy += __shfl_xor(sy, dlt, LPP);
        sz += __shfl_xor(sz, dlt, LPP);
      }
      if (j < 3 && npl > 0) sub[q * 8 + j] = (j == 0 ? sx : (j == 1 ? sy : sz)) / sub[q * 8 + 7];
    }
    const int rb = (R + 15) & ~15;
    for (int r = lane; r < rb; r += 64) {
      int q = 0;
#pragma unroll
      for (int j = 0; j < PC; ++j) q += ends[j] <= r ? 1 : 0;
      int info = 0;  // rows past the chunk's last one: zero rows that no walk reads
      if (r < R) {
        const int st = q > 0 ? ends[q - 1] : 0;
        info = q | ((r - st) << 4) | ((r + 1 == ends[q] ? 1 : 0) << 13);
      }
      rinfo[r] = info;
    }
    wave_lds_order();
    // running maxima of the pillar the walk is in (lane = channel); they start from the padded row's values
    // where the pillar has padded rows
    if (IDX && ch + stride < nchunks) issue_pts(ch + stride);  // in flight while this chunk's blocks are multiplied
    unsigned live = live0;
    int cur = live ? __builtin_ctz(live) : 0;
    float m1 = (notfull >> cur) & 1u ? y1pad : -INFINITY;
    float m2 = (notfull >> cur) & 1u ? tpad : -INFINITY;
    const int nblk = rb >> 4;
    for (int b = 0; b < nblk; ++b) {
      const int info = rinfo[b * 16 + r16];
      const unsigned endmask = (unsigned)(__ballot((info >> 13) & 1) & 0xffffull);
      const int q = info & 15, k = (info >> 4) & 255;
      // ---- layer 1 on the matrix cores; Y1 = relu(bn1(X W1)) -> y1T --------------------------------
      pfn_f32x4 acc1[CB1];
#pragma unroll
      for (int cb = 0; cb < CB1; ++cb) acc1[cb] = (pfn_f32x4){0.f, 0.f, 0.f, 0.f};
      // rows past the chunk's last one (info = 0) read pillar 0's slot: whatever they hold stays in their own rows of
      // the products (MFMA rows are independent) and no walk reads those rows, so the loads need no predicate
      float av[3];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) av[ks] = ln[q * pd + k * D + fsrc[ks]] - sub[q * 8 + fsub[ks]];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const float v = fzero[ks] ? 0.f : av[ks];
#pragma unroll
        for (int cb = 0; cb < CB1; ++cb)
          acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, w1r[ks][cb], acc1[cb], 0, 0, 0);
      }
#pragma unroll
      for (int cb = 0; cb < CB1; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          y1T[(4 * g + r) * YS + cb * 16 + r16] = fmaxf(fmaf(acc1[cb][r], sc1[cb], sh1[cb]), 0.f);
      wave_lds_order();
      // ---- layer 2, point-wise half of the concat: T = Y1 W2[0:32] (columns sign-folded) -> accT ------
      pfn_f32x4 acc2[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc2[cb] = (pfn_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        const float av = y1T[r16 * YS + ks * 4 + g];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w2a[ks][cb], acc2[cb], 0, 0, 0);
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) accT[(4 * g + r) * AS + cb * 16 + r16] = acc2[cb][r];
      wave_lds_order();
      // ---- walk the block's rows: lane = channel; a pillar's last row parks its maxima ------------------
      float tv[16], yv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        tv[r] = accT[r * AS + lane];
        yv[r] = y1T[r * YS + (lane & (C1 - 1))];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        m2 = pk_max(m2, tv[r]);
        m1 = pk_max(m1, yv[r]);
        if ((endmask >> r) & 1u) {
          if (lane < C1) park[cur * pst + lane] = m1;
          park[cur * pst + C1 + lane] = m2;
          live &= live - 1;
          cur = live ? __builtin_ctz(live) : 0;
          m1 = (notfull >> cur) & 1u ? y1pad : -INFINITY;
          m2 = (notfull >> cur) & 1u ? tpad : -INFINITY;
        }
      }
      wave_lds_order();  // the next block overwrites the tiles
    }
    if (live0 == 0u) continue;
    // ---- base = max_rows(Y1) W2[C1:2 C1] (the row-independent half of the concat, PFNLayer :100-104, VFELayer
    // :128-138) for the chunk's pillars as one more MFMA block: row = pillar (rows PC..15 repeat 0..PC-1), then bn2 +
    // ReLU once per pillar and channel: copy `fin` of the pillars finishes column blocks fin * CBF ..
    {
      pfn_f32x4 accb[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) accb[cb] = (pfn_f32x4){0.f, 0.f, 0.f, 0.f};
      // four K steps at a time; with W2G the sixteen B values of a group are global loads, and the loop stays rolled:
      // all KS2 * 4 of them in flight at once would not fit the register file
      auto base_group = [&](int ks0) {
        float av[4], bw[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          av[i] = park[(r16 & (PC - 1)) * pst + (ks0 + i) * 4 + g];
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) bw[i][cb] = w2b[((ks0 + i) * 4 + g) * 64 + cb * 16 + r16];
        }
        if (W2G) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
            accb[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bw[i][cb], accb[cb], 0, 0, 0);
        if (W2G) __builtin_amdgcn_sched_barrier(0);
      };
      if (W2G) {
#pragma unroll 1
        for (int ks0 = 0; ks0 < KS2; ks0 += 4) base_group(ks0);
      } else {
#pragma unroll
        for (int ks0 = 0; ks0 < KS2; ks0 += 4) base_group(ks0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = (4 * g + r) & (PC - 1);
        if ((live0 >> q) & 1u) {
#pragma unroll
          for (int cb = 0; cb < CBF; ++cb) {
            const int c = (fin * CBF + cb) * 16 + r16;
            float bs;  // accb[fin * CBF + cb][r]: `fin` differs between lanes, so it is a select, not an index
            if (REP == 2) bs = fin ? accb[2 + cb][r] : accb[cb][r];
            else bs = fin == 0 ? accb[0][r] : (fin == 1 ? accb[1][r] : (fin == 2 ? accb[2][r] : accb[3][r]));
            const float t = park[q * pst + C1 + c];
            const float x = (sc2m[cb] < 0.f ? -t : t) + bs;
            a.out[(p0 + q) * 64 + c] = fmaxf(fmaf(x, sc2m[cb], sh2m[cb]), 0.f);
          }
        }
      }
    }
    wave_lds_order();  // the next chunk's raw points overwrite the parked maxima
  }
}

__global__ __launch_bounds__(256) void voxel_mean_kernel(const float* __restrict__ voxels,
                                                         const int32_t* __restrict__ num_points,
                                                         int64_t m, int p, int d,
                                                         float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m * d) return;
  const int64_t v = e / d;
  const int c = (int)(e - v * d);
  const float* src = voxels + v * p * d + c;
  float s = 0.f;
  for (int k = 0; k < p; ++k) s += src[k * d];  // padded rows are zero, as in the reference's sum
  out[e] = s / (float)num_points[v];
}

}  // namespace pd3

using namespace pd3;

static int pfn_dispatch(const float* voxels, const int32_t* num_points, const int32_t* coors,
                        int64_t num_pillars, int max_points, int num_point_dim, int voxel_center_dims,
                        float vx, float vy, float vz, float x_offset, float y_offset, float z_offset,
                        const float* w1, const float* scale1, const float* shift1, int c1, const float* w2,
                        const float* scale2, const float* shift2, int c2, float* out, int path,
                        void* stream) {
  if (path < 0 || path > 2) return PD3_EINVAL;
  if (num_pillars < 0 || max_points <= 0 || num_point_dim < 3) return PD3_EINVAL;
  if (voxel_center_dims != 2 && voxel_center_dims != 3) return PD3_EINVAL;
  if (num_pillars == 0) return 0;
  if (!voxels || !num_points || !coors || !w1 || !scale1 || !shift1 || !out) return PD3_EINVAL;
  if (c1 <= 0 || c1 > 64 || (c1 % 4) != 0) return PD3_EUNSUPPORTED;
  if (w2 && (!scale2 || !shift2 || c2 <= 0 || c2 > 64)) return PD3_EUNSUPPORTED;
  PfnArgs a;
  a.voxels = voxels;
  a.num_points = num_points;
  a.coors = coors;
  a.m = num_pillars;
  a.p = max_points;
  a.d = num_point_dim;
  a.vx = vx;
  a.vy = vy;
  a.vz = vz;
  a.x_off = x_offset;
  a.y_off = y_offset;
  a.z_off = z_offset;
  a.center_dims = voxel_center_dims;
  a.w1 = w1;
  a.scale1 = scale1;
  a.shift1 = shift1;
  a.c1 = c1;
  a.w2 = w2;
  a.scale2 = scale2;
  a.shift2 = shift2;
  a.c2 = w2 ? c2 : 0;
  a.out = out;
  a.in_dim = num_point_dim + 3 + voxel_center_dims;
  a.in_pad = (a.in_dim + 3) / 4 * 4;
  // xs must also be able to hold c1 maxima (see mx_store)
  if (max_points * a.in_pad < c1) return PD3_EUNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!w2 && c1 == 64 && max_points * num_point_dim <= 128 && (num_point_dim == 4 || num_point_dim == 5)) {
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(num_pillars, 4), 256 * 16);
    if (num_point_dim == 4 && voxel_center_dims == 2) pfn_single64_kernel<4, 2><<<blocks, 256, 0, s>>>(a);
    else if (num_point_dim == 4) pfn_single64_kernel<4, 3><<<blocks, 256, 0, s>>>(a);
    else if (voxel_center_dims == 2) pfn_single64_kernel<5, 2><<<blocks, 256, 0, s>>>(a);
    else pfn_single64_kernel<5, 3><<<blocks, 256, 0, s>>>(a);
    return launch_status();
  }
#define PD3_PFN_MFMA(DD, CDV, C1V, MAXPV, LINEV)                                                                \
  do {                                                                                                        \
    constexpr size_t lds_ = ((size_t)4 * (LINEV + MAXPV * (C1V + 2) + C1V + 64 + 4 * 64) +                     \
                             (C1V == 32 ? C1V * 64 : 0)) * sizeof(float);                                      \
    if (lds_ > 48 * 1024) {                                                                                    \
      hipError_t e_ = pd3_max_dynamic_lds(reinterpret_cast<const void*>(pfn_two_mfma_kernel<DD, CDV, C1V, MAXPV, LINEV>), (int)lds_);              \
      if (e_ != hipSuccess) return (int)e_;                                                                    \
    }                                                                                                          \
    pfn_two_mfma_kernel<DD, CDV, C1V, MAXPV, LINEV><<<blocks, 256, lds_, s>>>(a);                              \
  } while (0)
  // packed form (path 0 picks it where it applies; path 1 = the per-pillar forms below; path 2 = this or nothing)
  const bool packed_ok = w2 && c1 == 32 && c2 == 64 && max_points <= 32 && (num_point_dim == 4 || num_point_dim == 5) &&
                         a.in_dim <= 12 && kPkPillars * max_points * num_point_dim <= 20 * 64;
  // the same form for HardVFE's widths (Linear(10, 64), Linear([64 | 64], 64), up to 64 points of 4 floats): four
  // pillars per chunk
  const bool packed64_ok = w2 && c1 == 64 && c2 == 64 && max_points <= 64 && num_point_dim == 4 &&
                           voxel_center_dims == 3 && a.in_dim <= 12 && 8 * max_points * num_point_dim <= 32 * 64;
  if (path == 2 && !packed_ok && !packed64_ok) return PD3_EUNSUPPORTED;
  if (packed64_ok && path != 1) {
    // eight pillars per chunk (8 KB of raw floats per wave), W2[64:128] from L1 instead of LDS: 75 KB per workgroup,
    // two workgroups per CU
    constexpr int kPc = 8, kNv = 32, kC1 = 64;
    const int64_t nchunks = ceil_div(num_pillars, (int64_t)kPc);
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(nchunks, 4), 256 * 2);
    const int rcap = (kPc * max_points + 15) & ~15;
    const size_t lds = ((size_t)4 * (kNv * 64 + 16 * 68 + 16 * 68 + kPc * 8 + rcap + 16 +
                                     (max_points * num_point_dim >= kC1 + 64 ? 0 : kPc * (kC1 + 64)))) * sizeof(float);
    if (lds > 48 * 1024) {
      hipError_t e_ = pd3_max_dynamic_lds(reinterpret_cast<const void*>(pfn_packed_kernel<4, 3, kNv, kC1, kPc, true>), (int)lds);
      if (e_ != hipSuccess) return (int)e_;
    }
    pfn_packed_kernel<4, 3, kNv, kC1, kPc, true><<<blocks, 256, lds, s>>>(a);
    return launch_status();
  }
  if (packed_ok && path != 1) {
    const int64_t nchunks = ceil_div(num_pillars, (int64_t)kPkPillars);
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(nchunks, 4), 256 * 3);
    const int nv = (kPkPillars * max_points * num_point_dim + 63) / 64 <= 13 ? 13 : 20;
    const int rcap = (kPkPillars * max_points + 15) & ~15;
    const size_t lds = ((size_t)4 * (nv * 64 + 16 * 34 + 16 * 68 + kPkPillars * 8 + rcap + 16 +
                                     (max_points * num_point_dim >= 96 ? 0 : kPkPillars * 96)) + 32 * 64) * sizeof(float);
#define PD3_PFN_PACKED(DD, CDV, NVV)                                                                               \
  do {                                                                                                            \
    if (lds > 48 * 1024) {                                                                                        \
      hipError_t e_ = pd3_max_dynamic_lds(reinterpret_cast<const void*>(pfn_packed_kernel<DD, CDV, NVV>), (int)lds);                  \
      if (e_ != hipSuccess) return (int)e_;                                                                       \
    }                                                                                                             \
    pfn_packed_kernel<DD, CDV, NVV><<<blocks, 256, lds, s>>>(a);                                                  \
  } while (0)
    if (nv == 13) {
      if (num_point_dim == 4 && voxel_center_dims == 2) PD3_PFN_PACKED(4, 2, 13);
      else if (num_point_dim == 4) PD3_PFN_PACKED(4, 3, 13);
      else if (voxel_center_dims == 2) PD3_PFN_PACKED(5, 2, 13);
      else PD3_PFN_PACKED(5, 3, 13);
    } else {
      if (num_point_dim == 4 && voxel_center_dims == 2) PD3_PFN_PACKED(4, 2, 20);
      else if (num_point_dim == 4) PD3_PFN_PACKED(4, 3, 20);
      else if (voxel_center_dims == 2) PD3_PFN_PACKED(5, 2, 20);
      else PD3_PFN_PACKED(5, 3, 20);
    }
#undef PD3_PFN_PACKED
    return launch_status();
  }
  if (w2 && c1 == 32 && c2 == 64 && max_points * num_point_dim <= 128 && max_points <= 32 &&
      (num_point_dim == 4 || num_point_dim == 5) && a.in_dim <= 12) {
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(num_pillars, 4), 256 * 8);
    if (num_point_dim == 4 && voxel_center_dims == 2) PD3_PFN_MFMA(4, 2, 32, 32, 128);
    else if (num_point_dim == 4) PD3_PFN_MFMA(4, 3, 32, 32, 128);
    else if (voxel_center_dims == 2) PD3_PFN_MFMA(5, 2, 32, 32, 128);
    else PD3_PFN_MFMA(5, 3, 32, 32, 128);
    return launch_status();
  }
  if (w2 && c1 == 64 && c2 == 64 && max_points * num_point_dim <= 256 && max_points <= 64 &&
      num_point_dim == 4 && voxel_center_dims == 3 && a.in_dim <= 12) {  // HardVFE (BEVFusion LiDAR stream)
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(num_pillars, 4), 256 * 8);
    PD3_PFN_MFMA(4, 3, 64, 64, 256);
    return launch_status();
  }
#undef PD3_PFN_MFMA
  const size_t w_floats = (size_t)a.in_pad * c1 + (w2 ? (size_t)2 * c1 * c2 : 0);
  const size_t wave_floats = (size_t)max_points * a.in_pad + (size_t)max_points * c1;
  int waves = kPfnMaxWaves;
  while (waves > 1 && (w_floats + waves * wave_floats) * sizeof(float) > 160 * 1024) waves >>= 1;
  const size_t bytes = (w_floats + waves * wave_floats) * sizeof(float);
  if (bytes > 160 * 1024) return PD3_EUNSUPPORTED;
  if (bytes > 48 * 1024) {
    hipError_t e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(pfn_kernel), (int)bytes);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t blocks = std::min<int64_t>(ceil_div(num_pillars, waves), 256 * 8);
  pfn_kernel<<<(unsigned)blocks, waves * 64, bytes, s>>>(a);
  return launch_status();
}

extern "C" int pd3_pillar_feature_net_indexed(const float* points, int64_t points_per_frame, const int32_t* vox_span,
                                              const int32_t* point_list, int64_t list_stride, const int32_t* coors,
                                              int64_t num_pillars, int pillars_per_frame, int max_points,
                                              int num_point_dim, int voxel_center_dims, float vx, float vy, float vz,
                                              float x_offset, float y_offset, float z_offset, const float* w1,
                                              const float* scale1, const float* shift1, int c1, const float* w2,
                                              const float* scale2, const float* shift2, int c2, float* out,
                                              void* stream) {
  if (num_pillars < 0 || max_points <= 0 || num_point_dim < 3 || pillars_per_frame <= 0 || points_per_frame <= 0 ||
      list_stride <= 0)
    return PD3_EINVAL;
  if (voxel_center_dims != 2 && voxel_center_dims != 3) return PD3_EINVAL;
  if (num_pillars == 0) return 0;
  if (!points || !vox_span || !point_list || !coors || !w1 || !scale1 || !shift1 || !w2 || !scale2 || !shift2 || !out)
    return PD3_EINVAL;
  const int in_dim = num_point_dim + 3 + voxel_center_dims;
  // the shapes of the packed two-layer kernel (PillarFeatureNet of the pillar models), whole chunks per frame, grid
  // coordinates that fit 16 bits, pillar ids the fp32 frame lookup is exact for
  if (!(c1 == 32 && c2 == 64 && max_points <= 32 && (num_point_dim == 4 || num_point_dim == 5) && in_dim <= 12 &&
        kPkPillars * max_points * num_point_dim <= 20 * 64 && pillars_per_frame % kPkPillars == 0 &&
        num_pillars % pillars_per_frame == 0 && num_pillars < ((int64_t)1 << 22)))
    return PD3_EUNSUPPORTED;
  PfnArgs a{};
  a.voxels = nullptr;
  a.num_points = nullptr;
  a.coors = coors;
  a.m = num_pillars;
  a.p = max_points;
  a.d = num_point_dim;
  a.vx = vx;
  a.vy = vy;
  a.vz = vz;
  a.x_off = x_offset;
  a.y_off = y_offset;
  a.z_off = z_offset;
  a.center_dims = voxel_center_dims;
  a.w1 = w1;
  a.scale1 = scale1;
  a.shift1 = shift1;
  a.c1 = c1;
  a.w2 = w2;
  a.scale2 = scale2;
  a.shift2 = shift2;
  a.c2 = c2;
  a.out = out;
  a.in_dim = in_dim;
  a.in_pad = (in_dim + 3) / 4 * 4;
  a.points = points;
  a.n_pts = points_per_frame;
  a.span = reinterpret_cast<const uint2*>(vox_span);
  a.plist = reinterpret_cast<const uint32_t*>(point_list);
  a.list_stride = list_stride;
  a.vper = pillars_per_frame;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t nchunks = ceil_div(num_pillars, (int64_t)kPkPillars);
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(nchunks, 4), 256 * 3);
  const int nv = (kPkPillars * max_points * num_point_dim + 63) / 64 <= 13 ? 13 : 20;
  const int rcap = (kPkPillars * max_points + 15) & ~15;
  const size_t lds = ((size_t)4 * (nv * 64 + 16 * 34 + 16 * 68 + kPkPillars * 8 + rcap + 16 +
                                   (max_points * num_point_dim >= 96 ? 0 : kPkPillars * 96)) + 32 * 64) * sizeof(float);
#define PD3_PFN_INDEXED_N(DD, CDV, NVV, NS)                                                                           \
  do {                                                                                                               \
    if (lds > 48 * 1024) {                                                                                           \
      hipError_t e_ = pd3_max_dynamic_lds(reinterpret_cast<const void*>(pfn_packed_kernel<DD, CDV, NVV, 32, kPkPillars, false, true, NS>), (int)lds);                                                     \
      if (e_ != hipSuccess) return (int)e_;                                                                          \
    }                                                                                                                \
    pfn_packed_kernel<DD, CDV, NVV, 32, kPkPillars, false, true, NS><<<blocks, 256, lds, s>>>(a);                    \
  } while (0)
#define PD3_PFN_INDEXED(DD, CDV, NVV)                                                                                 \
  do {                                                                                                               \
    if (kPkPillars * max_points <= 192) PD3_PFN_INDEXED_N(DD, CDV, NVV, 3);                                          \
    else PD3_PFN_INDEXED_N(DD, CDV, NVV, 4);                                                                         \
  } while (0)
  if (nv == 13) {
    if (num_point_dim == 4 && voxel_center_dims == 2) PD3_PFN_INDEXED(4, 2, 13);
    else if (num_point_dim == 4) PD3_PFN_INDEXED(4, 3, 13);
    else if (voxel_center_dims == 2) PD3_PFN_INDEXED(5, 2, 13);
    else PD3_PFN_INDEXED(5, 3, 13);
  } else {
    if (num_point_dim == 4 && voxel_center_dims == 2) PD3_PFN_INDEXED(4, 2, 20);
    else if (num_point_dim == 4) PD3_PFN_INDEXED(4, 3, 20);
    else if (voxel_center_dims == 2) PD3_PFN_INDEXED(5, 2, 20);
    else PD3_PFN_INDEXED(5, 3, 20);
  }
#undef PD3_PFN_INDEXED
#undef PD3_PFN_INDEXED_N
  return launch_status();
}

extern "C" int pd3_pillar_feature_net(const float* voxels, const int32_t* num_points,
                                      const int32_t* coors, int64_t num_pillars, int max_points,
                                      int num_point_dim, int voxel_center_dims, float vx, float vy,
                                      float vz, float x_offset, float y_offset, float z_offset,
                                      const float* w1, const float* scale1,
                                      const float* shift1, int c1, const float* w2,
                                      const float* scale2, const float* shift2, int c2, float* out,
                                      void* stream) {
  return pfn_dispatch(voxels, num_points, coors, num_pillars, max_points, num_point_dim, voxel_center_dims, vx, vy,
                      vz, x_offset, y_offset, z_offset, w1, scale1, shift1, c1, w2, scale2, shift2, c2, out, 0,
                      stream);
}

extern "C" int pd3_pillar_feature_net_path(const float* voxels, const int32_t* num_points,
                                           const int32_t* coors, int64_t num_pillars, int max_points,
                                           int num_point_dim, int voxel_center_dims, float vx, float vy,
                                           float vz, float x_offset, float y_offset, float z_offset,
                                           const float* w1, const float* scale1,
                                           const float* shift1, int c1, const float* w2,
                                           const float* scale2, const float* shift2, int c2, float* out,
                                           int path, void* stream) {
  return pfn_dispatch(voxels, num_points, coors, num_pillars, max_points, num_point_dim, voxel_center_dims, vx, vy,
                      vz, x_offset, y_offset, z_offset, w1, scale1, shift1, c1, w2, scale2, shift2, c2, out, path,
                      stream);
}

extern "C" int pd3_voxel_mean(const float* voxels, const int32_t* num_points, int64_t num_voxels,
                              int max_points, int num_point_dim, float* out, void* stream) {
  if (num_voxels < 0 || max_points <= 0 || num_point_dim <= 0) return PD3_EINVAL;
  if (num_voxels == 0) return 0;
  if (!voxels || !num_points || !out) return PD3_EINVAL;
  const int64_t n = num_voxels * num_point_dim;
  voxel_mean_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      voxels, num_points, num_voxels, max_points, num_point_dim, out);
  return launch_status();
}
