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

extern "C" int pd3_pillar_feature_net(const float* voxels, const int32_t* num_points,
                                      const int32_t* coors, int64_t num_pillars, int max_points,
                                      int num_point_dim, int voxel_center_dims, float vx, float vy,
                                      float vz, float x_offset, float y_offset, float z_offset,
                                      const float* w1, const float* scale1,
                                      const float* shift1, int c1, const float* w2,
                                      const float* scale2, const float* shift2, int c2, float* out,
                                      void* stream) {
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
  const size_t w_floats = (size_t)a.in_pad * c1 + (w2 ? (size_t)2 * c1 * c2 : 0);
  const size_t wave_floats = (size_t)max_points * a.in_pad + (size_t)max_points * c1;
  int waves = kPfnMaxWaves;
  while (waves > 1 && (w_floats + waves * wave_floats) * sizeof(float) > 160 * 1024) waves >>= 1;
  const size_t bytes = (w_floats + waves * wave_floats) * sizeof(float);
  if (bytes > 160 * 1024) return PD3_EUNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pfn_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t blocks = std::min<int64_t>(ceil_div(num_pillars, waves), 256 * 8);
  pfn_kernel<<<(unsigned)blocks, waves * 64, bytes, s>>>(a);
  return launch_status();
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
