// bev_pool_v2 forward / backward for gfx950.
// (reference: paddle3d/ops/bev_pool_v2/bev_pool_cuda.cu:18-44, paddle3d/ops/bev_pool_v2_backward/
//  bev_pool_cuda_bkwd.cu:44-94; op wrappers bev_pool.cc:30-54, bev_pool_bkwd.cc:24-57.)
//
// Forward: one lane per (interval, channel); consecutive lanes walk consecutive channels, so every
// gathered feature row is a contiguous read and the output row a contiguous write.  The depth weight
// and the three rank words are wave-uniform per interval row and come from the scalar/L1 path.
// Accumulation is fp32 in interval order without FMA contraction, i.e. bit-identical to the reference
// kernel.  Backward splits the reference's one-thread-per-interval loop into its two independent
// halves: depth_grad per frustum point (serial over channels, reference order), feat_grad per
// (interval, channel).
#include "../../include/paddle3d_amd.h"
#include "common.hpp"

namespace pd3 {

__global__ __launch_bounds__(256) void bev_pool_fwd_kernel(
    int c, int n_intervals, const float* __restrict__ depth, const float* __restrict__ feat,
    const int* __restrict__ ranks_depth, const int* __restrict__ ranks_feat,
    const int* __restrict__ ranks_bev, const int* __restrict__ interval_starts,
    const int* __restrict__ interval_lengths, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int iv = (int)(idx / c);
  const int ch = (int)(idx - (int64_t)iv * c);
  if (iv >= n_intervals) return;
  const int s = interval_starts[iv], len = interval_lengths[iv];
  float acc = 0.f;
  for (int i = 0; i < len; ++i)
    acc += feat[(int64_t)ranks_feat[s + i] * c + ch] * depth[ranks_depth[s + i]];
  out[(int64_t)ranks_bev[s] * c + ch] = acc;
}

// depth_grad[ranks_depth[p]] = sum_c out_grad[ranks_bev[p]][c] * feat[ranks_feat[p]][c]
__global__ __launch_bounds__(256) void bev_pool_bwd_depth_kernel(
    int c, int n_points, const float* __restrict__ out_grad, const float* __restrict__ feat,
    const int* __restrict__ ranks_depth, const int* __restrict__ ranks_feat,
    const int* __restrict__ ranks_bev, float* __restrict__ depth_grad) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_points) return;
  const float* g = out_grad + (int64_t)ranks_bev[p] * c;
  const float* f = feat + (int64_t)ranks_feat[p] * c;
  float acc = 0.f;
  for (int ch = 0; ch < c; ++ch) acc += g[ch] * f[ch];
  depth_grad[ranks_depth[p]] = acc;
}

__global__ __launch_bounds__(256) void bev_pool_bwd_feat_kernel(
    int c, int n_intervals, const float* __restrict__ out_grad, const float* __restrict__ depth,
    const int* __restrict__ ranks_depth, const int* __restrict__ ranks_feat,
    const int* __restrict__ ranks_bev, const int* __restrict__ interval_starts,
    const int* __restrict__ interval_lengths, float* __restrict__ feat_grad) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int iv = (int)(idx / c);
  const int ch = (int)(idx - (int64_t)iv * c);
  if (iv >= n_intervals) return;
  const int s = interval_starts[iv], len = interval_lengths[iv];
  float acc = 0.f;
  for (int i = 0; i < len; ++i)
    acc += out_grad[(int64_t)ranks_bev[s + i] * c + ch] * depth[ranks_depth[s + i]];
  feat_grad[(int64_t)ranks_feat[s] * c + ch] = acc;
}

}  // namespace pd3

using namespace pd3;

extern "C" int pd3_bev_pool_v2(const float* depth, const float* feat, const int32_t* ranks_depth,
                               const int32_t* ranks_feat, const int32_t* ranks_bev,
                               const int32_t* interval_lengths, const int32_t* interval_starts,
                               int n_intervals, int channels, int64_t out_elems, float* out,
                               void* stream) {
  if (!out || out_elems < 0 || n_intervals < 0 || channels <= 0) return PD3_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)out_elems, s);  // bev_pool.cc:48-49
  if (e != hipSuccess) return (int)e;
  if (n_intervals == 0) return 0;
  if (!depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev || !interval_lengths ||
      !interval_starts)
    return PD3_EINVAL;
  const int64_t total = (int64_t)n_intervals * channels;
  bev_pool_fwd_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, s>>>(
      channels, n_intervals, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
      interval_lengths, out);
  return launch_status();
}

extern "C" int pd3_bev_pool_v2_bkwd(const float* out_grad, const float* depth, const float* feat,
                                    const int32_t* ranks_depth, const int32_t* ranks_feat,
                                    const int32_t* ranks_bev, const int32_t* interval_lengths,
                                    const int32_t* interval_starts, int n_intervals,
                                    int64_t n_points, int channels, int64_t depth_elems,
                                    int64_t feat_elems, float* depth_grad, float* feat_grad,
                                    void* stream) {
  if (!depth_grad || !feat_grad || depth_elems < 0 || feat_elems < 0 || n_intervals < 0 ||
      n_points < 0 || n_points >= ((int64_t)1 << 31) || channels <= 0)
    return PD3_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(depth_grad, 0, sizeof(float) * (size_t)depth_elems, s);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(feat_grad, 0, sizeof(float) * (size_t)feat_elems, s);  // bev_pool_bkwd.cc:41-46
  if (e != hipSuccess) return (int)e;
  if (n_intervals == 0 || n_points == 0) return 0;
  if (!out_grad || !depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev ||
      !interval_lengths || !interval_starts)
    return PD3_EINVAL;
  // The intervals tile the rank arrays (voxel_pooling_prepare_v2, bevdet_transformer.py:230-274), so
  // "for every point of every interval" is "for every listed point".
  bev_pool_bwd_depth_kernel<<<(unsigned)ceil_div(n_points, 256), 256, 0, s>>>(
      channels, (int)n_points, out_grad, feat, ranks_depth, ranks_feat, ranks_bev, depth_grad);
  const int64_t total = (int64_t)n_intervals * channels;
  bev_pool_bwd_feat_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, s>>>(
      channels, n_intervals, out_grad, depth, ranks_depth, ranks_feat, ranks_bev, interval_starts,
      interval_lengths, feat_grad);
  return launch_status();
}
