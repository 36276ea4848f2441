// bev_pool_v2 forward / backward for gfx950.
// (reference: paddle3d/ops/bev_pool_v2/bev_pool_cuda.cu:18-44, paddle3d/ops/bev_pool_v2_backward/
//  bev_pool_cuda_bkwd.cu:44-94; op wrappers bev_pool.cc:30-54, bev_pool_bkwd.cc:24-57.)
//
// Forward: one lane per (interval, channel); consecutive lanes walk consecutive channels, so every
// gathered feature row is a contiguous read and the output row a contiguous write.  The depth weight
// and the three rank words are wave-uniform per interval row and come from the scalar/L1 path.
// Accumulation is fp32 in interval order without FMA contraction, i.e. bit-identical to the reference
// kernel; the loads of sixteen consecutive points are issued together (the walk is latency-bound: the long
// intervals near the cameras set the kernel's duration).  Backward splits the reference's one-thread-per-interval loop into its two independent
// halves: depth_grad per frustum point (serial over channels, reference order), feat_grad per
// (interval, channel).
#include "../../include/paddle3d_amd.h"
#include "common.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"

#include <algorithm>

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
  // the walk is a chain of dependent loads (rank -> row): sixteen points are fetched together, the sum itself
  // stays in the reference's order
  constexpr int U = 16;
  int i = 0;
  for (; i + U <= len; i += U) {
    int rf[U], rd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rf[u] = ranks_feat[s + i + u];
      rd[u] = ranks_depth[s + i + u];
    }
    float f[U], d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f[u] = feat[(int64_t)rf[u] * c + ch];
      d[u] = depth[rd[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += f[u] * d[u];
  }
  // the remainder (most intervals are shorter than one batch: BEVFusion's camera pooling has a median of 6 points per
  // cell) the same way: all loads of the up to 15 points together (indices clamped to the interval's last point), the
  // sum in order over the points that exist -- a point-by-point tail is a chain of two dependent memory round trips per
  // point and set the kernel's time (round 6: 349 -> see profiles/r06_camera_pool.txt)
  if (i < len) {
    const int rem = len - i;
    int rf[U - 1], rd[U - 1];
#pragma unroll
    for (int u = 0; u < U - 1; ++u) {
      const int q = s + i + min(u, rem - 1);
      rf[u] = ranks_feat[q];
      rd[u] = ranks_depth[q];
    }
    float f[U - 1], d[U - 1];
#pragma unroll
    for (int u = 0; u < U - 1; ++u) {
      f[u] = feat[(int64_t)rf[u] * c + ch];
      d[u] = depth[rd[u]];
    }
#pragma unroll
    for (int u = 0; u < U - 1; ++u)
      if (u < rem) acc += f[u] * d[u];
  }
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
  constexpr int U = 16;  // as in the forward kernel: loads in batches, the sum in the reference's order
  int i = 0;
  for (; i + U <= len; i += U) {
    int rb[U], rd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rb[u] = ranks_bev[s + i + u];
      rd[u] = ranks_depth[s + i + u];
    }
    float g[U], d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      g[u] = out_grad[(int64_t)rb[u] * c + ch];
      d[u] = depth[rd[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += g[u] * d[u];
  }
  if (i < len) {  // (the remainder as in the forward kernel: loads together, the sum in order)
    const int rem = len - i;
    int rb[U - 1], rd[U - 1];
#pragma unroll
    for (int u = 0; u < U - 1; ++u) {
      const int q = s + i + min(u, rem - 1);
      rb[u] = ranks_bev[q];
      rd[u] = ranks_depth[q];
    }
    float g[U - 1], d[U - 1];
#pragma unroll
    for (int u = 0; u < U - 1; ++u) {
      g[u] = out_grad[(int64_t)rb[u] * c + ch];
      d[u] = depth[rd[u]];
    }
#pragma unroll
    for (int u = 0; u < U - 1; ++u)
      if (u < rem) acc += g[u] * d[u];
  }
  feat_grad[(int64_t)ranks_feat[s] * c + ch] = acc;
}


// ---------------------------------------------------------------------------------------------------------
// voxel_pooling_prepare: the index build in front of bev_pool_v2, on the device.
// (reference: LSSViewTransformer.voxel_pooling_prepare_v2, paddle3d/models/transformers/bevdet_transformer.py:
//  230-274 -- quantise every frustum point, drop the ones outside the grid, argsort by BEV rank, run-length
//  encode; and the same steps inside LiftSplatShoot.voxel_pooling, cam_stream_lss.py:318-346.)
//   1. vpp_key_kernel     point -> key = output cell (or `cells`, one past the last cell, outside the grid)
//   2. stable radix sort  (key, point index): points of a cell stay in index order
//   3. vpp_head_kernel    head flag per sorted position (first of its cell) + number of kept points
//   4. flag scan          interval id of every head; its epilogue writes interval_starts
//   5. vpp_finish_kernel  ranks_bev / ranks_depth / ranks_feat in sorted order, interval_lengths, counts
struct VppGrid {
  float lo[3], step[3], size[3];  // lower bound, interval, grid size as the reference's float tensors hold them
  int gx, gy, gz;                 // int(size)
};

// mode 0: BEVDet  rank = b * (gz*gy*gx) + z * (gy*gx) + y * gx + x        (bevdet_transformer.py:256-259)
// mode 1: LSS     cell = ((b * gz + z) * gx + x) * gy + y                  (cam_stream_lss.py:358-361)
// mode 2: LSS cell order with BEVDet's split operands (ranks_depth = point, ranks_feat = camera pixel): the form
//         that pools depth [B*N, D, H, W] and feat [B*N, H, W, C] without the lifted depth (x) feat tensor
__global__ __launch_bounds__(256) void vpp_key_kernel(const float* __restrict__ coor, int64_t n, int64_t per_batch,
                                                      VppGrid g, int mode, uint32_t outside,
                                                      uint32_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long long c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // ((coor - lower) / interval).cast('int64'): fp32 subtract and divide, truncation toward zero -- a point up to
    // one interval below the lower bound lands in cell 0, exactly as in the reference
    const float q = (coor[i * 3 + a] - g.lo[a]) / g.step[a];
    c[a] = (q >= -9.2e18f && q <= 9.2e18f) ? (long long)q : (long long)0x8000000000000000ull;  // NaN/inf: x86 cvttss2si
  }
  const bool kept = c[0] >= 0 && (float)c[0] < g.size[0] && c[1] >= 0 && (float)c[1] < g.size[1] && c[2] >= 0 &&
                    (float)c[2] < g.size[2];
  uint32_t key = outside;
  if (kept) {
    const long long b = i / per_batch;
    key = mode == 0 ? (uint32_t)(((b * g.gz + c[2]) * g.gy + c[1]) * g.gx + c[0])
                    : (uint32_t)(((b * g.gz + c[2]) * g.gx + c[0]) * g.gy + c[1]);
  }
  keys[i] = key;
}

__global__ __launch_bounds__(256) void vpp_head_kernel(const uint32_t* __restrict__ skey, int64_t n, uint32_t outside,
                                                       int* __restrict__ head, int* __restrict__ counts) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t k = skey[j];
  const bool valid = k != outside;
  head[j] = (valid && (j == 0 || skey[j - 1] != k)) ? 1 : 0;
  if (valid && (j + 1 == n || skey[j + 1] == outside)) counts[0] = (int)(j + 1);  // number of kept points
}

struct EpiIntervalStart {
  int* starts;
  __device__ __forceinline__ void operator()(int, int64_t i, int flag, int prefix, int) const {
    if (flag) starts[prefix] = (int)i;
  }
};

__global__ __launch_bounds__(256) void vpp_finish_kernel(const uint32_t* __restrict__ skey,
                                                         const uint32_t* __restrict__ sidx, int64_t n, int depth_bins,
                                                         int feat_hw, int mode, const int* __restrict__ scan_total,
                                                         int32_t* __restrict__ ranks_bev, int32_t* __restrict__ ranks_depth,
                                                         int32_t* __restrict__ ranks_feat,
                                                         const int32_t* __restrict__ starts,
                                                         int32_t* __restrict__ lengths, int* __restrict__ counts) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int n_int = scan_total[0], n_kept = counts[0];
  if (j == 0) counts[1] = n_int;
  if (j < n_int) lengths[j] = (j + 1 < n_int ? starts[j + 1] : n_kept) - starts[j];
  if (j >= n_kept) return;
  const uint32_t idx = sidx[j];
  ranks_bev[j] = (int32_t)skey[j];
  ranks_depth[j] = (int32_t)idx;
  // ranks_feat = arange(num_points // D).reshape(B, N, 1, H, W).expand(B, N, D, H, W) (:236-239)
  const uint32_t dhw = (uint32_t)depth_bins * (uint32_t)feat_hw;
  ranks_feat[j] = mode != 1 ? (int32_t)((idx / dhw) * (uint32_t)feat_hw + idx % (uint32_t)feat_hw) : (int32_t)idx;
}

// frustum template -> ego-frame coordinates of every frustum point (LSSViewTransformer.get_lidar_coor,
// bevdet_transformer.py:142-192): undo the image-space augmentation, un-project with the depth, camera -> ego,
// BEV augmentation.  One lane per point; the per-camera 3x3 matrices are wave-uniform scalar loads.
__global__ __launch_bounds__(256) void frustum_to_lidar_kernel(
    const float* __restrict__ frustum, int64_t per_cam, int cams_per_batch, int64_t n,
    const float* __restrict__ inv_post_rot, const float* __restrict__ post_trans,
    const float* __restrict__ cam_to_ego, const float* __restrict__ trans, const float* __restrict__ bda,
    float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t cam = i / per_cam, f = i - cam * per_cam;
  const float* a = inv_post_rot + cam * 9;
  const float* c = cam_to_ego + cam * 9;
  const float* bd = bda + (cam / cams_per_batch) * 9;
  float p[3], q[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = frustum[f * 3 + k] - post_trans[cam * 3 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) q[k] = a[k * 3 + 0] * p[0] + a[k * 3 + 1] * p[1] + a[k * 3 + 2] * p[2];
  p[0] = q[0] * q[2];
  p[1] = q[1] * q[2];
  p[2] = q[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) q[k] = c[k * 3 + 0] * p[0] + c[k * 3 + 1] * p[1] + c[k * 3 + 2] * p[2] + trans[cam * 3 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) out[i * 3 + k] = bd[k * 3 + 0] * q[0] + bd[k * 3 + 1] * q[1] + bd[k * 3 + 2] * q[2];
}

struct VppWorkspace {
  uint32_t *keys_a, *vals_a, *keys_b, *vals_b;
  int *head, *hist, *partial, *total;
  size_t bytes;
};

static VppWorkspace vpp_carve(void* base, int64_t n, const RadixPlan& plan) {
  Carver c(base);
  VppWorkspace w;
  w.keys_a = c.take<uint32_t>((size_t)n);
  w.vals_a = c.take<uint32_t>((size_t)n);
  w.keys_b = c.take<uint32_t>((size_t)n);
  w.vals_b = c.take<uint32_t>((size_t)n);
  w.head = c.take<int>((size_t)n);
  w.hist = c.take<int>(radix_hist_ints(plan));
  w.partial = c.take<int>((size_t)std::max(scan_num_tiles((int64_t)radix_hist_ints(plan)), scan_num_tiles(n)));
  w.total = c.take<int>(1);
  w.bytes = c.off;
  return w;
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

extern "C" size_t pd3_voxel_pooling_prepare_workspace(int64_t num_points) {
  if (num_points <= 0) return 0;
  // the widest [digit][tile] table a plan can ask for: 10-bit digits (a 20-bit key; wider keys take narrower digits)
  return vpp_carve(nullptr, num_points, radix_plan(0xFFFFFu, num_points)).bytes;
}

extern "C" int pd3_voxel_pooling_prepare(const float* coor, int64_t num_points, int batch, int depth_bins,
                                         int feat_hw, const float* grid_lower, const float* grid_interval,
                                         const float* grid_size, int mode, int32_t* ranks_bev, int32_t* ranks_depth,
                                         int32_t* ranks_feat, int32_t* interval_starts, int32_t* interval_lengths,
                                         int32_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
  if (!coor || !grid_lower || !grid_interval || !grid_size || !ranks_bev || !ranks_depth || !ranks_feat ||
      !interval_starts || !interval_lengths || !counts || !workspace)
    return PD3_EINVAL;
  if (num_points <= 0 || num_points >= ((int64_t)1 << 31) || batch <= 0 || num_points % batch != 0 ||
      depth_bins <= 0 || feat_hw <= 0 || mode < 0 || mode > 2)
    return PD3_EINVAL;
  VppGrid g;
  for (int a = 0; a < 3; ++a) {
    g.lo[a] = grid_lower[a];
    g.step[a] = grid_interval[a];
    g.size[a] = grid_size[a];
  }
  g.gx = (int)grid_size[0];
  g.gy = (int)grid_size[1];
  g.gz = (int)grid_size[2];
  if (g.gx <= 0 || g.gy <= 0 || g.gz <= 0) return PD3_EINVAL;
  const int64_t cells = (int64_t)batch * g.gx * g.gy * g.gz;
  if (cells >= (int64_t)0xFFFFFFFFll) return PD3_EUNSUPPORTED;
  const RadixPlan full = radix_plan((uint32_t)cells, num_points);  // keys 0 .. cells, `cells` = outside the grid
  VppWorkspace w = vpp_carve(workspace, num_points, full);
  if (workspace_bytes < w.bytes) return PD3_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), s);
  if (e != hipSuccess) return (int)e;
  const unsigned blocks = (unsigned)ceil_div(num_points, 256);
  vpp_key_kernel<<<blocks, 256, 0, s>>>(coor, num_points, num_points / batch, g, mode, (uint32_t)cells, w.keys_a);
  const int where = enqueue_radix_sort(w.keys_a, w.vals_a, w.keys_b, w.vals_b, num_points, num_points, 1, full,
                                       /*identity_vals=*/true, w.hist, w.partial, s);
  const uint32_t* skey = where ? w.keys_b : w.keys_a;
  const uint32_t* sidx = where ? w.vals_b : w.vals_a;
  vpp_head_kernel<<<blocks, 256, 0, s>>>(skey, num_points, (uint32_t)cells, w.head, counts);
  EpiIntervalStart epi{interval_starts};
  enqueue_exclusive_scan(w.head, num_points, num_points, 1, w.partial, w.total, (int*)nullptr, LoadIdentity{}, epi, s);
  vpp_finish_kernel<<<blocks, 256, 0, s>>>(skey, sidx, num_points, depth_bins, feat_hw, mode, w.total, ranks_bev,
                                           ranks_depth, ranks_feat, interval_starts, interval_lengths, counts);
  return launch_status();
}

extern "C" int pd3_frustum_to_lidar(const float* frustum, int64_t points_per_camera, int batch, int num_cams,
                                    const float* inv_post_rots, const float* post_trans, const float* cam_to_ego,
                                    const float* trans, const float* bda, float* coor, void* stream) {
  if (!frustum || !inv_post_rots || !post_trans || !cam_to_ego || !trans || !bda || !coor) return PD3_EINVAL;
  if (points_per_camera <= 0 || batch <= 0 || num_cams <= 0) return PD3_EINVAL;
  const int64_t n = points_per_camera * batch * num_cams;
  frustum_to_lidar_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      frustum, points_per_camera, num_cams, n, inv_post_rots, post_trans, cam_to_ego, trans, bda, coor);
  return launch_status();
}
