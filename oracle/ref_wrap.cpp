// oracle/ref_wrap.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Thin extern "C" wrappers around the reference's OWN arithmetic.  The reference code itself is NOT
// in this repository: `make -C oracle ref` extracts the Paddle-free line ranges listed in
// oracle/extract_ref.sh from /root/reference into oracle/_ref/gen_*.inc (git-ignored) and this file
// #includes them.  What is written here is only the driver glue that the reference keeps inside
// Paddle-dependent functions (tensor allocation + initial values), restated with the file:line it follows.
//
// Barrier-free __global__ kernels of the reference (decode_kernel, bev_pool_v2_kernel,
// bev_pool_grad_kernel) are executed serially on the CPU by defining blockIdx/blockDim/threadIdx as
// plain globals and looping over the launch grid.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <math.h>
#include <stdio.h>
#include <vector>

#define __device__
#define __global__

namespace ref_vox {
#include "gen_voxelize_cpu.inc"
}  // namespace ref_vox

namespace ref_iou {
#include "gen_iou3d_cpu.inc"
#include "gen_iou_normal.inc"
}  // namespace ref_iou

namespace ref_cuda {
struct Dim3 {
  int x, y, z;
};
static Dim3 blockIdx, blockDim, threadIdx;
#include "gen_decode_kernel.inc"
#include "gen_bev_pool_kernel.inc"
namespace bkwd {
#include "gen_bev_pool_grad_kernel.inc"
}
}  // namespace ref_cuda

extern "C" {

// Driver restating hard_voxelize_cpu (voxelize_op.cc:84-140): grid = round((max-min)/size) (:97-102),
// coords / num_points_per_voxel zero-filled (:108-117), num_voxels 0 (:119-121), dense
// grid_idx_to_voxel_idx filled with -1 (:123-126); the kernel itself zero-fills voxels (:29-31).
int ref_hard_voxelize(const float *points, int64_t num_points, int num_point_dim,
                      const float *voxel_size, const float *point_cloud_range,
                      int max_num_points_in_voxel, int max_voxels, float *voxels, int *coords,
                      int *num_points_per_voxel, int *num_voxels) {
  const float voxel_size_x = voxel_size[0];
  const float voxel_size_y = voxel_size[1];
  const float voxel_size_z = voxel_size[2];
  int grid_size_x =
      static_cast<int>(round((point_cloud_range[3] - point_cloud_range[0]) / voxel_size_x));
  int grid_size_y =
      static_cast<int>(round((point_cloud_range[4] - point_cloud_range[1]) / voxel_size_y));
  int grid_size_z =
      static_cast<int>(round((point_cloud_range[5] - point_cloud_range[2]) / voxel_size_z));
  std::fill(coords, coords + (size_t)max_voxels * 3, 0);
  std::fill(num_points_per_voxel, num_points_per_voxel + max_voxels, 0);
  num_voxels[0] = 0;
  std::vector<int> grid((size_t)grid_size_x * grid_size_y * grid_size_z, -1);
  ref_vox::hard_voxelize_cpu_kernel<float, int>(
      points, point_cloud_range[0], point_cloud_range[1], point_cloud_range[2], voxel_size_x,
      voxel_size_y, voxel_size_z, grid_size_x, grid_size_y, grid_size_z, num_points, num_point_dim,
      max_num_points_in_voxel, max_voxels, voxels, coords, num_points_per_voxel, grid.data(),
      num_voxels);
  return 0;
}

// The same with the kernel instantiated for double (PD_DISPATCH_FLOATING_TYPES, voxelize_op.cc:128).
int ref_hard_voxelize_f64(const double *points, int64_t num_points, int num_point_dim,
                          const float *voxel_size, const float *point_cloud_range,
                          int max_num_points_in_voxel, int max_voxels, double *voxels, int *coords,
                          int *num_points_per_voxel, int *num_voxels) {
  const float voxel_size_x = voxel_size[0], voxel_size_y = voxel_size[1], voxel_size_z = voxel_size[2];
  int grid_size_x = static_cast<int>(round((point_cloud_range[3] - point_cloud_range[0]) / voxel_size_x));
  int grid_size_y = static_cast<int>(round((point_cloud_range[4] - point_cloud_range[1]) / voxel_size_y));
  int grid_size_z = static_cast<int>(round((point_cloud_range[5] - point_cloud_range[2]) / voxel_size_z));
  std::fill(coords, coords + (size_t)max_voxels * 3, 0);
  std::fill(num_points_per_voxel, num_points_per_voxel + max_voxels, 0);
  num_voxels[0] = 0;
  std::vector<int> grid((size_t)grid_size_x * grid_size_y * grid_size_z, -1);
  ref_vox::hard_voxelize_cpu_kernel<double, int>(
      points, point_cloud_range[0], point_cloud_range[1], point_cloud_range[2], voxel_size_x,
      voxel_size_y, voxel_size_z, grid_size_x, grid_size_y, grid_size_z, num_points, num_point_dim,
      max_num_points_in_voxel, max_voxels, voxels, coords, num_points_per_voxel, grid.data(),
      num_voxels);
  return 0;
}

// boxes_iou_bev_cpu loop (iou3d_cpu.cpp:257-262) over the extracted iou_bev.
void ref_boxes_iou_bev(const float *boxes_a, int num_a, const float *boxes_b, int num_b,
                       float *ans_iou) {
  for (int i = 0; i < num_a; i++)
    for (int j = 0; j < num_b; j++)
      ans_iou[(size_t)i * num_b + j] = ref_iou::iou_bev(boxes_a + i * 7, boxes_b + j * 7);
}

// boxes_overlap_kernel semantics (iou3d_nms_kernel.cu:275-290) over the extracted box_overlap.
void ref_boxes_overlap_bev(const float *boxes_a, int num_a, const float *boxes_b, int num_b,
                           float *ans) {
  for (int i = 0; i < num_a; i++)
    for (int j = 0; j < num_b; j++)
      ans[(size_t)i * num_b + j] = ref_iou::box_overlap(boxes_a + i * 7, boxes_b + j * 7);
}

float ref_iou_bev_pair(const float *a, const float *b) { return ref_iou::iou_bev(a, b); }
float ref_iou_normal_pair(const float *a, const float *b) { return ref_iou::iou_normal(a, b); }

// nms_gpu (iou3d_nms.cpp:86-141): suppression bit (i,j) as nms_kernel computes it
// (iou3d_nms_kernel.cu:345-361: diagonal tile only j>i, strict '>'), then the reference's own host
// sweep, included verbatim from iou3d_nms.cpp:119-137.
static const int THREADS_PER_BLOCK_NMS = sizeof(int64_t) * 8;
#define DIVUP(m, n) ((m) / (n) + ((m) % (n) > 0))

static void build_mask(const float *boxes, int boxes_num, float thresh, bool normal,
                       std::vector<int64_t> &mask) {
  const int col_blocks = DIVUP(boxes_num, THREADS_PER_BLOCK_NMS);
  mask.assign((size_t)boxes_num * col_blocks, 0);
  for (int i = 0; i < boxes_num; i++) {
    for (int cb = 0; cb < col_blocks; cb++) {
      int col_size = std::min(boxes_num - cb * THREADS_PER_BLOCK_NMS, THREADS_PER_BLOCK_NMS);
      int start = (i / THREADS_PER_BLOCK_NMS == cb) ? (i % THREADS_PER_BLOCK_NMS) + 1 : 0;
      uint64_t t = 0;
      for (int k = start; k < col_size; k++) {
        const float *other = boxes + (size_t)(cb * THREADS_PER_BLOCK_NMS + k) * 7;
        float v = normal ? ref_iou::iou_normal(boxes + (size_t)i * 7, other)
                         : ref_iou::iou_bev(boxes + (size_t)i * 7, other);
        if (v > thresh) t |= 1ULL << k;
      }
      mask[(size_t)i * col_blocks + cb] = (int64_t)t;
    }
  }
}

static void sweep(const int64_t *mask_cpu, int boxes_num, int *keep_data, int *num_to_keep_data) {
  const int col_blocks = DIVUP(boxes_num, THREADS_PER_BLOCK_NMS);
#include "gen_nms_sweep.inc"
}

void ref_nms(const float *boxes, int boxes_num, float thresh, int normal, int *keep,
             int *num_to_keep) {
  std::vector<int64_t> mask;
  build_mask(boxes, boxes_num, thresh, normal != 0, mask);
  if (boxes_num == 0) {
    num_to_keep[0] = 0;
    return;
  }
  sweep(mask.data(), boxes_num, keep, num_to_keep);
}

// decode_kernel (postprocess.cu:32-80) executed serially over DecodeLauncher's grid (:93-94).
void ref_centerpoint_decode(const float *score, const float *reg, const float *height,
                            const float *dim, const float *vel, const float *rot,
                            float score_threshold, int feat_w, float down_ratio, float voxel_size_x,
                            float voxel_size_y, float pc_x_min, float pc_y_min, const float *pcr,
                            int num_bboxes, int with_velocity, float *bboxes, unsigned char *mask,
                            int *score_idx) {
  using namespace ref_cuda;
  const int dims = with_velocity ? 9 : 7;
  std::vector<char> m((size_t)num_bboxes, 0);
  blockDim.x = THREADS_PER_BLOCK_NMS;
  int blocks = DIVUP(num_bboxes, THREADS_PER_BLOCK_NMS);
  for (int b = 0; b < blocks; b++)
    for (int t = 0; t < THREADS_PER_BLOCK_NMS; t++) {
      blockIdx.x = b;
      threadIdx.x = t;
      decode_kernel(score, reg, height, dim, vel, rot, score_threshold, feat_w, down_ratio,
                    voxel_size_x, voxel_size_y, pc_x_min, pc_y_min, pcr[0], pcr[1], pcr[2], pcr[3],
                    pcr[4], pcr[5], num_bboxes, with_velocity != 0, dims, bboxes,
                    reinterpret_cast<bool *>(m.data()), score_idx);
    }
  for (int i = 0; i < num_bboxes; i++) mask[i] = m[i] ? 1 : 0;
}

// bev_pool_v2_kernel (bev_pool_cuda.cu:18-44) executed serially over its launch grid (:98-105);
// `out` must be zero-filled by the caller as bev_pool.cc:48-49 does.
void ref_bev_pool_v2(int c, int n_intervals, const float *depth, const float *feat,
                     const int *ranks_depth, const int *ranks_feat, const int *ranks_bev,
                     const int *interval_starts, const int *interval_lengths, float *out) {
  using namespace ref_cuda;
  blockDim.x = 256;
  long total = (long)n_intervals * c;
  int blocks = (int)ceil(((double)n_intervals * c / 256));
  (void)total;
  for (int b = 0; b < blocks; b++)
    for (int t = 0; t < 256; t++) {
      blockIdx.x = b;
      threadIdx.x = t;
      bev_pool_v2_kernel(c, n_intervals, depth, feat, ranks_depth, ranks_feat, ranks_bev,
                         interval_starts, interval_lengths, out);
    }
}

// bev_pool_grad_kernel (bev_pool_cuda_bkwd.cu:44-94) over its grid (:105-114); grads zero-filled by
// the caller as bev_pool_bkwd.cc:41-46 does.
void ref_bev_pool_v2_bkwd(int c, int n_intervals, const float *out_grad, const float *depth,
                          const float *feat, const int *ranks_depth, const int *ranks_feat,
                          const int *ranks_bev, const int *interval_starts,
                          const int *interval_lengths, float *depth_grad, float *feat_grad) {
  using namespace ref_cuda;
  blockDim.x = 256;
  int blocks = (int)ceil(((double)n_intervals / 256));
  for (int b = 0; b < blocks; b++)
    for (int t = 0; t < 256; t++) {
      blockIdx.x = b;
      threadIdx.x = t;
      bkwd::bev_pool_grad_kernel(c, n_intervals, out_grad, depth, feat, ranks_depth, ranks_feat,
                                 ranks_bev, interval_starts, interval_lengths, depth_grad,
                                 feat_grad);
    }
}

}  // extern "C"
