/* paddle3d_amd.h -- C ABI of libpaddle3d_amd.so, the MI355X (gfx950) LiDAR-detection op library.
 *
 * Drop-in boundary: every entry point replaces one custom operator (or one layer built on Paddle-core
 * scatter) of the reference's `paddle3d.ops` plugin API for the LiDAR hot path.  The reference
 * registers its operators with PD_BUILD_OP(name).Inputs/.Outputs/.Attrs (file:line cited per
 * function below); a maintainer binds these symbols from a Paddle custom-op shim or via ctypes --
 * see INTEGRATION.md.  the modules under paddle3d_amd/ops/ are the ctypes binding used by this repository.
 *
 * Conventions
 *   - All pointers are DEVICE pointers unless a parameter is documented "host".
 *   - Plain C types only; `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - The library never allocates or frees device memory: outputs and `workspace` are caller-owned.
 *     Query the workspace size with the matching *_workspace() function (bytes; 256-B aligned base).
 *   - Every call is asynchronous on `stream` unless documented otherwise, re-entrant and thread-safe
 *     given disjoint buffers.
 *   - Return value: 0 ok; <0 argument error (PD3_E*); >0 a hipError_t from the launch.
 *   - fp32 only, like the reference ops (voxelize_op.cc:104-106 hard-codes FLOAT32 outputs).
 */
#ifndef PADDLE3D_AMD_H
#define PADDLE3D_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PD3_OK 0
#define PD3_EINVAL (-1)
#define PD3_EWORKSPACE (-2)
#define PD3_EUNSUPPORTED (-3)

/* Library/ABI version (major*10000 + minor*100 + patch) and the gfx target it was compiled for. */
int pd3_version(void);
const char *pd3_target_arch(void);

/* Self-check of the one undocumented hardware property the default hard_voxelize path relies on: lanes of one
 * wave-wide returning LDS add on the same word are served in ascending lane order, a wave's LDS instructions in
 * program order (csrc/selfcheck.hip).  Every wave (blocks * waves_per_block of them) walks `rounds` x 64 addresses
 * (addr [waves][rounds][64] uint32 < table, 0xFFFFFFFF = lane skips) through its own LDS table and writes what each
 * add returned (old, same shape; 0xFFFFFFFF for skipped lanes); the caller compares with a sequential count per
 * wave.  No reference counterpart (the reference's GPU path uses global atomics and is not order-exact). */
int pd3_selfcheck_lds_atomic_order(const uint32_t *addr, uint32_t *old, int blocks, int waves_per_block, int rounds,
                                   int table, void *stream);

/* ---------------------------------------------------------------------------------------------
 * hard_voxelize -- replaces PD_BUILD_OP(hard_voxelize), paddle3d/ops/voxel/voxelize_op.cc:183-191
 * (kernel fn hard_voxelize :149-166; CPU semantics hard_voxelize_cpu_kernel :19-82, which this
 * implementation reproduces bit-exactly, unlike the reference's own CUDA path voxelize_op.cu:208-346
 * whose atomics make voxel order and in-voxel point choice nondeterministic).
 *
 *   points            [batch, max_points, num_point_dim] fp32, row-major; frame b uses its first
 *                     num_points[b] rows (num_points == NULL: all max_points rows).
 *   voxel_size        host float[3] (x, y, z);  point_cloud_range host float[6] (xmin..zmin, xmax..zmax)
 *   voxels            [batch, max_voxels, max_num_points_in_voxel, num_point_dim] fp32, zero padded
 *   coords            [batch, max_voxels, 3] int32 (z, y, x), zero padded
 *   num_points_per_voxel [batch, max_voxels] int32, zero padded
 *   num_voxels        [batch] int32
 *   coors_batched     optional (NULL to skip) [batch, max_voxels, 4] int32 (batch, z, y, x) with batch = -1 on
 *                     padding rows: the `coors_pad` HardVoxelizer.single_forward builds with cast + F.pad
 *                     (voxelize.py:51-57), written by the same pass
 * batch = 1 is exactly the reference op; batch > 1 is the reference's HardVoxelizer python loop
 * (paddle3d/models/voxelizers/voxelize.py:60-82) in one launch sequence.
 */
size_t pd3_hard_voxelize_workspace(int batch, int64_t max_points, int num_point_dim,
                                   const float *voxel_size, const float *point_cloud_range,
                                   int max_num_points_in_voxel, int max_voxels);
int pd3_hard_voxelize(const float *points, const int32_t *num_points, int batch, int64_t max_points,
                      int num_point_dim, const float *voxel_size, const float *point_cloud_range,
                      int max_num_points_in_voxel, int max_voxels, float *voxels, int32_t *coords,
                      int32_t *num_points_per_voxel, int32_t *num_voxels, int32_t *coors_batched,
                      void *workspace, size_t workspace_bytes, void *stream);
/* Same operator with the implementation chosen by the caller (diagnostics and the parity tests, which run
 * every case on all of them): path 0 = automatic (what pd3_hard_voxelize does: 5 where the grid qualifies, else 3,
 * else 1), 1 = generic radix-sort path (any grid below 2^31 cells), 2 = tiled path, payload copied once into a
 * cell-ordered compact array, 3 = tiled path, rows gathered from the points through a cell-ordered index list (2 and
 * 3: BEV-sized grids up to 2^22 cells), 5 = wave form of the tiled path (one wave per group of 1024 cells, grids up to
 * 2^20 cells; 6 .. 10 = the same with the route kernel's tile shape forced: 4096 / 8192 / 10240 / 5120 / 5120 points;
 * 11 = 5 with heavy group waves at raised issue priority (the automatic choice), 12 / 13 = two half batches on two
 * streams), 14 = the wave form for 3-D grids of 2^20 .. 2^28 cells (15 / 16: its route tile forced), 17 = a measurement
 * form: path 6 with the points' payload carried through the route kernel's LDS slice.
 * PD3_EUNSUPPORTED when the shape does not qualify for a forced path.  Identical bytes out on every path. */
int pd3_hard_voxelize_path(const float *points, const int32_t *num_points, int batch, int64_t max_points,
                           int num_point_dim, const float *voxel_size, const float *point_cloud_range,
                           int max_num_points_in_voxel, int max_voxels, float *voxels, int32_t *coords,
                           int32_t *num_points_per_voxel, int32_t *num_voxels, int32_t *coors_batched,
                           void *workspace, size_t workspace_bytes, void *stream, int path);

/* hard_voxelize WITHOUT the padded [V, P, D] tensor (the model path; reference chain paddle3d/models/voxelizers/
 * voxelize.py:39-58 -> voxel_encoders/pillar_encoder.py:156-210): the same voxel ids, coords, counts and batched
 * coors as pd3_hard_voxelize, and in place of copies of the points an INDEX of them:
 *   vox_span   [batch, max_voxels, 2] int32: (start, count) -- voxel v of frame b holds the points
 *              point_list[b * max_points + start + 0 .. count - 1] (indices into frame b's rows of `points`,
 *              ascending = the reference's order inside a voxel); count == num_points_per_voxel
 *   point_list [pd3_hard_voxelize_index_list_entries(batch, max_points)] int32
 * A consumer reads a voxel's points from `points` itself (pd3_pillar_feature_net_indexed): the 78 %-zeros tensor is
 * neither written nor read.  Grids the wave forms do not serve (see pd3_hard_voxelize_path) return -3: run
 * pd3_hard_voxelize there.  Workspace: pd3_hard_voxelize_workspace. */
int64_t pd3_hard_voxelize_index_list_entries(int batch, int64_t max_points);
int pd3_hard_voxelize_index(const float *points, const int32_t *num_points, int batch, int64_t max_points,
                            int num_point_dim, const float *voxel_size, const float *point_cloud_range,
                            int max_num_points_in_voxel, int max_voxels, int32_t *vox_span, int32_t *point_list,
                            int32_t *coords, int32_t *num_points_per_voxel, int32_t *num_voxels,
                            int32_t *coors_batched, void *workspace, size_t workspace_bytes, void *stream);

/* hard_voxelize for double points: PD_DISPATCH_FLOATING_TYPES (voxelize_op.cc:128) instantiates the reference's CPU
 * kernel for float and double; with T = double the cell index is floor((p - (double)range_min) / (double)voxel_size)
 * (:37-45) and `voxels` is double.  Generic sort path (any grid below 2^31 cells); the workspace of
 * pd3_hard_voxelize_workspace suffices.  points / voxels fp64 device, everything else as pd3_hard_voxelize. */
int pd3_hard_voxelize_f64(const double *points, const int32_t *num_points, int batch, int64_t max_points,
                          int num_point_dim, const float *voxel_size, const float *point_cloud_range,
                          int max_num_points_in_voxel, int max_voxels, double *voxels, int32_t *coords,
                          int32_t *num_points_per_voxel, int32_t *num_voxels, int32_t *coors_batched,
                          void *workspace, size_t workspace_bytes, void *stream);

/* dynamic_voxelize -- per-point voxel coordinates without the per-voxel cap.  The reference has no such
 * operator (only a "dynamic voxelization" comment at transforms/ for num_points == -1);
 * BASELINE.json's north star names it, so it is provided with hard_voxelize's own cell rule
 * (voxelize_op.cc:37-45): coors[i] = (z, y, x) of point i, (-1, -1, -1) outside point_cloud_range.
 *   points [num_points, num_point_dim] fp32 device;  coors [num_points, 3] int32 device */
int pd3_dynamic_voxelize(const float *points, int64_t num_points, int num_point_dim, const float *voxel_size,
                         const float *point_cloud_range, int32_t *coors, void *stream);

/* ---------------------------------------------------------------------------------------------
 * pointpillars_scatter -- replaces PointPillarsScatter.forward_batch,
 * paddle3d/models/middle_encoders/pillar_scatter.py:57-93 (zeros canvas + paddle.scatter(overwrite) +
 * transpose + concat), fused into one canvas-parallel pass.
 *
 *   voxel_features [num_pillars, channels] fp32;  coords [num_pillars, 4] int32 (batch, z, y, x)
 *   canvas         [batch, channels, ny, nx] fp32 -- fully written (zero where no pillar)
 * Pillars with batch index outside [0, batch) are ignored.  Duplicate (batch, y, x): last index wins,
 * as paddle.scatter(overwrite=True) documents; the hot path never produces duplicates.
 */
size_t pd3_pointpillars_scatter_workspace(int batch, int ny, int nx);
int pd3_pointpillars_scatter(const float *voxel_features, const int32_t *coords,
                             int64_t num_pillars, int channels, int batch, int ny, int nx,
                             float *canvas, void *workspace, size_t workspace_bytes, void *stream);

/* PointPillarsScatter fused into the convolution that consumes it (round 3): the canvas is never written.
 *   pd3_pointpillars_inverse_map: inv [batch, ny*nx] int32 = the pillar row (index into voxel_features) whose coords
 *     name the cell, -1 for an empty cell; on duplicates the highest row wins (paddle.scatter(overwrite=True),
 *     pillar_scatter.py:83-90).
 *   pd3_scatter_conv3x3_bias_relu: conv3x3 / pad 1 / stride 2 + bias + ReLU (a SECOND block's first convolution,
 *     second_backbone.py:84-98, BatchNorm folded) over the canvas those two arguments describe; the kernel stages the
 *     occupied cells' channels straight from voxel_features [pillars, cin].  w_packed as pd3_conv3x3_bias_relu;
 *     out [batch, cout, ny/2, out_w] (out_w % 4 == 0, columns >= nx/2 written as zeros).  Results are bit-identical to
 *     pd3_pointpillars_scatter followed by pd3_conv3x3_bias_relu.  cin % 8 == 0, cout % 64 == 0, ny, nx even. */
int pd3_pointpillars_inverse_map(const int32_t *coords, int64_t num_pillars, int batch, int ny, int nx,
                                 int32_t *inv, void *stream);
int pd3_scatter_conv3x3_bias_relu(const float *voxel_features, const int32_t *inv, const float *w_packed,
                                  const float *bias, int batch, int cin, int cout, int ny, int nx, int stride,
                                  int relu, float *out, int out_w, void *stream);

/* The same pair -- PointPillarsScatter (pillar_scatter.py:57-93) + the strided 3x3 / pad 1 convolution that opens
 * SecondBackbone (second_backbone.py:84-98) -- as a SPARSE convolution over the occupied pillars (round 6): a nuScenes
 * canvas is 11 % occupied, an output pixel sees 2.5 of its nine cells on average and 60 % of the pixels see none.
 *   pd3_pillar_conv_rulebook: from the inverse map [batch, ny * nx] (pd3_pointpillars_inverse_map) the rulebook of the
 *     active output pixels in raster order: nbr [capacity, 9] int32 (pillar row of tap ky * 3 + kx or -1), out_cell
 *     [capacity] (pixel b * ho * wo + oy * wo + ox of a row), cell_row [batch, ho * wo] (row of a pixel or -1), n_out [1]
 *     (device: the row count; rows beyond `capacity` are dropped -- batch * ho * wo can never overflow), ho = (ny - 1) /
 *     stride + 1.  The rulebook feeds pd3_sparse_tile_order + pd3_sparse_conv3d_features_bf16x3 / _f16 (kernel_volume 9,
 *     in_feats = the pillar features, n_out as the device row count).  order (optional, [pd3_sparse_tile_order_entries(
 *     capacity)] int32): the rulebook's tile order by a counting sort over the 512 possible tap masks -- the same
 *     scheduling hint pd3_sparse_tile_order computes with its general sorting network.
 *   pd3_rows_to_dense_fill: rows [n, channels] fp32 -> out [batch, channels, h, w] NCHW through cell_row; a pixel without
 *     a row holds fill[c] (= relu(bias[c]): what the dense convolution computes from nine zeros); h * w % 4 == 0,
 *     channels % 4 == 0. */
size_t pd3_pillar_conv_rulebook_workspace(int batch, int ny, int nx, int stride);
int pd3_pillar_conv_rulebook(const int32_t *inverse_map, int batch, int ny, int nx, int stride, int32_t *nbr,
                             int32_t *out_cell, int32_t *cell_row, int32_t *n_out, int capacity, int32_t *order,
                             void *workspace, size_t workspace_bytes, void *stream);
int pd3_rows_to_dense_fill(const float *rows, const int32_t *cell_row, const float *fill, int batch, int channels, int h,
                           int w, float *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * pillar feature net (PFN) -- replaces PillarFeatureNet.forward / PFNLayer.forward in eval mode,
 * paddle3d/models/voxel_encoders/pillar_encoder.py:156-210 / :81-105 (decorate with cluster and
 * pillar-centre offsets, mask padded slots, Linear(no bias) -> BatchNorm1D(eps 1e-3) -> ReLU -> max
 * over the points of a pillar; the non-last layer concatenates the max back).  BatchNorm is folded by
 * the caller: scale = gamma / sqrt(var + eps), shift = beta - mean * scale.
 *
 * The same entry point covers HardVFE.forward with with_cluster_center = with_voxel_center = True
 * (voxel_encoders/voxel_encoder.py:142-283, the BEVFusion LiDAR stream): voxel_center_dims = 3 adds the z
 * offset z - (coor_z * vz + z_offset) and layer 1 keeps all its C1 units (w2 is [2*C1, C2] either way).
 *
 *   voxels [M, P, D], num_points [M] int32, coors [M, 4] int32 (b, z, y, x)
 *   voxel_center_dims 2 (PillarFeatureNet: x, y) or 3 (HardVFE: x, y, z)
 *   w1 [D+3+voxel_center_dims, C1] (Paddle Linear layout [in, out]), scale1/shift1 [C1]
 *   w2 [2*C1, C2], scale2/shift2 [C2]   (w2 == NULL: single-layer PFN, output [M, C1])
 *   out [M, C2]
 * legacy == 0 only (the nuScenes configs); with_distance unsupported (no config on the path uses it).
 * Rows with num_points <= 0 (padding of a fixed-shape batch) produce zeros.
 */
int pd3_pillar_feature_net(const float *voxels, const int32_t *num_points, const int32_t *coors,
                           int64_t num_pillars, int max_points, int num_point_dim,
                           int voxel_center_dims, float vx, float vy, float vz, float x_offset,
                           float y_offset, float z_offset, const float *w1, const float *scale1,
                           const float *shift1, int c1, const float *w2, const float *scale2,
                           const float *shift2, int c2, float *out, void *stream);

/* Same operator with the kernel form named by the caller (diagnostics and the parity tests, which run every form on
 * one input; there is no environment switch): path 0 = what pd3_pillar_feature_net picks, 1 = the per-pillar forms
 * (one wave per pillar), 2 = the packed form (a wave packs the rows of 8 consecutive pillars into 16-row MFMA blocks;
 * two-layer C1 = 32, C2 = 64 nets with P <= 32 only, PD3_EUNSUPPORTED otherwise). */
int pd3_pillar_feature_net_path(const float *voxels, const int32_t *num_points, const int32_t *coors,
                                int64_t num_pillars, int max_points, int num_point_dim,
                                int voxel_center_dims, float vx, float vy, float vz, float x_offset,
                                float y_offset, float z_offset, const float *w1, const float *scale1,
                                const float *shift1, int c1, const float *w2, const float *scale2,
                                const float *shift2, int c2, float *out, int path, void *stream);

/* pd3_pillar_feature_net reading the pillars' points through pd3_hard_voxelize_index's (vox_span, point_list) from
 * the point cloud itself: points [frames, points_per_frame, D], list_stride = rows of point_list per frame
 * (= max_points of the voxelizer call), coors [num_pillars, 4], pillars_per_frame = max_voxels.  Bit-identical to
 * pd3_hard_voxelize + pd3_pillar_feature_net.  Serves the two-layer 32 / 64 net with up to 32 points of 4 / 5
 * floats per pillar and pillars_per_frame a multiple of 8; other shapes return -3 (run the pair). */
int pd3_pillar_feature_net_indexed(const float *points, int64_t points_per_frame, const int32_t *vox_span,
                                   const int32_t *point_list, int64_t list_stride, const int32_t *coors,
                                   int64_t num_pillars, int pillars_per_frame, int max_points, int num_point_dim,
                                   int voxel_center_dims, float vx, float vy, float vz, float x_offset,
                                   float y_offset, float z_offset, const float *w1, const float *scale1,
                                   const float *shift1, int c1, const float *w2, const float *scale2,
                                   const float *shift2, int c2, float *out, void *stream);

/* VoxelMean.forward, paddle3d/models/voxel_encoders/voxel_encoder.py:44-57: sum over P / count. */
int pd3_voxel_mean(const float *voxels, const int32_t *num_points, int64_t num_voxels,
                   int max_points, int num_point_dim, float *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * iou3d_nms -- replaces the five ops of paddle3d/ops/iou3d_nms/iou3d_nms_api.cpp:73-108.
 * Boxes are [N, 7] fp32 (x, y, z, dx, dy, dz, heading).
 *
 * pd3_nms_bev / pd3_nms_normal: nms_gpu / nms_normal_gpu (iou3d_nms.cpp:86-141 / :143-204).
 * The suppression bit-matrix AND the greedy sweep run on the device (the reference copies the mask to
 * the host and sweeps there).  keep [N] int32 (first *num_to_keep entries valid) and num_to_keep [1]
 * int32 are DEVICE buffers; the Python shim returns them as CPU int32 tensors like the reference.
 */
size_t pd3_nms_workspace(int num_boxes);
int pd3_nms_bev(const float *boxes, int num_boxes, float nms_overlap_thresh, int32_t *keep,
                int32_t *num_to_keep, void *workspace, size_t workspace_bytes, void *stream);
int pd3_nms_normal(const float *boxes, int num_boxes, float nms_overlap_thresh, int32_t *keep,
                   int32_t *num_to_keep, void *workspace, size_t workspace_bytes, void *stream);
/* boxes_iou_bev_gpu / boxes_overlap_bev_gpu (iou3d_nms.cpp:44-84): dense [N, M] matrices. */
int pd3_boxes_iou_bev(const float *boxes_a, int num_a, const float *boxes_b, int num_b,
                      float *ans_iou, void *stream);
int pd3_boxes_overlap_bev(const float *boxes_a, int num_a, const float *boxes_b, int num_b,
                          float *ans_overlap, void *stream);
/* Diagnostic: the float math routines the geometry / decode kernels use, over an array -- glibc's sinf / cosf / expf /
 * atanf / atan2f bit for bit (csrc/libm_exact.hpp; what the reference's cos / sin / atan2 on floats call,
 * iou3d_cpu.cpp:77-79,128-129).  op: 0 sinf(x), 1 cosf(x), 2 expf(x), 3 atanf(x), 4 atan2f(x, y).  x, y, out device. */
int pd3_libm_eval(int op, const float *x, const float *y, float *out, int64_t n, void *stream);

/* ---------------------------------------------------------------------------------------------
 * centerpoint_postprocess -- replaces PD_BUILD_OP(centerpoint_postprocess),
 * paddle3d/ops/centerpoint_postprocess/postprocess.cc:91-104 (postprocess_gpu, postprocess.cu:104-280).
 * All tasks of one frame in one launch sequence on one stream, no host round trip inside.
 *
 *   hm/reg/height/dim/vel/rot: host arrays of `num_tasks` device pointers to contiguous fp32 NCHW
 *     maps: hm[t] [batch, hm_channels[t], H, W], reg [batch,2,H,W], height [batch,1,H,W],
 *     dim [batch,3,H,W], vel [batch,2,H,W], rot [batch,2,H,W].  batch = 1 is exactly the reference op
 *     (it rejects anything else, postprocess.cu:19-20,138); batch > 1 runs the per-frame op for every
 *     frame in the same launch sequence.
 *   hm_channels: host int[num_tasks];  label_offsets: host int[num_tasks] (the op's `num_classes` attr)
 *   out_bboxes  [batch, num_tasks * max(nms_post_max_size,1), 9 or 7] fp32
 *   out_scores  [batch, same rows] fp32;  out_labels [batch, same rows] int64
 *   out_count   [batch] int32: number of valid leading rows per frame (tasks concatenated in order;
 *               a task with no candidate contributes the reference's fake row: zeros box, score -1,
 *               label 0).
 */
size_t pd3_centerpoint_postprocess_workspace(int batch, int num_tasks, int feat_h, int feat_w,
                                             int nms_pre_max_size, int nms_post_max_size);
int pd3_centerpoint_postprocess(const float *const *hm, const float *const *reg,
                                const float *const *height, const float *const *dim,
                                const float *const *vel, const float *const *rot, int batch,
                                int num_tasks, const int *hm_channels, int feat_h, int feat_w,
                                const float *voxel_size, const float *point_cloud_range,
                                const float *post_center_range, const int *label_offsets,
                                int down_ratio, float score_threshold, float nms_iou_threshold,
                                int nms_pre_max_size, int nms_post_max_size, int with_velocity,
                                float *out_bboxes, float *out_scores, int64_t *out_labels,
                                int32_t *out_count, void *workspace, size_t workspace_bytes,
                                void *stream);

/* Same, for head tensors that are channel slices (views) of one wider [batch, C, H, W] map: every head pointer
 * addresses frame 0 of its slice and `head_batch_stride` (> 0, in elements, the same for all heads) leads to the
 * next frame -- what a fused CenterHead produces; saves the per-head contiguous copies. */
int pd3_centerpoint_postprocess_strided(const float *const *hm, const float *const *reg,
                                const float *const *height, const float *const *dim,
                                const float *const *vel, const float *const *rot, int64_t head_batch_stride, int batch,
                                int num_tasks, const int *hm_channels, int feat_h, int feat_w,
                                const float *voxel_size, const float *point_cloud_range,
                                const float *post_center_range, const int *label_offsets,
                                int down_ratio, float score_threshold, float nms_iou_threshold,
                                int nms_pre_max_size, int nms_post_max_size, int with_velocity,
                                float *out_bboxes, float *out_scores, int64_t *out_labels,
                                int32_t *out_count, void *workspace, size_t workspace_bytes,
                                void *stream, int selection);
/* Same as the strided form (automatic selection), plus `out_records` [batch, max_per_img, 11] fp32: the rows once
 * more in the fixed shape of the multi-GPU result hand-off -- box (7 or 9 values, zero padded to 9), score, label as
 * float, rows >= count zero -- so that the RCCL all-gather of a batch's detections (paddle3d_amd/dist.py; no
 * counterpart in the reference, whose evaluation is single device, apis/trainer.py:47-51) starts from the operator's
 * own output.  All four outputs read zero behind the last row; none needs clearing by the caller. */
int pd3_centerpoint_postprocess_records(const float *const *hm, const float *const *reg,
                                        const float *const *height, const float *const *dim,
                                        const float *const *vel, const float *const *rot,
                                        int64_t head_batch_stride, int batch, int num_tasks,
                                        const int *hm_channels, int feat_h, int feat_w, const float *voxel_size,
                                        const float *point_cloud_range, const float *post_center_range,
                                        const int *label_offsets, int down_ratio, float score_threshold,
                                        float nms_iou_threshold, int nms_pre_max_size, int nms_post_max_size,
                                        int with_velocity, float *out_bboxes, float *out_scores,
                                        int64_t *out_labels, int32_t *out_count, float *out_records,
                                        int max_per_img, void *workspace, size_t workspace_bytes, void *stream);
/* selection: how the nms_pre_max_size best cells of a task are found -- 0 automatic (an exact in-LDS top-K
 * selection where the map allows, else a full sort), 1 always the full stable radix sort of all cells (what the
 * reference does; the tests run both and require identical output). */

/* ---------------------------------------------------------------------------------------------
 * bev_pool_v2 / bev_pool_v2_bkwd -- replace PD_BUILD_OP(bev_pool_v2) (bev_pool_v2/bev_pool.cc:111-118,
 * kernel bev_pool_cuda.cu:18-44) and PD_BUILD_OP(bev_pool_v2_bkwd)
 * (bev_pool_v2_backward/bev_pool_bkwd.cc:75-80, kernel bev_pool_cuda_bkwd.cu:44-94).
 * `out` / grads are fully written (zero where no interval lands), fp32 accumulation in interval order.
 * n_points = length of the ranks_* arrays; the intervals tile [0, n_points).
 */
int pd3_bev_pool_v2(const float *depth, const float *feat, const int32_t *ranks_depth,
                    const int32_t *ranks_feat, const int32_t *ranks_bev,
                    const int32_t *interval_lengths, const int32_t *interval_starts,
                    int n_intervals, int channels, int64_t out_elems, float *out, void *stream);
int pd3_bev_pool_v2_bkwd(const float *out_grad, const float *depth, const float *feat,
                         const int32_t *ranks_depth, const int32_t *ranks_feat,
                         const int32_t *ranks_bev, const int32_t *interval_lengths,
                         const int32_t *interval_starts, int n_intervals, int64_t n_points,
                         int channels, int64_t depth_elems, int64_t feat_elems, float *depth_grad,
                         float *feat_grad, void *stream);

/* ---------------------------------------------------------------------------------------------
 * frustum_to_lidar -- LSSViewTransformer.get_lidar_coor (paddle3d/models/transformers/bevdet_transformer.py:
 * 142-192): ego-frame coordinates of every point of the camera frustums.
 *   frustum        [points_per_camera, 3] fp32 device: (u, v, depth) template of create_frustum (:126-140)
 *   inv_post_rots  [batch*num_cams, 3, 3]  inverse of the image-space augmentation rotation
 *   post_trans     [batch*num_cams, 3];  cam_to_ego [batch*num_cams, 3, 3] = rots @ inverse(cam2imgs)
 *   trans          [batch*num_cams, 3];  bda [batch, 3, 3]                (all fp32, device)
 *   coor           [batch*num_cams*points_per_camera, 3] fp32
 */
int pd3_frustum_to_lidar(const float *frustum, int64_t points_per_camera, int batch, int num_cams,
                         const float *inv_post_rots, const float *post_trans, const float *cam_to_ego,
                         const float *trans, const float *bda, float *coor, void *stream);

/* ---------------------------------------------------------------------------------------------
 * voxel_pooling_prepare -- the index build in front of bev_pool_v2, on the device: replaces
 * LSSViewTransformer.voxel_pooling_prepare_v2 (paddle3d/models/transformers/bevdet_transformer.py:230-274: quantise
 * every frustum point, keep the ones inside the grid, argsort by BEV rank, run-length encode into intervals) and
 * the same steps of LiftSplatShoot.voxel_pooling (models/detection/bevfusion/cam_stream_lss.py:318-346).
 *   coor           [num_points, 3] fp32 frustum points in the ego frame, num_points = B * N * D * H * W
 *   depth_bins, feat_hw   D and H * W (ranks_feat of mode 0 drops the depth axis)
 *   grid_lower / grid_interval / grid_size   host float[3] each, as the reference's float tensors
 *   mode 0 (BEVDet)  rank = b * (Z*Y*X) + z * (Y*X) + y * X + x;  ranks_feat = camera-pixel index
 *   mode 1 (LSS)     rank = ((b * Z + z) * X + x) * Y + y  (the cell of the reference's [B, Z, X, Y] output);
 *                    ranks_feat = point index
 *   mode 2 (LSS, split operands)  rank as mode 1, ranks_depth = point index, ranks_feat = camera-pixel index as
 *                    mode 0: pools depth [B*N, D, H, W] and feat [B*N, H, W, C] directly -- the lifted
 *                    depth (x) feat tensor of CamEncode.get_depth_feat (cam_stream_lss.py:166) is never formed
 *   ranks_bev / ranks_depth / ranks_feat [num_points] int32 (first counts[0] valid, sorted by rank, points of
 *   a cell in index order = a stable argsort), interval_starts / interval_lengths [num_points] int32 (first
 *   counts[1] valid), counts [2] int32 (device): kept points, intervals.
 */
size_t pd3_voxel_pooling_prepare_workspace(int64_t num_points);
int pd3_voxel_pooling_prepare(const float *coor, int64_t num_points, int batch, int depth_bins, int feat_hw,
                              const float *grid_lower, const float *grid_interval, const float *grid_size, int mode,
                              int32_t *ranks_bev, int32_t *ranks_depth, int32_t *ranks_feat,
                              int32_t *interval_starts, int32_t *interval_lengths, int32_t *counts, void *workspace,
                              size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * sparse_conv3d -- replaces the Paddle-core sparse ops the CenterPoint-Voxel middle encoder is built
 * from: paddle.sparse.nn.SubmConv3D / Conv3D (+ BatchNorm, ReLU, sparse.add fused into the epilogue) and
 * SparseCooTensor.to_dense (call sites paddle3d/models/middle_encoders/sparse_resnet.py:31-59, :115-206;
 * sparsenet.py:31-64).  The reference arithmetic is in the paddlepaddle wheel (>= 2.4.0, not vendored):
 * parity is pinned on the public definition against a dense conv3d oracle.
 *
 * A convolution = pd3_sparse_conv3d_indices (output coordinate set + neighbour table; reusable by every
 * conv that shares the same input set, kernel, stride and padding -- the reference's `key=` hint) followed
 * by pd3_sparse_conv3d_features (gather-GEMM with fused epilogue).
 *   in_coords  [n_in, 4] int32 (batch, z, y, x); rows with batch < 0 are padding and ignored
 *   spatial_shape host int[3] (D, H, W);  kernel_size / stride / padding host int[3]
 *   subm != 0: output set = input set (same rows, same order); requires stride 1, padding = k/2
 *   subm == 0: output set = every position reached by an active input, rows sorted by (b, z, y, x)
 *   out_coords [out_cap, 4], nbr [out_cap, kd*kh*kw] int32 (input row or -1), n_out [1] int32 (device)
 *   weight [kd, kh, kw, Cin, Cout] (Paddle layout); bias / scale+shift / residual may be NULL
 *   out = relu?( (sum_k W[k].in[nbr[.,k]] + bias) * scale + shift + residual ),  Cout <= 128
 */
size_t pd3_sparse_conv3d_workspace(int n_in, const int *kernel_size, int subm, int out_cap);
int pd3_sparse_conv3d_indices(const int32_t *in_coords, int n_in, int batch, const int *spatial_shape,
                              const int *kernel_size, const int *stride, const int *padding, int subm,
                              int32_t *out_coords, int32_t *nbr, int32_t *n_out, int out_cap,
                              void *workspace, size_t workspace_bytes, void *stream);
int pd3_sparse_conv3d_features(const float *in_feats, const int32_t *nbr, const int32_t *n_out,
                               int n_out_cap, int kernel_volume, int cin, int cout,
                               const float *weight, const float *bias, const float *scale,
                               const float *shift, const float *residual, int relu, float *out,
                               void *stream);
/* Tile order (a scheduling hint, never a change of results): the rows of every window of 8192 consecutive output
 * rows sorted by their neighbour mask, so that the 16-row blocks of the gather-GEMM hold rows that need the same
 * kernel offsets (a block runs an offset if ANY of its rows has that neighbour: 1.3x - 7x more steps than pairs
 * exist with rows in raster order, 1.2x - 2x in tile order).
 *   order [pd3_sparse_tile_order_entries(n_out_cap)] int32: slot -> output row, -1 past the row count
 *   n_out may be NULL (= n_out_cap); kernel_volume <= 31
 * pd3_sparse_conv3d_features_ordered = pd3_sparse_conv3d_features taking `order` (NULL: raster order); same bytes
 * out for any order. */
int64_t pd3_sparse_tile_order_entries(int n_out_cap);
int pd3_sparse_tile_order(const int32_t *nbr, const int32_t *n_out, int n_out_cap, int kernel_volume,
                          int32_t *order, void *stream);
int pd3_sparse_conv3d_features_ordered(const float *in_feats, const int32_t *nbr, const int32_t *n_out,
                                       int n_out_cap, int kernel_volume, int cin, int cout,
                                       const float *weight, const float *bias, const float *scale,
                                       const float *shift, const float *residual, int relu,
                                       const int32_t *order, float *out, void *stream);
/* The fp32 gather-GEMM on the bf16 matrix cores (round 5; the default of the fp32 encoder from 16 -> 32 channels on): every
 * fp32 operand is cut into three bf16 pieces (hi + mid + lo = the fp32 value exactly: 3 x 8 significand bits) and six of
 * the nine piece products are accumulated in fp32 (the three dropped ones: <= 2^-23 of a product, 2^-28 on average) -- the error against exact arithmetic is that of the fp32 matrix-core kernel
 * (tests/test_sparse_conv_gpu.py::test_features_bf16x3_is_fp32_arithmetic), the rate 2.7 x the fp32 pipe's.
 *   pd3_sparse_pack_weight_bf16x3       weight [K, Cin, Cout] fp32 (Paddle layout) -> 3 * K * Cin * Cout bf16 in the
 *                                       kernel's operand order; Cin % 16 == 0, Cout in {32, 64, 128}
 *   pd3_sparse_conv3d_features_bf16x3   as pd3_sparse_conv3d_features_ordered (all fp32 rows in and out) with that
 *                                       packed weight.  Other shapes return -3 (run the fp32 entry). */
int pd3_sparse_pack_weight_bf16x3(const float *weight, int kernel_volume, int cin, int cout, void *packed, void *stream);
int pd3_sparse_conv3d_features_bf16x3(const float *in_feats, const int32_t *nbr, const int32_t *n_out, int n_out_cap,
                                      int kernel_volume, int cin, int cout, const void *weight_packed,
                                      const float *bias, const float *scale, const float *shift, const float *residual,
                                      int relu, const int32_t *order, float *out, void *stream);
/* Mixed precision (the reference's amp_cfg level O2 for the CenterPoint-Voxel encoder): fp16 feature rows and weights
 * on the fp16 matrix cores, fp32 accumulation; index sets, rulebooks and tile order are the fp32 path's.
 *   pd3_sparse_pack_weight_f16   weight [K, Cin, Cout] fp32 (Paddle layout) -> packed fp16 (K * Cin * Cout halfs) in the
 *                                operand order of the kernel; Cin % 16 == 0, Cout in {32, 64, 128}
 *   pd3_sparse_conv3d_features_f16   as pd3_sparse_conv3d_features_ordered with in_feats / residual / out fp16
 *                                (out fp32 when out_f32 != 0: the encoder's last layer feeds pd3_sparse_to_dense);
 *                                bias / scale / shift stay fp32.  Other shapes return -3 (run the fp32 entry). */
int pd3_sparse_pack_weight_f16(const float *weight, int kernel_volume, int cin, int cout, void *packed, void *stream);
int pd3_sparse_conv3d_features_f16(const void *in_feats_f16, const int32_t *nbr, const int32_t *n_out, int n_out_cap,
                                   int kernel_volume, int cin, int cout, const void *weight_packed_f16,
                                   const float *bias, const float *scale, const float *shift,
                                   const void *residual_f16, int relu, const int32_t *order, void *out, int out_f32,
                                   void *stream);
/* The same kernel as a general gather-GEMM: out[row, out_off + co] = epilogue(sum_k W[k] . in[nbr[row, k]]) into a
 * channel slice of a wider row-major matrix (row stride out_ld).  Under AMP the FPN levels of SecondFPN
 * (paddle3d/models/necks/second_fpn.py:99-157: kernel = stride convolutions / transposed convolutions) are this with a
 * static neighbour table over the pixels of an fp16 NHWC map, each level writing its slice of the concatenated map. */
int pd3_gather_gemm_f16(const void *in_feats_f16, const int32_t *nbr, const int32_t *n_out, int n_out_cap,
                        int kernel_volume, int cin, int cout, const void *weight_packed_f16, const float *bias,
                        const float *scale, const float *shift, const void *residual_f16, int relu,
                        const int32_t *order, void *out, int out_f32, int out_ld, int out_off, void *stream);
/* Plan path: the index sets of a whole encoder without a host round trip between the convolutions (the
 * reference's layers read nnz on the host after every sparse op).  An index set is a SORTED array of keys
 * ((b*D + z)*H + y)*W + x (raster order; 0xFFFFFFFF = padding, at the end) with its length in device memory.
 *   pd3_sparse_sort_coords   coords [n, 4] (rows with batch < 0 are padding) -> keys_sorted [n], order [n]
 *                            (keys_sorted[i] belongs to input row order[i]), n_valid [1]
 *   pd3_sparse_conv_outputs  the output set of a regular convolution: every position an active input reaches
 *                            (paddle.sparse.nn.Conv3D), sorted; n_in / n_out are device counts, out_cap >= the
 *                            number of outputs (min(n_in_cap * prod(ceil(k/s)), batch * D' * H' * W') always is)
 *   pd3_sparse_rulebook      nbr [n_out_cap, kd*kh*kw] (input row or -1) and, if out_coords != NULL, the (b, z,
 *                            y, x) rows of the output set; in_keys / out_keys as produced above (subm != 0: pass
 *                            the same array twice); n_in / n_out may be NULL (= the caps).  A neighbour is looked
 *                            up by a binary search inside its (b, z, y) line of the sorted input array.
 * Workspaces: pd3_sparse_plan_workspace(n) for the sort, pd3_sparse_conv_outputs_workspace (a byte map of the
 * output grid) and pd3_sparse_rulebook_workspace (first row of every (b, z, y) line of the input set). */
size_t pd3_sparse_plan_workspace(int n_cap);
size_t pd3_sparse_conv_outputs_workspace(int batch, const int *spatial_shape, const int *kernel_size,
                                         const int *stride, const int *padding);
int pd3_sparse_sort_coords(const int32_t *coords, int n, int batch, const int *spatial_shape,
                           uint32_t *keys_sorted, int32_t *order, int32_t *n_valid, void *workspace,
                           size_t workspace_bytes, void *stream);
int pd3_sparse_conv_outputs(const uint32_t *in_keys, const int32_t *n_in, int n_in_cap, int batch,
                            const int *spatial_shape, const int *kernel_size, const int *stride,
                            const int *padding, uint32_t *out_keys, int32_t *n_out, int out_cap,
                            void *workspace, size_t workspace_bytes, void *stream);
size_t pd3_sparse_rulebook_workspace(int batch, const int *spatial_shape);  /* spatial_shape of the INPUT set */
int pd3_sparse_rulebook(const uint32_t *in_keys, const int32_t *n_in, int n_in_cap, const uint32_t *out_keys,
                        const int32_t *n_out, int n_out_cap, int batch, const int *spatial_shape,
                        const int *kernel_size, const int *stride, const int *padding, int subm, int32_t *nbr,
                        int32_t *out_coords, void *workspace, size_t workspace_bytes, void *stream);
/* values [n, C] at coords -> dense [batch, C*D, H, W] fp32 (to_dense + transpose + reshape of
 * sparse_resnet.py:202-205); `dense` is fully written (zero where inactive). n may be NULL (= n_cap). */
int pd3_sparse_to_dense(const float *feats, const int32_t *coords, const int32_t *n, int n_cap,
                        int channels, int batch, const int *spatial_shape, float *dense, void *stream);

/* ---------------------------------------------------------------------------------------------
 * merge_sweeps -- the caller-side step right before hard_voxelize on the nuScenes path:
 * LoadPointCloud.__call__'s multi-sweep merge, paddle3d/transforms/reader.py:118-164.
 *   points         [sweep_offsets[num_sweeps], dim_in] fp32: key frame first, then the sweeps (device)
 *   sweep_offsets  host int64[num_sweeps + 1]; sweep 0 is the key frame (kept untouched)
 *   ref_from_curr  host double[num_sweeps][16] row-major 4x4 (NULL: no transform); has_transform host
 *                  int32[num_sweeps] (NULL: every sweep has one; 0 = this sweep's `ref_from_curr` is None,
 *                  reader.py:150, e.g. the key frame repeated as padding at a scene start); time_lag host
 *                  float[num_sweeps] (NULL: zeros); remove_radius = sweep_remove_radius
 *   out            [<= total points, use_dim (+1 if use_time_lag)] fp32, rows in the reference's
 *                  concatenation order (given the sweep order); num_out [1] int32 (device)
 */
size_t pd3_merge_sweeps_workspace(int64_t num_points);
int pd3_merge_sweeps(const float *points, const int64_t *sweep_offsets, int num_sweeps, int dim_in,
                     int use_dim, const double *ref_from_curr, const int32_t *has_transform,
                     const float *time_lag, int use_time_lag,
                     float remove_radius, float *out, int32_t *num_out, void *workspace,
                     size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * conv3x3_bias_relu -- dense 3x3 / pad 1 convolution, stride 1 or 2, with fused bias and ReLU on the fp32
 * matrix cores: the convolutions of SecondBackbone (paddle3d/models/backbones/second_backbone.py:72-120)
 * and CenterHead / SeparateHead (detection/centerpoint/center_head.py:43-220) with BatchNorm folded into
 * weight and bias (cuDNN convolutions in the reference).
 *   x [batch, cin, h, w] fp32 NCHW (16-byte aligned);  out [batch, cout, h/stride, out_w];  bias [cout] or NULL
 *   w_packed: the [cout, cin, 3, 3] weight re-ordered to [cout/64][cin/8][4 channel pairs][9 taps][2][64]
 *             (row = (pair*9 + ky*3 + kx)*2 + channel-of-pair; see paddle3d_amd/ops/conv.py)
 *   w / w_valid / out_w: maps whose width is not a multiple of 4 (CenterPoint-Voxel's 90 x 90 stage) live in rows
 *             padded with zeros to a multiple of 4: w is the row pitch of x, w_valid <= w its real width (columns
 *             >= w_valid hold zeros), out_w >= w_valid / stride the row pitch of out, whose columns >=
 *             w_valid / stride are written as zeros -- the next layer reads them as its own zero padding.
 *             Ordinary maps: w_valid = w, out_w = w / stride.
 *   requires cin % 8 == 0, cout % 64 == 0, h and w_valid multiples of stride, w % 4 == 0, out_w % 4 == 0 (rows
 *   are staged as aligned float4); otherwise PD3_EUNSUPPORTED.  Outputs that are not a multiple of the 4 x 32
 *   pixel tile get partial border tiles (masked stores).
 */
int pd3_conv3x3_bias_relu(const float *x, const float *w_packed, const float *bias, int batch, int cin,
                          int cout, int h, int w, int w_valid, int stride, int relu, float *out, int out_w,
                          void *stream);

/* ---------------------------------------------------------------------------------------------
 * conv3x3_winograd43_bias_relu -- the same stride-1 convolution by Winograd F(4x4, 3x3) (4x fewer multiplies;
 * all fp32; ~1e-5 absolute from the direct form for |y| ~ 1).  One fused kernel.
 *   x [batch, cin, h, w] fp32 NCHW (16-byte aligned);  out [batch, cout, h, w] (16-byte aligned);  bias or NULL
 *   channels_per_tile: 32 or 64 output channels per workgroup (64: one 512-thread workgroup per CU, half the
 *             patch-transform work per MFMA; 32: two 256-thread workgroups per CU, for cout % 64 != 0)
 *   u_packed: U = G g G^T (6x6) of the [cout, cin, 3, 3] weight, packed [cout/T][cin/4][T/16][4][16][36] for
 *             T = channels_per_tile (16-channel block, input channel, channel, component xi*6+nu;
 *             paddle3d_amd/ops/conv.py:pack_winograd43_weight)
 *   w / w_valid: row pitch of x and out, and the real width (see conv3x3_bias_relu); w_valid = w for ordinary maps
 *   requires cin % 4 == 0, cout % T == 0, w % 4 == 0 (any h; partial 8 x 64 tiles at the border are masked)
 */
int pd3_conv3x3_winograd43_bias_relu(const float *x, const float *u_packed, const float *bias, int batch,
                                     int cin, int cout, int h, int w, int w_valid, int relu, float *out,
                                     int channels_per_tile, void *stream);

/* conv3x3_winograd43_pp_bias_relu -- the same convolution, same F(4x4, 3x3) arithmetic, as a two-group ping-pong kernel
 * whose multiply waves issue only MFMAs and LDS reads; U and the raw input rows reach LDS by buffer_load ... lds.
 *   u_lane: U = G g G^T of the [cout, cin, 3, 3] weight in lane order, [cout/64][cin/8][2 trips][4 blocks][9][64 lanes][4]:
 *           lane l of block cb holds channel 64 ct + 16 cb + (l & 15), input channel 8 slot + 4 trip + (l >> 4), float4 q =
 *           components 4q .. 4q+3 of xi*6+nu (paddle3d_amd/ops/conv.py:pack_winograd43_lane_weight); 16-byte aligned
 *   requires cin % 8 == 0, cout % 64 == 0, w % 4 == 0; the same U values as the packed form give the same sums up to the
 *   order of accumulation over input channels (identical here: ascending) */
int pd3_conv3x3_winograd43_pp_bias_relu(const float *x, const float *u_lane, const float *bias, int batch, int cin,
                                        int cout, int h, int w, int w_valid, int relu, float *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * conv3x3_winograd43_ppv_bias_relu -- conv3x3_winograd43_pp_bias_relu for a layer with many output-channel blocks over
 * one input (CenterHead's 36 first-stage convolutions as one 64 -> 2304 layer, center_head.py:99-118): the input transform
 * V = B^T d B is computed ONCE by pd3_winograd43_input_transform and fetched by the convolution instead of being redone
 * by each of its cout / 64 channel blocks (csrc/conv_winograd43_ppv.hip).  Same u_lane, same arguments otherwise, the
 * same bytes out as the pp form.
 *   v_pre: pd3_winograd43_input_transform_floats(batch, cin, h, w) floats (16-byte aligned), [pixel tile][cin / 8][tile
 *          row 2][ci 8][tile 16][36]; cin % 8 == 0 (0 floats / PD3_EUNSUPPORTED otherwise)
 */
size_t pd3_winograd43_input_transform_floats(int batch, int cin, int h, int w);
int pd3_winograd43_input_transform(const float *x, int batch, int cin, int h, int w, int w_valid, float *v_pre,
                                   void *stream);
int pd3_conv3x3_winograd43_ppv_bias_relu(const float *v_pre, const float *u_lane, const float *bias, int batch, int cin,
                                         int cout, int h, int w, int w_valid, int relu, float *out, void *stream);
/* measurement hook: + cycle counters of one workgroup (blockIdx 8), dbg int64 [8 waves][4] (device): transform /
 * multiply / barrier-wait / kernel cycles (the last with the SIMD id in bits 56+; tools/prof/prof_wino_trace.py) */
int pd3_conv3x3_winograd43_pp_trace(const float *x, const float *u_lane, const float *bias, int batch, int cin, int cout,
                                    int h, int w, int relu, float *out, long long *dbg, void *stream);

/* ---------------------------------------------------------------------------------------------
 * stable_argsort -- the index order the host glue of two reference functions needs, on the library's radix sort:
 * rotate_nms_pcdet's `paddle.argsort(scores, descending=True)` (models/layers/layer_libs.py:230-236) and the re-sort
 * by ranks_feat in front of bev_pool_v2_bkwd (models/transformers/bevdet_transformer.py:60-68).  Stable: equal keys
 * keep their input order.
 *   mode 0: keys int32 >= 0 (max_key bounds them: fewer passes), ascending;  mode 1: keys fp32, descending
 *   order [n] int32 (device);  workspace: pd3_stable_argsort_workspace(n, max_key) bytes
 */
size_t pd3_stable_argsort_workspace(int64_t n, uint32_t max_key);
int pd3_stable_argsort(const void *keys, int64_t n, int mode, uint32_t max_key, int32_t *order, void *workspace,
                       size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * conv3x3_f16_bias_relu -- the stride-1 3x3 / pad 1 convolutions of SecondBackbone and CenterHead in MIXED PRECISION:
 * fp16 activations and weights on the fp16 matrix cores, fp32 accumulation, bias + ReLU fused.  The reference's AMP
 * configuration of the model (configs/centerpoint/centerpoint_pillars_02voxel_nuscenes_10sweep_ampO2_ultra.yml:5-9,
 * `amp_cfg: level O2`, cuDNN fp16 convolutions there); an option of the host classes (`model.set_amp(True)`), never the
 * fp32 default.
 *   x            [batch, h, w, cin] fp16 NHWC (16-byte aligned; pd3_f32_nchw_to_f16_nhwc makes it from an fp32 map)
 *   w_packed_f16 [cout / T][cin / 16][9 taps (dy * 3 + dx)][2 halves][T][8] fp16, T = channels_per_tile: element
 *                (ct, c, t, kh, co, e) = W[ct T + co][16 c + 8 kh + e][dy][dx] -- a 16-channel sub-chunk of a channel tile
 *                is one contiguous piece in exactly its LDS order (round 6; paddle3d_amd/ops/conv.py:
 *                pack_conv3x3_f16_weight), bias [cout] fp32 or NULL
 *   out          out_mode 0: [batch, h, w, cout] fp16 NHWC (the next fp16 layer's input);
 *                out_mode 1: [batch, cout, h, w] fp32 NCHW (what the fp32 kernels of the graph read)
 *                out_mode 3: [batch, cout / 64, h, w, 64] fp16 "group-major" (round 6): the 64 channels of each branch of
 *                            the head pixel after pixel -- what pd3_grouped_conv3x3_small_f16_gm reads
 *   channels_per_tile 128 or 64 (workgroup = T channels x 16 rows x 32 columns, walking several (pixel tile, channel
 *                tile) items)
 *   requires cin % 16 == 0, cout % T == 0; else PD3_EUNSUPPORTED (the caller runs fp32).  Maps that are not whole
 *   tiles (config 4's 180 x 180) run with masked border tiles.
 */
int pd3_conv3x3_f16_bias_relu(const void *x_f16_nhwc, const void *w_packed_f16, const float *bias, int batch, int cin,
                              int cout, int h, int w, int relu, void *out, int out_mode, int channels_per_tile,
                              void *stream);
/* out_mode "both": a block's last stride-1 layer under AMP leaves fp16 NHWC (the next block's stride-2 convolution reads
 * it) AND fp32 NCHW (the FPN level reads it) in one pass.  Arguments as pd3_conv3x3_f16_bias_relu. */
int pd3_conv3x3_f16_bias_relu_dual(const void *x_f16_nhwc, const void *w_packed_f16, const float *bias, int batch,
                                   int cin, int cout, int h, int w, int relu, void *out_f16_nhwc, float *out_f32_nchw,
                                   int channels_per_tile, void *stream);
/* The stride-2 3x3 / pad 1 convolutions that open SecondBackbone's blocks (second_backbone.py:84-113) under AMP:
 * x [batch, h, w, cin] fp16 NHWC -> out [batch, (h - 1) / 2 + 1, (w - 1) / 2 + 1, cout] fp16 NHWC; weights packed as for
 * pd3_conv3x3_f16_bias_relu with T = 128; cin % 16 == 0, cout % 128 == 0 (else -3: the caller runs the fp32 kernel). */
int pd3_conv3x3_s2_f16_bias_relu(const void *x_f16_nhwc, const void *w_packed_f16, const float *bias, int batch, int cin,
                                 int cout, int h, int w, int relu, void *out_f16_nhwc, void *stream);
/* PointPillarsScatter fused into that stride-2 convolution (the layer that opens SecondBackbone's block 0 under AMP, the
 * fp16 sibling of pd3_scatter_conv3x3_bias_relu): features_f16 [M, cin] fp16 pillar features, inverse_map [batch, ny * nx]
 * int32 (pd3_pointpillars_inverse_map: pillar row of a cell or -1) -> out [batch, ny / 2, nx / 2, cout] fp16 NHWC; the
 * canvas is never written.  channels_per_tile 64 or 128 (weights packed with that T), cin % 16 == 0. */
int pd3_scatter_conv3x3_s2_f16_bias_relu(const void *features_f16, const int32_t *inverse_map, const void *w_packed_f16,
                                         const float *bias, int batch, int cin, int cout, int ny, int nx, int relu,
                                         void *out_f16_nhwc, int channels_per_tile, void *stream);
/* The final SeparateHead convolutions under AMP (center_head.py:99-118): grouped 3x3 / pad 1 convolution + bias reading
 * the first stage's fp16 NHWC output.
 *   x_f16_nhwc [batch, h, w, groups * 64] fp16; w_f16 [groups][9 taps][out_per_group][64] fp16; bias [groups *
 *   out_per_group] fp32 or NULL; out [batch, out_groups * out_per_group, h, w] fp32 NCHW, the slice's maps at groups
 *   [out_group0, out_group0 + groups); out_per_group 1 .. 4, 64 channels per group (else -3) */
int pd3_grouped_conv3x3_small_f16(const void *x_f16_nhwc, const void *w_f16, const float *bias, int batch, int groups,
                                  int channels_per_group, int out_per_group, int h, int w, float *out, int out_groups,
                                  int out_group0, void *stream);
/* The same convolution on the group-major form of the first stage's output (out_mode 3 of pd3_conv3x3_f16_bias_relu):
 * x [batch, groups, h, w, 64] fp16.  A group's patch rows are contiguous there -- in NHWC its pixels lie groups * 128 bytes
 * apart and the fetch of a tile runs at a third of the rate of contiguous lines.  Persistent kernel, patches by LDS-DMA
 * with the XOR swizzle done by the fetch (csrc/conv_f16.hip).  Same arguments, same results bit for bit; the input below
 * 2 GB (else -3). */
int pd3_grouped_conv3x3_small_f16_gm(const void *x_f16_group_major, const void *w_f16, const float *bias, int batch,
                                     int groups, int channels_per_group, int out_per_group, int h, int w, float *out,
                                     int out_groups, int out_group0, void *stream);
/* x [batch, channels, h, w] fp32 NCHW -> out [batch, h, w, channels] fp16 NHWC (round to nearest even) */
int pd3_f32_nchw_to_f16_nhwc(const float *x, int batch, int channels, int h, int w, void *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * grouped_conv3x3_small -- grouped 3x3 / stride 1 / pad 1 convolution with 1..4 output channels per group and
 * bias, no activation: the final convolutions of all SeparateHead branches (center_head.py:99-118) as one
 * launch over the concatenated first-stage maps.
 *   x [batch, groups*cin_per_group, h, w];  out [batch, groups*cout_per_group, h, w];  bias [groups*cout_per_group] or NULL
 *   w_grouped: [groups][cin_per_group][cout_per_group][9] (the [groups*cout, cin, 3, 3] weight with the two
 *              channel axes swapped inside each group)
 *   requires cin_per_group % 4 == 0, w % 4 == 0 (any h; partial 8 x 128 tiles at the border are masked)
 */
int pd3_grouped_conv3x3_small(const float *x, const float *w_grouped, const float *bias, int batch, int groups,
                              int cin_per_group, int cout_per_group, int h, int w, float *out, void *stream);
/* The same over a SLICE of the groups: x, w_grouped and bias describe `groups` consecutive groups, whose outputs land
 * at groups [out_group0, out_group0 + groups) of out [batch, out_groups*cout_per_group, h, w].  CenterHead runs its
 * 36 branches a few at a time this way, so that a slice's first-stage map (67 MB per branch and 16 frames) is read
 * back while it is still in the last-level cache instead of after 2.4 GB have gone by. */
int pd3_grouped_conv3x3_small_slice(const float *x, const float *w_grouped, const float *bias, int batch, int groups,
                                    int cin_per_group, int cout_per_group, int h, int w, float *out, int out_groups,
                                    int out_group0, void *stream);

/* ---------------------------------------------------------------------------------------------
 * conv3x3_winograd_bias_relu -- the same stride-1 convolution as conv3x3_bias_relu computed by Winograd
 * F(2x2, 3x3) on the fp32 matrix cores (2.25x fewer multiplies, all fp32; results differ from the direct
 * form by fp32 rounding only, ~1e-6 relative).  One fused kernel: input transform, 16 GEMMs, output transform.
 *   x [batch, cin, h, w] fp32 NCHW (16-byte aligned);  out [batch, cout, h, w];  bias [cout] or NULL
 *   u_packed: U = G g G^T of the [cout, cin, 3, 3] weight, packed [cout/32][cin/8][2][8][16][16]
 *             (16-channel block, input channel, channel, component xi*4+nu; see paddle3d_amd/ops/conv.py)
 *   requires cin % 8 == 0, cout % 32 == 0, w % 4 == 0 (any h; partial 8 x 32 tiles at the border are masked)
 */
int pd3_conv3x3_winograd_bias_relu(const float *x, const float *u_packed, const float *bias, int batch,
                                   int cin, int cout, int h, int w, int relu, float *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * patch_conv_bias_relu -- the non-overlapping-patch convolutions of SecondFPN (paddle3d/models/necks/
 * second_fpn.py:99-157; Conv2D / Conv2DTranspose with kernel = stride, BatchNorm folded) as one fp32-MFMA
 * GEMM with bias + ReLU, written at a channel offset of a wider output tensor (the concat of the FPN levels).
 *   mode 0: Conv2D kernel 2 stride 2      x [batch, cin, h, w] -> out[:, off:off+cout] of [batch, ctot, h/2, w/2]
 *           w_packed = A[co][ci*4 + py*2 + px] from the [cout, cin, 2, 2] weight; needs h % 4 == 0, w % 4 == 0
 *   mode 1: 1x1 convolution               -> [batch, ctot, h, w];  A[co][ci]; needs (h*w) % 4 == 0
 *   mode 2: Conv2DTranspose kernel 2 stride 2 -> [batch, ctot, 2h, 2 w_valid];  A[co*4 + dy*2 + dx][ci] from the
 *           [cin, cout, 2, 2] weight; needs (h*w) % 4 == 0; w is the row pitch of x, w_valid <= w its real width
 *           (see conv3x3_bias_relu); modes 0 and 1 need w_valid == w
 *   mode 3: Conv2DTranspose kernel 4 stride 4 -> [batch, ctot, 4h, 4 w_valid];  A[co*16 + dy*4 + dx][ci] from the
 *           [cin, cout, 4, 4] weight (PointPillars' third FPN level, upsample_strides [1, 2, 4]); out 16-byte aligned
 *   A is packed [M/64][K/16][16][64] (paddle3d_amd/ops/conv.py:pack_patch_weight); K % 16 == 0, M % 64 == 0 --
 *   except mode 1, which takes any cout >= 1 with A zero-padded to ceil(cout / 64) * 64 rows (SSDHead's 1x1
 *   convolutions, pointpillars_head.py:62-69; rows >= cout are never stored)
 */
int pd3_patch_conv_bias_relu(const float *x, const float *w_packed, const float *bias, int mode, int batch,
                             int cin, int cout, int h, int w, int w_valid, int relu, float *out,
                             int out_channels_total, int out_channel_offset, void *stream);

/* ---------------------------------------------------------------------------------------------
 * patch_conv_x3_bias_relu -- modes 0, 1, 2 of patch_conv_bias_relu (same reference layers, second_fpn.py:99-157, same
 * tensors and output placement) in fp32 arithmetic on the bf16 matrix cores: every fp32 operand as three bf16 pieces whose
 * sum IS the value, six piece products accumulated in fp32 (csrc/sparse_conv_x3.hip; error vs fp64 = that of the fp32
 * kernel, tested).  One persistent kernel, 128 GEMM rows x 256 pixels per work item.
 *   w_packed: bf16 [row tile][step][16384]: a step's A pieces as the LDS image the kernel fetches, [piece 3][row 128][40]
 *   (32 values of K, 8 of padding) + 1024 of padding (paddle3d_amd/ops/conv.py:pack_patch_weight_x3):
 *     mode 0: row tile = 128 output channels, step = (dy, 16 input channels), k = (ci, dx); needs cin % 16 == 0,
 *             cout % 128 == 0, h % 2 == 0, w % 64 == 0
 *     mode 1: row tile = 128 output channels, step = 32 input channels; needs cin % 32 == 0, cout % 128 == 0
 *     mode 2: row tile = (dy, 64 output channels), rows (dx, co), step = 32 input channels; needs cin % 32 == 0,
 *             cout % 64 == 0; w = row pitch of x, w_valid its real width
 *   bias [cout] or NULL; x 16-byte aligned; out 8-byte aligned; every tensor below 2 GB; PD3_EUNSUPPORTED otherwise
 */
int pd3_patch_conv_x3_bias_relu(const float *x, const void *w_packed, const float *bias, int mode, int batch, int cin,
                                int cout, int h, int w, int w_valid, int relu, float *out, int out_channels_total,
                                int out_channel_offset, void *stream);

/* ---------------------------------------------------------------------------------------------
 * conv3x3_s2_x3_bias_relu -- the stride-2 3x3 / pad 1 convolution + bias + ReLU that opens a SECOND block
 * (second_backbone.py:72-120) in fp32 arithmetic on the bf16 matrix cores (three bf16 pieces per operand, six products,
 * fp32 accumulation: csrc/conv_s2_x3.hip; same tensors as conv3x3_bias_relu with stride 2).
 *   x [batch, cin, h, w] fp32 NCHW (16-byte aligned; w = row pitch, w_valid <= w the real width, pad columns zero: the
 *   convention of conv3x3_bias_relu) -> out [batch, cout, h/2, out_w], w_valid/2 real columns, the rest of the pitch zero
 *   w_packed: bf16 [cout/128][step = (16-channel chunk, ky)][24576]: a step's A pieces as the LDS image the kernel fetches,
 *             [piece 3][row 128][56] (k = kx * 16 + channel: 48 values, 8 of padding) + 3072 of padding
 *             (paddle3d_amd/ops/conv.py:pack_conv3x3_s2_x3_weight)
 *   needs cin % 16 == 0, cout % 128 == 0 (<= 1024), h % 2 == 0, w_valid % 2 == 0, w % 4 == 0, every tensor below 2 GB;
 *   PD3_EUNSUPPORTED otherwise (the caller runs conv3x3_bias_relu)
 */
int pd3_conv3x3_s2_x3_bias_relu(const float *x, const void *w_packed, const float *bias, int batch, int cin, int cout,
                                int h, int w, int w_valid, int relu, float *out, int out_w, void *stream);

/*
 * ssd_postprocess -- SSDHead.post_process of PointPillars for a whole batch (paddle3d/models/detection/
 * pointpillars/pointpillars_head.py:86-196: PointPillarsCoder.decode (pointpillars_coder.py:126-148), the
 * anchors_mask selection, sigmoid -> max / argmax, direction argmax, score (>=) and centre-range filter, heading
 * flip, rotate_nms_pcdet (models/layers/layer_libs.py:210-249), index_select) together with
 * AnchorGenerator.generate_anchors_mask (anchors_generator.py:103-121, :191-210) that test_forward runs per frame
 * in front of it (pointpillars.py:120-127).
 *   head_map     [batch, C, feat_h, feat_w] fp32, the 1x1 head convolutions' output as they write it (NCHW);
 *                batch_stride = elements between frames; class logits of anchor j at channels
 *                cls_channel0 + j*(num_classes + !encode_background_as_zeros) + k, box code at box_channel0 + j*7 + k,
 *                direction logits at dir_channel0 + j*2 + k (dir_channel0 < 0: use_direction_classifier=False)
 *   anchors      [A, 7] fp32 (x, y, z, w, l, h, r), A = feat_h*feat_w*anchors_per_loc, index (y*feat_w + x)*apl + j
 *   anchors_bv   [A, 4] int32 pillar-index boxes (xmin, ymin, xmax, ymax) (AnchorGenerator.anchors_bv)
 *   coors        [num_coors, 4] int32 (batch, z, y, x) of the batch's pillars; rows with batch outside [0, batch) skipped
 *   center_limit_range  6 host floats or NULL (prediction_center_limit_range=None)
 * Outputs (device): out_boxes [batch, max(post,1), 7], out_scores [batch, max(post,1)], out_labels int64, out_count
 * [batch] int32.  A frame without detections (no anchor passes the area test, or none the score / range test) has
 * count 0 and the reference's `_box_empty` row (zeros, -1, -1) in row 0.  No host synchronisation.
 * selection: how the top nms_pre_max_size anchors by score are found -- 0 = radix select + ordered compaction
 * (nms_pre_max_size <= 1024; larger caps take the sort), 1 = a full stable sort of every anchor's key (the
 * reference's argsort); identical results, the argument exists so that tests run both.
 */
size_t pd3_ssd_postprocess_workspace(int batch, int feat_h, int feat_w, int anchors_per_loc, int grid_x, int grid_y,
                                     int nms_pre_max_size);
int pd3_ssd_postprocess(const float *head_map, int64_t batch_stride, int cls_channel0, int box_channel0,
                        int dir_channel0, int batch, int feat_h, int feat_w, int anchors_per_loc, int num_classes,
                        int encode_background_as_zeros, const float *anchors, const int32_t *anchors_bv,
                        const int32_t *coors, int64_t num_coors, int grid_x, int grid_y,
                        float anchor_area_threshold, float score_threshold, const float *center_limit_range,
                        float nms_iou_threshold, int nms_pre_max_size, int nms_post_max_size, float *out_boxes,
                        float *out_scores, int64_t *out_labels, int32_t *out_count, void *workspace,
                        size_t workspace_bytes, void *stream, int selection);

#ifdef __cplusplus
}
#endif
#endif /* PADDLE3D_AMD_H */
