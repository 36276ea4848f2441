/* oracle/port/oracle_port.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("port") of the reference algorithms on the LiDAR-detection hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (paddle3d_amd/) never does.  Parity of this restatement is PINNED against the
 * reference's own code compiled from /root/reference (oracle/_ref, see oracle/ref_wrap.cpp) by
 * tests/test_oracle.py and by the golden vectors under tests/golden/ that were generated from it.
 */
#ifndef ORACLE_PORT_H
#define ORACLE_PORT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int port_hard_voxelize(const float *points, int64_t n, int d, const float *voxel_size,
                       const float *pc_range, int max_pts, int max_voxels, float *voxels,
                       int32_t *coords, int32_t *num_pts, int32_t *num_voxels);

float port_box_overlap(const float *a, const float *b);
float port_iou_bev(const float *a, const float *b);
float port_iou_normal(const float *a, const float *b);
void port_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out);
void port_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out);
void port_nms(const float *boxes, int n, float thresh, int normal, int32_t *keep, int32_t *num);

void port_pillar_scatter(const float *feats, const int32_t *coords, int64_t m, int c, int batch,
                         int ny, int nx, float *canvas);

/* centerpoint_postprocess for ONE task; returns number of output rows (>=1: fake row if empty). */
int port_centerpoint_postprocess_task(const float *hm, int ncls, const float *reg,
                                      const float *height, const float *dim, const float *vel,
                                      const float *rot, int feat_h, int feat_w,
                                      const float *voxel_size, const float *pc_range,
                                      const float *post_center_range, int label_offset,
                                      int down_ratio, float score_threshold,
                                      float nms_iou_threshold, int nms_pre_max_size,
                                      int nms_post_max_size, int with_velocity, float *out_boxes,
                                      float *out_scores, int64_t *out_labels,
                                      float *min_margin /* optional diagnostics[2] */);

void port_bev_pool_v2(int c, int n_intervals, const float *depth, const float *feat,
                      const int32_t *ranks_depth, const int32_t *ranks_feat,
                      const int32_t *ranks_bev, const int32_t *interval_starts,
                      const int32_t *interval_lengths, float *out);
void port_bev_pool_v2_bkwd(int c, int n_intervals, const float *out_grad, const float *depth,
                           const float *feat, const int32_t *ranks_depth,
                           const int32_t *ranks_feat, const int32_t *ranks_bev,
                           const int32_t *interval_starts, const int32_t *interval_lengths,
                           float *depth_grad, float *feat_grad);

int port_hard_voxelize_f64(const double *points, int64_t n, int d, const float *voxel_size,
                           const float *pc_range, int max_pts, int max_voxels, double *voxels,
                           int32_t *coords, int32_t *num_pts, int32_t *num_voxels);
void port_libm_eval(int op, const float *x, const float *y, float *out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
