/* oracle/port/oracle_port.c -- TEST INFRASTRUCTURE ONLY (see oracle_port.h).
 *
 * CPU restatement of the reference algorithms, plain C11, IEEE fp32, no FMA contraction
 * (built with -ffp-contract=off and no -march flags, see oracle/Makefile).  Every function cites
 * the reference file:line it follows (paths relative to /root/reference/paddle3d/ops).
 * Parity status: PINNED -- tests/test_oracle.py checks every function here bit-for-bit against the
 * reference's own code compiled into oracle/_ref, and against tests/golden/ vectors made from it.
 */
#include "oracle_port.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * hard_voxelize  (voxel/voxelize_op.cc:19-82 kernel, :84-140 driver)
 * ---------------------------------------------------------------------------------------------- */
static int grid_extent(float lo, float hi, float step) {
  /* voxelize_op.cc:97-102: static_cast<int>(round((max - min) / size)), fp32 operands, double round */
  return (int)round((double)((hi - lo) / step));
}

int port_hard_voxelize(const float *points, int64_t n, int d, const float *voxel_size,
                       const float *pc_range, int max_pts, int max_voxels, float *voxels,
                       int32_t *coords, int32_t *num_pts, int32_t *num_voxels) {
  const int gx = grid_extent(pc_range[0], pc_range[3], voxel_size[0]);
  const int gy = grid_extent(pc_range[1], pc_range[4], voxel_size[1]);
  const int gz = grid_extent(pc_range[2], pc_range[5], voxel_size[2]);
  const size_t cells = (size_t)gx * gy * gz;
  int32_t *cell_to_voxel = (int32_t *)malloc(cells * sizeof(int32_t));
  if (!cell_to_voxel) return -1;
  for (size_t i = 0; i < cells; ++i) cell_to_voxel[i] = -1; /* :123-126 */
  memset(voxels, 0, sizeof(float) * (size_t)max_voxels * max_pts * d); /* :29-31 */
  memset(coords, 0, sizeof(int32_t) * (size_t)max_voxels * 3);         /* :108-110 */
  memset(num_pts, 0, sizeof(int32_t) * (size_t)max_voxels);            /* :112-117 */
  int made = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float *p = points + i * d;
    /* :37-45  fp32 subtract, fp32 divide, floor, truncate to int */
    const int cx = (int)floorf((p[0] - pc_range[0]) / voxel_size[0]);
    const int cy = (int)floorf((p[1] - pc_range[1]) / voxel_size[1]);
    const int cz = (int)floorf((p[2] - pc_range[2]) / voxel_size[2]);
    if (cx < 0 || cx >= gx || cy < 0 || cy >= gy || cz < 0 || cz >= gz) continue; /* :47-55 */
    const size_t cell = ((size_t)cz * gy + cy) * gx + cx;                          /* :57-58 */
    int v = cell_to_voxel[cell];
    if (v < 0) {
      if (made >= max_voxels) continue; /* :60-64 cap: new voxels refused, old ones still fed */
      v = made++;
      cell_to_voxel[cell] = v;
      coords[v * 3 + 0] = cz; /* :67-69 (z, y, x) */
      coords[v * 3 + 1] = cy;
      coords[v * 3 + 2] = cx;
    }
    const int k = num_pts[v];
    if (k < max_pts) { /* :71-79 */
      memcpy(voxels + ((size_t)v * max_pts + k) * d, p, sizeof(float) * d);
      num_pts[v] = k + 1;
    }
  }
  num_voxels[0] = made;
  free(cell_to_voxel);
  return 0;
}

/* the same scan with T = double (PD_DISPATCH_FLOATING_TYPES, voxelize_op.cc:128): the float attributes are promoted,
 * the cell index is floor((p - (double)min) / (double)size) in double (:37-45) */
int port_hard_voxelize_f64(const double *points, int64_t n, int d, const float *voxel_size,
                           const float *pc_range, int max_pts, int max_voxels, double *voxels,
                           int32_t *coords, int32_t *num_pts, int32_t *num_voxels) {
  const int gx = grid_extent(pc_range[0], pc_range[3], voxel_size[0]);
  const int gy = grid_extent(pc_range[1], pc_range[4], voxel_size[1]);
  const int gz = grid_extent(pc_range[2], pc_range[5], voxel_size[2]);
  const size_t cells = (size_t)gx * gy * gz;
  int32_t *cell_to_voxel = (int32_t *)malloc(cells * sizeof(int32_t));
  if (!cell_to_voxel) return -1;
  for (size_t i = 0; i < cells; ++i) cell_to_voxel[i] = -1;
  memset(voxels, 0, sizeof(double) * (size_t)max_voxels * max_pts * d);
  memset(coords, 0, sizeof(int32_t) * (size_t)max_voxels * 3);
  memset(num_pts, 0, sizeof(int32_t) * (size_t)max_voxels);
  int made = 0;
  for (int64_t i = 0; i < n; ++i) {
    const double *p = points + i * d;
    const int cx = (int)floor((p[0] - pc_range[0]) / voxel_size[0]);
    const int cy = (int)floor((p[1] - pc_range[1]) / voxel_size[1]);
    const int cz = (int)floor((p[2] - pc_range[2]) / voxel_size[2]);
    if (cx < 0 || cx >= gx || cy < 0 || cy >= gy || cz < 0 || cz >= gz) continue;
    const size_t cell = ((size_t)cz * gy + cy) * gx + cx;
    int v = cell_to_voxel[cell];
    if (v < 0) {
      if (made >= max_voxels) continue;
      v = made++;
      cell_to_voxel[cell] = v;
      coords[v * 3 + 0] = cz;
      coords[v * 3 + 1] = cy;
      coords[v * 3 + 2] = cx;
    }
    const int k = num_pts[v];
    if (k < max_pts) {
      memcpy(voxels + ((size_t)v * max_pts + k) * d, p, sizeof(double) * d);
      num_pts[v] = k + 1;
    }
  }
  num_voxels[0] = made;
  free(cell_to_voxel);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * rotated BEV IoU  (iou3d_nms/iou3d_cpu.cpp:36-239)
 * ---------------------------------------------------------------------------------------------- */
#define GEOM_EPS 1e-8f   /* iou3d_cpu.cpp:35 */
#define IN_MARGIN 1e-2f  /* iou3d_cpu.cpp:75 */

typedef struct {
  float x, y;
} v2;

static float lo2(float a, float b) { return a > b ? b : a; } /* :31 */
static float hi2(float a, float b) { return a > b ? a : b; } /* :33 */

/* cross(p1, p2, p0)  :61-63 */
static float tri(v2 p1, v2 p2, v2 p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

/* intersection(p1, p0, q1, q0, ans)  :88-118 */
static int seg_hit(v2 p1, v2 p0, v2 q1, v2 q0, v2 *ans) {
  /* check_rect_cross(p0, p1, q0, q1)  :65-72 */
  if (!(lo2(p0.x, p1.x) <= hi2(q0.x, q1.x) && lo2(q0.x, q1.x) <= hi2(p0.x, p1.x) &&
        lo2(p0.y, p1.y) <= hi2(q0.y, q1.y) && lo2(q0.y, q1.y) <= hi2(p0.y, p1.y)))
    return 0;
  const float s1 = tri(q0, p1, p0);
  const float s2 = tri(p1, q1, p0);
  const float s3 = tri(p0, q1, q0);
  const float s4 = tri(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  const float s5 = tri(q1, p1, p0);
  if (fabsf(s5 - s1) > GEOM_EPS) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

/* check_in_box2d  :74-86 */
static int inside(const float *box, v2 p) {
  const float c = cosf(-box[6]), s = sinf(-box[6]);
  const float rx = (p.x - box[0]) * c + (p.y - box[1]) * (-s);
  const float ry = (p.x - box[0]) * s + (p.y - box[1]) * c;
  return fabsf(rx) < box[3] / 2 + IN_MARGIN && fabsf(ry) < box[4] / 2 + IN_MARGIN;
}

static void corners(const float *box, v2 *out /*[5]*/) {
  /* :139-160 axis-aligned corners then rotate_around_center (:120-127) */
  const float hx = box[3] / 2, hy = box[4] / 2;
  const float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
  const float c = cosf(box[6]), s = sinf(box[6]);
  const float px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
  for (int k = 0; k < 4; ++k) {
    out[k].x = (px[k] - box[0]) * c + (py[k] - box[1]) * (-s) + box[0];
    out[k].y = (px[k] - box[0]) * s + (py[k] - box[1]) * c + box[1];
  }
  out[4] = out[0]; /* :162-163 */
}

float port_box_overlap(const float *a, const float *b) { /* box_overlap :134-229 */
  v2 ca[5], cb[5], poly[24]; /* reference uses 16; 24 only guards the degenerate overflow */
  corners(a, ca);
  corners(b, cb);
  /* NB: the reference rotates a and b corner k alternately; the values do not depend on that order */
  int cnt = 0;
  v2 ctr = {0.f, 0.f};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { /* :170-181 */
      if (seg_hit(ca[i + 1], ca[i], cb[j + 1], cb[j], &poly[cnt])) {
        ctr.x = ctr.x + poly[cnt].x;
        ctr.y = ctr.y + poly[cnt].y;
        ++cnt;
      }
    }
  for (int k = 0; k < 4; ++k) { /* :184-195 */
    if (inside(a, cb[k])) {
      ctr.x = ctr.x + cb[k].x;
      ctr.y = ctr.y + cb[k].y;
      poly[cnt++] = cb[k];
    }
    if (inside(b, ca[k])) {
      ctr.x = ctr.x + ca[k].x;
      ctr.y = ctr.y + ca[k].y;
      poly[cnt++] = ca[k];
    }
  }
  ctr.x /= cnt; /* :197-198 (0/0 -> NaN when cnt==0; unused then) */
  ctr.y /= cnt;
  /* bubble sort by centroid angle, point_cmp :129-132, loop :201-210 */
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i) {
      if (atan2f(poly[i].y - ctr.y, poly[i].x - ctr.x) >
          atan2f(poly[i + 1].y - ctr.y, poly[i + 1].x - ctr.x)) {
        v2 t = poly[i];
        poly[i] = poly[i + 1];
        poly[i + 1] = t;
      }
    }
  float area = 0; /* :213-217 shoelace fan around poly[0] */
  for (int k = 0; k < cnt - 1; ++k) {
    const v2 u = {poly[k].x - poly[0].x, poly[k].y - poly[0].y};
    const v2 w = {poly[k + 1].x - poly[0].x, poly[k + 1].y - poly[0].y};
    area += u.x * w.y - u.y * w.x;
  }
  return (float)(fabsf(area) / 2.0); /* :219 */
}

float port_iou_bev(const float *a, const float *b) { /* iou_bev :222-229 */
  const float sa = a[3] * a[4], sb = b[3] * b[4];
  const float so = port_box_overlap(a, b);
  return so / fmaxf(sa + sb - so, GEOM_EPS);
}

float port_iou_normal(const float *a, const float *b) { /* iou3d_nms_kernel.cu:365-378 */
  const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2);
  const float right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2);
  const float bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
  const float inter = w * h;
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, GEOM_EPS);
}

void port_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out) {
  for (int i = 0; i < na; ++i) /* iou3d_cpu.cpp:257-262 */
    for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = port_iou_bev(a + i * 7, b + j * 7);
}

void port_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out) {
  for (int i = 0; i < na; ++i) /* iou3d_nms_kernel.cu:275-290 */
    for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = port_box_overlap(a + i * 7, b + j * 7);
}

/* Greedy NMS = suppression bits of nms_kernel (iou3d_nms_kernel.cu:310-363: bit (i,j) for j>i when
 * iou > thresh, strict) followed by the host sweep of nms_gpu (iou3d_nms.cpp:119-137).  Restated
 * without the bit matrix: box i is kept iff no earlier KEPT box suppresses it. */
static void greedy_nms(const float *boxes, int n, float thresh, int normal, int32_t *keep,
                       int32_t *num, float *min_margin) {
  unsigned char *dead = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    if (dead[i]) continue;
    keep[kept++] = i;
    for (int j = i + 1; j < n; ++j) {
      const float v = normal ? port_iou_normal(boxes + (size_t)i * 7, boxes + (size_t)j * 7)
                             : port_iou_bev(boxes + (size_t)i * 7, boxes + (size_t)j * 7);
      if (min_margin) {
        const float m = fabsf(v - thresh);
        if (m < *min_margin) *min_margin = m;
      }
      if (v > thresh) dead[j] = 1;
    }
  }
  *num = kept;
  free(dead);
}

void port_nms(const float *boxes, int n, float thresh, int normal, int32_t *keep, int32_t *num) {
  greedy_nms(boxes, n, thresh, normal, keep, num, NULL);
}

/* ------------------------------------------------------------------------------------------------
 * PointPillarsScatter.forward_batch  (models/middle_encoders/pillar_scatter.py:57-93)
 * canvas[b, :, y, x] = feats[m, :] for coords[m] = (b, z, y, x); everything else zero.
 * ---------------------------------------------------------------------------------------------- */
void port_pillar_scatter(const float *feats, const int32_t *coords, int64_t m, int c, int batch,
                         int ny, int nx, float *canvas) {
  const size_t plane = (size_t)ny * nx;
  memset(canvas, 0, sizeof(float) * (size_t)batch * c * plane);
  for (int64_t i = 0; i < m; ++i) {
    const int b = coords[i * 4 + 0];
    if (b < 0 || b >= batch) continue;
    const size_t cell = (size_t)coords[i * 4 + 2] * nx + coords[i * 4 + 3]; /* :77 */
    for (int k = 0; k < c; ++k) canvas[((size_t)b * c + k) * plane + cell] = feats[i * c + k];
  }
}

/* ------------------------------------------------------------------------------------------------
 * centerpoint_postprocess, one task  (centerpoint_postprocess/postprocess.cu:32-80 decode,
 * :137-278 orchestration; iou3d_nms_kernel.cu:294-308 box remap for NMS)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  float score;
  int pos;
} sel_t;

static int by_score_desc_stable(const void *pa, const void *pb) {
  const sel_t *a = (const sel_t *)pa, *b = (const sel_t *)pb;
  if (a->score > b->score) return -1;
  if (a->score < b->score) return 1;
  return a->pos - b->pos; /* stable: equal scores keep masked_select order */
}

int port_centerpoint_postprocess_task(const float *hm, int ncls, const float *reg,
                                      const float *height, const float *dim, const float *vel,
                                      const float *rot, int feat_h, int feat_w,
                                      const float *voxel_size, const float *pc_range,
                                      const float *pcr, int label_offset, int down_ratio_i,
                                      float score_threshold, float nms_iou_threshold,
                                      int nms_pre_max_size, int nms_post_max_size,
                                      int with_velocity, float *out_boxes, float *out_scores,
                                      int64_t *out_labels, float *min_margin) {
  const int hw = feat_h * feat_w;
  const int dims = with_velocity ? 9 : 7;
  const float down_ratio = (float)down_ratio_i; /* int attr received as float, postprocess.cu:35,85 */
  float *boxes = (float *)malloc(sizeof(float) * (size_t)hw * dims);
  float *score = (float *)malloc(sizeof(float) * (size_t)hw);
  int *label = (int *)malloc(sizeof(int) * (size_t)hw);
  sel_t *sel = (sel_t *)malloc(sizeof(sel_t) * (size_t)hw);
  int *cell_of = (int *)malloc(sizeof(int) * (size_t)hw); /* selected_score_idx */
  int nsel = 0;
  if (min_margin) min_margin[0] = min_margin[1] = INFINITY;
  for (int i = 0; i < hw; ++i) {
    /* :145-149 sigmoid -> max / argmax over the class axis (first maximum wins) */
    float best = -1.f;
    int arg = 0;
    for (int k = 0; k < ncls; ++k) {
      const float s = 1.0f / (1.0f + expf(-hm[(size_t)k * hw + i]));
      if (k == 0 || s > best) {
        best = s;
        arg = k;
      }
    }
    score[i] = best;
    label[i] = arg;
    /* decode_kernel :41-70 */
    const int xs = i % feat_w, ys = i / feat_w;
    const float x = reg[i], y = reg[i + hw], z = height[i];
    float *bx = boxes + (size_t)i * dims;
    bx[0] = (x + xs) * down_ratio * voxel_size[0] + pc_range[0];
    bx[1] = (y + ys) * down_ratio * voxel_size[1] + pc_range[1];
    bx[2] = z;
    bx[3] = expf(dim[i]); /* :151 exp(dim) */
    bx[4] = expf(dim[i + hw]);
    bx[5] = expf(dim[i + 2 * hw]);
    if (with_velocity) {
      bx[6] = vel[i];
      bx[7] = vel[i + hw];
      bx[8] = atan2f(rot[i], rot[i + hw]);
    } else {
      bx[6] = atan2f(rot[i], rot[i + hw]);
    }
    /* :72-77 mask on the RAW reg/height values */
    if (min_margin) {
      const float m = fabsf(best - score_threshold);
      if (m < min_margin[0]) min_margin[0] = m;
    }
    if (best > score_threshold && x <= pcr[3] && y <= pcr[4] && z <= pcr[5] && x >= pcr[0] &&
        y >= pcr[1] && z >= pcr[2]) {
      /* masked_select keeps ascending cell order (:184-188): entry nsel <-> cell i */
      sel[nsel].score = best;
      sel[nsel].pos = nsel;
      cell_of[nsel] = i;
      ++nsel;
    }
  }
  int rows = 0;
  if (nsel == 0) { /* :190-201 fake output */
    memset(out_boxes, 0, sizeof(float) * dims);
    out_scores[0] = -1.f;
    out_labels[0] = 0;
    rows = 1;
  } else {
    qsort(sel, (size_t)nsel, sizeof(sel_t), by_score_desc_stable); /* :204-205 argsort desc */
    const int n = nsel > nms_pre_max_size ? nms_pre_max_size : nsel; /* :206-207 */
    float *nb = (float *)malloc(sizeof(float) * (size_t)n * 7);
    for (int r = 0; r < n; ++r) { /* iou3d_nms_kernel.cu:294-308 */
      const float *bx = boxes + (size_t)cell_of[sel[r].pos] * dims;
      nb[r * 7 + 0] = bx[0];
      nb[r * 7 + 1] = bx[1];
      nb[r * 7 + 2] = bx[2];
      nb[r * 7 + 3] = bx[4];
      nb[r * 7 + 4] = bx[3];
      nb[r * 7 + 5] = bx[5];
      nb[r * 7 + 6] = (float)(-bx[dims - 1] - 3.141592653589793 / 2);
    }
    int32_t *keep = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t nkeep = 0;
    float mm = INFINITY;
    greedy_nms(nb, n, nms_iou_threshold, 0, keep, &nkeep, min_margin ? &mm : NULL);
    if (min_margin) min_margin[1] = mm;
    rows = nkeep > nms_post_max_size ? nms_post_max_size : nkeep; /* :247-248 */
    for (int r = 0; r < rows; ++r) { /* :254-271 gathers */
      const int cell = cell_of[sel[keep[r]].pos];
      memcpy(out_boxes + (size_t)r * dims, boxes + (size_t)cell * dims, sizeof(float) * dims);
      out_scores[r] = sel[keep[r]].score;
      out_labels[r] = (int64_t)label[cell] + label_offset;
    }
    free(keep);
    free(nb);
  }
  free(cell_of);
  free(sel);
  free(label);
  free(score);
  free(boxes);
  return rows;
}

/* ------------------------------------------------------------------------------------------------
 * bev_pool_v2 forward / backward  (bev_pool_v2/bev_pool_cuda.cu:18-44,
 * bev_pool_v2_backward/bev_pool_cuda_bkwd.cu:44-94); fp32 accumulation in interval order.
 * `out` / grads are zero-filled here as bev_pool.cc:48-49 / bev_pool_bkwd.cc:41-46 do -- the caller
 * passes the element counts through the shapes it allocated, so zero-filling is the caller's job.
 * ---------------------------------------------------------------------------------------------- */
void port_bev_pool_v2(int c, int n_intervals, const float *depth, const float *feat,
                      const int32_t *ranks_depth, const int32_t *ranks_feat,
                      const int32_t *ranks_bev, const int32_t *interval_starts,
                      const int32_t *interval_lengths, float *out) {
  for (int iv = 0; iv < n_intervals; ++iv) {
    const int s = interval_starts[iv], len = interval_lengths[iv];
    for (int ch = 0; ch < c; ++ch) {
      float acc = 0;
      for (int i = 0; i < len; ++i)
        acc += feat[(size_t)ranks_feat[s + i] * c + ch] * depth[ranks_depth[s + i]];
      out[(size_t)ranks_bev[s] * c + ch] = acc;
    }
  }
}

void port_bev_pool_v2_bkwd(int c, int n_intervals, const float *out_grad, const float *depth,
                           const float *feat, const int32_t *ranks_depth,
                           const int32_t *ranks_feat, const int32_t *ranks_bev,
                           const int32_t *interval_starts, const int32_t *interval_lengths,
                           float *depth_grad, float *feat_grad) {
  for (int iv = 0; iv < n_intervals; ++iv) {
    const int s = interval_starts[iv], len = interval_lengths[iv];
    for (int i = 0; i < len; ++i) { /* :62-75 */
      const float *g = out_grad + (size_t)ranks_bev[s + i] * c;
      const float *f = feat + (size_t)ranks_feat[s + i] * c;
      float acc = 0;
      for (int ch = 0; ch < c; ++ch) acc += g[ch] * f[ch];
      depth_grad[ranks_depth[s + i]] = acc;
    }
    for (int ch = 0; ch < c; ++ch) { /* :79-92 */
      float acc = 0;
      for (int i = 0; i < len; ++i)
        acc += out_grad[(size_t)ranks_bev[s + i] * c + ch] * depth[ranks_depth[s + i]];
      feat_grad[(size_t)ranks_feat[s] * c + ch] = acc;
    }
  }
}

/* the host libm's float routines over an array (what the reference's cos / sin / atan2 / exp on floats call):
 * the tests hold the device's restatement (csrc/libm_exact.hpp) to them.  op as pd3_libm_eval. */
void port_libm_eval(int op, const float *x, const float *y, float *out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    volatile float a = x[i];
    switch (op) {
      case 0: out[i] = sinf(a); break;
      case 1: out[i] = cosf(a); break;
      case 2: out[i] = expf(a); break;
      case 3: out[i] = atanf(a); break;
      default: out[i] = atan2f(a, y[i]); break;
    }
  }
}
