// Rotated-rectangle BEV overlap / IoU for gfx950, value-compatible with the reference CPU code
// (reference: paddle3d/ops/iou3d_nms/iou3d_cpu.cpp:36-239; the CUDA twin iou3d_nms_kernel.cu:28-273).
//
// Arithmetic contract (what makes the keep indices match the CPU reference):
//   * identical fp32 operation order for every product / sum (the library is built with
//     -ffp-contract=off, so nothing below is fused into FMA);
//   * sin / cos / atan2 are glibc's cosf / sinf / atan2f bit for bit (libm_exact.hpp: the Arm Optimized Routines
//     fp64 polynomials with the fused operations of glibc's x86-64 FMA build, fdlibm's float atan2f), which is
//     what the reference's `cos(-box[6])` etc. on floats call (iou3d_cpu.cpp:77-79,128-129);
//   * per-box quantities (corners, cos/sin of -heading, half extents + MARGIN) are computed once per
//     box instead of once per pair: they are pure functions of the box, so the values are the same;
//   * the centroid-angle bubble sort keeps the reference's comparison order (angles are computed once
//     per vertex; the reference recomputes the same atan2 at every comparison).
// An exact early-out skips pairs whose circumscribed circles (plus slack far larger than MARGIN) are
// disjoint: the reference finds no vertex for them and returns exactly 0.
#pragma once
#include "common.hpp"
#include "libm_exact.hpp"

namespace pd3 {

constexpr float kGeomEps = 1e-8f;   // iou3d_cpu.cpp:35
constexpr float kInMargin = 1e-2f;  // iou3d_cpu.cpp:75

struct Pt {
  float x, y;
};

// Everything box_overlap needs about one box.
struct BoxPre {
  float cx, cy;      // centre
  float area;        // box[3] * box[4]
  float lim_x, lim_y;  // box[3]/2 + MARGIN, box[4]/2 + MARGIN   (check_in_box2d :84-85)
  float ncos, nsin;  // cos(-heading), sin(-heading)             (check_in_box2d :79-81)
  float rad;         // circumscribed radius (for the exact early-out only)
  Pt c[4];           // rotated corners                            (:139-160)
};

// glibc's bits, not the correctly rounded value (libm_exact.hpp): the reference's cos / sin / atan2 on floats
__device__ __forceinline__ float cos_rn(float a) { return lm::cosf(a); }
__device__ __forceinline__ float sin_rn(float a) { return lm::sinf(a); }
__device__ __forceinline__ float atan2_rn(float y, float x) { return lm::atan2f(y, x); }

__device__ __forceinline__ BoxPre box_prepare(const float* b) {
  BoxPre p;
  p.cx = b[0];
  p.cy = b[1];
  p.area = b[3] * b[4];
  const float hx = b[3] / 2, hy = b[4] / 2;
  p.lim_x = hx + kInMargin;
  p.lim_y = hy + kInMargin;
  p.ncos = cos_rn(-b[6]);
  p.nsin = sin_rn(-b[6]);
  const float x1 = b[0] - hx, y1 = b[1] - hy, x2 = b[0] + hx, y2 = b[1] + hy;
  const float ac = cos_rn(b[6]), as = sin_rn(b[6]);
  const float px[4] = {x1, x2, x2, x1};
  const float py[4] = {y1, y1, y2, y2};
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // rotate_around_center :120-127
    p.c[k].x = (px[k] - b[0]) * ac + (py[k] - b[1]) * (-as) + b[0];
    p.c[k].y = (px[k] - b[0]) * as + (py[k] - b[1]) * ac + b[1];
  }
  p.rad = sqrtf(hx * hx + hy * hy);
  return p;
}

__device__ __forceinline__ float lo2(float a, float b) { return a > b ? b : a; }  // :31
__device__ __forceinline__ float hi2(float a, float b) { return a > b ? a : b; }  // :33

// cross(p1, p2, p0) :61-63
__device__ __forceinline__ float tri(Pt p1, Pt p2, Pt p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

// intersection(p1, p0, q1, q0, ans) :88-118
__device__ __forceinline__ bool seg_hit(Pt p1, Pt p0, Pt q1, Pt q0, Pt& ans) {
  if (!(lo2(p0.x, p1.x) <= hi2(q0.x, q1.x) && lo2(q0.x, q1.x) <= hi2(p0.x, p1.x) &&
        lo2(p0.y, p1.y) <= hi2(q0.y, q1.y) && lo2(q0.y, q1.y) <= hi2(p0.y, p1.y)))
    return false;
  const float s1 = tri(q0, p1, p0);
  const float s2 = tri(p1, q1, p0);
  const float s3 = tri(p0, q1, q0);
  const float s4 = tri(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = tri(q1, p1, p0);
  if (fabsf(s5 - s1) > kGeomEps) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / D;
    ans.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

// check_in_box2d :74-86
__device__ __forceinline__ bool inside(const BoxPre& box, Pt p) {
  const float rx = (p.x - box.cx) * box.ncos + (p.y - box.cy) * (-box.nsin);
  const float ry = (p.x - box.cx) * box.nsin + (p.y - box.cy) * box.ncos;
  return fabsf(rx) < box.lim_x && fabsf(ry) < box.lim_y;
}

// The vertex list of box_overlap (iou3d_cpu.cpp: `Point cross_points[16]`) lives in LDS, not in a private array: a
// dynamically indexed private array is scratch memory on this machine (every append, every compare of the bubble
// sort a round trip through the vector memory path), and an overlap evaluation was a ~90 k-cycle latency chain.
// Layout of one wave's region: [x | y | angle][16 vertices][64 lanes] floats; a lane passes `region + lane`.
constexpr int kPolyCap = 16;
constexpr int kPolyWaveFloats = 3 * kPolyCap * 64;

// box_overlap :134-229
__device__ inline float box_overlap(const BoxPre& a, const BoxPre& b, float* __restrict__ st) {
  {
    const float dx = a.cx - b.cx, dy = a.cy - b.cy, r = a.rad + b.rad + 0.25f;
    if (dx * dx + dy * dy > r * r) return 0.0f;  // exact: the reference builds an empty polygon
  }
  float* vx = st;
  float* vy = st + kPolyCap * 64;
  float* va = st + 2 * kPolyCap * 64;
  int cnt = 0;
  float sx = 0.f, sy = 0.f;
  // A 17th vertex (16 crossings + 8 contained corners are possible for degenerate / garbage boxes only) is dropped:
  // the reference writes past `cross_points[16]` there (undefined behaviour, nothing to match); here the slot would
  // alias the next array or another wave's polygon, so the append is bounded.
  auto push = [&](float x, float y) {
    if (cnt < kPolyCap) {
      sx = sx + x;
      sy = sy + y;
      vx[cnt * 64] = x;
      vy[cnt * 64] = y;
      ++cnt;
    }
  };
  for (int i = 0; i < 4; ++i) {
    const Pt a0 = a.c[i], a1 = a.c[(i + 1) & 3];
    for (int j = 0; j < 4; ++j) {
      Pt hit;
      if (seg_hit(a1, a0, b.c[(j + 1) & 3], b.c[j], hit)) push(hit.x, hit.y);
    }
  }
  for (int k = 0; k < 4; ++k) {  // :184-195
    if (inside(a, b.c[k])) push(b.c[k].x, b.c[k].y);
    if (inside(b, a.c[k])) push(a.c[k].x, a.c[k].y);
  }
  if (cnt == 0) return 0.0f;
  sx /= cnt;  // :197-198
  sy /= cnt;
  for (int k = 0; k < cnt; ++k) va[k * 64] = atan2_rn(vy[k * 64] - sy, vx[k * 64] - sx);  // point_cmp :129-132
  float area = 0.f;
  if (cnt <= 8) {
    // two convex quadrilaterals in general position meet in at most eight vertices: sort and sum in registers.  The
    // fixed eight-element bubble network does the reference's compare-swaps (:201-210) in the reference's order; the
    // extra ones only ever look at an element that is already in its final place or at a +inf pad, and never swap.
    float px[8], py[8], an[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool in = k < cnt;
      px[k] = in ? vx[k * 64] : 0.f;
      py[k] = in ? vy[k * 64] : 0.f;
      an[k] = in ? va[k * 64] : INFINITY;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
      for (int i = 0; i < 7 - j; ++i) {
        const bool sw = an[i] > an[i + 1];
        const float tx = px[i], ty = py[i], ta = an[i];
        px[i] = sw ? px[i + 1] : tx;
        py[i] = sw ? py[i + 1] : ty;
        an[i] = sw ? an[i + 1] : ta;
        px[i + 1] = sw ? tx : px[i + 1];
        py[i + 1] = sw ? ty : py[i + 1];
        an[i + 1] = sw ? ta : an[i + 1];
      }
#pragma unroll
    for (int k = 0; k < 7; ++k) {  // :213-217
      if (k < cnt - 1) {
        const float ux = px[k] - px[0], uy = py[k] - py[0];
        const float wx = px[k + 1] - px[0], wy = py[k + 1] - py[0];
        area += ux * wy - uy * wx;
      }
    }
  } else {
    for (int j = 0; j < cnt - 1; ++j)  // :201-210
      for (int i = 0; i < cnt - j - 1; ++i) {
        const float a0 = va[i * 64], a1 = va[(i + 1) * 64];
        if (a0 > a1) {
          const float x0 = vx[i * 64], y0 = vy[i * 64];
          vx[i * 64] = vx[(i + 1) * 64];
          vy[i * 64] = vy[(i + 1) * 64];
          va[i * 64] = a1;
          vx[(i + 1) * 64] = x0;
          vy[(i + 1) * 64] = y0;
          va[(i + 1) * 64] = a0;
        }
      }
    const float x0 = vx[0], y0 = vy[0];
    for (int k = 0; k < cnt - 1; ++k) {  // :213-217
      const float ux = vx[k * 64] - x0, uy = vy[k * 64] - y0;
      const float wx = vx[(k + 1) * 64] - x0, wy = vy[(k + 1) * 64] - y0;
      area += ux * wy - uy * wx;
    }
  }
  return fabsf(area) / 2.0f;  // :219 (the fp64 division by 2 is exact)
}

// iou_bev :222-229
__device__ __forceinline__ float iou_bev(const BoxPre& a, const BoxPre& b, float* __restrict__ st) {
  const float so = box_overlap(a, b, st);
  return so / fmaxf(a.area + b.area - so, kGeomEps);
}

// iou_normal, iou3d_nms_kernel.cu:365-378 (axis aligned, raw boxes)
__device__ __forceinline__ float iou_normal(const float* a, const float* b) {
  const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2);
  const float right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2);
  const float bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
  const float inter = w * h;
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, kGeomEps);
}

}  // namespace pd3
