// iou3d_nms ops for gfx950 (reference: paddle3d/ops/iou3d_nms/iou3d_nms_api.cpp:73-108,
// iou3d_nms.cpp:44-204, kernels iou3d_nms_kernel.cu:275-482).
#include "../../include/paddle3d_amd.h"
#include "common.hpp"
#include "nms_kernels.hpp"

namespace pd3 {

// boxes_overlap_kernel / boxes_iou_bev_kernel (iou3d_nms_kernel.cu:275-308): one thread per (a, b).
template <bool IOU>
__global__ __launch_bounds__(256) void pairwise_kernel(const float* __restrict__ boxes_a, int num_a,
                                                       const float* __restrict__ boxes_b, int num_b,
                                                       float* __restrict__ ans) {
  const int b_idx = blockIdx.x * 16 + (threadIdx.x & 15);
  const int a_idx = blockIdx.y * 16 + (threadIdx.x >> 4);
  __shared__ BoxPre pa[16], pb[16];
  __shared__ float poly_s[4 * kPolyWaveFloats];
  float* st = poly_s + wave_id() * kPolyWaveFloats + lane_id();
  if (threadIdx.x < 16) {
    const int a = blockIdx.y * 16 + threadIdx.x;
    if (a < num_a) pa[threadIdx.x] = box_prepare(boxes_a + (int64_t)a * 7);
  } else if (threadIdx.x < 32) {
    const int b = blockIdx.x * 16 + threadIdx.x - 16;
    if (b < num_b) pb[threadIdx.x - 16] = box_prepare(boxes_b + (int64_t)b * 7);
  }
  __syncthreads();
  if (a_idx >= num_a || b_idx >= num_b) return;
  const BoxPre& A = pa[threadIdx.x >> 4];
  const BoxPre& B = pb[threadIdx.x & 15];
  ans[(int64_t)a_idx * num_b + b_idx] = IOU ? iou_bev(A, B, st) : box_overlap(A, B, st);
}

template <bool NORMAL>
static int run_nms(const float* boxes, int n, float thresh, int32_t* keep, int32_t* num_to_keep,
                   void* workspace, size_t workspace_bytes, void* stream) {
  if (n < 0 || !keep || !num_to_keep) return PD3_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_to_keep, 0, sizeof(int32_t), s);
    return e == hipSuccess ? 0 : (int)e;
  }
  if (!boxes || !workspace) return PD3_EINVAL;
  const int cb = (n + 63) / 64;
  if (cb > kNmsMaxWords) return PD3_EUNSUPPORTED;
  if (workspace_bytes < pd3_nms_workspace(n)) return PD3_EWORKSPACE;
  unsigned long long* mask = static_cast<unsigned long long*>(workspace);
  dim3 grid(cb, cb, 1);
  nms_mask_kernel<NORMAL><<<grid, 64, 0, s>>>(boxes, nullptr, n, n, cb, thresh, mask);
  {
    const size_t lds = nms_sweep_lds(n);
    if (lds > 48 * 1024) {
      hipError_t e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(nms_sweep_kernel), (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    nms_sweep_kernel<<<1, kNmsSweepThreads, lds, s>>>(mask, nullptr, n, n, cb, keep, num_to_keep);
  }
  return launch_status();
}

}  // namespace pd3

using namespace pd3;

extern "C" size_t pd3_nms_workspace(int num_boxes) {
  if (num_boxes <= 0) return 256;
  const size_t cb = ((size_t)num_boxes + 63) / 64;
  return align_up((size_t)num_boxes * cb * sizeof(unsigned long long), 256);
}

extern "C" int pd3_nms_bev(const float* boxes, int num_boxes, float thresh, int32_t* keep,
                           int32_t* num_to_keep, void* workspace, size_t workspace_bytes,
                           void* stream) {
  return run_nms<false>(boxes, num_boxes, thresh, keep, num_to_keep, workspace, workspace_bytes, stream);
}

extern "C" int pd3_nms_normal(const float* boxes, int num_boxes, float thresh, int32_t* keep,
                              int32_t* num_to_keep, void* workspace, size_t workspace_bytes,
                              void* stream) {
  return run_nms<true>(boxes, num_boxes, thresh, keep, num_to_keep, workspace, workspace_bytes, stream);
}

template <bool IOU>
static int run_pairwise(const float* a, int na, const float* b, int nb, float* ans, void* stream) {
  if (na < 0 || nb < 0) return PD3_EINVAL;
  if (na == 0 || nb == 0) return 0;
  if (!a || !b || !ans) return PD3_EINVAL;
  dim3 grid((nb + 15) / 16, (na + 15) / 16);
  pairwise_kernel<IOU><<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(a, na, b, nb, ans);
  return launch_status();
}

extern "C" int pd3_boxes_iou_bev(const float* boxes_a, int num_a, const float* boxes_b, int num_b,
                                 float* ans_iou, void* stream) {
  return run_pairwise<true>(boxes_a, num_a, boxes_b, num_b, ans_iou, stream);
}

extern "C" int pd3_boxes_overlap_bev(const float* boxes_a, int num_a, const float* boxes_b,
                                     int num_b, float* ans_overlap, void* stream) {
  return run_pairwise<false>(boxes_a, num_a, boxes_b, num_b, ans_overlap, stream);
}

// Diagnostic entry point: the device's sinf / cosf / expf / atanf / atan2f (libm_exact.hpp) over an array, so that
// the tests can hold them to glibc bit for bit on the GPU too.  op: 0 sinf, 1 cosf, 2 expf, 3 atanf, 4 atan2f(x, y).
namespace pd3 {
static __global__ void libm_eval_kernel(int op, const float* __restrict__ x, const float* __restrict__ y,
                                        float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r;
  switch (op) {
    case 0: r = lm::sinf(x[i]); break;
    case 1: r = lm::cosf(x[i]); break;
    case 2: r = lm::expf(x[i]); break;
    case 3: r = lm::atanf(x[i]); break;
    default: r = lm::atan2f(x[i], y[i]); break;
  }
  out[i] = r;
}
}  // namespace pd3

extern "C" int pd3_libm_eval(int op, const float* x, const float* y, float* out, int64_t n, void* stream) {
  if (op < 0 || op > 4 || n < 0 || (n > 0 && (!x || !out || (op == 4 && !y)))) return PD3_EINVAL;
  if (n == 0) return 0;
  libm_eval_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(op, x, y, out, n);
  return launch_status();
}
