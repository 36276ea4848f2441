// Dense 3x3 / stride 1 / pad 1 convolution + bias + ReLU on the fp32 matrix cores, NCHW in and out.
// (reference layers: the SECOND backbone and CenterHead convolutions, paddle3d/models/backbones/
//  second_backbone.py:72-120 and detection/centerpoint/center_head.py:43-220 -- cuDNN convs in the reference,
//  89 % of the dense graph's 127 GFLOP per nuScenes scene.)
//
// Implicit GEMM, D[co][pixel] = sum_k W[co][k] * X[k][pixel], k = (ci, ky, kx), on
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 157 TFLOP/s peak -- there is no TF32 on gfx950).
// Workgroup tile: 64 output channels x 128 pixels (R rows x WT columns of one image, R * WT = 128), four
// waves, wave w owns pixel columns [32w, 32w+32) for both 32-channel row blocks (2 accumulators of 16
// registers).  K is walked 8 input channels (72 taps) at a time:
//   X chunk  -> LDS as [8][R+2][WT+2] (halo included, zero outside the image): for tap (ky,kx) the 32 pixels
//               of a wave are consecutive floats => conflict-free ds_read_b32 for the B operand;
//   W chunk  -> LDS as [72][64], pre-packed on the host in exactly this order so the copy is linear and the
//               A operand (lane = output channel) is conflict-free.
// Several workgroups per CU (30 KB of LDS each) overlap one another's staging with MFMA issue; the
// epilogue adds the bias, applies ReLU and writes 128-byte row segments.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"

namespace pd3 {

typedef float cv_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kCvCo = 64;    // output channels per tile
constexpr int kCvPix = 128;  // pixels per tile
constexpr int kCvCi = 8;     // input channels per K chunk
constexpr int kCvK = kCvCi * 9;

template <int R, int WT>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ wp,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ out, int cin, int cout,
                                                           int h, int w, int relu) {
  static_assert(R * WT == kCvPix, "tile must hold 128 pixels");
  constexpr int XR = R + 2, XW = WT + 2;
  __shared__ float Xs[kCvCi * XR * XW];
  __shared__ __attribute__((aligned(16))) float Ws[kCvK * kCvCo];
  __shared__ int koff[kCvK];
  const int lane = lane_id(), wave = wave_id();
  const int tiles_x = w / WT, tiles_y = h / R;
  const int pt = blockIdx.x;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int ct = blockIdx.y;
  const int y0 = ty * R, x0 = tx * WT;
  if (threadIdx.x < kCvK) {
    const int cil = threadIdx.x / 9, tap = threadIdx.x % 9;
    koff[threadIdx.x] = cil * (XR * XW) + (tap / 3) * XW + (tap % 3);
  }
  const int pj = wave * 32 + (lane & 31);
  const int pr = pj / WT, px = pj % WT;
  const int pbase = pr * XW + px;
  const int kk = lane >> 5;
  cv_f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc0[i] = 0.f;
    acc1[i] = 0.f;
  }
  const int chunks = cin / kCvCi;
  const float* xin = x + (int64_t)n * cin * h * w;
  const float4* wsrc = reinterpret_cast<const float4*>(wp + (int64_t)ct * chunks * (kCvK * kCvCo));
  for (int cc = 0; cc < chunks; ++cc) {
    __syncthreads();  // previous chunk fully consumed (also publishes koff on the first trip)
    for (int e = threadIdx.x; e < kCvCi * XR * XW; e += 256) {
      const int ci = e / (XR * XW), rem = e - ci * (XR * XW);
      const int r = rem / XW, c = rem - r * XW;
      const int gy = y0 - 1 + r, gx = x0 - 1 + c;
      float v = 0.f;
      if (gy >= 0 && gy < h && gx >= 0 && gx < w) v = xin[((int64_t)(cc * kCvCi + ci) * h + gy) * w + gx];
      Xs[e] = v;
    }
    float4* wdst = reinterpret_cast<float4*>(Ws);
    for (int e = threadIdx.x; e < kCvK * kCvCo / 4; e += 256) wdst[e] = wsrc[(int64_t)cc * (kCvK * kCvCo / 4) + e];
    __syncthreads();
#pragma unroll 4
    for (int k2 = 0; k2 < kCvK / 2; ++k2) {
      const int k = 2 * k2 + kk;
      const float b = Xs[koff[k] + pbase];
      const float a0 = Ws[k * kCvCo + (lane & 31)];
      const float a1 = Ws[k * kCvCo + 32 + (lane & 31)];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
    }
  }
  // epilogue: D layout col = lane & 31 (pixel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* obase = out + (int64_t)n * cout * h * w + (int64_t)(y0 + pr) * w + x0 + px;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kk;
    {
      const int co = ct * kCvCo + row;
      float v = acc0[reg] + (bias ? bias[co] : 0.f);
      if (relu) v = fmaxf(v, 0.f);
      obase[(int64_t)co * h * w] = v;
    }
    {
      const int co = ct * kCvCo + 32 + row;
      float v = acc1[reg] + (bias ? bias[co] : 0.f);
      if (relu) v = fmaxf(v, 0.f);
      obase[(int64_t)co * h * w] = v;
    }
  }
}

}  // namespace pd3

using namespace pd3;

extern "C" int pd3_conv3x3_bias_relu(const float* x, const float* w_packed, const float* bias,
                                     int batch, int cin, int cout, int h, int w, int relu, float* out,
                                     void* stream) {
  if (!x || !w_packed || !out || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0)
    return PD3_EINVAL;
  if (cin % kCvCi != 0 || cout % kCvCo != 0) return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(w_packed) % 16 != 0) return PD3_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(0, cout / kCvCo);
  if (w % 128 == 0) {
    grid.x = (unsigned)((int64_t)batch * h * (w / 128));
    conv3x3_mfma_kernel<1, 128><<<grid, 256, 0, s>>>(x, w_packed, bias, out, cin, cout, h, w, relu);
  } else if (w % 64 == 0 && h % 2 == 0) {
    grid.x = (unsigned)((int64_t)batch * (h / 2) * (w / 64));
    conv3x3_mfma_kernel<2, 64><<<grid, 256, 0, s>>>(x, w_packed, bias, out, cin, cout, h, w, relu);
  } else if (w % 32 == 0 && h % 4 == 0) {
    grid.x = (unsigned)((int64_t)batch * (h / 4) * (w / 32));
    conv3x3_mfma_kernel<4, 32><<<grid, 256, 0, s>>>(x, w_packed, bias, out, cin, cout, h, w, relu);
  } else {
    return PD3_EUNSUPPORTED;
  }
  return launch_status();
}
