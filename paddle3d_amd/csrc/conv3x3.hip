// Dense 3x3 / stride 1 / pad 1 convolution + bias + ReLU on the fp32 matrix cores, NCHW in and out.
// (reference layers: the SECOND backbone and CenterHead convolutions, paddle3d/models/backbones/
//  second_backbone.py:72-120 and detection/centerpoint/center_head.py:43-220 -- cuDNN convs in the reference,
//  89 % of the dense graph's 127 GFLOP per nuScenes scene.)
//
// Implicit GEMM, D[co][pixel] = sum_k W[co][k] * X[k][pixel], k = (ci, ky, kx), on
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 157 TFLOP/s peak -- there is no TF32 on gfx950).
// Workgroup tile: 64 output channels x (R rows x WT columns) pixels of one image, R * WT = 128 or 256; four
// waves, wave w owns NB = R*WT/128 blocks of 32 pixels for both 32-channel row blocks (2*NB accumulators of
// 16 registers).  K is walked 8 input channels (72 taps) at a time, as 36 MFMA steps (channel pair cp, tap):
// the two K lanes of the instruction (lane >> 5) take channels 2cp and 2cp+1 of the SAME tap, so every LDS
// address of the step is a per-lane base plus a compile-time constant -- the unrolled loop is ds_read with
// immediate offsets and MFMA only, no integer VALU work.
//   X chunk  -> LDS as [8][R+2][WT+8] (staged with aligned float4 loads from column x0-4, zero outside the
//               image): for tap (ky,kx) the 32 pixels of a block are consecutive floats => conflict-free
//               ds_read_b32 for the B operand;
//   W chunk  -> LDS as [36 steps][2 K lanes][64], pre-packed on the host in exactly this order so the copy
//               is linear and the A operand (lane = output channel) is conflict-free.
// Software pipeline: chunk c+1 travels global -> registers while the matrix cores work on chunk c out of LDS
// buffer c & 1; the registers are parked in the other buffer after the MFMA loop (one barrier per chunk).
// The epilogue adds the bias, applies ReLU and writes 128-byte row segments.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"


namespace pd3 {

typedef float cv_f32x16 __attribute__((ext_vector_type(16)));
typedef float cv_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kCvCo = 64;  // output channels per tile
constexpr int kCvCi = 8;   // input channels per K chunk
constexpr int kCvK = kCvCi * 9;

// XB = number of X buffers: 2 (one barrier per trip), or 1 for the 256-pixel stride-2 tile whose staged input
// (42 KB) would not leave room for two workgroups per CU if doubled -- one more barrier per trip instead.
// FUSED: the input map is never materialised -- `x` is the pillar feature matrix [pillars, cin] and `imap` the
// inverse map cell -> pillar row (-1: empty) of PointPillarsScatter (pillar_scatter.py:57-93); the staging of a
// chunk gathers the 8 channels of the tile's occupied cells (32 contiguous bytes of a pillar's row each) straight
// into the LDS image, whose empty cells were zeroed once.  A nuScenes canvas is 11 % occupied: the first backbone
// convolution stages less than it did from the dense canvas, and the 67 MB per frame the scatter wrote are gone.
template <int R, int WT, int S, int XB, bool FUSED = false>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ wp,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ out, int cin, int cout,
                                                           int h, int w, int wi, int wv, int relu, int ptiles,
                                                           const int* __restrict__ imap) {
  // h, w: OUTPUT rows and row pitch; the input map has S*h rows of pitch wi, S = stride (1 or 2), padding 1.
  // Output columns >= wv (the valid width) are written as zeros: a map whose width is not a multiple of 4 lives
  // in rows padded with zeros, which the next layer reads as its own zero padding.
  static_assert(R * WT == 128 || R * WT == 256, "tile must hold 128 or 256 pixels");
  constexpr int NB = R * WT / 128;           // 32-pixel blocks per wave
  constexpr int XR = (R - 1) * S + 3, XW = S * WT + 8;  // staged input columns S*x0-4 .. S*x0+S*WT+3
  constexpr int XQ = XW / 4;                 // float4 per staged row
  constexpr int XPL = XR * XW;               // floats per staged channel plane
  constexpr int XN4 = kCvCi * XR * XQ;       // float4 of one X chunk
  constexpr int XPT = (XN4 + 255) / 256;
  constexpr int WN4 = kCvK * kCvCo / 4;      // 1152 float4 of one W chunk
  constexpr int WPT = (WN4 + 255) / 256;
  constexpr int XSZ = kCvCi * XPL;           // floats per X buffer
  constexpr int WSZ = kCvK * kCvCo;          // floats per W buffer
  extern __shared__ __attribute__((aligned(16))) float cv_smem[];  // X[XB][XSZ] then W[2][WSZ]
  const int lane = lane_id(), wave = wave_id();
  const int tiles_x = (w + WT - 1) / WT, tiles_y = (h + R - 1) / R;  // border tiles are partial: masked stores
  // XCD-aware tile order (workgroups are dealt round-robin over the 8 XCDs): pixel tile pt lives on XCD
  // pt % 8 and its channel tiles follow each other there -> the input tile is fetched from HBM once per XCD
  const int nct = cout / kCvCo;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ct = slot % nct, pt = (slot / nct) * 8 + xcd;
  if (pt >= ptiles) return;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int y0 = ty * R, x0 = tx * WT;
  const int kk = lane >> 5;
  // per-lane LDS bases (floats): B operand of pixel block t, A operand of channel block 0
  int xb[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const int pj = (wave * NB + t) * 32 + (lane & 31);
    xb[t] = kk * XPL + (pj / WT) * S * XW + (pj % WT) * S + 3;
  }
  const int wb = XB * XSZ + kk * kCvCo + (lane & 31);
  cv_f32x16 acc[2][NB];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
  const int chunks = cin / kCvCi;
  const int64_t plane = (int64_t)h * w;           // output plane
  const int hi = S * h;
  const int64_t iplane = (int64_t)hi * wi;        // input plane
  const float* xin = x + (int64_t)n * cin * iplane;
  // staging pattern, identical for every chunk: float4 e of the LDS image <- global offset inside the chunk
  // (clamped to a valid address; `live` bit i says whether the value or zero is kept).  Threads past the end
  // of the image repeat its last float4 (same value to the same address).
  int gofs[XPT], ldst[XPT];
  unsigned live = 0;
  // FUSED: thread t owns cells t, t + 256, ... of the staged [XR][XW] window; cpid = pillar row of the cell or -1
  constexpr int NCELL = XR * XW, CPT = (NCELL + 255) / 256;
  int cpid[FUSED ? CPT : 1], cdst[FUSED ? CPT : 1];
  if (FUSED) {
    const int* im = imap + (int64_t)n * iplane;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int e = (int)threadIdx.x + i * 256;
      const int r = e / XW, c = e - r * XW;
      const int gy = S * y0 - 1 + r, gx = S * x0 - 4 + c;
      const bool ok = e < NCELL && gy >= 0 && gy < hi && gx >= 0 && gx < wi;
      cpid[i] = ok ? im[(int64_t)gy * wi + gx] : -1;
      cdst[i] = min(e, NCELL - 1);
    }
    for (int e = threadIdx.x; e < XB * XSZ / 4; e += 256)
      reinterpret_cast<cv_f32x4*>(cv_smem)[e] = (cv_f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int e = min((int)threadIdx.x + i * 256, XN4 - 1);
    const int ci = e / (XR * XQ), rem = e - ci * (XR * XQ);
    const int r = rem / XQ, c4 = rem - r * XQ;
    const int gy = S * y0 - 1 + r, gx = S * x0 - 4 + c4 * 4;  // a float4 is entirely inside or outside (w % 4 == 0)
    const bool ok = gy >= 0 && gy < hi && gx >= 0 && gx < wi;
    gofs[i] = ok ? (int)(ci * iplane + (int64_t)gy * wi + gx) : 0;
    live |= ok ? (1u << i) : 0u;
    ldst[i] = e * 4;
  }
  int wofs[WPT];
#pragma unroll
  for (int i = 0; i < WPT; ++i) wofs[i] = min((int)threadIdx.x + i * 256, WN4 - 1);
  const cv_f32x4* wsrc = reinterpret_cast<const cv_f32x4*>(wp) + (int64_t)ct * chunks * WN4;
  cv_f32x4 xr[FUSED ? 2 * CPT : XPT], wr[WPT];

#define CV_FETCH(cc)                                                                     \
  {                                                                                      \
    if (FUSED) {                                                                         \
      _Pragma("unroll") for (int i = 0; i < CPT; ++i)                                    \
        if (cpid[i] >= 0) {                                                              \
          const float* f_ = x + (int64_t)cpid[i] * cin + (cc) * kCvCi;                   \
          xr[2 * i] = *reinterpret_cast<const cv_f32x4*>(f_);                            \
          xr[2 * i + 1] = *reinterpret_cast<const cv_f32x4*>(f_ + 4);                    \
        }                                                                                \
    } else {                                                                             \
      const float* xc_ = xin + (int64_t)(cc) * kCvCi * iplane;                           \
      _Pragma("unroll") for (int i = 0; i < XPT; ++i)                                    \
          xr[i] = *reinterpret_cast<const cv_f32x4*>(xc_ + gofs[i]);                     \
    }                                                                                    \
    const cv_f32x4* wc_ = wsrc + (int64_t)(cc) * WN4;                                      \
    _Pragma("unroll") for (int i = 0; i < WPT; ++i) wr[i] = wc_[wofs[i]];                \
  }
#define CV_STASH(buf)                                                                    \
  {                                                                                      \
    float* xd_ = cv_smem + (XB == 2 ? (buf) : 0) * XSZ;                                  \
    float* wd_ = cv_smem + XB * XSZ + (buf) * WSZ;                                       \
    if (FUSED) {                                                                         \
      _Pragma("unroll") for (int i = 0; i < CPT; ++i)                                    \
        if (cpid[i] >= 0) {                                                              \
          float* d_ = xd_ + cdst[i];                                                     \
          _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) {                             \
            d_[k_ * XPL] = xr[2 * i][k_];                                                \
            d_[(4 + k_) * XPL] = xr[2 * i + 1][k_];                                      \
          }                                                                              \
        }                                                                                \
    } else {                                                                             \
      _Pragma("unroll") for (int i = 0; i < XPT; ++i) {                                  \
        const bool on_ = (live >> i) & 1u;                                               \
        const cv_f32x4 z_ = {0.f, 0.f, 0.f, 0.f};                                        \
        *reinterpret_cast<cv_f32x4*>(xd_ + ldst[i]) = on_ ? xr[i] : z_;                  \
      }                                                                                  \
    }                                                                                    \
    _Pragma("unroll") for (int i = 0; i < WPT; ++i)                                      \
        *reinterpret_cast<cv_f32x4*>(wd_ + wofs[i] * 4) = wr[i];                         \
  }

  CV_FETCH(0)
  CV_STASH(0)
  __syncthreads();
  for (int cc = 0; cc < chunks; ++cc) {
    // the last trip re-fetches its own chunk into the idle buffer: no control flow around the pipeline
    const int nx = min(cc + 1, chunks - 1);
    CV_FETCH(nx)
    __builtin_amdgcn_sched_barrier(0);  // keep the loads in flight ahead of the MFMA block
    const float* Xs = cv_smem + (XB == 2 ? (cc & 1) : 0) * XSZ;
    const float* Ws = cv_smem + (cc & 1) * WSZ + wb;
    // 36 steps, operands read two steps ahead of their MFMAs through a ring of three register sets (the
    // scheduling fences pin that order: the compiler would otherwise issue each read right before its use)
    float a0_[3], a1_[3], b_[3][NB];
#define CV_OPERANDS(s_, slot_)                                                           \
  {                                                                                      \
    const int cp_ = (s_) / 9, tap_ = (s_) % 9;                                           \
    const int xo_ = 2 * cp_ * XPL + (tap_ / 3) * XW + (tap_ % 3);                        \
    a0_[slot_] = Ws[(s_) * 2 * kCvCo];                                                   \
    a1_[slot_] = Ws[(s_) * 2 * kCvCo + 32];                                              \
    _Pragma("unroll") for (int t = 0; t < NB; ++t) b_[slot_][t] = Xs[xb[t] + xo_];       \
  }
    CV_OPERANDS(0, 0)
    CV_OPERANDS(1, 1)
#pragma unroll
    for (int s = 0; s < kCvK / 2; ++s) {
      if (s + 2 < kCvK / 2) CV_OPERANDS(s + 2, (s + 2) % 3)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_[s % 3], b_[s % 3][t], acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_[s % 3], b_[s % 3][t], acc[1][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef CV_OPERANDS
    __builtin_amdgcn_sched_barrier(0);
    if (XB == 1) __syncthreads();  // the single X buffer is still being read
    CV_STASH((cc + 1) & 1)
    __syncthreads();
  }
#undef CV_FETCH
#undef CV_STASH
  // epilogue: D layout col = lane & 31 (pixel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float bv[2][16];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) bv[m][reg] = 0.f;
  if (bias) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        bv[m][reg] = bias[ct * kCvCo + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk];
  }
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const int pj = (wave * NB + t) * 32 + (lane & 31);
    if (y0 + pj / WT >= h || x0 + pj % WT >= w) continue;  // pixel of a partial border tile
    float* obase = out + (int64_t)n * cout * plane + (int64_t)(y0 + pj / WT) * w + x0 + (pj % WT);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int co = ct * kCvCo + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
        float v = acc[m][t][reg] + bv[m][reg];
        if (relu) v = fmaxf(v, 0.f);
        obase[(int64_t)co * plane] = (x0 + pj % WT < wv) ? v : 0.f;
      }
  }
}

template <int R, int WT, int S, int XB = 2, bool FUSED = false>
static int launch_conv3x3(int64_t tiles, hipStream_t s, const float* x, const float* wp, const float* bias,
                          float* out, int cin, int cout, int h, int w, int wi, int wv, int relu,
                          const int* imap = nullptr) {
  constexpr size_t lds = (size_t)(XB * kCvCi * ((R - 1) * S + 3) * (S * WT + 8) + 2 * kCvK * kCvCo) * sizeof(float);
  // raise the dynamic-LDS cap (per device and per instantiation: set on every launch, it is a host-side table write)
  hipError_t e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(conv3x3_mfma_kernel<R, WT, S, XB, FUSED>), (int)lds);
  if (e != hipSuccess) return (int)e;
  const int64_t nwg = (tiles + 7) / 8 * 8 * (cout / kCvCo);
  if (nwg >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  conv3x3_mfma_kernel<R, WT, S, XB, FUSED><<<(unsigned)nwg, 256, lds, s>>>(x, wp, bias, out, cin, cout, h, w, wi, wv,
                                                                           relu, (int)tiles, imap);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// Grouped 3x3 / stride 1 / pad 1 convolution with very few output channels per group: the 36 final
// SeparateHead convolutions 64 -> {1,2,3} of CenterHead run as one launch on the [B, 36*64, H, W] first-stage
// map (center_head.py:99-118).  1.4 GFLOP but 1.2 GB of input per 8 scenes: HBM-bound, so plain VALU FMAs
// fed from LDS.  Workgroup = (image, group, 8 x 128 pixel tile); thread = 4 adjacent pixels x CO channels;
// the weights of the group are wave-uniform and come in through scalar loads.
constexpr int kGcCi = 4;              // input channels staged per trip
constexpr int kGcR = 8, kGcW = 128;   // pixel tile
constexpr int kGcXR = kGcR + 2, kGcXW = kGcW + 8;

template <int CO>
__global__ __launch_bounds__(256) void grouped_conv3x3_small_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ wg,
                                                                    const float* __restrict__ bias,
                                                                    float* __restrict__ out, int groups,
                                                                    int cg, int h, int w, int out_groups,
                                                                    int out_group0) {
  __shared__ __attribute__((aligned(16))) float Xs[kGcCi * kGcXR * kGcXW];
  const int tiles_x = (w + kGcW - 1) / kGcW, tiles_y = (h + kGcR - 1) / kGcR;
  const int pt = blockIdx.x;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int g = blockIdx.y;
  const int y0 = ty * kGcR, x0 = tx * kGcW;
  const int tr = threadIdx.x >> 5, tc = (threadIdx.x & 31) * 4;  // pixel row, first pixel column of the thread
  const int64_t plane = (int64_t)h * w;
  const float* xin = x + ((int64_t)n * groups + g) * cg * plane;
  const float* wgp = wg + (int64_t)g * cg * CO * 9;  // [cg][CO][9]
  float acc[CO][4];
#pragma unroll
  for (int c = 0; c < CO; ++c)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[c][p] = 0.f;
  constexpr int XQ = kGcXW / 4, XN4 = kGcCi * kGcXR * XQ;  // 1360 float4 per trip
  constexpr int XPT = (XN4 + 255) / 256;                   // 6 per thread
  // staging pattern (identical for every trip): element -> offset in a channel-plane quadruple, zero outside
  int gofs[XPT];
  unsigned live = 0;
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int e = min((int)threadIdx.x + i * 256, XN4 - 1);
    const int ci = e / (kGcXR * XQ), rem = e - ci * (kGcXR * XQ);
    const int r = rem / XQ, c4 = rem - r * XQ;
    const int gy = y0 - 1 + r, gx = x0 - 4 + c4 * 4;
    const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
    gofs[i] = ok ? (int)(ci * plane + (int64_t)gy * w + gx) : 0;
    live |= ok ? 1u << i : 0u;
  }
  cv_f32x4 xr[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) xr[i] = *reinterpret_cast<const cv_f32x4*>(xin + gofs[i]);
  for (int c0 = 0; c0 < cg; c0 += kGcCi) {
    __syncthreads();  // the previous trip's reads of Xs are done
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int e = min((int)threadIdx.x + i * 256, XN4 - 1);
      const cv_f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<cv_f32x4*>(Xs + e * 4) = ((live >> i) & 1u) ? xr[i] : z;
    }
    if (c0 + kGcCi < cg) {  // the next trip's loads fly while this one is multiplied
      const float* xn = xin + (int64_t)(c0 + kGcCi) * plane;
#pragma unroll
      for (int i = 0; i < XPT; ++i) xr[i] = *reinterpret_cast<const cv_f32x4*>(xn + gofs[i]);
    }
    __syncthreads();
#pragma unroll
    for (int ci = 0; ci < kGcCi; ++ci) {
      const float* wc = wgp + (int64_t)(c0 + ci) * CO * 9;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        // input columns tc-1 .. tc+4 of row tr+ky live at LDS columns tc+3 .. tc+8
        const float* row = Xs + (ci * kGcXR + tr + ky) * kGcXW + tc;
        const cv_f32x4 mid = *reinterpret_cast<const cv_f32x4*>(row + 4);
        float in[6];
        in[1] = mid[0];
        in[2] = mid[1];
        in[3] = mid[2];
        in[4] = mid[3];
        in[0] = row[3];
        in[5] = row[8];
#pragma unroll
        for (int c = 0; c < CO; ++c)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float wv = wc[c * 9 + ky * 3 + kx];
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[c][p] = __builtin_fmaf(wv, in[p + kx], acc[c][p]);
          }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    const float b = bias ? bias[g * CO + c] : 0.f;
    cv_f32x4 v = {acc[c][0] + b, acc[c][1] + b, acc[c][2] + b, acc[c][3] + b};
    if (y0 + tr < h && x0 + tc < w)  // partial tiles at the border (w % 4 == 0: a quad is in or out)
      *reinterpret_cast<cv_f32x4*>(out + ((int64_t)n * out_groups * CO + (out_group0 + g) * CO + c) * plane +
                                   (int64_t)(y0 + tr) * w + x0 + tc) = v;
  }
}

}  // namespace pd3

using namespace pd3;

extern "C" int pd3_conv3x3_bias_relu(const float* x, const float* w_packed, const float* bias,
                                     int batch, int cin, int cout, int h, int w, int w_valid, int stride, int relu,
                                     float* out, int out_w, void* stream) {
  if (!x || !w_packed || !out || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || w_valid <= 0 ||
      w_valid > w || out_w <= 0)
    return PD3_EINVAL;
  if (stride != 1 && stride != 2) return PD3_EUNSUPPORTED;
  if (cin % kCvCi != 0 || cout % kCvCo != 0 || h % stride != 0 || w_valid % stride != 0) return PD3_EUNSUPPORTED;
  if (w % 4 != 0 || out_w % 4 != 0 || out_w < w_valid / stride) return PD3_EUNSUPPORTED;  // rows: aligned float4
  if (reinterpret_cast<uintptr_t>(w_packed) % 16 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0)
    return PD3_EINVAL;
  if ((int64_t)cin * h * w >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;  // 32-bit staging offsets
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int ho = h / stride, wo = out_w, wv = w_valid / stride;  // (h + 2 - 3) / stride + 1 for even h
#define PD3_CV(R, WT, S)                                                                                     \
  launch_conv3x3<R, WT, S>((int64_t)batch * ceil_div(ho, R) * ceil_div(wo, WT), s, x, w_packed, bias, out, cin, \
                           cout, ho, wo, w, wv, relu)
  if (stride == 1) {
    if (wo % 128 == 0 && ho % 2 == 0) return PD3_CV(2, 128, 1);
    if (wo % 64 == 0 && ho % 4 == 0) return PD3_CV(4, 64, 1);
    if (wo % 128 == 0) return PD3_CV(1, 128, 1);
    if (wo % 64 == 0 && ho % 2 == 0) return PD3_CV(2, 64, 1);
    return PD3_CV(4, 32, 1);  // any size: partial tiles at the right / bottom border
  } else {  // the staged input tile is 4x larger: a 256-pixel tile with ONE X buffer, or 128-pixel tiles
    if (wo % 128 == 0 && ho % 2 == 0)
      return launch_conv3x3<2, 128, 2, 1>((int64_t)batch * ho * wo / 256, s, x, w_packed, bias, out, cin, cout, ho, wo,
                                          w, wv, relu);
    if (wo % 64 == 0 && ho % 2 == 0) return PD3_CV(2, 64, 2);
    return PD3_CV(4, 32, 2);  // any size
  }
#undef PD3_CV
}

// PointPillarsScatter fused into a stride-2 3x3 convolution (the first SECOND block of the pillar models): see FUSED
// above.  feats [pillars, cin], imap [batch, ny * nx] (pd3_pointpillars_inverse_map), out [batch, cout, ny/2, out_w].
extern "C" int pd3_scatter_conv3x3_bias_relu(const float* feats, const int32_t* imap, const float* w_packed,
                                             const float* bias, int batch, int cin, int cout, int ny, int nx,
                                             int stride, int relu, float* out, int out_w, void* stream) {
  if (!feats || !imap || !w_packed || !out || batch <= 0 || cin <= 0 || cout <= 0 || ny <= 0 || nx <= 0 || out_w <= 0)
    return PD3_EINVAL;
  if (stride != 2) return PD3_EUNSUPPORTED;
  if (cin % kCvCi != 0 || cout % kCvCo != 0 || ny % 2 != 0 || nx % 2 != 0 || out_w % 4 != 0 || out_w < nx / 2)
    return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(w_packed) % 16 != 0 || reinterpret_cast<uintptr_t>(feats) % 16 != 0 || cin % 4 != 0)
    return PD3_EINVAL;
  if ((int64_t)ny * nx >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int ho = ny / 2, wo = out_w, wv = nx / 2;
  const int* im = reinterpret_cast<const int*>(imap);
  if (wo % 128 == 0 && ho % 2 == 0)
    return launch_conv3x3<2, 128, 2, 1, true>((int64_t)batch * ho * wo / 256, s, feats, w_packed, bias, out, cin, cout, ho,
                                              wo, nx, wv, relu, im);
  if (wo % 64 == 0 && ho % 2 == 0)
    return launch_conv3x3<2, 64, 2, 2, true>((int64_t)batch * ceil_div(ho, 2) * ceil_div(wo, 64), s, feats, w_packed, bias,
                                             out, cin, cout, ho, wo, nx, wv, relu, im);
  return launch_conv3x3<4, 32, 2, 2, true>((int64_t)batch * ceil_div(ho, 4) * ceil_div(wo, 32), s, feats, w_packed, bias,
                                           out, cin, cout, ho, wo, nx, wv, relu, im);
}

static int grouped_small_launch(const float* x, const float* w_grouped, const float* bias, int batch, int groups,
                                int cin_per_group, int cout_per_group, int h, int w, float* out, int out_groups,
                                int out_group0, void* stream) {
  if (!x || !w_grouped || !out || batch <= 0 || groups <= 0 || cin_per_group <= 0 || h <= 0 || w <= 0 ||
      out_group0 < 0 || out_group0 + groups > out_groups)
    return PD3_EINVAL;
  if (cout_per_group < 1 || cout_per_group > 4 || cin_per_group % kGcCi != 0 || w % 4 != 0)
    return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0) return PD3_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((int64_t)batch * ceil_div(h, kGcR) * ceil_div(w, kGcW)), (unsigned)groups);
#define PD3_GC(CO)                                                                                                  \
  grouped_conv3x3_small_kernel<CO><<<grid, 256, 0, s>>>(x, w_grouped, bias, out, groups, cin_per_group, h, w,        \
                                                        out_groups, out_group0)
  switch (cout_per_group) {
    case 1: PD3_GC(1); break;
    case 2: PD3_GC(2); break;
    case 3: PD3_GC(3); break;
    default: PD3_GC(4); break;
  }
#undef PD3_GC
  return launch_status();
}

extern "C" int pd3_grouped_conv3x3_small(const float* x, const float* w_grouped, const float* bias, int batch,
                                         int groups, int cin_per_group, int cout_per_group, int h, int w,
                                         float* out, void* stream) {
  return grouped_small_launch(x, w_grouped, bias, batch, groups, cin_per_group, cout_per_group, h, w, out, groups, 0,
                              stream);
}

extern "C" int pd3_grouped_conv3x3_small_slice(const float* x, const float* w_grouped, const float* bias, int batch,
                                               int groups, int cin_per_group, int cout_per_group, int h, int w,
                                               float* out, int out_groups, int out_group0, void* stream) {
  return grouped_small_launch(x, w_grouped, bias, batch, groups, cin_per_group, cout_per_group, h, w, out, out_groups,
                              out_group0, stream);
}
