// 3x3 / stride 1 / pad 1 convolution + bias + ReLU by Winograd F(2x2, 3x3) on the fp32 matrix cores,
// NCHW in and out, everything fused in one kernel (no transformed tensors in HBM).
// (reference layers: the stride-1 convolutions of SecondBackbone, paddle3d/models/backbones/second_backbone.py:
//  72-120, and of CenterHead / SeparateHead, detection/centerpoint/center_head.py:43-220 -- 115 of the dense
//  graph's 127 GFLOP per nuScenes scene; cuDNN picks its own algorithm for them in the reference.)
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          per 2x2 output tile, d = its 4x4 input patch
// i.e. 16 independent GEMMs  M[xi][co][tile] = sum_ci U[xi][co][ci] * V[xi][ci][tile]  with 2.25x fewer
// multiplies than the direct form.  All of it stays fp32 (v_mfma_f32_16x16x4_f32 is an exact fp32 fma chain).
//
// Workgroup = 32 output channels x 64 tiles (4 tile rows x 16 tile columns = 8 x 32 output pixels) x all 16
// components; K walks 8 input channels per trip:
//   U trip   : pre-transformed on the host and packed [2 co blocks][8 ci][16 co][16 xi]; copied to LDS with
//              the component axis padded 16 -> 20 floats;
//   raw X    : [8 ci][10 rows][40 cols] (aligned float4 loads from column x0-4, zero outside the image);
//   V trip   : each thread transforms two (ci, tile) patches B^T d B out of the raw buffer and scatters the 16
//              components to LDS as [4 tile rows][8 ci][16 tile cols][16 xi + 4 pad] (four b128 writes);
//   MFMA     : wave w owns co block (w & 1) and tile rows 2(w >> 1), 2(w >> 1) + 1 for ALL 16 components
//              (32 accumulator quads): A = U (row = co), B = V (col = tile); a lane reads four components of
//              its (k, co) / (k, tile) element with one ds_read_b128 (stride 20 floats: conflict-free), every
//              address is base + constant.
//   epilogue : one lane holds all 16 components of its (co, tile) -> A^T M A in registers, + bias, ReLU,
//              float2 stores (16 lanes = one 128-byte row segment).
// The next trip's global loads are issued before the MFMA block and parked in LDS after it; two workgroups
// per CU (73 KB of LDS, 128 accumulator registers) overlap one's transform phase with the other's MFMAs.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"

namespace pd3 {

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));

constexpr int kWgCi = 8;                 // input channels per trip
constexpr int kWgCo = 32;                // output channels per workgroup
constexpr int kWgTR = 4, kWgTC = 16;     // tile rows / columns per workgroup (2x2 outputs each)
constexpr int kWgRawR = 2 * kWgTR + 2;   // 10 staged input rows
constexpr int kWgRawW = 2 * kWgTC + 8;   // 40 staged input columns: x0-4 .. x0+35
constexpr int kWgRawPl = kWgRawR * kWgRawW;
constexpr int kWgCs = 20;                                 // 16 components + 4 pad: conflict-free b128 access
constexpr int kWgUsz = kWgCo * kWgCi * kWgCs;             // 5120 floats  [2 co blocks][8 ci][16 co][20]
constexpr int kWgRawSz = kWgCi * kWgRawPl;                // 3200 floats
constexpr int kWgVsz = kWgTR * kWgCi * kWgTC * kWgCs;     // 10240 floats [4 tile rows][8 ci][16 tile cols][20]
constexpr int kWgUN4 = kWgCo * kWgCi * 16 / 4;             // 1024 float4 of U per trip in global memory (no pad)
constexpr int kWgUPT = kWgUN4 / 256;                      // 4 per thread
constexpr size_t kWgLds = (size_t)(kWgUsz + kWgRawSz + kWgVsz) * sizeof(float);  // 74240 B: two workgroups per CU
constexpr int kWgXN4 = kWgRawSz / 4;                      // 800 float4
constexpr int kWgXPT = (kWgXN4 + 255) / 256;              // 4 (the tail repeats element 799)

__global__ __launch_bounds__(256, 2) void conv3x3_winograd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ up,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ out, int cin, int cout,
                                                                  int h, int w, int relu, int ptiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Us = smem;
  float* Raw = smem + kWgUsz;
  float* Vs = smem + kWgUsz + kWgRawSz;
  const int lane = lane_id(), wave = wave_id();
  const int tiles_x = (w + 2 * kWgTC - 1) / (2 * kWgTC), tiles_y = (h + 2 * kWgTR - 1) / (2 * kWgTR);
  // XCD-aware tile order: the dispatcher deals consecutive workgroups round-robin over the 8 XCDs (private
  // L2s).  Pixel tile pt lives on XCD pt % 8 and its channel tiles follow each other there, so the input
  // patch is fetched from HBM once per XCD and each L2 only sees 1/8 of the image.
  const int nct = cout / kWgCo;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ct = slot % nct, pt = (slot / nct) * 8 + xcd;
  if (pt >= ptiles) return;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int y0 = ty * 2 * kWgTR, x0 = tx * 2 * kWgTC;
  const int chunks = cin / kWgCi;
  const int64_t plane = (int64_t)h * w;
  const float* xin = x + (int64_t)n * cin * plane;
  const wg_f32x4* usrc = reinterpret_cast<const wg_f32x4*>(up) + (int64_t)ct * chunks * kWgUN4;

  // staging pattern of the raw patch (identical for every trip)
  int gofs[kWgXPT], ldst[kWgXPT];
  unsigned live = 0;
#pragma unroll
  for (int i = 0; i < kWgXPT; ++i) {
    const int e = min((int)threadIdx.x + i * 256, kWgXN4 - 1);
    const int ci = e / (kWgRawR * (kWgRawW / 4)), rem = e - ci * (kWgRawR * (kWgRawW / 4));
    const int r = rem / (kWgRawW / 4), c4 = rem - r * (kWgRawW / 4);
    const int gy = y0 - 1 + r, gx = x0 - 4 + c4 * 4;  // a float4 is entirely inside or outside (w % 4 == 0)
    const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
    gofs[i] = ok ? (int)(ci * plane + (int64_t)gy * w + gx) : 0;
    live |= ok ? (1u << i) : 0u;
    ldst[i] = e * 4;
  }
  // transform assignment: two (ci, tile) patches per thread
  const int t16 = lane & 15;
  const int q = (lane >> 5) | (wave << 1);  // 0..7
  int rsrc[2], vdst[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = q + 8 * it;
    const int tb = idx & 3, ci = (idx >> 2) * 2 + ((lane >> 4) & 1);
    rsrc[it] = ci * kWgRawPl + (2 * tb) * kWgRawW + 2 * t16 + 3;
    vdst[it] = ((tb * kWgCi + ci) * kWgTC + t16) * kWgCs;
  }
  // MFMA operand bases
  const int cb = wave & 1, tb0 = 2 * (wave >> 1);
  const int abase = ((cb * kWgCi + (lane >> 4)) * 16 + (lane & 15)) * kWgCs;
  const int bbase = ((tb0 * kWgCi + (lane >> 4)) * kWgTC + (lane & 15)) * kWgCs;

  wg_f32x4 acc[16][2];
#pragma unroll
  for (int c = 0; c < 16; ++c)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[c][b] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};

  wg_f32x4 xr[kWgXPT], ur[kWgUPT];
  // U float4 e = threadIdx.x + 256 i of the trip: element e / 4 (16 components each), quarter e % 4
  const int udst = (threadIdx.x >> 2) * kWgCs + (threadIdx.x & 3) * 4;

#define WG_FETCH(cc)                                                                     \
  {                                                                                      \
    const float* xc_ = xin + (int64_t)(cc) * kWgCi * plane;                              \
    _Pragma("unroll") for (int i = 0; i < kWgXPT; ++i)                                   \
        xr[i] = *reinterpret_cast<const wg_f32x4*>(xc_ + gofs[i]);                       \
    const wg_f32x4* uc_ = usrc + (int64_t)(cc) * kWgUN4;                                 \
    _Pragma("unroll") for (int i = 0; i < kWgUPT; ++i) ur[i] = uc_[threadIdx.x + i * 256]; \
  }
#define WG_STASH()                                                                       \
  {                                                                                      \
    _Pragma("unroll") for (int i = 0; i < kWgXPT; ++i) {                                 \
      const bool on_ = (live >> i) & 1u;                                                 \
      const wg_f32x4 z_ = {0.f, 0.f, 0.f, 0.f};                                          \
      *reinterpret_cast<wg_f32x4*>(Raw + ldst[i]) = on_ ? xr[i] : z_;                    \
    }                                                                                    \
    _Pragma("unroll") for (int i = 0; i < kWgUPT; ++i)                                   \
        *reinterpret_cast<wg_f32x4*>(Us + udst + i * 64 * kWgCs) = ur[i];                \
  }
  // V = B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
#define WG_TRANSFORM()                                                                   \
  {                                                                                      \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                   \
      const float* d_ = Raw + rsrc[it];                                                  \
      float t_[4][4];                                                                    \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                    \
        const float d0 = d_[j], d1 = d_[kWgRawW + j], d2 = d_[2 * kWgRawW + j],          \
                    d3 = d_[3 * kWgRawW + j];                                            \
        t_[0][j] = d0 - d2;                                                              \
        t_[1][j] = d1 + d2;                                                              \
        t_[2][j] = d2 - d1;                                                              \
        t_[3][j] = d1 - d3;                                                              \
      }                                                                                  \
      float* v_ = Vs + vdst[it];                                                         \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                      \
          *reinterpret_cast<wg_f32x4*>(v_ + i * 4) =                                     \
              (wg_f32x4){t_[i][0] - t_[i][2], t_[i][1] + t_[i][2], t_[i][2] - t_[i][1],  \
                         t_[i][1] - t_[i][3]};                                           \
    }                                                                                    \
  }

  // 8 groups (k half, 4 components): three b128 reads feed eight MFMAs; the reads of group g+1 are issued
  // while group g runs (the scheduling fences keep the compiler from hoisting every read: registers)
#define WG_LOAD(g_, a_, b0_, b1_)                                                        \
  {                                                                                      \
    const int o_ = ((g_) >> 2) * (4 * 16 * kWgCs) + ((g_) & 3) * 4;                      \
    a_ = *reinterpret_cast<const wg_f32x4*>(Us + abase + o_);                            \
    b0_ = *reinterpret_cast<const wg_f32x4*>(Vs + bbase + o_);                           \
    b1_ = *reinterpret_cast<const wg_f32x4*>(Vs + bbase + o_ + kWgCi * kWgTC * kWgCs);   \
  }
#define WG_MFMA()                                                                        \
  {                                                                                      \
    wg_f32x4 a_, b0_, b1_;                                                               \
    WG_LOAD(0, a_, b0_, b1_)                                                             \
    _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                   \
      wg_f32x4 an_ = a_, b0n_ = b0_, b1n_ = b1_;                                         \
      if (g_ + 1 < 8) WG_LOAD(g_ + 1, an_, b0n_, b1n_)                                   \
      __builtin_amdgcn_sched_barrier(0);                                                 \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                 \
        const int c_ = (g_ & 3) * 4 + j_;                                                \
        acc[c_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[j_], b0_[j_], acc[c_][0], 0, 0, 0); \
        acc[c_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[j_], b1_[j_], acc[c_][1], 0, 0, 0); \
      }                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                 \
      a_ = an_;                                                                          \
      b0_ = b0n_;                                                                        \
      b1_ = b1n_;                                                                        \
    }                                                                                    \
  }

  WG_FETCH(0)
  WG_STASH()
  __syncthreads();
  WG_TRANSFORM()
  __syncthreads();
  // steady state (no conditionals, or the compiler sinks the prefetch behind the MFMA block); last trip peeled
  for (int cc = 0; cc + 1 < chunks; ++cc) {
    WG_FETCH(cc + 1)
    __builtin_amdgcn_sched_barrier(0);  // keep the loads in flight ahead of the MFMA block
    WG_MFMA()
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // every wave is done with U and V of this trip
    WG_STASH()
    __syncthreads();
    WG_TRANSFORM()
    __syncthreads();
  }
  WG_MFMA()
#undef WG_MFMA
#undef WG_LOAD
#undef WG_FETCH
#undef WG_STASH
#undef WG_TRANSFORM

  // epilogue: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]; lane: tile column lane & 15, channels 4 (lane >> 4) + r
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = 0.f;
  const int co0 = ct * kWgCo + cb * 16 + 4 * (lane >> 4);
  if (bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[co0 + r];
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int oy = y0 + 2 * (tb0 + b), ox = x0 + 2 * (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s[2][4];  // A^T M
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[0][j] = acc[j][b][r] + acc[4 + j][b][r] + acc[8 + j][b][r];
        s[1][j] = acc[4 + j][b][r] - acc[8 + j][b][r] - acc[12 + j][b][r];
      }
      float* o = out + ((int64_t)n * cout + co0 + r) * plane + (int64_t)oy * w + ox;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v0 = s[i][0] + s[i][1] + s[i][2] + bv[r];
        float v1 = s[i][1] - s[i][2] - s[i][3] + bv[r];
        if (relu) {
          v0 = fmaxf(v0, 0.f);
          v1 = fmaxf(v1, 0.f);
        }
        if (oy + i < h && ox < w)  // partial tiles at the bottom / right border (w is even: a pair is in or out)
          *reinterpret_cast<wg_f32x2*>(o + (int64_t)i * w) = (wg_f32x2){v0, v1};
      }
    }
  }
}

}  // namespace pd3

using namespace pd3;

extern "C" int pd3_conv3x3_winograd_bias_relu(const float* x, const float* u_packed, const float* bias,
                                              int batch, int cin, int cout, int h, int w, int relu, float* out,
                                              void* stream) {
  if (!x || !u_packed || !out || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return PD3_EINVAL;
  if (cin % kWgCi != 0 || cout % kWgCo != 0 || w % 4 != 0)  // float4 staging: a quad is inside or outside a row
    return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(u_packed) % 16 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out) % 8 != 0)
    return PD3_EINVAL;
  if ((int64_t)cin * h * w >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;  // 32-bit staging offsets
  const int64_t ptiles = (int64_t)batch * ceil_div(h, 2 * kWgTR) * ceil_div(w, 2 * kWgTC);
  const int64_t nwg = (ptiles + 7) / 8 * 8 * (cout / kWgCo);
  if (nwg >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  dim3 grid((unsigned)nwg);
  {  // dynamic-LDS cap: per device, so it is set on every launch (a host-side table write)
    hipError_t e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(conv3x3_winograd_kernel), (int)kWgLds);
    if (e != hipSuccess) return (int)e;
  }
  conv3x3_winograd_kernel<<<grid, 256, kWgLds, static_cast<hipStream_t>(stream)>>>(x, u_packed, bias, out, cin,
                                                                                   cout, h, w, relu, (int)ptiles);
  return launch_status();
}
