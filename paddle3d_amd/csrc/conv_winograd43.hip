// 3x3 / stride 1 / pad 1 convolution + bias + ReLU by Winograd F(4x4, 3x3) on the fp32 matrix cores, NCHW in
// and out, fully fused (same layers as conv_winograd.hip; reference: second_backbone.py:72-120 and
// center_head.py:43-220, cuDNN in the reference).
//
//   Y(4x4) = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A,   d = the 6x6 input patch of a 4x4 output tile
// 36 independent GEMMs M[xi] = U[xi] V[xi]: 4x fewer multiplies than the direct form (F(2x2,3x3): 2.25x) for
// 1.5x the transform work per pixel.  fp32 throughout; the transform constants (4, 5, 8, 1/24 ...) cost about one
// decimal digit against the direct form (measured ~1e-5 absolute for |y| ~ 1.5), far inside the 1e-3 contract.
//
// Workgroup = 32 output channels x 32 tiles (2 tile rows x 16 tile columns = 8 x 64 output pixels) x all 36
// components; K walks 4 input channels per trip = one K step of v_mfma_f32_16x16x4_f32:
//   U trip : pre-transformed on the host, packed [2 co blocks][4 ci][16 co][36 xi] -> linear copy to LDS;
//   raw X  : [4 ci][10 rows][72 cols], aligned float4 loads from column x0-4, zero outside the image;
//   V trip : a PAIR of lanes transforms one (channel, tile) patch: lane h takes columns 3h..3h+2 through the
//            row pass, the halves are swapped with one DPP quad_perm per value, lane h then runs the column
//            pass for rows 3h..3h+2 and writes its 18 components to LDS as [2 tile rows][4 ci][16 tile cols][36];
//   MFMA   : wave w owns co block (w & 1) and tile row (w >> 1) for all 36 components (36 accumulator quads);
//            the 36-float component axis is read with conflict-free ds_read_b128 (36 l mod 64 are 16 distinct
//            multiples of 4), 9 + 9 reads for 36 MFMAs, reads running two groups ahead of their MFMAs;
//   epilogue: one lane holds all 36 components of its (co, tile): A^T M A in registers, + bias, ReLU, one float4
//            store per output row (16 lanes = 256 contiguous bytes).
// Two workgroups per CU (47 KB LDS); the next trip's global loads are issued before the MFMA block.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"
#include "conv_winograd43.hpp"

namespace pd3 {

// CB = 16-channel blocks per workgroup.  CB = 2: 256 threads, two workgroups per CU.  CB = 4: 512 threads, one
// workgroup per CU -- every transformed patch then feeds 64 output channels (half the transform work per
// MFMA) and the two waves of a SIMD run their MFMA phases together, back to back on the matrix pipe; waves
// 0-3 transform while waves 4-7 park the (twice as large) U slice.
template <int CB>
__global__ __launch_bounds__(128 * CB, CB == 2 ? 2 : 1) void conv3x3_winograd43_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ up,
                                                                    const float* __restrict__ bias,
                                                                    float* __restrict__ out, int cin, int cout,
                                                                    int h, int w, int wv, int relu, int ptiles) {
  // w = row pitch of input and output (a multiple of 4), wv <= w the valid width: output columns >= wv are
  // written as zeros (and the caller guarantees the same of the input), i.e. they are the next layer's padding
  constexpr int THREADS = 128 * CB;
  constexpr int CO = 16 * CB;                       // output channels per workgroup
  constexpr int USZ = CB * kW4Ci * 16 * kW4Cs;      // floats of one U trip
  constexpr int UN4 = USZ / 4;                      // CB = 2: 1152, CB = 4: 2304 float4
  constexpr int UT0 = CB == 2 ? 0 : 256;            // first of the 256 threads that move U
  constexpr int UPT = (UN4 + 255) / 256;            // 5 (tail repeats the last one) / 9
  constexpr int XPT = (kW4XN4 + THREADS - 1) / THREADS;  // 3 / 2
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NBUF = CB == 4 ? 2 : 1;              // CB = 4: U and V double-buffered (see the trip loop)
  float* Us = smem;                                  // [NBUF][USZ]
  float* Vs = smem + NBUF * USZ;                     // [NBUF][kW4Vsz]
  float* Raw = Vs + NBUF * kW4Vsz;
  float *Ucur = Us, *Vcur = Vs, *Unext = Us, *Vnext = Vs;  // buffers of the trip being multiplied / prepared
  const int lane = lane_id(), wave = wave_id();
  const int tiles_x = (w + 4 * kW4TC - 1) / (4 * kW4TC), tiles_y = (h + 4 * kW4TR - 1) / (4 * kW4TR);
  // XCD-aware tile order (see conv_winograd.hip): pixel tile pt lives on XCD pt % 8 with all its channel tiles
  const int nct = cout / CO;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ct = slot % nct, pt = (slot / nct) * 8 + xcd;
  if (pt >= ptiles) return;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int y0 = ty * 4 * kW4TR, x0 = tx * 4 * kW4TC;
  const int chunks = cin / kW4Ci;
  const int64_t plane = (int64_t)h * w;
  const float* xin = x + (int64_t)n * cin * plane;
  const w4_f32x4* usrc = reinterpret_cast<const w4_f32x4*>(up) + (int64_t)ct * chunks * UN4;

  // staging pattern of the raw patch (identical for every trip)
  int gofs[XPT], ldst[XPT];
  unsigned live = 0;
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int e = min((int)threadIdx.x + i * THREADS, kW4XN4 - 1);
    const int ci = e / (kW4RawR * (kW4RawW / 4)), rem = e - ci * (kW4RawR * (kW4RawW / 4));
    const int r = rem / (kW4RawW / 4), c4 = rem - r * (kW4RawW / 4);
    const int gy = y0 - 1 + r, gx = x0 - 4 + c4 * 4;  // a float4 is entirely inside or outside (w % 4 == 0)
    const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
    gofs[i] = ok ? (int)(ci * plane + (int64_t)gy * w + gx) : 0;
    live |= ok ? (1u << i) : 0u;
    ldst[i] = e * 4;
  }
  const bool moves_u = (int)threadIdx.x >= UT0;  // wave-uniform
  int uofs[UPT];
#pragma unroll
  for (int i = 0; i < UPT; ++i) uofs[i] = min(max((int)threadIdx.x - UT0, 0) + i * 256, UN4 - 1);
  const bool transforms = threadIdx.x < 256;     // wave-uniform
  // transform assignment: thread pair (2p, 2p+1) owns patch p = (ci, tile); half hf = columns / rows 3hf..3hf+2
  const int pidx = threadIdx.x >> 1, hf = threadIdx.x & 1;
  const int pci = pidx >> 5, ptile = pidx & 31;
  const int rsrc = pci * kW4RawPl + (4 * (ptile >> 4)) * kW4RawW + 4 * (ptile & 15) + 3 + 3 * hf;
  const int vdst = (((ptile >> 4) * kW4Ci + pci) * kW4TC + (ptile & 15)) * kW4Cs + 18 * hf;
  // MFMA operand bases
  const int cb = wave % CB, tb = wave / CB;
  const int abase = ((cb * kW4Ci + (lane >> 4)) * 16 + (lane & 15)) * kW4Cs;
  const int bbase = ((tb * kW4Ci + (lane >> 4)) * kW4TC + (lane & 15)) * kW4Cs;

  w4_f32x4 acc[36];
#pragma unroll
  for (int c = 0; c < 36; ++c) acc[c] = (w4_f32x4){0.f, 0.f, 0.f, 0.f};

  w4_f32x4 xr[XPT], ur[UPT];

#define W4_FETCH_X(cc)                                                                   \
  {                                                                                      \
    const float* xc_ = xin + (int64_t)(cc) * kW4Ci * plane;                              \
    _Pragma("unroll") for (int i = 0; i < XPT; ++i)                                      \
        xr[i] = *reinterpret_cast<const w4_f32x4*>(xc_ + gofs[i]);                       \
  }
#define W4_FETCH_U(cc)                                                                   \
  if (CB == 2 || moves_u) {                                                              \
    const w4_f32x4* uc_ = usrc + (int64_t)(cc) * UN4;                                    \
    _Pragma("unroll") for (int i = 0; i < UPT; ++i) ur[i] = uc_[uofs[i]];                \
  }
#define W4_STASH_X()                                                                     \
  {                                                                                      \
    _Pragma("unroll") for (int i = 0; i < XPT; ++i) {                                    \
      const bool on_ = (live >> i) & 1u;                                                 \
      const w4_f32x4 z_ = {0.f, 0.f, 0.f, 0.f};                                          \
      *reinterpret_cast<w4_f32x4*>(Raw + ldst[i]) = on_ ? xr[i] : z_;                    \
    }                                                                                    \
  }
#define W4_STASH_U()                                                                     \
  if (CB == 2 || moves_u) {                                                              \
    _Pragma("unroll") for (int i = 0; i < UPT; ++i)                                      \
        *reinterpret_cast<w4_f32x4*>(Unext + uofs[i] * 4) = ur[i];                       \
  }
  // V = B^T d B.  Row pass on this lane's three columns, halves swapped between the pair, column pass on this
  // lane's three rows; components (row, nu) -> 6 row + nu.
#define W4_TRANSFORM()                                                                   \
  if (CB == 2 || transforms) {                                                                                      \
    float lo_[3][3], hi_[3][3]; /* (B^T d)[row a or 3 + a][my column b] */               \
    _Pragma("unroll") for (int b = 0; b < 3; ++b) {                                      \
      const float* d_ = Raw + rsrc + b;                                                  \
      float t_[6];                                                                       \
      w4_in(d_[0], d_[kW4RawW], d_[2 * kW4RawW], d_[3 * kW4RawW], d_[4 * kW4RawW], d_[5 * kW4RawW], t_); \
      _Pragma("unroll") for (int a = 0; a < 3; ++a) {                                    \
        lo_[a][b] = t_[a];                                                               \
        hi_[a][b] = t_[3 + a];                                                           \
      }                                                                                  \
    }                                                                                    \
    /* the even lane runs the column pass for rows 0..2, the odd lane for rows 3..5; what a lane lacks are \
       the other three columns of its rows, i.e. the partner's lo_ (even lane) or hi_ (odd lane): one      \
       select with a DPP-swapped operand per value */                                    \
    float* v_ = Vnext + vdst;                                                            \
    _Pragma("unroll") for (int a = 0; a < 3; ++a) {                                      \
      float f_[3], l_[3]; /* columns 0..2 / 3..5 of row 3 hf + a of B^T d */             \
      _Pragma("unroll") for (int b = 0; b < 3; ++b) {                                    \
        const float ph_ = w4_swap_pair(hi_[a][b]), pl_ = w4_swap_pair(lo_[a][b]);        \
        f_[b] = hf ? ph_ : lo_[a][b];                                                    \
        l_[b] = hf ? hi_[a][b] : pl_;                                                    \
      }                                                                                  \
      float o_[6];                                                                       \
      w4_in(f_[0], f_[1], f_[2], l_[0], l_[1], l_[2], o_);                               \
      *reinterpret_cast<w4_f32x2*>(v_ + a * 6 + 0) = (w4_f32x2){o_[0], o_[1]};           \
      *reinterpret_cast<w4_f32x2*>(v_ + a * 6 + 2) = (w4_f32x2){o_[2], o_[3]};           \
      *reinterpret_cast<w4_f32x2*>(v_ + a * 6 + 4) = (w4_f32x2){o_[4], o_[5]};           \
    }                                                                                    \
  }
  // 9 groups of 4 components: two b128 reads feed four MFMAs; reads run two groups ahead (ring of three)
#define W4_LOAD(g_, slot_)                                                               \
  {                                                                                      \
    a_[slot_] = *reinterpret_cast<const w4_f32x4*>(Ucur + abase + (g_) * 4);             \
    b_[slot_] = *reinterpret_cast<const w4_f32x4*>(Vcur + bbase + (g_) * 4);             \
  }
#define W4_MFMA()                                                                        \
  {                                                                                      \
    w4_f32x4 a_[3], b_[3];                                                               \
    W4_LOAD(0, 0)                                                                        \
    W4_LOAD(1, 1)                                                                        \
    _Pragma("unroll") for (int g_ = 0; g_ < 9; ++g_) {                                   \
      if (g_ + 2 < 9) W4_LOAD(g_ + 2, (g_ + 2) % 3)                                      \
      __builtin_amdgcn_sched_barrier(0);                                                 \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                   \
          acc[g_ * 4 + j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[g_ % 3][j_], b_[g_ % 3][j_], acc[g_ * 4 + j_], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                 \
    }                                                                                    \
  }

  // The U slice of a trip is 37 KB (CB = 4): its loads back-pressure at issue for ~1700 cycles (the L2 -> CU
  // path delivers ~16 B/clk), and a wave issues in order -- so they are issued where their waves have slack, right
  // after parking the previous slice during the other waves' transform, not in front of the MFMA phase.
  W4_FETCH_X(0)
  W4_FETCH_U(0)
  W4_STASH_X()
  __syncthreads();
  W4_TRANSFORM()
  W4_STASH_U()
  if (chunks > 1) W4_FETCH_U(1)
  if (CB == 4 && chunks > 1) W4_FETCH_X(1)
  __syncthreads();
  if (CB == 4) {
    // 64-channel form.  Wave w and wave w + 4 share a SIMD and take turns on its pipes inside a trip:
    //   waves 0-3:  transform(c+1) -> V[next]           then  MFMA(c)
    //   waves 4-7:  MFMA(c)                             then  park U(c+1) -> U[next], issue the loads of U(c+2)
    // so one wave of every SIMD feeds the matrix pipe while its partner runs the VALU / LDS chain (a wave issues
    // in order and does not overlap its own MFMAs with its own VALU work; two waves of a SIMD do).  Two barriers
    // per trip: after the raw patch of trip c+1 is staged, and at the end.  (The first round-2 attempt at such a
    // ping-pong gave every group its own transform and lost: the chain then ran twice per trip.)
    for (int cc = 0; cc < chunks; ++cc) {
      const bool more = cc + 1 < chunks;
      Ucur = Us + (cc & 1) * USZ;
      Vcur = Vs + (cc & 1) * kW4Vsz;
      Unext = Us + ((cc + 1) & 1) * USZ;
      Vnext = Vs + ((cc + 1) & 1) * kW4Vsz;
      if (more) {
        W4_STASH_X()  // X(cc + 1), fetched a trip ago
        if (cc + 2 < chunks) W4_FETCH_X(cc + 2)
      }
      __syncthreads();
      if (transforms) {
        if (more) W4_TRANSFORM()
        __builtin_amdgcn_s_setprio(1);
        W4_MFMA()
        __builtin_amdgcn_s_setprio(0);
      } else {
        __builtin_amdgcn_s_setprio(1);
        W4_MFMA()
        __builtin_amdgcn_s_setprio(0);
        if (more) W4_STASH_U()
        if (cc + 2 < chunks) W4_FETCH_U(cc + 2)
      }
      __syncthreads();
    }
  } else {
    // steady state (no conditionals around the activation loads); last trip peeled
    for (int cc = 0; cc + 1 < chunks; ++cc) {
      W4_FETCH_X(cc + 1)
      __builtin_amdgcn_sched_barrier(0);  // keep the loads in flight ahead of the MFMA block
      __builtin_amdgcn_s_setprio(1);  // the matrix phase outranks the co-resident waves' transform VALU
      W4_MFMA()
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // every wave is done with U and V of this trip
      W4_STASH_X()
      __syncthreads();
      W4_TRANSFORM()
      W4_STASH_U()
      if (cc + 2 < chunks) W4_FETCH_U(cc + 2)
      __syncthreads();
    }
    W4_MFMA()
  }
#undef W4_MFMA
#undef W4_LOAD
#undef W4_FETCH_X
#undef W4_FETCH_U
#undef W4_STASH_X
#undef W4_STASH_U
#undef W4_TRANSFORM

  // epilogue: Y = A^T M A; lane: tile column lane & 15, channels 4 (lane >> 4) + r of the co block
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = 0.f;
  const int co0 = ct * CO + cb * 16 + 4 * (lane >> 4);
  if (bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[co0 + r];
  }
  const int oy = y0 + 4 * tb, ox = x0 + 4 * (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s[4][6];  // A^T M: column j of M through the row pass
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float c4[4];
      w4_out(acc[0 * 6 + j][r], acc[1 * 6 + j][r], acc[2 * 6 + j][r], acc[3 * 6 + j][r], acc[4 * 6 + j][r],
             acc[5 * 6 + j][r], c4);
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k][j] = c4[k];
    }
    float* o = out + ((int64_t)n * cout + co0 + r) * plane + (int64_t)oy * w + ox;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float y4[4];
      w4_out(s[k][0], s[k][1], s[k][2], s[k][3], s[k][4], s[k][5], y4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y4[j] += bv[r];
        if (relu) y4[j] = fmaxf(y4[j], 0.f);
        if (ox + j >= wv) y4[j] = 0.f;
      }
      if (oy + k < h && ox < w)  // partial tiles at the border (w % 4 == 0: a quad is in or out)
        __builtin_nontemporal_store((w4_f32x4){y4[0], y4[1], y4[2], y4[3]},
                                    reinterpret_cast<w4_f32x4*>(o + (int64_t)k * w));
    }
  }
}


}  // namespace pd3

using namespace pd3;

template <int CB>
static int launch_wino43(const float* x, const float* u_packed, const float* bias, int batch, int cin, int cout,
                         int h, int w, int wv, int relu, float* out, hipStream_t s) {
  constexpr size_t lds = (size_t)((CB == 4 ? 2 : 1) * (CB * kW4Ci * 16 * kW4Cs + kW4Vsz) + kW4RawSz) * sizeof(float);
  {  // dynamic-LDS cap: per device, so it is set on every launch (a host-side table write)
    hipError_t e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(conv3x3_winograd43_kernel<CB>), (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t ptiles = (int64_t)batch * ceil_div(h, 4 * kW4TR) * ceil_div(w, 4 * kW4TC);
  const int64_t nwg = (ptiles + 7) / 8 * 8 * (cout / (16 * CB));
  if (nwg >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  conv3x3_winograd43_kernel<CB><<<(unsigned)nwg, 128 * CB, lds, s>>>(x, u_packed, bias, out, cin, cout, h, w, wv,
                                                                       relu, (int)ptiles);
  return launch_status();
}

// channels_per_tile selects the workgroup shape the weights were packed for: 32 (two workgroups per CU) or 64
extern "C" int pd3_conv3x3_winograd43_bias_relu(const float* x, const float* u_packed, const float* bias,
                                                int batch, int cin, int cout, int h, int w, int w_valid,
                                                int relu, float* out, int channels_per_tile, void* stream) {
  if (!x || !u_packed || !out || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || w_valid <= 0 ||
      w_valid > w)
    return PD3_EINVAL;
  if (channels_per_tile != 32 && channels_per_tile != 64) return PD3_EINVAL;
  if (cin % kW4Ci != 0 || cout % channels_per_tile != 0 || w % 4 != 0) return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(u_packed) % 16 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out) % 16 != 0)
    return PD3_EINVAL;
  if ((int64_t)cin * h * w >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;  // 32-bit staging offsets
  hipStream_t s = static_cast<hipStream_t>(stream);
  return channels_per_tile == 64 ? launch_wino43<4>(x, u_packed, bias, batch, cin, cout, h, w, w_valid, relu, out, s)
                                 : launch_wino43<2>(x, u_packed, bias, batch, cin, cout, h, w, w_valid, relu, out, s);
}
