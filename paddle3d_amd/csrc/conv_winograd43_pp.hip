// 3x3 / stride 1 / pad 1 convolution + bias + ReLU by Winograd F(4x4, 3x3) on the fp32 matrix cores: the ping-pong form
// (round 4).  Same arithmetic and tile shape as conv_winograd43.hip (reference layers: second_backbone.py:72-120,
// center_head.py:43-220, cuDNN there); see the comment in front of the kernel for what differs.
// Built with -fno-slp-vectorize (paddle3d_amd/build.py): packed v_pk_* arithmetic beside MFMAs costs more than it saves.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"
#include "conv_winograd43.hpp"

namespace pd3 {

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the ping-pong form with two-trip slots and U computed on the fly (`conv3x3_winograd43_pp_kernel`).
//
// What bounds the kernel above: a trip is the input-transform chain of waves 0-3 (~2300-2700 cycles whatever
// the amount of data: LDS read -> row pass -> DPP exchange -> column pass -> LDS write, a latency chain on one wave per
// SIMD) FOLLOWED by those waves' own 36 MFMAs (~1350), against 2304 cycles of matrix work per SIMD and trip.  Round 2's
// ping-pong gave every group its own transform per trip and lost: a slot was as long as the chain (2700), not as its 1152
// cycles of MFMAs, and there were two slots per trip.  Here a slot covers TWO trips (8 input channels):
//   * group g (waves 4g .. 4g+3, one per SIMD) owns tile row g for all 64 channels; in a transform slot its 256 threads
//     turn the 8 channels x 16 tiles of ITS tile row into V (one patch per thread pair: the same per-thread chain as above),
//     in a multiply slot its waves issue 2 x 36 MFMAs (2304 cycles) -- while the other group, half a period off, does the
//     opposite.  Per SIMD one wave always feeds the matrix pipe while its partner runs the chain; a slot lasts
//     max(chain, 2304) and two slots cover two trips.
//   * the A operand is no longer a 36.9 KB slice of pre-transformed U per trip through L2 -> registers -> LDS ->
//     registers: lane (co, ci) loads its 9 raw weights a slot ahead and computes U = G g G^T (36 values, ~90 VALU
//     operations) in the shadow of the MFMAs -- a quarter of the ingest, no U in LDS (64 KB instead of 122 KB), no
//     parking role.  The weights are the plain folded [cout][cin][3][3] tensor: no host-side packing.
// One workgroup-wide barrier per slot, reached by both groups.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPpKT = 2;                                   // trips per slot
constexpr int kPpCi = kPpKT * kW4Ci;                       // 8 input channels per slot
constexpr int kPpRawR = 6;                                 // staged input rows of one tile row
constexpr int kPpRawPl = kPpRawR * kW4RawW;                // 432 floats per channel
constexpr int kPpRawSz = kPpCi * kPpRawPl;                 // 3456 floats per group
constexpr int kPpVsz = kPpCi * kW4TC * kW4Cs;              // 4608 floats per group
constexpr int kPpXN4 = kPpRawSz / 4;                       // 864 float4 per group
constexpr int kPpXPT = (kPpXN4 + 255) / 256;               // 4 per thread

// G applied to three values (one column / one row of the 3x3 kernel): F(4x4, 3x3)'s kernel transform
__device__ __forceinline__ void w4_gg(const float a, const float b, const float c, float (&t)[6]) {
  const float s = a + c;
  t[0] = a * 0.25f;
  t[1] = (s + b) * (-1.f / 6.f);
  t[2] = (s - b) * (-1.f / 6.f);
  const float p = __builtin_fmaf(a, 1.f / 24.f, c * (1.f / 6.f)), q = b * (1.f / 12.f);
  t[3] = p + q;
  t[4] = p - q;
  t[5] = c;
}

template <bool FENCE>
__global__ __launch_bounds__(512, 1) void conv3x3_winograd43_pp_kernel(const float* __restrict__ x,
                                                                       const float* __restrict__ wraw,
                                                                       const float* __restrict__ bias,
                                                                       float* __restrict__ out, int cin, int cout, int h,
                                                                       int w, int wv, int relu, int ptiles, int prio,
                                                                       long long* __restrict__ dbg) {
  constexpr int CO = 64;
  long long t_tr = 0, t_mu = 0, t_ba = 0, t_all = dbg ? clock64() : 0;  // phase cycles of this wave (measurement)
  long long t_u0 = 0, t_m0 = 0, t_m1 = 0, t_st = 0;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = lane_id(), wave = wave_id();
  const int grp = wave >> 2, cb = wave & 3;  // tile row / 16-channel block of this wave; waves w and w + 4 share a SIMD
  const int gt = threadIdx.x & 255;          // thread inside its group
  float* Raw = smem + grp * (kPpRawSz + kPpVsz);  // [8 ci][6 rows][72 cols] of this group's tile row
  float* Vs = Raw + kPpRawSz;                      // [8 ci][16 tiles][36]
  const int tiles_x = (w + 4 * kW4TC - 1) / (4 * kW4TC), tiles_y = (h + 4 * kW4TR - 1) / (4 * kW4TR);
  const int nct = cout / CO;
  const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
  const int ct = slot_id % nct, pt = (slot_id / nct) * 8 + xcd;
  if (pt >= ptiles) return;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int y0 = ty * 4 * kW4TR, x0 = tx * 4 * kW4TC;
  const int slots = cin / kPpCi;
  const int64_t plane = (int64_t)h * w;
  const float* xin = x + (int64_t)n * cin * plane;

  // staging pattern of this group's raw rows (identical for every slot)
  int gofs[kPpXPT], ldst[kPpXPT];
  unsigned live = 0;
#pragma unroll
  for (int i = 0; i < kPpXPT; ++i) {
    const int e = min(gt + i * 256, kPpXN4 - 1);
    const int ci = e / (kPpRawR * (kW4RawW / 4)), rem = e - ci * (kPpRawR * (kW4RawW / 4));
    const int r = rem / (kW4RawW / 4), c4 = rem - r * (kW4RawW / 4);
    const int gy = y0 + 4 * grp - 1 + r, gx = x0 - 4 + c4 * 4;
    const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
    gofs[i] = ok ? (int)(ci * plane + (int64_t)gy * w + gx) : 0;
    live |= ok ? (1u << i) : 0u;
    ldst[i] = e * 4;
  }
  // transform assignment: thread pair (2p, 2p+1) of the group owns patch p = (ci 0..7, tile column 0..15)
  const int pidx = gt >> 1, hf = gt & 1;
  const int pci = pidx >> 4, ptile = pidx & 15;
  const int rsrc = pci * kPpRawPl + 4 * ptile + 3 + 3 * hf;
  const int vdst = (pci * kW4TC + ptile) * kW4Cs + 18 * hf;
  // MFMA operands: B = V[(trip * 4 + k) ci][tile][component]; A = U of (co, ci) = (lane & 15, lane >> 4), on the fly
  const int bbase = ((lane >> 4) * kW4TC + (lane & 15)) * kW4Cs;
  const float* wlane = wraw + ((int64_t)(ct * CO + cb * 16 + (lane & 15)) * cin + (lane >> 4)) * 9;

  w4_f32x4 acc[36];
#pragma unroll
  for (int c = 0; c < 36; ++c) acc[c] = (w4_f32x4){0.f, 0.f, 0.f, 0.f};
  w4_f32x4 xr[kPpXPT];
  float gw[kPpKT][9];

  auto fetch_x = [&](int s) {
    const float* xc = xin + (int64_t)s * kPpCi * plane;
#pragma unroll
    for (int i = 0; i < kPpXPT; ++i) xr[i] = *reinterpret_cast<const w4_f32x4*>(xc + gofs[i]);
  };
  auto fetch_g = [&](int s) {
#pragma unroll
    for (int kt = 0; kt < kPpKT; ++kt)
#pragma unroll
      for (int q = 0; q < 9; ++q) gw[kt][q] = wlane[(int64_t)(s * kPpCi + kt * kW4Ci) * 9 + q];
  };
  auto stash_x = [&]() {
#pragma unroll
    for (int i = 0; i < kPpXPT; ++i) {
      const w4_f32x4 z = {0.f, 0.f, 0.f, 0.f};
      if (kPpXN4 % 256 == 0 || gt + i * 256 < kPpXN4)
        *reinterpret_cast<w4_f32x4*>(Raw + ldst[i]) = ((live >> i) & 1u) ? xr[i] : z;
    }
  };
  auto transform = [&]() {  // V = B^T d B of this thread pair's patch (as W4_TRANSFORM above)
    float lo[3][3], hi[3][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float* d = Raw + rsrc + b;
      float t[6];
      w4_in(d[0], d[kW4RawW], d[2 * kW4RawW], d[3 * kW4RawW], d[4 * kW4RawW], d[5 * kW4RawW], t);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a][b] = t[a];
        hi[a][b] = t[3 + a];
      }
    }
    float* v = Vs + vdst;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float f[3], l[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float ph = w4_swap_pair(hi[a][b]), pl = w4_swap_pair(lo[a][b]);
        f[b] = hf ? ph : lo[a][b];
        l[b] = hf ? hi[a][b] : pl;
      }
      float o[6];
      w4_in(f[0], f[1], f[2], l[0], l[1], l[2], o);
      *reinterpret_cast<w4_f32x2*>(v + a * 6 + 0) = (w4_f32x2){o[0], o[1]};
      *reinterpret_cast<w4_f32x2*>(v + a * 6 + 2) = (w4_f32x2){o[2], o[3]};
      *reinterpret_cast<w4_f32x2*>(v + a * 6 + 4) = (w4_f32x2){o[4], o[5]};
    }
  };
  // U = G g G^T of this lane's (co, ci) for trip kt, in registers
  auto make_u = [&](int kt, float (&u)[36]) {
    float tc[3][6];  // G g: column j of the kernel through the transform
#pragma unroll
    for (int j = 0; j < 3; ++j) w4_gg(gw[kt][j], gw[kt][3 + j], gw[kt][6 + j], tc[j]);
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) {
      float r6[6];
      w4_gg(tc[0][xi], tc[1][xi], tc[2][xi], r6);
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) u[xi * 6 + nu] = r6[nu];
    }
  };
  // element xi of G applied to (a, b, c) (w4_gg, one output at a time: xi is a compile-time constant where it is used)
  auto gg1 = [](int xi, float a, float b, float c) -> float {
    switch (xi) {
      case 0: return a * 0.25f;
      case 1: return ((a + c) + b) * (-1.f / 6.f);
      case 2: return ((a + c) - b) * (-1.f / 6.f);
      case 3: return __builtin_fmaf(a, 1.f / 24.f, c * (1.f / 6.f)) + b * (1.f / 12.f);
      case 4: return __builtin_fmaf(a, 1.f / 24.f, c * (1.f / 6.f)) - b * (1.f / 12.f);
      default: return c;
    }
  };
  // one trip: 9 LDS reads of V feed 36 MFMAs.  ROLL: behind every third group (12 MFMAs = 384 matrix-pipe cycles in
  // flight) two rows of the NEXT trip's U are computed into the twelve registers those groups have just released --
  // ~30 VALU operations that do not depend on the MFMAs and run in their shadow, so only a slot's first U is exposed.
  auto mfma_trip = [&](int kt, float (&u)[36], bool roll) {
    const float* vb = Vs + kt * (kW4Ci * kW4TC * kW4Cs) + bbase;
    w4_f32x4 b[3];
    b[0] = *reinterpret_cast<const w4_f32x4*>(vb);
    b[1] = *reinterpret_cast<const w4_f32x4*>(vb + 4);
#pragma unroll
    for (int g = 0; g < 9; ++g) {
      if (g + 2 < 9) b[(g + 2) % 3] = *reinterpret_cast<const w4_f32x4*>(vb + (g + 2) * 4);
      if (FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[g * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[g * 4 + j], b[g % 3][j], acc[g * 4 + j], 0, 0, 0);
      if (FENCE) __builtin_amdgcn_sched_barrier(0);
      if (roll && g % 3 == 2) {
        const int xi0 = 2 * (g / 3);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int xi = xi0 + r;
          float r6[6];
          w4_gg(gg1(xi, gw[kt + 1][0], gw[kt + 1][3], gw[kt + 1][6]), gg1(xi, gw[kt + 1][1], gw[kt + 1][4], gw[kt + 1][7]),
                gg1(xi, gw[kt + 1][2], gw[kt + 1][5], gw[kt + 1][8]), r6);
#pragma unroll
          for (int nu = 0; nu < 6; ++nu) u[xi * 6 + nu] = r6[nu];
        }
        if (FENCE) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // group g runs T(0) M(0) T(1) M(1) ... one slot behind group g - 1: 2 * slots + 1 time slots, ONE barrier each.  A
  // multiply slot ends by staging the raw rows of the group's next transform slot (fetched at its start), so a transform
  // slot is the chain alone.  (A first version staged at the start of the transform slot behind a second barrier: the
  // barrier lined the other group's first trip up with the staging and its second trip with the transform, i.e. the
  // slot was one trip + the chain instead of their maximum: 15-20 % slower than the packed form.)
  fetch_x(0);
  stash_x();
  __syncthreads();
  const int nt = 2 * slots + 1;
  for (int tau = 0; tau < nt; ++tau) {
    const int k = tau - grp;
    const bool active = k >= 0 && k < 2 * slots;
    const int s = k >> 1;
    if (active && (k & 1) == 0) {  // transform slot s
      const long long c0 = dbg ? clock64() : 0;
      fetch_g(s);                  // needed by the multiply slot that follows
      if ((prio & 3) == 2) __builtin_amdgcn_s_setprio(2);
      transform();
      if ((prio & 3) == 2) __builtin_amdgcn_s_setprio(0);
      if (dbg) t_tr += clock64() - c0;
    } else if (active) {           // multiply slot s
      const long long c0 = dbg ? clock64() : 0;
      float u[36];
      make_u(0, u);  // (first use of the weights fetched a slot ago: the wait must not cover the loads issued next)
      fetch_x(min(s + 1, slots - 1));  // the next transform slot's raw rows travel during the MFMAs (the last slot
                                       // re-reads its own: no branch, so the load counter stays exact)
      const long long c_a = dbg ? clock64() : 0;
      if ((prio & 3) == 1) __builtin_amdgcn_s_setprio(1);
      mfma_trip(0, u, true);   // + U of trip 1, rolled in behind the MFMA groups
      const long long c_b = dbg ? clock64() : 0;
      mfma_trip(1, u, false);
      if ((prio & 3) == 1) __builtin_amdgcn_s_setprio(0);
      const long long c_c = dbg ? clock64() : 0;
      stash_x();
      if (dbg) {
        const long long c_d = clock64();
        t_u0 += c_a - c0, t_m0 += c_b - c_a, t_m1 += c_c - c_b, t_st += c_d - c_c;
      }
      if (dbg) t_mu += clock64() - c0;
    }
    const long long c1 = dbg ? clock64() : 0;
    __syncthreads();
    if (dbg) t_ba += clock64() - c1;
  }
  if (dbg && blockIdx.x == 8 && lane == 0) {
    dbg[wave * 4 + 0] = t_tr;
    dbg[wave * 4 + 1] = t_mu;
    dbg[wave * 4 + 2] = t_ba;
    dbg[wave * 4 + 3] = (clock64() - t_all) | ((long long)__builtin_amdgcn_s_getreg(2308) << 56);  // + SIMD id (HW_ID[5:4])
    dbg[32 + wave * 4 + 0] = t_u0;
    dbg[32 + wave * 4 + 1] = t_m0;
    dbg[32 + wave * 4 + 2] = t_m1;
    dbg[32 + wave * 4 + 3] = t_st;
  }

  // epilogue (as above): Y = A^T M A; lane: tile column lane & 15, channels 4 (lane >> 4) + r of the co block
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = 0.f;
  const int co0 = ct * CO + cb * 16 + 4 * (lane >> 4);
  if (bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[co0 + r];
  }
  const int oy = y0 + 4 * grp, ox = x0 + 4 * (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float sm[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float c4[4];
      w4_out(acc[0 * 6 + j][r], acc[1 * 6 + j][r], acc[2 * 6 + j][r], acc[3 * 6 + j][r], acc[4 * 6 + j][r],
             acc[5 * 6 + j][r], c4);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) sm[kk][j] = c4[kk];
    }
    float* o = out + ((int64_t)n * cout + co0 + r) * plane + (int64_t)oy * w + ox;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float y4[4];
      w4_out(sm[kk][0], sm[kk][1], sm[kk][2], sm[kk][3], sm[kk][4], sm[kk][5], y4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y4[j] += bv[r];
        if (relu) y4[j] = fmaxf(y4[j], 0.f);
        if (ox + j >= wv) y4[j] = 0.f;
      }
      if (oy + kk < h && ox < w)
        __builtin_nontemporal_store((w4_f32x4){y4[0], y4[1], y4[2], y4[3]},
                                    reinterpret_cast<w4_f32x4*>(o + (int64_t)kk * w));
    }
  }
}

}  // namespace pd3

using namespace pd3;

// The ping-pong form: weights are the plain folded [cout][cin][3][3] tensor (U is computed in the kernel).
// variant (measurement): bit 2 = scheduling fences around the MFMA groups; bits 0-1 = 0 no priorities, 1 multiply slots
// high, 2 transform slots high
static int launch_wino43_pp(const float* x, const float* w_raw, const float* bias, int batch, int cin, int cout, int h,
                            int w, int w_valid, int relu, float* out, int variant, hipStream_t s,
                            long long* dbg = nullptr) {
  constexpr size_t lds = (size_t)2 * (kPpRawSz + kPpVsz) * sizeof(float);
  const bool fence = (variant & 4) != 0;
  const void* fn = fence ? reinterpret_cast<const void*>(conv3x3_winograd43_pp_kernel<true>)
                         : reinterpret_cast<const void*>(conv3x3_winograd43_pp_kernel<false>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int64_t ptiles = (int64_t)batch * ceil_div(h, 4 * kW4TR) * ceil_div(w, 4 * kW4TC);
  const int64_t nwg = (ptiles + 7) / 8 * 8 * (cout / 64);
  if (nwg >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  if (fence)
    conv3x3_winograd43_pp_kernel<true><<<(unsigned)nwg, 512, lds, s>>>(x, w_raw, bias, out, cin, cout, h, w, w_valid, relu,
                                                                       (int)ptiles, variant & 3, dbg);
  else
    conv3x3_winograd43_pp_kernel<false><<<(unsigned)nwg, 512, lds, s>>>(x, w_raw, bias, out, cin, cout, h, w, w_valid, relu,
                                                                        (int)ptiles, variant & 3, dbg);
  return launch_status();
}

extern "C" int pd3_conv3x3_winograd43_raw_bias_relu_variant(const float* x, const float* w_raw, const float* bias,
                                                            int batch, int cin, int cout, int h, int w, int w_valid,
                                                            int relu, float* out, int variant, void* stream) {
  if (!x || !w_raw || !out || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || w_valid <= 0 || w_valid > w)
    return PD3_EINVAL;
  if (cin % kPpCi != 0 || cout % 64 != 0 || w % 4 != 0) return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0) return PD3_EINVAL;
  if ((int64_t)cin * h * w >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;  // 32-bit staging offsets
  return launch_wino43_pp(x, w_raw, bias, batch, cin, cout, h, w, w_valid, relu, out, variant,
                          static_cast<hipStream_t>(stream));
}

extern "C" int pd3_conv3x3_winograd43_raw_bias_relu(const float* x, const float* w_raw, const float* bias, int batch,
                                                    int cin, int cout, int h, int w, int w_valid, int relu, float* out,
                                                    void* stream) {
  return pd3_conv3x3_winograd43_raw_bias_relu_variant(x, w_raw, bias, batch, cin, cout, h, w, w_valid, relu, out, 1,
                                                      stream);
}

// measurement hook: the variant entry + per-wave phase cycle counters of one workgroup (dbg [8 waves][4]: transform,
// multiply, barrier wait, whole kernel; device memory)
extern "C" int pd3_conv3x3_winograd43_raw_trace(const float* x, const float* w_raw, const float* bias, int batch, int cin,
                                                int cout, int h, int w, int relu, float* out, int variant,
                                                long long* dbg, void* stream) {
  if (!x || !w_raw || !out || !dbg || cin % kPpCi != 0 || cout % 64 != 0 || w % 4 != 0) return PD3_EINVAL;
  return launch_wino43_pp(x, w_raw, bias, batch, cin, cout, h, w, w, relu, out, variant, static_cast<hipStream_t>(stream),
                          dbg);
}
