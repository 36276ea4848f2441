// 3x3 / stride 1 / pad 1 convolution + bias + ReLU by Winograd F(4x4, 3x3) on the fp32 matrix cores: the pipelined form
// (round 4).  Same arithmetic, tile shape and results (bit for bit) as conv_winograd43.hip (reference layers:
// second_backbone.py:72-120, center_head.py:43-220, cuDNN there).
#include "../../include/paddle3d_amd.h"
#include "common.hpp"
#include "conv_winograd43.hpp"

namespace pd3 {

// ---------------------------------------------------------------------------------------------------------------------
// What the hardware does (tools/hwcheck/mfma_valu_overlap.hip, pingpong_skeleton.hip; measured on gfx950):
//   * a wave's own VALU instructions are NOT hidden behind its v_mfma_f32_16x16x4_f32 (32.4 / 36.9 / 44.5 cycles per MFMA
//     with 0 / 1 / 3 fmas behind each): a filler costs about its issue time, but no latency;
//   * a wave that streams fp32 MFMAs back to back starves its partner on the SIMD: the partner's LDS reads return and its
//     VALU instructions issue only when the stream ends.  A "transform wave beside a multiply wave" (the ping-pong forms
//     of rounds 2 and 4) therefore runs its latency chain (LDS read -> row pass -> DPP exchange -> column pass, ~1600
//     cycles for ~150 instructions) AFTER the partner's 2304 cycles of MFMAs, not beside them: 57 % matrix-pipe use.
// So the chain is cut into micro-steps of three or four VALU instructions and one step stands behind every second MFMA of
// the SAME wave: it costs its issue slots (~10 cycles per step) and none of its latency, and the two waves of a SIMD run
// the same mixed stream, so one wave's MFMAs fill the other's filler slots.
//
// All eight waves have one role.  Wave (g, cb): tile row g, output channels 16 cb .. 16 cb + 15 of the workgroup's 64, and
// the transform of channels 2 cb, 2 cb + 1 of every slot (8 input channels = two trips of MFMAs) for its tile row.
//   slot s:  72 MFMAs of slot s (A: lane-packed U from LDS, B: V from LDS, both through rings of ds_read_b128)
//            + the transform of slot s + 1 in micro-steps, results in 18 registers
//            + the fetches (buffer_load ... lds): the wave's raw rows of slot s + 2, one half of a U block of slot s + 1
//   then     barrier, V of slot s + 1 out of the registers into LDS, barrier.
// LDS (138 KB): U of one slot [2 trips][4 cb][9][64 lanes][4] (one buffer: trip h of block cb is refilled by wave (1 - h,
// cb) right after the barrier behind its last read -- in the middle of the slot for trip 0, at its end for trip 1 -- and
// has more than a trip to land); per tile row Raw [8][6][72] (wave-private planes) and V [8][16][36].
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPlKT = 2;                                   // trips per slot
constexpr int kPlCi = kPlKT * kW4Ci;                       // 8 input channels per slot
constexpr int kPlRawR = 6;                                 // staged input rows of one tile row
constexpr int kPlRawPl = kPlRawR * kW4RawW;                // 432 floats per channel
constexpr int kPlRawSz = kPlCi * kPlRawPl;                 // 3456 floats per group
constexpr int kPlVsz = kPlCi * kW4TC * kW4Cs;              // 4608 floats per group
constexpr int kPlXPT = 4;                                  // raw-row pieces per wave (216 float4)
constexpr int kPlUHalf = 9 * 64 * 4;                       // 2304 floats: U of one trip for one wave (9 float4 per lane)
constexpr int kPlUsz = kPlKT * 4 * kPlUHalf;               // 18432 floats per slot: [trip][cb][q][lane][4]

__global__ __launch_bounds__(512, 1) void conv3x3_winograd43_pl_kernel(const float* __restrict__ x,
                                                                       const float* __restrict__ ulane,
                                                                       const float* __restrict__ bias,
                                                                       float* __restrict__ out, int cin, int cout, int h,
                                                                       int w, int wv, int relu, int ptiles,
                                                                       long long* __restrict__ dbg) {
  const long long t_all = dbg ? clock64() : 0;  // (measurement: stamps of one workgroup's first slots)
  const bool tl = dbg && blockIdx.x == 8 && (threadIdx.x & 63) == 0;
  auto stamp = [&](int s, int id) {
    if (tl && s < 8) dbg[((threadIdx.x >> 6) * 8 + s) * 8 + id] = clock64() - t_all;
  };
  constexpr int CO = 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(wave_id());  // (uniform: scalar slot control)
  const int grp = wave >> 2, cb = wave & 3;  // tile row / 16-channel block of this wave; waves w and w + 4 share a SIMD
  const int gt = cb * 64 + lane;             // thread inside its group
  float* Us = smem;                                          // [2 trips][4 cb][9][64 lanes][4]: U of the current slot
  float* Raw = smem + kPlUsz + grp * (kPlRawSz + kPlVsz);    // [8 ci][6 rows][72 cols] of this group's tile row
  float* Vs = Raw + kPlRawSz;                      // [8 ci][16 tiles][36]
  const int tiles_x = (w + 4 * kW4TC - 1) / (4 * kW4TC), tiles_y = (h + 4 * kW4TR - 1) / (4 * kW4TR);
  const int nct = cout / CO;
  const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
  const int ct = slot_id % nct, pt = (slot_id / nct) * 8 + xcd;
  if (pt >= ptiles) return;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int y0 = ty * 4 * kW4TR, x0 = tx * 4 * kW4TC;
  const int slots = cin / kPlCi;
  const int64_t plane = (int64_t)h * w;
  const float* xin = x + (int64_t)n * cin * plane;

  // staging pattern of this WAVE's raw rows (identical for every slot).  Wave cb transforms channels 2 cb and 2 cb + 1 of
  // the slot and nothing else, so those two planes of Raw ([2][6][72] = 216 float4) are private to it: it fetches them
  // itself and needs no barrier between its reads and the next fetch.  float4 e = lane + 64 i comes from byte offset
  // gofs[i] of the slot's 8 channel planes, or from beyond the buffer's range (-> zeros) for the padding.
  constexpr int kWvN4 = 2 * kPlRawPl / 4;  // 216
  unsigned gofs[kPlXPT];
#pragma unroll
  for (int i = 0; i < kPlXPT; ++i) {
    const int e = min(lane + i * 64, kWvN4 - 1);
    const int cl = e / (kPlRawR * (kW4RawW / 4)), rem = e - cl * (kPlRawR * (kW4RawW / 4));
    const int r = rem / (kW4RawW / 4), c4 = rem - r * (kW4RawW / 4);
    const int gy = y0 + 4 * grp - 1 + r, gx = x0 - 4 + c4 * 4;
    const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
    gofs[i] = ok ? (unsigned)(4 * ((2 * cb + cl) * plane + (int64_t)gy * w + gx)) : 0x7ffffff0u;
  }
  // transform assignment: thread pair (2p, 2p+1) of the group owns patch p = (ci 0..7, tile column 0..15)
  const int pidx = gt >> 1, hf = gt & 1;
  const int pci = pidx >> 4, ptile = pidx & 15;
  const int rsrc = pci * kPlRawPl + 4 * ptile + 3 + 3 * hf;
  const int vdst = (pci * kW4TC + ptile) * kW4Cs + 18 * hf;
  // MFMA operands: B = V[(trip * 4 + k) ci][tile][component]; A = U of (co, ci) = (lane & 15, lane >> 4), float4 q of the
  // lane's 36 components at Us[((trip * 4 + cb) * 9 + q) * 256 + 4 lane]
  const int bbase = ((lane >> 4) * kW4TC + (lane & 15)) * kW4Cs;
  const float* uct = ulane + (int64_t)ct * slots * kPlUsz;  // this workgroup's 64 output channels, all slots

  w4_f32x4 acc[36];
#pragma unroll
  for (int c = 0; c < 36; ++c) acc[c] = (w4_f32x4){0.f, 0.f, 0.f, 0.f};

  // the wave's raw rows of slot s go from global memory straight into its planes of Raw (buffer_load_dwordx4 ... lds: no
  // staging registers, no store pass)
  auto fetch_x = [&](int s) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(xin + (int64_t)s * kPlCi * plane), 0, (int)(kPlCi * plane * 4), 0x00020000);
#pragma unroll
    for (int i = 0; i < kPlXPT; ++i) {
      if (lane + i * 64 < kWvN4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rs, (__attribute__((address_space(3))) void*)(Raw + 2 * cb * kPlRawPl + i * 256), 16, gofs[i], 0, 0, 0);
    }
  };
  // half hh of slot s of this wave's U block: 9 KB, contiguous in global memory and in LDS alike
  const __amdgpu_buffer_rsrc_t urs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uct), 0, (int)((int64_t)slots * kPlUsz * 4), 0x00020000);
  auto fetch_u = [&](int s, int hh) {
    const int blk = (hh * 4 + cb) * kPlUHalf;
#pragma unroll
    for (int q = 0; q < 9; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(urs, (__attribute__((address_space(3))) void*)(Us + blk + q * 256), 16,
                                               lane * 16, (s * kPlUsz + blk + q * 256) * 4, 0, 0);
  };

  // ---- the transform of one slot for this thread pair's patch, as state + micro-steps -------------------------------
  float rv[3][6];            // the patch's six rows, this thread's three columns
  float lo[3][3], hi[3][3];  // after the row pass: rows 0-2 / 3-5 of B^T d
  float fl[3][6];            // after the exchange with the pair: this thread's three ROWS, six columns
  float vo[3][6];            // V values of this thread: rows 3 hf .. 3 hf + 2, six components each
  float ta = 0.f, tb = 0.f, tc = 0.f, te = 0.f;
  auto read_rows = [&]() {
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float* d = Raw + rsrc + b;
#pragma unroll
      for (int r = 0; r < 6; ++r) rv[b][r] = d[r * kW4RawW];
    }
  };
  // w4_in(d[0..5]) -> t[0..5] in four steps of three operations (the same operations in the same order)
  auto in4 = [&](int part, const float (&d)[6], float (&t)[6]) {
    if (part == 0) {
      ta = __builtin_fmaf(-4.f, d[2], d[4]);
      tb = __builtin_fmaf(-4.f, d[1], d[3]);
      tc = d[4] - d[2];
    } else if (part == 1) {
      te = d[3] - d[1];
      t[1] = ta + tb;
      t[2] = ta - tb;
    } else if (part == 2) {
      t[3] = __builtin_fmaf(2.f, te, tc);
      t[4] = __builtin_fmaf(-2.f, te, tc);
      t[0] = __builtin_fmaf(-5.f, d[2], d[4]);
    } else {
      t[0] = __builtin_fmaf(4.f, d[0], t[0]);
      t[5] = __builtin_fmaf(4.f, d[1], __builtin_fmaf(-5.f, d[3], d[5]));
    }
  };
  float trow[6];
  // step k of 33: 0-11 row pass (column b = k / 4), 12-20 exchange with the pair (a = (k - 12) / 3, b = (k - 12) % 3),
  // 21-32 column pass (row a = (k - 21) / 4)
  auto tstep = [&](int k) {
    if (k < 12) {
      const int b = k / 4;
      in4(k % 4, rv[b], trow);
      if (k % 4 == 3) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          lo[a][b] = trow[a];
          hi[a][b] = trow[3 + a];
        }
      }
    } else if (k < 21) {
      const int a = (k - 12) / 3, b = (k - 12) % 3;
      const float ph = w4_swap_pair(hi[a][b]), pl = w4_swap_pair(lo[a][b]);
      fl[a][b] = hf ? ph : lo[a][b];
      fl[a][3 + b] = hf ? hi[a][b] : pl;
    } else if (k < 33) {
      const int a = (k - 21) / 4;
      in4((k - 21) % 4, fl[a], vo[a]);
    }
  };
  auto write_v = [&]() {
    float* v = Vs + vdst;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      *reinterpret_cast<w4_f32x2*>(v + a * 6 + 0) = (w4_f32x2){vo[a][0], vo[a][1]};
      *reinterpret_cast<w4_f32x2*>(v + a * 6 + 2) = (w4_f32x2){vo[a][2], vo[a][3]};
      *reinterpret_cast<w4_f32x2*>(v + a * 6 + 4) = (w4_f32x2){vo[a][4], vo[a][5]};
    }
  };

  auto vptr = [&](int g) { return Vs + (g / 9) * (kW4Ci * kW4TC * kW4Cs) + bbase + (g % 9) * 4; };
  auto uptr = [&](int g) { return Us + (((g / 9) * 4 + cb) * 9 + (g % 9)) * 256 + lane * 4; };

  // prologue: the wave's rows of slots 0 and 1, its half of its U block of slot 0, the transform of slot 0
  fetch_x(0);
  fetch_u(0, 1 - grp);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  read_rows();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the rows are in registers before their planes are overwritten
  fetch_x(min(1, slots - 1));
#pragma unroll
  for (int k = 0; k < 33; ++k) tstep(k);
  write_v();
  w4_lds_barrier();

  for (int s = 0; s < slots; ++s) {
    const int sn = min(s + 1, slots - 1), sn2 = min(s + 2, slots - 1);  // (past the end: refetched, never used)
    // the wave's rows of slot s + 1 have landed (fetched a slot ago; only group 0's U pieces, issued after them, may
    // still be on their way)
    if (grp == 0 && s > 0) __builtin_amdgcn_s_waitcnt(0x0f70 | 9);  // vmcnt(9)
    else __builtin_amdgcn_s_waitcnt(0x0f70);                        // vmcnt(0)
    stamp(s, 0);
    read_rows();
    w4_f32x4 a[3], b[3];
    a[0] = *reinterpret_cast<const w4_f32x4*>(uptr(0));
    b[0] = *reinterpret_cast<const w4_f32x4*>(vptr(0));
    a[1] = *reinterpret_cast<const w4_f32x4*>(uptr(1));
    b[1] = *reinterpret_cast<const w4_f32x4*>(vptr(1));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int g9 = 0; g9 < 9; ++g9) {
        const int g = t * 9 + g9;
        if (g == 7) {
          // every read of trip 0's U has been issued (the ring runs two groups ahead); behind this barrier trip 1's U --
          // refilled by group 0 at the end of the last slot, waited for here -- is read, and trip 0's block is refilled
          // by group 1 with the next slot's values
          __builtin_amdgcn_sched_barrier(0);
          stamp(s, 1);
          if (grp == 0) __builtin_amdgcn_s_waitcnt(0x0f70 | 4);  // vmcnt(4): all but the raw rows issued in this slot
          w4_lds_barrier();
          stamp(s, 2);
          if (grp == 1) fetch_u(sn, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (g + 2 < 18) {
          a[(g + 2) % 3] = *reinterpret_cast<const w4_f32x4*>(uptr(g + 2));
          b[(g + 2) % 3] = *reinterpret_cast<const w4_f32x4*>(vptr(g + 2));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[g9 * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g % 3][j], b[g % 3][j], acc[g9 * 4 + j], 0, 0, 0);
          const int m = g * 4 + j;  // MFMA 0 .. 71 of the slot
          if (m >= 4 && m % 2 == 0 && (m - 4) / 2 < 33) tstep((m - 4) / 2);
          __builtin_amdgcn_sched_barrier(0);
        }
        // the rows were read in front of the first group's operands (LDS returns in order): their planes are free
        if (g == 0) fetch_x(sn2);
      }
    }
    stamp(s, 3);
    if (grp == 1) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): trip 0's refill has landed before the barriers below
    w4_lds_barrier();                                   // every wave is done with V and with trip 1's U of slot s
    stamp(s, 4);
    write_v();
    if (grp == 0) fetch_u(sn, 1);
    w4_lds_barrier();
    stamp(s, 5);
  }
  stamp(7, 6);

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

static int check_wino43_pl(const float* x, const float* u_lane, const float* out, int batch, int cin, int cout, int h,
                           int w, int w_valid) {
  if (!x || !u_lane || !out || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || w_valid <= 0 || w_valid > w)
    return PD3_EINVAL;
  if (cin % kPlCi != 0 || cout % 64 != 0 || w % 4 != 0) return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(u_lane) % 16 != 0)
    return PD3_EINVAL;
  if ((int64_t)kPlCi * h * w >= (int64_t)1 << 29) return PD3_EUNSUPPORTED;           // 32-bit byte offsets inside a slot
  if ((int64_t)(cin / kPlCi) * kPlUsz >= (int64_t)1 << 29) return PD3_EUNSUPPORTED;  // and inside a channel tile's U
  return PD3_OK;
}

extern "C" int pd3_conv3x3_winograd43_pl_bias_relu(const float* x, const float* u_lane, const float* bias, int batch,
                                                   int cin, int cout, int h, int w, int w_valid, int relu, float* out,
                                                   void* stream) {
  const int st = check_wino43_pl(x, u_lane, out, batch, cin, cout, h, w, w_valid);
  if (st != PD3_OK) return st;
  constexpr size_t lds = ((size_t)kPlUsz + 2 * (kPlRawSz + kPlVsz)) * sizeof(float);  // 138,240 B
  const void* fn = reinterpret_cast<const void*>(conv3x3_winograd43_pl_kernel);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int64_t ptiles = (int64_t)batch * ceil_div(h, 4 * kW4TR) * ceil_div(w, 4 * kW4TC);
  const int64_t nwg = (ptiles + 7) / 8 * 8 * (cout / 64);
  if (nwg >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  conv3x3_winograd43_pl_kernel<<<(unsigned)nwg, 512, lds, static_cast<hipStream_t>(stream)>>>(
      x, u_lane, bias, out, cin, cout, h, w, w_valid, relu, (int)ptiles, nullptr);
  return launch_status();
}

// measurement hook: + cycle stamps of one workgroup's first eight slots, dbg [8 waves][8 slots][8] int64 (device)
extern "C" int pd3_conv3x3_winograd43_pl_trace(const float* x, const float* u_lane, const float* bias, int batch, int cin,
                                               int cout, int h, int w, int relu, float* out, long long* dbg,
                                               void* stream) {
  const int st = check_wino43_pl(x, u_lane, out, batch, cin, cout, h, w, w);
  if (st != PD3_OK || !dbg) return st != PD3_OK ? st : PD3_EINVAL;
  constexpr size_t lds = ((size_t)kPlUsz + 2 * (kPlRawSz + kPlVsz)) * sizeof(float);
  const void* fn = reinterpret_cast<const void*>(conv3x3_winograd43_pl_kernel);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int64_t ptiles = (int64_t)batch * ceil_div(h, 4 * kW4TR) * ceil_div(w, 4 * kW4TC);
  const int64_t nwg = (ptiles + 7) / 8 * 8 * (cout / 64);
  conv3x3_winograd43_pl_kernel<<<(unsigned)nwg, 512, lds, static_cast<hipStream_t>(stream)>>>(
      x, u_lane, bias, out, cin, cout, h, w, w, relu, (int)ptiles, dbg);
  return launch_status();
}
