// The ping-pong F(4x4, 3x3) kernel (conv_winograd43_pp.hip) for a layer with MANY output-channel blocks over the same input
// -- CenterHead's 36 first-stage convolutions run as one 64 -> 2304 layer (center_head.py:99-118) -- with the input
// transform taken OUT of it (round 6): V = B^T d B of every (image, tile, input channel) is computed ONCE by
// w43_input_transform_kernel into global memory, in exactly the order a transform slot of the ping-pong kernel leaves it
// in LDS, and a transform slot here is nothing but the fetch of its 18 KB.
// Why: every workgroup of the ping-pong kernel transforms its pixel tile's input for itself -- 36 times the same work for
// the head -- and 13-15 % of that kernel's time is the transform's arithmetic, its row reads and V stores (measured with
// them switched off).  The kernel is not bound by what it fetches (1-5 % with the fetches switched off), so
// fetching V (18.4 KB per tile row and slot) instead of raw rows (13.8 KB) costs nothing; the pass itself reads 67 MB and
// writes 151 MB for the head's input (16 frames, 64 channels at 128 x 128).
// Same U (pack_winograd43_lane_weight), same V bits (the same w4_in sequence), same MFMA order, same output transform:
// the same bytes out as conv_winograd43_pp.hip (tested).
#include "../../include/paddle3d_amd.h"
#include "common.hpp"
#include "conv_winograd43.hpp"

#include <type_traits>

namespace pd3 {

constexpr int kPvKT = 2;                                   // trips per slot
constexpr int kPvCi = kPvKT * kW4Ci;                       // 8 input channels per slot
constexpr int kPvVsz = kPvCi * kW4TC * kW4Cs;              // 4608 floats per (tile row, slot): [8 ci][16 tiles][36]
constexpr int kPvUHalf = 9 * 64 * 4;                       // 2304 floats: U of one trip for one wave
constexpr int kPvUsz = kPvKT * 4 * kPvUHalf;               // 18432 floats per slot
constexpr int kPvMaxBlocks = 18;                           // channel blocks a workgroup walks at most (their bias sits in LDS)

// V[pixel tile pt][slot s][tile row g][ci 8][tile 16][36]: one workgroup = one (pt, s, g) block of 18 432 bytes; thread =
// (ci, tile): 6 x 6 input values (zero outside the image), B^T along the rows of every column, then along the columns --
// the order of the ping-pong kernel's thread pairs, so the bits are the same.  The block leaves through LDS as one
// contiguous piece.
__global__ __launch_bounds__(128) void w43_input_transform_kernel(const float* __restrict__ x, int cin, int h, int w,
                                                                  int wv, int tiles_x, int tiles_y, int slots,
                                                                  float* __restrict__ v) {
  __shared__ __attribute__((aligned(16))) float blk[kPvVsz];
  const int g = blockIdx.x & 1, s = (blockIdx.x >> 1) % slots, pt = (blockIdx.x >> 1) / slots;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int ci = threadIdx.x >> 4, tile = threadIdx.x & 15;
  const int y0 = ty * 4 * kW4TR + 4 * g - 1, x0 = tx * 4 * kW4TC + 4 * tile - 1;
  const float* xin = x + ((int64_t)n * cin + s * kPvCi + ci) * (int64_t)h * w;
  float d[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int gy = y0 + r, gx = x0 + c;
      d[r][c] = gy >= 0 && gy < h && gx >= 0 && gx < wv ? xin[(int64_t)gy * w + gx] : 0.f;
    }
  float t[6][6];  // t[a][c] = (B^T d)[a][c]
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    float o[6];
    w4_in(d[0][c], d[1][c], d[2][c], d[3][c], d[4][c], d[5][c], o);
#pragma unroll
    for (int a = 0; a < 6; ++a) t[a][c] = o[a];
  }
  float* dst = blk + (ci * kW4TC + tile) * kW4Cs;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    float o[6];
    w4_in(t[a][0], t[a][1], t[a][2], t[a][3], t[a][4], t[a][5], o);
#pragma unroll
    for (int c = 0; c < 6; ++c) dst[a * 6 + c] = o[c];
  }
  __syncthreads();
  w4_f32x4* out4 = reinterpret_cast<w4_f32x4*>(v + (int64_t)blockIdx.x * kPvVsz);  // block index = (pt * slots + s) * 2 + g
  const w4_f32x4* b4 = reinterpret_cast<const w4_f32x4*>(blk);
#pragma unroll
  for (int i = 0; i < kPvVsz / 4 / 128; ++i) out4[threadIdx.x + i * 128] = b4[threadIdx.x + i * 128];
}

// 64 lanes x 16 bytes from base + voff + soff to lds .. lds + 1023
__device__ __forceinline__ void pv_dma(const float* base, unsigned bytes, float* lds, unsigned voff, unsigned soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, (int)soff, 0, 0);
}

__global__ __launch_bounds__(512, 1) void conv3x3_winograd43_ppv_kernel(const float* __restrict__ vpre,
                                                                        const float* __restrict__ ulane,
                                                                        const float* __restrict__ bias,
                                                                        float* __restrict__ out, int cin, int cout, int h,
                                                                        int w, int wv, int relu, int ptiles, int ipw) {
  constexpr int CO = 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int grp = wave >> 2, cb = wave & 3;  // tile row / 16-channel block of this wave; waves w and w + 4 share a SIMD
  float* Us = smem;                                  // [2 trips][4 cb][9][64 lanes][4]: U of the current slot
  float* Vs = smem + kPvUsz + grp * 2 * kPvVsz;      // [2 buffers][8 ci][16 tiles][36] of this group's tile row
  float* bias_s = smem + kPvUsz + 4 * kPvVsz;        // [ipw][64]: the bias of this workgroup's channel blocks
  const int tiles_x = (w + 4 * kW4TC - 1) / (4 * kW4TC), tiles_y = (h + 4 * kW4TR - 1) / (4 * kW4TR);
  // a workgroup walks `ipw` consecutive channel blocks of ONE pixel tile (they read the same V): the first fetches of block
  // i + 1 travel under the output transform of block i, whose stores drain under block i + 1's first slots (cycle stamps of
  // the one-block form, the head's 64 -> 1152 slice: prologue 5.9 k, eight slot pairs 41.7 k, epilogue 7-9 k cycles)
  const int ncg = (cout / CO) / ipw;
  const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
  const int ct0 = (slot_id % ncg) * ipw, pt = (slot_id / ncg) * 8 + xcd;
  if (pt >= ptiles) return;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int y0 = ty * 4 * kW4TR, x0 = tx * 4 * kW4TC;
  const int slots = cin / kPvCi;
  const bool whole_tiles = h % (4 * kW4TR) == 0 && w % (4 * kW4TC) == 0;
  const int64_t plane = (int64_t)h * w;
  const int bbase = ((lane >> 4) * kW4TC + (lane & 15)) * kW4Cs;
  // this pixel tile's V: [slot][tile row][4608]; the group's block of slot s = 18 fetches of 1 KB, wave cb sends pieces
  // cb, cb + 4, .. (five for cb < 2, four otherwise)
  const float* vpt = vpre + (int64_t)pt * slots * 2 * kPvVsz;
  const unsigned vbytes = (unsigned)(slots * 2 * kPvVsz * 4);
  const int nv = cb < 2 ? 5 : 4;

  w4_f32x4 acc[36];
#pragma unroll
  for (int c = 0; c < 36; ++c) acc[c] = (w4_f32x4){0.f, 0.f, 0.f, 0.f};

  auto fetch_v = [&](int s) {
    float* dst = Vs + (s & 1) * kPvVsz;
    const unsigned so = (unsigned)((s * 2 + grp) * kPvVsz * 4);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int piece = cb + 4 * i;
      if (i < 4 || cb < 2) pv_dma(vpt, vbytes, dst + piece * 256, lane * 16, so + piece * 1024);
    }
  };
  const unsigned ubytes = (unsigned)((int64_t)(cout / CO) * slots * kPvUsz * 4);
  int ct = ct0;  // the channel block being multiplied
  auto fetch_u = [&](int c, int s, int hh) {  // half hh of slot s of channel block c
    const int blk = (hh * 4 + cb) * kPvUHalf;
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(((c * slots + s) * kPvUsz + blk) * 4));
#pragma unroll
    for (int q = 0; q < 9; ++q) pv_dma(ulane, ubytes, Us + blk + q * 256, lane * 16, so + (unsigned)(q * 1024));
  };
  // a multiply slot: 72 MFMAs, one stream over both trips, fed by ds_read_b128 alone (as conv_winograd43_pp.hip; no fetch
  // is waited for inside it)
  auto multiply_ring = [&](const float* V) {
    auto vptr = [&](int g) { return V + (g / 9) * (kW4Ci * kW4TC * kW4Cs) + bbase + (g % 9) * 4; };
    auto uptr = [&](int g) { return Us + (((g / 9) * 4 + cb) * 9 + (g % 9)) * 256 + lane * 4; };
    w4_f32x4 a[3], b[3];
    a[0] = *reinterpret_cast<const w4_f32x4*>(uptr(0));
    b[0] = *reinterpret_cast<const w4_f32x4*>(vptr(0));
    a[1] = *reinterpret_cast<const w4_f32x4*>(uptr(1));
    b[1] = *reinterpret_cast<const w4_f32x4*>(vptr(1));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int g9 = 0; g9 < 9; ++g9) {
        const int g = t * 9 + g9;
        if (g + 2 < 18) {
          a[(g + 2) % 3] = *reinterpret_cast<const w4_f32x4*>(uptr(g + 2));
          b[(g + 2) % 3] = *reinterpret_cast<const w4_f32x4*>(vptr(g + 2));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[g9 * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g % 3][j], b[g % 3][j], acc[g9 * 4 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  w4_f32x4 ua[18];
  auto multiply_regs = [&](const float* V) {
    auto vptr = [&](int g) { return V + (g / 9) * (kW4Ci * kW4TC * kW4Cs) + bbase + (g % 9) * 4; };
    w4_f32x4 b[3];
    b[0] = *reinterpret_cast<const w4_f32x4*>(vptr(0));
    b[1] = *reinterpret_cast<const w4_f32x4*>(vptr(1));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int g9 = 0; g9 < 9; ++g9) {
        const int g = t * 9 + g9;
        if (g + 2 < 18) b[(g + 2) % 3] = *reinterpret_cast<const w4_f32x4*>(vptr(g + 2));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[g9 * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[g][j], b[g % 3][j], acc[g9 * 4 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // prologue: the bias of the workgroup's blocks, V of slot 0 (each group its own tile row) and slot 0's U (group 0's waves)
  for (int t = threadIdx.x; t < ipw * CO; t += 512) bias_s[t] = bias ? bias[ct0 * CO + t] : 0.f;
  fetch_v(0);
  if (grp == 0) {
    fetch_u(ct0, 0, 0);
    fetch_u(ct0, 0, 1);
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  w4_lds_barrier();
  // Time slots as in conv_winograd43_pp.hip: a group's "transform slot" s is now: send V of slot s + 1 into the other
  // buffer (group 0: refill U with slot s first -- group 0 read slot s - 1 in the time slot before, group 1 into ua[]),
  // group 1: U of slot s into registers; then wait for everything but the V fetches just sent (in-order completion: U of
  // this slot and V of slot s, sent a slot ago, have landed) and the barrier.  (every wave passes 2 * slots + 1 barriers:
  // group 1 waits out the first time slot, group 0 the last)
  auto run = [&](auto is_g0) {
    constexpr bool G0 = decltype(is_g0)::value;
    if (!G0) w4_lds_barrier();
    for (int s = 0; s < slots; ++s) {
      {  // "transform" slot s
        if (G0 && s > 0) {
          fetch_u(ct, s, 0);
          fetch_u(ct, s, 1);
        }
        const bool more = s + 1 < slots;
        if (more) fetch_v(s + 1);
        if (!G0) {
          auto uptr = [&](int g) { return Us + (((g / 9) * 4 + cb) * 9 + (g % 9)) * 256 + lane * 4; };
#pragma unroll
          for (int g = 0; g < 18; ++g) ua[g] = *reinterpret_cast<const w4_f32x4*>(uptr(g));
        }
        if (s == 0 && ct != ct0) {
          // (a later block's slot 0: its U and V landed before the barrier that followed the block before; waiting here
          // would wait for that block's stores)
        } else if (!more) {
          __builtin_amdgcn_s_waitcnt(0x0f70);
        } else if (nv == 5) {
          __builtin_amdgcn_s_waitcnt(0x0f70 | 5);
        } else {
          __builtin_amdgcn_s_waitcnt(0x0f70 | 4);
        }
      }
      w4_lds_barrier();
      {  // multiply slot s
        const float* V = Vs + (s & 1) * kPvVsz;
        if (G0) multiply_ring(V);
        else multiply_regs(V);
      }
      w4_lds_barrier();
    }
    if (G0) w4_lds_barrier();
  };
  // (one straight-line copy of the block loop per group: with the group's role chosen inside the loop the two roles'
  // registers meet at a join in every trip -- 181 spilled registers, 40 % slower than the one-block form)
  auto blocks = [&](auto is_g0) {
  for (int it = 0; it < ipw; ++it) {
  ct = ct0 + it;
  run(is_g0);
  // (behind run()'s last barrier every wave is done with U and with both V buffers)
  if (it + 1 < ipw) {  // the next block's first fetches travel under this block's output transform
    fetch_v(0);
    if (grp == 0) {
      fetch_u(ct + 1, 0, 0);
      fetch_u(ct + 1, 0, 1);
    }
  }

  // epilogue (as conv_winograd43_pp.hip): Y = A^T M A; lane: tile column lane & 15, channels 4 (lane >> 4) + r of the block
  float bv[4];
  const int co0 = ct * CO + cb * 16 + 4 * (lane >> 4);
  {
    const w4_f32x4 b4 = *reinterpret_cast<const w4_f32x4*>(bias_s + it * CO + cb * 16 + 4 * (lane >> 4));
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = b4[r];
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
  if (it + 1 < ipw) {
#pragma unroll
    for (int c = 0; c < 36; ++c) acc[c] = (w4_f32x4){0.f, 0.f, 0.f, 0.f};
    // the next block's fetches have landed; younger than them are only this block's 16 stores per lane, which drain under
    // the next block's first slots -- where every wave issued all 16 (whole tiles); with partial tiles a wave may have
    // skipped some, and the count would let a fetch through: everything then
    if (whole_tiles) __builtin_amdgcn_s_waitcnt(0x0f70 | (16 & 15) | ((16 >> 4) << 14));  // vmcnt(16)
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    w4_lds_barrier();
  }
  }  // blocks of this workgroup
  };
  if (grp == 0) blocks(std::true_type{});
  else blocks(std::false_type{});
}

}  // namespace pd3

using namespace pd3;

extern "C" size_t pd3_winograd43_input_transform_floats(int batch, int cin, int h, int w) {
  if (batch <= 0 || cin <= 0 || h <= 0 || w <= 0 || cin % kPvCi != 0) return 0;
  const int64_t ptiles = (int64_t)batch * ceil_div(h, 4 * kW4TR) * ceil_div(w, 4 * kW4TC);
  return (size_t)(ptiles * (cin / kPvCi) * 2 * kPvVsz);
}

extern "C" int pd3_winograd43_input_transform(const float* x, int batch, int cin, int h, int w, int w_valid, float* v,
                                              void* stream) {
  if (!x || !v || batch <= 0 || cin <= 0 || h <= 0 || w <= 0 || w_valid <= 0 || w_valid > w) return PD3_EINVAL;
  if (cin % kPvCi != 0) return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(v) % 16 != 0) return PD3_EINVAL;
  const int tiles_x = (int)ceil_div(w, 4 * kW4TC), tiles_y = (int)ceil_div(h, 4 * kW4TR);
  const int64_t blocks = (int64_t)batch * tiles_x * tiles_y * (cin / kPvCi) * 2;
  if (blocks >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  w43_input_transform_kernel<<<(unsigned)blocks, 128, 0, static_cast<hipStream_t>(stream)>>>(x, cin, h, w, w_valid, tiles_x,
                                                                                           tiles_y, cin / kPvCi, v);
  return launch_status();
}

extern "C" int pd3_conv3x3_winograd43_ppv_bias_relu(const float* v_pre, const float* u_lane, const float* bias, int batch,
                                                    int cin, int cout, int h, int w, int w_valid, int relu, float* out,
                                                    void* stream) {
  if (!v_pre || !u_lane || !out || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || w_valid <= 0 || w_valid > w)
    return PD3_EINVAL;
  if (cin % kPvCi != 0 || cout % 64 != 0 || w % 4 != 0) return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(v_pre) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(u_lane) % 16 != 0)
    return PD3_EINVAL;
  if ((int64_t)(cout / 64) * (cin / kPvCi) * kPvUsz >= (int64_t)1 << 29 ||
      (int64_t)(cin / kPvCi) * 2 * kPvVsz >= (int64_t)1 << 29)
    return PD3_EUNSUPPORTED;  // 32-bit byte offsets inside U and inside a pixel tile's V
  // channel blocks per workgroup: the largest divisor of cout / 64 (up to kPvMaxBlocks) that still leaves six workgroups per
  // CU (the head's 18 blocks per slice, 16 frames: 3 / 6 / 9 / 18 blocks per workgroup = 831 / 812 / 830 / 849 us)
  const int nct = cout / 64;
  const int64_t ptiles8 = (((int64_t)batch * ceil_div(h, 4 * kW4TR) * ceil_div(w, 4 * kW4TC)) + 7) / 8 * 8;
  int ipw = 1;
  for (int d = 2; d <= kPvMaxBlocks; ++d)
    if (nct % d == 0 && ptiles8 * (nct / d) >= 6 * 256) ipw = d;
  const size_t lds = ((size_t)kPvUsz + 4 * kPvVsz + kPvMaxBlocks * 64) * sizeof(float);  // 147 456 B + the blocks' bias
  const void* fn = reinterpret_cast<const void*>(conv3x3_winograd43_ppv_kernel);
  const hipError_t e = pd3_max_dynamic_lds(fn, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int64_t ptiles = (int64_t)batch * ceil_div(h, 4 * kW4TR) * ceil_div(w, 4 * kW4TC);
  const int64_t nwg = (ptiles + 7) / 8 * 8 * (nct / ipw);
  if (nwg >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  conv3x3_winograd43_ppv_kernel<<<(unsigned)nwg, 512, lds, static_cast<hipStream_t>(stream)>>>(
      v_pre, u_lane, bias, out, cin, cout, h, w, w_valid, relu, (int)ptiles, ipw);
  return launch_status();
}
