// The kernel = stride convolutions of SecondFPN (second_fpn.py:99-157) in fp32 arithmetic on the bf16 matrix cores
// (round 6): the same GEMMs as conv_patch.hip -- D[m][pixel] = sum_k A[m][k] B[k][pixel] + bias, ReLU, written at a channel
// offset of the concatenated map -- with every fp32 operand as THREE bf16 pieces and six piece products accumulated in fp32
// (sparse_conv_x3.hip explains the scheme and its error: that of an fp32 multiplication, not of a 16-bit format;
// tests/test_bf16x3_arith.py restates it on the CPU).  v_mfma_f32_32x32x16_bf16 runs at sixteen times the rate of
// v_mfma_f32_32x32x2_f32; six of them per fp32 product leave 2.7x, and these three layers (17 + 9 + 17 GFLOP per 16 frames,
// 0.87 GB in and out) were bound by the fp32 pipe (0.46 ms at 93 TFLOP/s).
//
//   mode 0  Conv2D k2 s2:          m = co,        a step = (dy, 16 input channels), k = (ci, dx): the two taps of an input
//                                                 row are one 8-byte LDS read
//   mode 1  1x1:                   m = co,        a step = 32 input channels
//   mode 2  Conv2DTranspose k2 s2: m = (dx, co) of 64 output channels for ONE dy; B as mode 1; the two dx of a channel sit
//                                                 in two accumulator blocks of the same lane and leave as one 8-byte store
//                                                 out[co][2 y + dy][2 x .. 2 x + 1]
// A workgroup of eight waves owns 256 pixels x 128 rows of A per work item, a wave 32 pixels (the MFMA's N) for all four
// 32-row blocks.  NCHW puts a lane's 8 values of one K-step (8 channels of ONE pixel) a plane apart; the first version
// loaded them with 4-byte loads straight into registers and was bound by the texture addresser (16 cycles per load
// instruction, 152 per step and CU: cycle stamps 2450 cycles per step before the younger half of the waves had its loads
// out).  Now NOTHING goes through registers on its way in: a wave's rows (32 channels x its 32 pixels, whole 128-byte lines)
// and the step's A pieces (cut and padded on the host: pack_patch_weight_x3) travel by buffer_load_dwordx4 ... lds, 4 + 4
// instructions per wave and step; the lane then reads its 8 channels from the wave's PRIVATE slot (no barrier between the
// fetch and the read, only the wave's own vmcnt) and cuts them.  Rows go out two steps ahead, A pieces one.
// These GEMMs are SHORT (4-8 steps per tile), so the kernel is persistent: a workgroup walks `ipw` consecutive work items
// (pixel tile, row tile; row tiles of one pixel tile follow each other) as ONE stream of steps -- the fetches run across
// the item boundary, and an item's stores leave while the next item's first MFMAs run.
// Summation order fixed (steps ascending, K-steps, pieces small to large): run-to-run identical.
#include "../../include/paddle3d_amd.h"
#include "bf16x3.hpp"
#include "common.hpp"

namespace pd3 {

constexpr int kPxThreads = 512;
constexpr int kPxPix = 256;                       // pixels per work item
constexpr int kPxM = 128;                         // rows of A per work item
constexpr int kPxKC = 32;                         // K per step
constexpr int kPxLine = kPxKC + 8;                // bf16 per (piece, row) line (16 bytes of padding: conflict-free b128 reads)
constexpr int kPxWBytes = 32768;                  // one step's A pieces in global memory AND in LDS: [3][128][40] bf16 =
                                                  // 30 720 bytes, padded to 32 fetch instructions of 1 KB
constexpr int kPxGroup = 1024 + 128;              // a fetch instruction's 1 KB of rows + 128 bytes (the two lane halves of
                                                  // a K-step read groups g and g + 1: 32 banks apart)
constexpr int kPxSlot = 4 * kPxGroup;             // a wave's rows of one step
constexpr size_t kPxLds = (size_t)2 * kPxWBytes + (size_t)8 * 2 * kPxSlot;  // 64 KB + 72 KB

struct PxArgs {
  const float* x;
  const __bf16* wpk;   // [row tile][step][32 KB]
  const float* bias;
  float* out;
  int cin, hi, wi, wv, ho, wo, ctot, coff, relu;
  int ptiles, tpp;     // pixel tiles, tiles per plane (modes 1, 2: of the input plane; mode 0: of the output plane)
  int nmt, nsteps;     // row tiles per pixel tile, steps per item
  int nslots, ipw;     // slots per XCD lane = ceil(ptiles / 8) * nmt; slots per workgroup
  int bias_n;          // output channels of the layer
  unsigned x_bytes, out_bytes, w_bytes;
};

template <int MODE>
__global__ __launch_bounds__(kPxThreads, 1) void patch_gemm_x3_kernel(PxArgs a) {
  constexpr int NC = kPxM / 32, S = kPxKC / 16;
  constexpr int NST = MODE == 2 ? 32 : 64;  // store instructions of an item's epilogue
  extern __shared__ __attribute__((aligned(16))) unsigned char px_smem[];
  unsigned char* Wl = px_smem;                                              // [2][32 KB]
  unsigned char* Bl = px_smem + 2 * kPxWBytes;                              // [8 waves][2 slots][kPxSlot]
  float* bias_s = reinterpret_cast<float*>(px_smem + kPxLds);               // [bias_n]
  const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(wave_id());  // (uniform: item arithmetic on the SALU)
  const int l31 = lane & 31, kh = lane >> 5;
  const int xcd = blockIdx.x & 7, s0 = (blockIdx.x >> 3) * a.ipw;
  const int nit = min(a.ipw, a.nslots - s0);  // items of this workgroup
  if (nit <= 0) return;
  const int iplane = a.hi * a.wi, oplane = a.ho * a.wo;
  const int nsteps = a.nsteps;
  unsigned char* myB = Bl + wave * 2 * kPxSlot;

  // item i of the workgroup -> (pixel tile, row tile); a pixel tile past the end is computed on zeros and never stored
  auto item_pt = [&](int i) { return ((s0 + i) / a.nmt) * 8 + xcd; };
  auto item_mt = [&](int i) { return (s0 + i) % a.nmt; };
  // Row fetches of item i.  One instruction = 64 lanes x 16 bytes: mode 0 -> 4 channels x 256 bytes (the 64 input pixels
  // under the wave's 32 output pixels, which share an output row: wo % 32 == 0), modes 1, 2 -> 8 channels x 128 bytes.
  // voff = the lane's (channel of the group, 16-byte piece) -- out of range where the piece lies past the plane --,
  // soff = (image, first pixel of the wave).
  auto b_base = [&](int i, unsigned& voff, unsigned& soff) {
    const int pt = item_pt(i);
    const int ptc = min(pt, a.ptiles - 1);
    const int n = ptc / a.tpp, p0 = (ptc - n * a.tpp) * kPxPix + wave * 32;
    if (MODE == 0) {
      const bool ok = pt < a.ptiles && p0 < oplane;
      const int q = min(p0, oplane - 32);
      const int oy = q / a.wo, ox = q - oy * a.wo;
      voff = ok ? 4u * (unsigned)((lane >> 4) * iplane + (lane & 15) * 4) : kPxOob;
      soff = 4u * ((unsigned)(n * a.cin) * (unsigned)iplane + (unsigned)(2 * oy * a.wi + 2 * ox));
    } else {
      const bool ok = pt < a.ptiles && p0 + (lane & 7) * 4 < iplane;
      voff = ok ? 4u * (unsigned)((lane >> 3) * iplane + (lane & 7) * 4) : kPxOob;
      soff = 4u * ((unsigned)(n * a.cin) * (unsigned)iplane + (unsigned)min(p0, iplane - 4));
    }
  };

  for (int t = threadIdx.x; t < a.bias_n; t += kPxThreads) bias_s[t] = a.bias ? a.bias[t] : 0.f;  // (published by the
                                                                                                // first barrier below)
  px_f32x16 acc[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  px_b8 bpa[S][3], bpb[S][3];

  // A pieces of (item i, step st): 32 KB, this wave's four 1 KB pieces of it
  // (sent by the younger half of the waves alone, eight pieces each: see the schedule below)
  auto fetch_w = [&](int i, int st, unsigned char* dst) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(item_mt(i) * nsteps + st) * (unsigned)kPxWBytes +
                                                       (unsigned)(wave - 4) * 8192u);
#pragma unroll
    for (int j = 0; j < 8; ++j) px_dma(a.wpk, a.w_bytes, dst + (wave - 4) * 8192 + j * 1024, lane * 16, so + j * 1024);
  };
  // the wave's rows of step st, always issued (a step past the end re-fetches the last one): four groups of 8 channels
  // (mode 0: of 4 channels x 2 taps)
  auto fetch_b = [&](unsigned voff, unsigned soff, int st, unsigned char* dst) {
    const unsigned ib = 4u * (unsigned)iplane;
    unsigned so;
    if (MODE == 0) {
      const int nch = a.cin >> 4;
      const int dy = st >= nch ? 1 : 0, c = st - dy * nch;
      so = soff + (unsigned)(c * 16) * ib + (unsigned)(dy * a.wi * 4);
    } else {
      so = soff + (unsigned)(st * kPxKC) * ib;
    }
    so = __builtin_amdgcn_readfirstlane(so);  // (uniform by construction; says so to the instruction selector, which
                                              // otherwise wraps every fetch in a waterfall loop over a VGPR offset)
#pragma unroll
    for (int g = 0; g < 4; ++g) px_dma(a.x, a.x_bytes, dst + g * kPxGroup, voff, so + (unsigned)(g * (MODE == 0 ? 4 : 8)) * ib);
  };
  // value e of K-step s = k = 16 s + 8 kh + e of the step: group 2 s + kh of the slot; modes 1, 2: channel e of the group at
  // the lane's pixel; mode 0: (channel e / 2, tap e & 1) = the 8 bytes under the lane's output pixel
  auto split_b = [&](const unsigned char* src, px_b8 (&bp)[S][3]) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const unsigned char* g = src + (2 * s + kh) * kPxGroup;
      px_f32x4 lo4, hi4;
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const px_f32x2 v = *reinterpret_cast<const px_f32x2*>(g + j * 256 + l31 * 8);
          if (j < 2) {
            lo4[2 * j] = v[0];
            lo4[2 * j + 1] = v[1];
          } else {
            hi4[2 * j - 4] = v[0];
            hi4[2 * j - 3] = v[1];
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo4[e] = *reinterpret_cast<const float*>(g + e * 128 + l31 * 4);
          hi4[e] = *reinterpret_cast<const float*>(g + (e + 4) * 128 + l31 * 4);
        }
      }
      px_split(lo4, hi4, bp[s][0], bp[s][1], bp[s][2]);
    }
  };
  // An item's stores: the lane's offset is its pixel (or out of range: dropped), the channel a uniform offset -- no branch,
  // no address arithmetic; the bias comes from LDS.
  auto epilogue = [&](int i) {
    const int pt = item_pt(i), mt = item_mt(i);
    const int ptc = min(pt, a.ptiles - 1);
    const int n = ptc / a.tpp, p0 = (ptc - n * a.tpp) * kPxPix;
    const int p = p0 + wave * 32 + l31;
    const unsigned ob = 4u * (unsigned)oplane;
    if (MODE == 2) {
      const int ncoh = a.nmt >> 1;                 // mt = dy * (cout / 64) + 64-channel block
      const int dy = mt / ncoh, cb = mt - dy * ncoh;
      const int pc = min(p, iplane - 1);
      const int y = pc / a.wi, xx = pc - y * a.wi;
      const bool ok = pt < a.ptiles && p < iplane && xx < a.wv;
      const unsigned voff = ok ? 4u * (unsigned)(4 * kh * oplane + 2 * y * a.wo + 2 * xx) : kPxOob;
      const unsigned so = ((unsigned)(n * a.ctot + a.coff + cb * 64)) * ob + 4u * (unsigned)(dy * a.wo);
      const float* bl = bias_s + cb * 64 + 4 * kh;
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const px_f32x4 b4 = *reinterpret_cast<const px_f32x4*>(bl + i2 * 32 + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v0 = acc[i2][4 * q + e] + b4[e], v1 = acc[i2 + 2][4 * q + e] + b4[e];
            if (a.relu) {
              v0 = fmaxf(v0, 0.f);
              v1 = fmaxf(v1, 0.f);
            }
            px_st2(a.out, a.out_bytes, voff, so + (unsigned)(i2 * 32 + 8 * q + e) * ob, v0, v1);
          }
        }
    } else {
      const bool ok = pt < a.ptiles && p < oplane;
      const unsigned voff = ok ? 4u * (unsigned)(4 * kh * oplane + p) : kPxOob;
      const unsigned so = ((unsigned)(n * a.ctot + a.coff + mt * kPxM)) * ob;
      const float* bl = bias_s + mt * kPxM + 4 * kh;
#pragma unroll
      for (int i2 = 0; i2 < NC; ++i2)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const px_f32x4 b4 = *reinterpret_cast<const px_f32x4*>(bl + i2 * 32 + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[i2][4 * q + e] + b4[e];
            if (a.relu) v = fmaxf(v, 0.f);
            px_st1(a.out, a.out_bytes, voff, so + (unsigned)(i2 * 32 + 8 * q + e) * ob, v);
          }
        }
    }
#pragma unroll
    for (int i2 = 0; i2 < NC; ++i2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i2][r] = 0.f;
  };

  // cursors over the workgroup's stream of steps: cur (multiplied now), nxt (its rows were sent a step ago and are cut
  // during this step), far (its rows are sent now).  Rows of stream step t live in slot t & 1 of the wave.
  int ci = 0, cs = 0, ni = 0, ns = 0, fi = 0, fs = 0, buf = 0;
  auto advance = [&](int& i, int& st) {
    if (++st == nsteps) {
      st = 0;
      ++i;
    }
  };
  advance(ni, ns);
  unsigned fvoff, fsoff;
  b_base(0, fvoff, fsoff);
  if (wave >= 4) fetch_w(0, 0, Wl);
  fetch_b(fvoff, fsoff, 0, myB);
  if (ni != 0) b_base(min(ni, nit - 1), fvoff, fsoff);
  fetch_b(fvoff, fsoff, ni < nit ? ns : nsteps - 1, myB + kPxSlot);
  fi = ni;
  fs = ns;
  PX_VMCNT(4);  // the A pieces and the rows of step 0 have landed (the rows of step 1 may still travel)
  split_b(myB, bpa);
  px_lds_barrier();
  // Waves w and w + 4 share a SIMD and the older one wins the matrix pipe: the two MFMA streams of a SIMD run one after the
  // other.  The older half multiplies at once and does everything else behind its MFMAs; the younger half does everything
  // else FIRST, under the older half's MFMAs:
  //   older  (waves 0-3): MFMAs | rows t + 1 have landed: cut them | send own rows t + 2        | stores | barrier
  //   younger (waves 4-7): send A t + 1 (all of it) and own rows t + 2 | cut rows t + 1 | MFMAs | A t + 1 landed | stores |
  // A fetch that finds the memory pipe's queue full HOLDS its wave (conv_f16.hip's grouped kernel measured it: 3100 cycles
  // of sending in front of 3300 of MFMAs): with all eight waves sending at the top of the step -- the first schedule --
  // nothing multiplied for 700-1700 cycles of every step (cycle stamps, profiles/r06_patch_x3_stamps.txt).
  // Fetches and stores complete in order, so the waits are counts of what is younger: the younger half needs rows t + 1
  // behind its 12 fetches (vmcnt(12); or, when step t - 1 ended an item, behind NST stores too: capped at 63); after its
  // MFMAs, A t + 1 with only the 4 row fetches younger (vmcnt(4)).  The older half sent rows t + 1 in the middle of step
  // t - 1, younger than them are only that step's stores (vmcnt(NST) or 0).
  // (Tried and dropped: the older half sending the A pieces behind its MFMAs, where it waits 1500-1900 cycles at the
  // barrier anyway -- 109 -> 124 us for the k2 s2 level: the pieces then arrive late for the barrier.)
  const bool cut_first = wave >= 4;
  bool after_epi = false;
  auto step = [&](unsigned char* near, unsigned char* far, px_b8 (&bcur)[S][3], px_b8 (&bnext)[S][3]) {
    const bool more = ni < nit;
    advance(fi, fs);
    if (fs == 0 && fi < nit) b_base(fi, fvoff, fsoff);
    if (cut_first) {
      fetch_w(more ? ni : ci, more ? ns : cs, Wl + (buf ^ 1) * kPxWBytes);
      fetch_b(fvoff, fsoff, fi < nit ? fs : nsteps - 1, far);
      if (more) {
        if (after_epi) PX_VMCNT((NST + 12 < 63 ? NST + 12 : 63));
        else PX_VMCNT(12);
        split_b(near, bnext);
      }
    }
    {
      // A: lane (m = l31, kh) of piece p, row block i, K-step s: 8 bf16 at line (p 128 + 32 i + l31), position 16 s + 8 kh
      const __bf16* wl = reinterpret_cast<const __bf16*>(Wl + buf * kPxWBytes) + l31 * kPxLine + kh * 8;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          px_b8 av[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) av[p] = *reinterpret_cast<const px_b8*>(wl + (p * kPxM + i * 32) * kPxLine + s * 16);
          // the small products first
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bcur[s][2], acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[2], bcur[s][0], acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bcur[s][1], acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bcur[s][1], acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bcur[s][0], acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bcur[s][0], acc[i], 0, 0, 0);
        }
      }
    }
    if (cut_first) {
      PX_VMCNT(4);
    } else {
      if (after_epi) PX_VMCNT((NST < 63 ? NST : 63));
      else PX_VMCNT(0);
      if (more) split_b(near, bnext);
      fetch_b(fvoff, fsoff, fi < nit ? fs : nsteps - 1, far);
    }
    after_epi = cs == nsteps - 1;
    if (after_epi) epilogue(ci);
    px_lds_barrier();
    buf ^= 1;
    ci = ni;
    cs = ns;
    advance(ni, ns);
  };
  while (true) {
    step(myB + kPxSlot, myB, bpa, bpb);
    if (ci >= nit) break;
    step(myB, myB + kPxSlot, bpb, bpa);
    if (ci >= nit) break;
  }
  PX_VMCNT(0);  // the fetches of the steps past the end still write this workgroup's LDS: they land before it is given up
}

template <int MODE>
static int launch_px(const PxArgs& a, hipStream_t s) {
  const size_t lds = kPxLds + (size_t)a.bias_n * sizeof(float);
  const hipError_t e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(patch_gemm_x3_kernel<MODE>), (int)(kPxLds + 4096));
  if (e != hipSuccess) return (int)e;
  const int64_t nwg = 8 * ceil_div(a.nslots, a.ipw);
  patch_gemm_x3_kernel<MODE><<<(unsigned)nwg, kPxThreads, lds, s>>>(a);
  return launch_status();
}

}  // namespace pd3

using namespace pd3;

extern "C" int pd3_patch_conv_x3_bias_relu(const float* x, const void* w_packed, const float* bias, int mode, int batch,
                                           int cin, int cout, int h, int w, int w_valid, int relu, float* out,
                                           int out_channels_total, int out_channel_offset, void* stream) {
  if (!x || !w_packed || !out || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || w_valid <= 0 || w_valid > w ||
      (mode < 2 && w_valid != w))
    return PD3_EINVAL;
  if (mode < 0 || mode > 2 || out_channel_offset < 0 || out_channel_offset + cout > out_channels_total) return PD3_EINVAL;
  if (reinterpret_cast<uintptr_t>(w_packed) % 16 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out) % 8 != 0)
    return PD3_EINVAL;
  PxArgs a;
  a.x = x;
  a.wpk = static_cast<const __bf16*>(w_packed);
  a.bias = bias;
  a.out = out;
  a.cin = cin;
  a.hi = h;
  a.wi = w;
  a.wv = w_valid;
  a.ctot = out_channels_total;
  a.coff = out_channel_offset;
  a.relu = relu;
  a.bias_n = cout;
  if (cout > 1024) return PD3_EUNSUPPORTED;  // (the layer's bias sits in LDS: 4 KB next to the 136 KB of operands)
  int64_t plane;
  if (mode == 0) {
    // (a wave's 32 output pixels share an output row; rows start 16-byte aligned)
    if (h % 2 != 0 || w % 64 != 0 || cin % 16 != 0 || cout % kPxM != 0) return PD3_EUNSUPPORTED;
    a.ho = h / 2;
    a.wo = w / 2;
    a.nmt = cout / kPxM;
    a.nsteps = 2 * (cin / 16);
    plane = (int64_t)a.ho * a.wo;
  } else if (mode == 1) {
    if (cin % kPxKC != 0 || cout % kPxM != 0 || ((int64_t)h * w) % 4 != 0) return PD3_EUNSUPPORTED;
    a.ho = h;
    a.wo = w;
    a.nmt = cout / kPxM;
    a.nsteps = cin / kPxKC;
    plane = (int64_t)h * w;
  } else {
    if (cin % kPxKC != 0 || cout % 64 != 0 || ((int64_t)h * w) % 4 != 0) return PD3_EUNSUPPORTED;
    a.ho = 2 * h;
    a.wo = 2 * w_valid;
    a.nmt = 2 * (cout / 64);
    a.nsteps = cin / kPxKC;
    plane = (int64_t)h * w;
  }
  const int64_t tpp = ceil_div(plane, kPxPix), ptiles = tpp * batch;
  // 32-bit buffer offsets, kPxOob beyond every tensor
  const int64_t xb = (int64_t)batch * cin * h * w * 4, ob = (int64_t)batch * out_channels_total * a.ho * a.wo * 4;
  const int64_t wb = (int64_t)a.nmt * a.nsteps * kPxWBytes;
  if (xb >= (int64_t)kPxOob || ob >= (int64_t)kPxOob || wb >= (int64_t)kPxOob || ptiles >= (int64_t)1 << 28)
    return PD3_EUNSUPPORTED;
  a.x_bytes = (unsigned)xb;
  a.out_bytes = (unsigned)ob;
  a.w_bytes = (unsigned)wb;
  a.tpp = (int)tpp;
  a.ptiles = (int)ptiles;
  const int64_t nslots = ceil_div(ptiles, 8) * a.nmt;
  a.nslots = (int)nslots;
  a.ipw = px_items_per_workgroup(nslots);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mode == 0) return launch_px<0>(a, s);
  if (mode == 1) return launch_px<1>(a, s);
  return launch_px<2>(a, s);
}
