// The stride-2 3x3 / pad 1 convolutions that open SECOND's blocks (second_backbone.py:72-120: 64 -> 128 at 256^2 -> 128^2,
// 128 -> 256 at 128^2 -> 64^2) in fp32 arithmetic on the bf16 matrix cores (round 6): every operand as three bf16 pieces,
// six piece products accumulated in fp32 (sparse_conv_x3.hip; error = that of the fp32 kernel, tested).  The fp32 implicit
// GEMM (conv3x3.hip) ran these at 0.71-0.73 of the fp32 pipe (342 + 328 us per 16 frames); the bf16 pipe gives six
// products at 2.7x that pipe's rate.
//
// Work item = 8 output rows x 32 columns x 128 output channels; wave w owns output row oy0 + w (32 pixels = the MFMA's
// N) for all four 32-channel blocks.  A step = (16 input channels, kernel row ky): K = 48 = three K-steps, one per kx, each
// over the 16 channels -- lane (pixel n, half kh) holds channels 8 kh .. 8 kh + 7 at input column 2 (ox0 + n) + kx - 1 of
// input row 2 oy + ky - 1.  Nothing goes through registers on its way in (conv_patch_x3.hip found 4-byte loads into
// registers bound by their issue): per step a wave fetches ITS input row -- 16 channels x the 64 aligned columns
// 2 ox0 .. 2 ox0 + 63, four buffer_load_dwordx4 ... lds, plus the one column left of them (kx = 0 of its first pixel; zeros
// at the image border) as a fifth, 4-byte one -- into its private slot, and a sixth of the step's A pieces (cut and padded on
// the host, [piece][128][48 + 8] bf16).  The slot is single: the rows of step t + 2 are sent right after the values of step
// t + 1 were read out of it and cut, a whole step before they are needed.  Padding needs no branch: an input row above the
// image is a fetch at an out-of-range offset (zeros).
// Persistent like conv_patch_x3.hip: a workgroup walks `ipw` consecutive (pixel tile, channel tile) items as one stream of
// steps.  Summation order fixed (channel chunks, ky, kx, pieces small to large): run-to-run identical.
#include "../../include/paddle3d_amd.h"
#include "bf16x3.hpp"
#include "common.hpp"

namespace pd3 {

constexpr int kS2Threads = 512;
constexpr int kS2M = 128;
constexpr int kS2K = 48;                            // K per step: 3 kx x 16 channels
constexpr int kS2Line = kS2K + 8;                   // bf16 per (piece, row) line: 112 bytes, conflict-free b128 reads
constexpr int kS2WBytes = 49152;                    // a step's A pieces: [3][128][56] bf16 = 43 008 bytes, padded to 48 fetches
constexpr int kS2Slot = 4096 + 256;                 // a wave's rows: [16 channels][64 columns] + the left column of each
constexpr size_t kS2Lds = (size_t)2 * kS2WBytes + (size_t)8 * kS2Slot;  // 96 KB + 34 KB

struct S2Args {
  const float* x;
  const __bf16* wpk;  // [channel tile][step][48 KB]
  const float* bias;
  float* out;
  int cin, hi, wi, ho, wo, wov, relu;  // wi / wo: row pitches of x / out; wov: real output width (columns behind it are zero)
  int ptiles, tx, ty;  // pixel tiles; tiles per output row, per image column
  int nmt, nsteps, nslots, ipw, bias_n;
  unsigned x_bytes, out_bytes, w_bytes;
};

__global__ __launch_bounds__(kS2Threads, 1) void conv3x3_s2_x3_kernel(S2Args a) {
  constexpr int NC = kS2M / 32, S = 3;  // (an item ends with 64 stores per lane: the capped vmcnt(63) below)
  extern __shared__ __attribute__((aligned(16))) unsigned char s2_smem[];
  unsigned char* Wl = s2_smem;                                             // [2][48 KB]
  float* bias_s = reinterpret_cast<float*>(s2_smem + kS2Lds);              // [bias_n]
  const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int l31 = lane & 31, kh = lane >> 5;
  unsigned char* myB = s2_smem + 2 * kS2WBytes + wave * kS2Slot;           // [16][64] floats, then [64] (16 used)
  const int xcd = blockIdx.x & 7, s0 = (blockIdx.x >> 3) * a.ipw;
  const int nit = min(a.ipw, a.nslots - s0);
  if (nit <= 0) return;
  const int iplane = a.hi * a.wi, oplane = a.ho * a.wo;
  const int nsteps = a.nsteps;

  auto item_pt = [&](int i) { return ((s0 + i) / a.nmt) * 8 + xcd; };
  auto item_mt = [&](int i) { return (s0 + i) % a.nmt; };
  // item -> (image, output row of this wave, first column)
  auto item_pos = [&](int i, int& n, int& oy, int& ox0) {
    const int pt = min(item_pt(i), a.ptiles - 1);
    const int per = a.tx * a.ty;
    n = pt / per;
    const int r = pt - n * per;
    oy = (r / a.tx) * 8 + wave;
    ox0 = (r - (r / a.tx) * a.tx) * 32;
  };

  for (int t = threadIdx.x; t < a.bias_n; t += kS2Threads) bias_s[t] = a.bias ? a.bias[t] : 0.f;
  px_f32x16 acc[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  px_b8 bpa[S][3], bpb[S][3];

  // (sent by the younger half of the waves alone, twelve pieces each: a fetch that finds the memory pipe's queue full holds
  // its wave, and the younger half waits for the older half's MFMAs anyway -- conv_patch_x3.hip)
  auto fetch_w = [&](int i, int st, unsigned char* dst) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(item_mt(i) * nsteps + st) * (unsigned)kS2WBytes +
                                                       (unsigned)(wave - 4) * 12288u);
#pragma unroll
    for (int j = 0; j < 12; ++j) px_dma(a.wpk, a.w_bytes, dst + (wave - 4) * 12288 + j * 1024, lane * 16, so + j * 1024);
  };
  // The wave's input row of step st = (chunk c, ky) of item i: lane l of fetch g = (channel 4 g + l / 16, columns
  // 2 ox0 + 4 (l % 16) ..); the fifth fetch = channel l's column 2 ox0 - 1 for l < 16.  A row above the image (or a tile
  // past the end) and the column left of the image are out-of-range offsets: zeros.
  auto fetch_b = [&](int i, int st) {
    int n, oy, ox0;
    item_pos(i, n, oy, ox0);
    const int c = st / 3, ky = st - 3 * c;
    const int iy = 2 * oy + ky - 1;
    const bool row_ok = item_pt(i) < a.ptiles && iy >= 0 && iy < a.hi;
    const unsigned ib = 4u * (unsigned)iplane;
    const unsigned row = __builtin_amdgcn_readfirstlane(
        4u * ((unsigned)(n * a.cin + c * 16) * (unsigned)iplane + (unsigned)(max(iy, 0) * a.wi + 2 * ox0)));
    // (a piece past the row's pitch is out of range; the pad columns inside the pitch hold zeros by convention)
    const unsigned voff = row_ok && 2 * ox0 + (lane & 15) * 4 < a.wi
                              ? (unsigned)(lane >> 4) * ib + (unsigned)(lane & 15) * 16u
                              : kPxOob;
#pragma unroll
    for (int g = 0; g < 4; ++g) px_dma(a.x, a.x_bytes, myB + g * 1024, voff, row + (unsigned)(4 * g) * ib);
    const bool left_ok = row_ok && ox0 > 0 && lane < 16;
    const unsigned voff1 = left_ok ? (unsigned)lane * ib : kPxOob;
    px_dma4(a.x, a.x_bytes, myB + 4096, voff1, ox0 > 0 ? row - 4u : row);
  };
  // K-step kx, value e: channel 8 kh + e at column 2 n + kx - 1 relative to 2 ox0 (column -1 = the slot's left column)
  const unsigned char* rd = myB + (8 * kh) * 256 + l31 * 8;
  auto split_b = [&](px_b8 (&bp)[S][3]) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      px_f32x4 lo4, hi4;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v;
        if (kx == 0) {
          const unsigned char* p = l31 == 0 ? myB + 4096 + (8 * kh + e) * 4 : rd + e * 256 - 4;
          v = *reinterpret_cast<const float*>(p);
        } else {
          v = *reinterpret_cast<const float*>(rd + e * 256 + (kx - 1) * 4);
        }
        if (e < 4) lo4[e] = v;
        else hi4[e - 4] = v;
      }
      px_split(lo4, hi4, bp[kx][0], bp[kx][1], bp[kx][2]);
    }
  };
  auto epilogue = [&](int i) {
    int n, oy, ox0;
    item_pos(i, n, oy, ox0);
    const int mt = item_mt(i);
    const bool ok = item_pt(i) < a.ptiles && oy < a.ho && ox0 + l31 < a.wo;
    const bool pad = ox0 + l31 >= a.wov;  // a pad column of the output's pitch: zero
    const unsigned ob = 4u * (unsigned)oplane;
    const unsigned voff = ok ? 4u * (unsigned)(4 * kh * oplane + oy * a.wo + ox0 + l31) : kPxOob;
    const unsigned so = (unsigned)(n * (a.nmt * kS2M) + mt * kS2M) * ob;
    const float* bl = bias_s + mt * kS2M + 4 * kh;
#pragma unroll
    for (int i2 = 0; i2 < NC; ++i2)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const px_f32x4 b4 = *reinterpret_cast<const px_f32x4*>(bl + i2 * 32 + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i2][4 * q + e] + b4[e];
          if (a.relu) v = fmaxf(v, 0.f);
          px_st1(a.out, a.out_bytes, voff, so + (unsigned)(i2 * 32 + 8 * q + e) * ob, pad ? 0.f : v);
        }
      }
#pragma unroll
    for (int i2 = 0; i2 < NC; ++i2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i2][r] = 0.f;
  };

  // cursors: cur (multiplied now), nxt (its row is in the slot or on its way, cut during this step), far (sent when the
  // slot has been read)
  int ci = 0, cs = 0, ni = 0, ns = 0, fi = 0, fs = 0, buf = 0;
  auto advance = [&](int& i, int& st) {
    if (++st == nsteps) {
      st = 0;
      ++i;
    }
  };
  advance(ni, ns);
  if (wave >= 4) fetch_w(0, 0, Wl);
  fetch_b(0, 0);
  PX_VMCNT(0);
  split_b(bpa);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (see cut_and_send)
  fi = ni;
  fs = ns;
  fetch_b(min(fi, nit - 1), fi < nit ? fs : nsteps - 1);  // (the slot was read: the row of step 1 goes out)
  px_lds_barrier();
  // Younger half (waves 4-7), in issue order: [row t + 1: 5] [stores of the item that ended with step t - 1: NST] | step t:
  // [A t + 1: 12, all of it] -- wait for the row (12 younger, or NST + 12: capped at 63), cut, send row t + 2 [5]; MFMAs; wait
  // for A t + 1 (5 younger).  Older half: MFMAs at once; then everything of its own has landed (vmcnt(0)); cut; send row
  // t + 2.
  const bool cut_first = wave >= 4;
  bool after_epi = false;
  auto step = [&](px_b8 (&bcur)[S][3], px_b8 (&bnext)[S][3]) {
    const bool more = ni < nit;
    if (cut_first) fetch_w(more ? ni : ci, more ? ns : cs, Wl + (buf ^ 1) * kS2WBytes);
    auto cut_and_send = [&]() {
      if (more) split_b(bnext);
      // the slot is single: its reads are complete before the next row is sent into it (the compiler does not see that the
      // fetch writes what split_b read -- without this it moved reads behind the fetch)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      advance(fi, fs);
      fetch_b(min(fi, nit - 1), fi < nit ? fs : nsteps - 1);
    };
    if (cut_first) {
      if (after_epi) PX_VMCNT(63);
      else PX_VMCNT(12);
      cut_and_send();
    }
    {
      const __bf16* wl = reinterpret_cast<const __bf16*>(Wl + buf * kS2WBytes) + l31 * kS2Line + kh * 8;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          px_b8 av[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) av[p] = *reinterpret_cast<const px_b8*>(wl + (p * kS2M + i * 32) * kS2Line + s * 16);
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
      PX_VMCNT(5);
    } else {
      PX_VMCNT(0);
      cut_and_send();
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
    step(bpa, bpb);
    if (ci >= nit) break;
    step(bpb, bpa);
    if (ci >= nit) break;
  }
  PX_VMCNT(0);  // the fetches past the end still write this workgroup's LDS
}

}  // namespace pd3

using namespace pd3;

extern "C" int pd3_conv3x3_s2_x3_bias_relu(const float* x, const void* w_packed, const float* bias, int batch, int cin,
                                           int cout, int h, int w, int w_valid, int relu, float* out, int out_w,
                                           void* stream) {
  if (!x || !w_packed || !out || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || w_valid <= 0 || w_valid > w ||
      out_w < w_valid / 2)
    return PD3_EINVAL;
  if (reinterpret_cast<uintptr_t>(w_packed) % 16 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out) % 4 != 0)
    return PD3_EINVAL;
  if (cin % 16 != 0 || cout % kS2M != 0 || cout > 1024 || h % 2 != 0 || w_valid % 2 != 0 || w % 4 != 0)
    return PD3_EUNSUPPORTED;
  S2Args a;
  a.x = x;
  a.wpk = static_cast<const __bf16*>(w_packed);
  a.bias = bias;
  a.out = out;
  a.cin = cin;
  a.hi = h;
  a.wi = w;
  a.ho = h / 2;
  a.wo = out_w;
  a.wov = w_valid / 2;
  a.relu = relu;
  a.bias_n = cout;
  a.nmt = cout / kS2M;
  a.nsteps = 3 * (cin / 16);
  a.tx = (int)ceil_div(a.wo, 32);  // (the pad columns of the output's pitch are written too: zeros)
  a.ty = (int)ceil_div(a.ho, 8);
  const int64_t ptiles = (int64_t)batch * a.tx * a.ty;
  const int64_t xb = (int64_t)batch * cin * h * w * 4, ob = (int64_t)batch * cout * a.ho * a.wo * 4;
  const int64_t wb = (int64_t)a.nmt * a.nsteps * kS2WBytes;
  if (xb >= (int64_t)kPxOob || ob >= (int64_t)kPxOob || wb >= (int64_t)kPxOob || ptiles >= (int64_t)1 << 28)
    return PD3_EUNSUPPORTED;
  a.x_bytes = (unsigned)xb;
  a.out_bytes = (unsigned)ob;
  a.w_bytes = (unsigned)wb;
  a.ptiles = (int)ptiles;
  const int64_t nslots = ceil_div(ptiles, 8) * a.nmt;
  a.nslots = (int)nslots;
  a.ipw = px_items_per_workgroup(nslots);
  const size_t lds = kS2Lds + (size_t)cout * sizeof(float);
  const hipError_t e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(conv3x3_s2_x3_kernel), (int)(kS2Lds + 4096));
  if (e != hipSuccess) return (int)e;
  const int64_t nwg = 8 * ceil_div(nslots, a.ipw);
  conv3x3_s2_x3_kernel<<<(unsigned)nwg, kS2Threads, lds, static_cast<hipStream_t>(stream)>>>(a);
  return launch_status();
}
