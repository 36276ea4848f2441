// The fp32 sparse gather-GEMM on the bf16 matrix cores (round 5): every fp32 operand travels as THREE bf16 pieces whose
// sum IS the fp32 value, and six of the nine piece products are accumulated in fp32 -- fp32 arithmetic, not a reduced
// precision: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); both differences are exact in fp32, each piece
// rounds to nearest and carries 8 significand bits, and 3 x 8 = the 24 bits of an fp32 significand: x = hi + mid + lo
// EXACTLY, with |mid| <= 2^-8 |x| and |lo| <= 2^-16 |x|.  A piece product (8 x 8 bits) is exact in the MFMA's fp32
// accumulator; kept are hi hi, hi mid, mid hi, hi lo, mid mid, lo hi; dropped are mid lo, lo mid, lo lo: at most 2^-23 of
// the product in the worst case, 2^-24.3 at most and 2^-28 on average over millions of random pairs (tests/
// test_bf16x3_arith.py restates the scheme on the CPU) -- the size of the rounding an fp32 multiplication does by itself
// (2^-24).  tests/test_sparse_conv_gpu.py measures the kernel: against an fp64 gather-GEMM its error is that of the fp32
// matrix-core kernel (sparse_conv.hip), not that of a 16-bit format.
// Why: v_mfma_f32_16x16x4_f32 runs at 64 flop per cycle and SIMD, v_mfma_f32_32x32x16_bf16 at 1024; six of the latter per
// fp32 product are 2.7 times the fp32 pipe's rate, and a sparse convolution has no Winograd form to shrink its products
// by -- the big layers of the CenterPoint-Voxel encoder (sparse_resnet.py:115-206: 64 -> 64 and 128 -> 128 over 27
// offsets) ran at 85-105 executed TFLOP/s of the fp32 pipe's 157 (profiles/r05_sparse_layers_fp32.txt).
// NaN / Inf inputs: x - bf16(x) is NaN for an infinite x, so an Inf in the input becomes a NaN in the output (the fp32
// kernel would carry the Inf); activations of a network are finite.
//
// Shape of the kernel = the fp16 form's (sparse_conv_f16.hip): tile = 256 output rows per workgroup, weights are the
// MFMA's A operand (M = 32 output channels) read from padded LDS lines, the gathered rows its B operand (N = 32 output
// rows) loaded straight into registers -- here as fp32 (lane (row n, kh) reads KC / 2 consecutive floats of input row
// nbr[n][k]: 64 contiguous bytes with KC = 32) and cut into pieces by the lane (5.5 VALU instructions per value:
// v_cvt_pk_bf16_f32, shifts, subtractions; the value then feeds 6 Cout / 32 MFMAs).  The weights are cut once per
// parameter version (pd3_sparse_pack_weight_bf16x3: [offset][chunk][piece][co][KC / 16][2][8]).  A wave owns RB 32-row
// blocks for all Cout: RB = 1 with eight waves for Cout = 128 (64 accumulator registers, two waves per SIMD, one 90 KB
// workgroup per CU), RB = 2 with four waves below.  Block-uniform offset skip and tile order as in the other forms;
// summation order fixed (offsets ascending, chunks, K-steps, pieces): run-to-run identical.
// Measured on the way (profiles/r05_sparse_x3_forms.txt): rows cut by the layer that computes them and
// gathered as pieces (6 B per value, no VALU in the loop) are slower than this form -- the gather bounds the kernel;
// an XCD-contiguous tile range is slower than round-robin tiles; gathers two steps ahead pay, but only with a constant
// number of loads in flight (see fetch_b).  Used from 64 output channels on (ops.sparse_conv3d.bf16x3_pays): 32 -> 32
// spends 7 VALU instructions per MFMA here and is faster on the fp32 kernel.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"

namespace pd3 {

typedef __bf16 sx_b8 __attribute__((ext_vector_type(8)));
typedef float sx_f32x16 __attribute__((ext_vector_type(16)));
typedef float sx_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSxRows = 256;  // output rows per workgroup
constexpr int kSxMaxK = 27;

struct SpGemmX3Args {
  const float* in;           // [n_in, cin]
  const int32_t* nbr;        // [n_out, K]
  const __bf16* wpk;         // packed weight pieces, see pd3_sparse_pack_weight_bf16x3
  const float *bias, *scale, *shift;  // [cout] or null
  const float* residual;     // [n_out, cout] or null
  float* out;                // [n_out, cout]
  const int* n_out_dev;
  int n_out_cap, K, cin, cout, relu;
  const int32_t* order;
};

// x = hi + mid + lo exactly, every piece a bf16 (round to nearest even)
__device__ __forceinline__ void sx_split(const sx_f32x4 lo4, const sx_f32x4 hi4, sx_b8& h, sx_b8& m, sx_b8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = e < 4 ? lo4[e & 3] : hi4[e & 3];
    const __bf16 a = (__bf16)x;
    const float r1 = x - (float)a;
    const __bf16 b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    h[e] = a;
    m[e] = b;
    l[e] = (__bf16)r2;
  }
}

template <int NC, int KC, int RB>
__global__ __launch_bounds__(64 * (8 / RB), RB == 1 ? 1 : 2) void sp_gemm_rows_x3_kernel(SpGemmX3Args a) {
  constexpr int THREADS = 64 * (8 / RB);
  constexpr int S = KC / 16;               // K-steps (MFMAs along K) per chunk
  constexpr int COUT = NC * 32;
  constexpr int LINE = KC + 8;             // bf16 per (piece, output channel) line in LDS (16 bytes of padding)
  constexpr int WSZ = 3 * COUT * LINE;     // bf16 of one staged chunk (three pieces)
  constexpr int WQ = 3 * COUT * KC / 8;    // 16-byte pieces of a chunk in global memory
  constexpr int WPT = (WQ + THREADS - 1) / THREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char sx_smem[];
  __bf16* Ws = reinterpret_cast<__bf16*>(sx_smem);                           // [2][WSZ]
  int* nbs = reinterpret_cast<int*>(Ws + 2 * WSZ);                           // [256][K]
  uint32_t* masks = reinterpret_cast<uint32_t*>(nbs + kSxRows * a.K);        // [0] workgroup, [1 + 32-row block]
  int* rows = reinterpret_cast<int*>(masks + 16);                            // [256]
  const int lane = lane_id(), wave = wave_id();
  const int l31 = lane & 31, kh = lane >> 5;
  const int K = a.K, cin = a.cin;
  const int n_out = a.n_out_dev ? min(*a.n_out_dev, a.n_out_cap) : a.n_out_cap;
  const int tile = sp_window_tile(blockIdx.x, (n_out + kSxRows - 1) / kSxRows, 8192 / kSxRows);  // (a window per XCD)
  if (tile < 0) return;
  const int row0 = tile * kSxRows;
  if (threadIdx.x < kSxRows) {
    int r = row0 + (int)threadIdx.x;
    if (a.order) r = a.order[row0 + threadIdx.x];
    rows[threadIdx.x] = r >= 0 && r < n_out ? r : -1;
    if (threadIdx.x < 9) masks[threadIdx.x] = 0u;
  }
  __syncthreads();
  // the tile's rulebook rows -> LDS.  Element e = (row i, offset k) in rulebook order, eight loads per thread in flight at a
  // time (round 6: the loop as first written -- one (row, k < 32) element per trip, the load behind an LDS read -- ran its
  // trips one memory round trip after the other: 35 k cycles per tile, unnoticed beside 54 steps of a 27-offset layer,
  // 40 % of a tile of the 9-offset pillar convolution)
  {
    const int total = kSxRows * K;
    for (int e0 = threadIdx.x; e0 < total; e0 += 8 * THREADS) {
      int v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * THREADS;
        const int i = min(e, total - 1) / K, k = min(e, total - 1) - i * K;
        const int r = rows[i];
        v[u] = a.nbr[(int64_t)max(r, 0) * K + k];
        v[u] = r >= 0 ? v[u] : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * THREADS;
        if (e < total) nbs[e] = v[u];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < kSxRows) {  // which offsets does each block of 32 rows need (thread = row)
    uint32_t m = 0;
    for (int k = 0; k < K; ++k) m |= nbs[threadIdx.x * K + k] >= 0 ? 1u << k : 0u;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) m |= (uint32_t)__shfl_xor((int)m, d, kWave);
    if (l31 == 0 && m) {
      atomicOr(&masks[1 + (threadIdx.x >> 5)], m);
      atomicOr(&masks[0], m);
    }
  }
  __syncthreads();
  const uint32_t wg_mask = masks[0];
  uint32_t blk_mask[RB], wave_mask = 0u;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    blk_mask[rb] = masks[1 + RB * wave + rb];
    wave_mask |= blk_mask[rb];
  }
  const int nchunks = cin / KC;

  sx_f32x16 acc[NC][RB];
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][rb][r] = 0.f;
  sx_b8 wreg[WPT];
  // the pieces of the current step's gathered values and (RB == 1) of the next step's: with a second set a wave can cut
  // the next step's values BEFORE its own MFMAs, i.e. while its SIMD partner multiplies (see `step`)
  sx_b8 bpa[RB][S][3], bpb[RB == 1 ? RB : 1][RB == 1 ? S : 1][3];
  sx_f32x4 raw_a[RB][S][2], raw_b[RB][S][2];     // the values of the next step and of the one behind it, as loaded
  const sx_f32x4 fz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int s = 0; s < S; ++s) raw_a[rb][s][0] = raw_a[rb][s][1] = raw_b[rb][s][0] = raw_b[rb][s][1] = fz;

  auto fetch_w = [&](int k, int c) {  // (always issued, see fetch_b)
    const sx_b8* wk = reinterpret_cast<const sx_b8*>(a.wpk) + ((int64_t)k * nchunks + c) * WQ;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = min((int)threadIdx.x + i * THREADS, WQ - 1);
      wreg[i] = wk[e];
    }
  };
  auto stash_w = [&](__bf16* dst) {  // 16-byte piece e = (piece p, co, q of its KC / 8): a plain copy into padded lines
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = threadIdx.x + i * THREADS;
      if (WQ % THREADS != 0 && e >= WQ) break;
      const int line = e / (KC / 8), q = e - line * (KC / 8);
      *reinterpret_cast<sx_b8*>(dst + line * LINE + q * 8) = wreg[i];
    }
  };
  // The loads of a step are issued WITHOUT a branch around them -- a row that is missing (or a block / a step that is not
  // needed) reads input row 0 and is zeroed when it is cut -- so that the number of loads in flight is a constant the
  // compiler can wait against (s_waitcnt vmcnt(n) with n = the loads of the step behind): with a branch it waits for
  // everything, and the second stage of the prefetch is gone.
  auto fetch_b = [&](int k, int c, sx_f32x4 (&dst)[RB][S][2], int (&jv)[RB]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const bool on = k >= 0 && ((blk_mask[rb] >> k) & 1u);
      const int j = on ? nbs[(wave * (32 * RB) + rb * 32 + l31) * K + k] : -1;
      jv[rb] = j;
      const sx_f32x4* src = reinterpret_cast<const sx_f32x4*>(a.in + (int64_t)max(j, 0) * cin + c * KC + kh * (KC / 2));
#pragma unroll
      for (int s = 0; s < S; ++s) {
        dst[rb][s][0] = src[2 * s];
        dst[rb][s][1] = src[2 * s + 1];
      }
    }
  };
  auto split_b = [&](int k, const sx_f32x4 (&src)[RB][S][2], const int (&jv)[RB], sx_b8 (&bcur)[RB][S][3]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      if (!((blk_mask[rb] >> k) & 1u)) continue;
      const bool live = jv[rb] >= 0;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        sx_f32x4 lo4 = src[rb][s][0], hi4 = src[rb][s][1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo4[e] = live ? lo4[e] : 0.f;
          hi4[e] = live ? hi4[e] : 0.f;
        }
        sx_split(lo4, hi4, bcur[rb][s][0], bcur[rb][s][1], bcur[rb][s][2]);
      }
    }
  };
  auto next_step = [&](int& k, int& c) {  // (k >= 0)
    if (++c < nchunks) return;
    c = 0;
    const uint32_t rest = k + 1 < 32 ? wg_mask >> (k + 1) : 0u;
    k = rest ? k + 1 + __builtin_ctz(rest) : -1;
  };

  // Steps (offset k, chunk c) in order; the loads of a step's gathered values go out TWO steps ahead (a step is 1500-3000
  // cycles of MFMAs, a gathered row comes from another XCD's lines in the Infinity Cache or from HBM: one step ahead, every
  // step waited for its rows -- measured, profiles/r05_sparse_x3_forms.txt), its weights one step ahead.
  int k = wg_mask ? __builtin_ctz(wg_mask) : -1, c = 0, buf = 0;
  int k2 = k, c2 = c;
  int j_a[RB], j_b[RB];
  if (k < 0) return;  // (a tile whose rows have no neighbour at all cannot exist: the centre offset is its own row)
  next_step(k2, c2);
  fetch_w(k, c);
  fetch_b(k, c, raw_a, j_a);
  fetch_b(k2, c2, raw_b, j_b);
  stash_w(Ws);
  if ((wave_mask >> k) & 1u) split_b(k, raw_a, j_a, bpa);
  __syncthreads();
  // one step: `near` holds the next step's values (in flight since the step before), `far` takes the loads of the step
  // behind it
  // RB == 1 (512 threads): waves w and w + 4 share a SIMD; the older one wins the matrix pipe, so the pair's two MFMA
  // streams run one after the other (cycle counters, 128 -> 128: MFMAs 1950 cycles for waves 0-3, which then cut their
  // next values and wait 1700 cycles at the barrier; waves 4-7: 2750 of MFMAs behind the partner's, THEN 700 cycles of
  // cutting with the pipe idle).  The younger half therefore cuts the next step's values BEFORE its MFMAs -- into the
  // second set of piece registers -- while the older half multiplies; the older half cuts after its MFMAs as before.
  const bool cut_first = RB == 1 && wave >= 4;
  auto step = [&](sx_f32x4 (&near)[RB][S][2], int (&jn)[RB], sx_f32x4 (&far)[RB][S][2], int (&jf)[RB],
                  sx_b8 (&bcur)[RB][S][3], sx_b8 (&bnext)[RB][S][3]) {
    const bool more = k2 >= 0;
    int k3 = k2, c3 = c2;
    if (more) next_step(k3, c3);
    fetch_w(more ? k2 : k, more ? c2 : c);
    fetch_b(k3, c3, far, jf);
    const bool cut = more && ((wave_mask >> k2) & 1u);
    if (cut_first && cut) split_b(k2, near, jn, bnext);
    if ((wave_mask >> k) & 1u) {
      // A: lane (m = l31, kh) of piece p, channel block i, K-step s: 8 bf16 at line (p COUT + 32 i + l31), piece 2 s + kh
      const __bf16* wl = Ws + buf * WSZ + l31 * LINE + kh * 8;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          sx_b8 av[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) av[p] = *reinterpret_cast<const sx_b8*>(wl + (p * COUT + i * 32) * LINE + s * 16);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            if (!((blk_mask[rb] >> k) & 1u)) continue;
            // the small products first
            acc[i][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bcur[rb][s][2], acc[i][rb], 0, 0, 0);
            acc[i][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[2], bcur[rb][s][0], acc[i][rb], 0, 0, 0);
            acc[i][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bcur[rb][s][1], acc[i][rb], 0, 0, 0);
            acc[i][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bcur[rb][s][1], acc[i][rb], 0, 0, 0);
            acc[i][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bcur[rb][s][0], acc[i][rb], 0, 0, 0);
            acc[i][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bcur[rb][s][0], acc[i][rb], 0, 0, 0);
          }
        }
      }
    }
    stash_w(Ws + (buf ^ 1) * WSZ);
    if (!cut_first && cut) split_b(k2, near, jn, bnext);
    __syncthreads();
    buf ^= 1;
    k = k2;
    c = c2;
    k2 = k3;
    c2 = c3;
  };
  if constexpr (RB == 1) {
    while (k >= 0) {
      step(raw_b, j_b, raw_a, j_a, bpa, bpb);
      if (k < 0) break;
      step(raw_a, j_a, raw_b, j_b, bpb, bpa);
    }
  } else {
    while (k >= 0) {  // (one set of pieces: the next step's values are cut into the set the finished MFMAs read)
      step(raw_b, j_b, raw_a, j_a, bpa, bpa);
      if (k < 0) break;
      step(raw_a, j_a, raw_b, j_b, bpa, bpa);
    }
  }
  // epilogue: D[m = (reg & 3) + 8 (reg >> 2) + 4 kh][n = l31] of block (i, rb): channel 32 i + m, output row n of the
  // wave's block rb; four consecutive channels (regs 4 q .. 4 q + 3) leave as one 16-byte store
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int row = rows[wave * (32 * RB) + rb * 32 + l31];
    if (row < 0) continue;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = i * 32 + 8 * q + 4 * kh;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][rb][4 * q + e];
        if (a.bias) {
          const sx_f32x4 b4 = *reinterpret_cast<const sx_f32x4*>(a.bias + co);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += b4[e];
        }
        if (a.scale) {
          const sx_f32x4 s4 = *reinterpret_cast<const sx_f32x4*>(a.scale + co);
          const sx_f32x4 h4 = *reinterpret_cast<const sx_f32x4*>(a.shift + co);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], s4[e], h4[e]);
        }
        if (a.residual) {
          const sx_f32x4 r4 = *reinterpret_cast<const sx_f32x4*>(a.residual + (int64_t)row * COUT + co);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += r4[e];
        }
        if (a.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<sx_f32x4*>(a.out + (int64_t)row * COUT + co) = sx_f32x4{v[0], v[1], v[2], v[3]};
      }
    }
  }
}

// weight [K, cin, cout] fp32 (Paddle layout, kd kh kw flattened) -> bf16 pieces [K][cin / KC][3][cout][KC / 16][2][8]:
// element (p, co, s, kh, e) of chunk c is piece p of W[k][c KC + kh KC / 2 + 8 s + e][co]
__global__ __launch_bounds__(256) void sp_pack_weight_x3_kernel(const float* __restrict__ w, int K, int cin, int cout,
                                                                int kc, __bf16* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)K * cin * cout;
  if (t >= total) return;
  const int e = (int)(t % 8);
  int64_t r = t / 8;
  const int khv = (int)(r % 2);
  r /= 2;
  const int s = (int)(r % (kc / 16));
  r /= (kc / 16);
  const int co = (int)(r % cout);
  r /= cout;
  const int c = (int)(r % (cin / kc));
  const int k = (int)(r / (cin / kc));
  const int ci = c * kc + khv * (kc / 2) + 8 * s + e;
  const float x = w[((int64_t)k * cin + ci) * cout + co];
  const __bf16 hi = (__bf16)x;
  const float r1 = x - (float)hi;
  const __bf16 mid = (__bf16)r1;
  const __bf16 lo = (__bf16)(r1 - (float)mid);
  // position inside the (k, c) chunk: [p][co][s][kh][e]
  const int64_t chunk = (int64_t)k * (cin / kc) + c;
  const int64_t inner = ((int64_t)co * (kc / 16) + s) * 16 + khv * 8 + e;
  const int64_t per_piece = (int64_t)cout * kc;
  __bf16* dst = out + chunk * 3 * per_piece + inner;
  dst[0] = hi;
  dst[per_piece] = mid;
  dst[2 * per_piece] = lo;
}

static inline int sx_chunk(int cin) { return cin % 32 == 0 ? 32 : 16; }

}  // namespace pd3

using namespace pd3;

extern "C" int pd3_sparse_pack_weight_bf16x3(const float* weight, int kernel_volume, int cin, int cout, void* packed,
                                             void* stream) {
  if (!weight || !packed || kernel_volume <= 0 || cin <= 0 || cout <= 0) return PD3_EINVAL;
  if (cin % 16 != 0 || (cout != 32 && cout != 64 && cout != 128)) return PD3_EUNSUPPORTED;
  const int64_t total = (int64_t)kernel_volume * cin * cout;
  sp_pack_weight_x3_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      weight, kernel_volume, cin, cout, sx_chunk(cin), static_cast<__bf16*>(packed));
  return launch_status();
}

extern "C" int pd3_sparse_conv3d_features_bf16x3(const float* in_feats, const int32_t* nbr, const int32_t* n_out,
                                                 int n_out_cap, int kernel_volume, int cin, int cout,
                                                 const void* weight_packed, const float* bias, const float* scale,
                                                 const float* shift, const float* residual, int relu,
                                                 const int32_t* order, float* out, void* stream) {
  if (!in_feats || !nbr || !weight_packed || !out || n_out_cap <= 0 || kernel_volume <= 0 || cin <= 0 || cout <= 0)
    return PD3_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return PD3_EINVAL;
  if (cin % 16 != 0 || (cout != 32 && cout != 64 && cout != 128) || kernel_volume > kSxMaxK) return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(in_feats) % 16 != 0 || reinterpret_cast<uintptr_t>(weight_packed) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out) % 16 != 0 || (residual && reinterpret_cast<uintptr_t>(residual) % 16 != 0) ||
      (bias && reinterpret_cast<uintptr_t>(bias) % 16 != 0) || (scale && reinterpret_cast<uintptr_t>(scale) % 16 != 0) ||
      (shift && reinterpret_cast<uintptr_t>(shift) % 16 != 0))
    return PD3_EINVAL;
  SpGemmX3Args a{in_feats, nbr, static_cast<const __bf16*>(weight_packed), bias, scale, shift, residual, out, n_out,
                 n_out_cap, kernel_volume, cin, cout, relu ? 1 : 0, order};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int kc = sx_chunk(cin), nc = cout / 32;
  const size_t lds = (size_t)2 * 3 * cout * (kc + 8) * sizeof(__bf16) +
                     ((size_t)kSxRows * kernel_volume + 16 + kSxRows) * sizeof(int);
  const unsigned grid = sp_window_grid(ceil_div(n_out_cap, kSxRows), 8192 / kSxRows);
  hipError_t e;
#define PD3_SX(NCV, KCV, RBV)                                                                        \
  do {                                                                                               \
    if (lds > 48 * 1024) {                                                                           \
      e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(sp_gemm_rows_x3_kernel<NCV, KCV, RBV>), (int)lds);                 \
      if (e != hipSuccess) return (int)e;                                                            \
    }                                                                                                \
    sp_gemm_rows_x3_kernel<NCV, KCV, RBV><<<grid, 64 * (8 / RBV), lds, s>>>(a);                      \
  } while (0)
  switch (nc * 100 + kc) {
    case 116: PD3_SX(1, 16, 2); break;
    case 132: PD3_SX(1, 32, 2); break;
    case 216: PD3_SX(2, 16, 2); break;
    case 232: PD3_SX(2, 32, 2); break;
    case 416: PD3_SX(4, 16, 1); break;
    case 432: PD3_SX(4, 32, 1); break;
    default: return PD3_EUNSUPPORTED;
  }
#undef PD3_SX
  return launch_status();
}
