// Mixed-precision (AMP) form of the sparse gather-GEMM: fp16 feature rows and weights on the fp16 matrix cores
// (v_mfma_f32_32x32x16_f16, fp32 accumulation), bias / folded BatchNorm / residual / ReLU fused, fp16 rows out.
// Reference: the sparse middle encoder of CenterPoint-Voxel (paddle3d/models/middle_encoders/sparse_resnet.py:115-206)
// under the reference's `amp_cfg` (level O2: fp16 activations and weights in the convolutions); the index sets, the
// rulebooks and the tile order are those of the fp32 path (sparse_conv.hip), only the arithmetic of
// pd3_sparse_conv3d_features changes.  Never the default: `SparseResNet3D.amp`, reported under its own bench workload.
//
// Shape of the kernel (the fp32 form's, re-cut for a 32x32x16 MFMA that is sixteen times faster than the fp32 one, so
// that what bounds it is the gather and LDS, not the matrix pipe):
//   * tile = 256 output rows per workgroup: wave w owns rows 64 w .. 64 w + 63 as two 32-row blocks, for all Cout;
//   * the MFMA's A operand is the WEIGHTS (M = 32 output channels), its B operand the GATHERED rows (N = 32 output
//     rows): lane (n, kh) needs 8 consecutive halfs of input row nbr[row n][k] -- one 16-byte global load straight into
//     the operand register, KC / 16 of them per step, 64 contiguous bytes of the row per lane with KC = 64 (the two kh
//     lanes of a row cover one 128-byte line) -- and D[channel][row] leaves as 8-byte stores of four channels;
//   * K assignment: MFMA k = 8 kh + e of K-step s stands for channel c KC + kh KC / 2 + 8 s + e; the weights are packed
//     on the host to match ([offset][chunk][co][s][kh][8]: pd3_sparse_pack_weight_f16), staged per (offset, chunk) step
//     in a double-buffered LDS tile (lines padded by 16 bytes: conflict-free ds_read_b128), one barrier per step;
//   * a 32-row block runs an offset if any of its rows has that neighbour (block-uniform skip), the tile order groups
//     rows with similar masks.  The summation order (offsets ascending, chunks, K-steps) is fixed: run-to-run identical.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"

namespace pd3 {

typedef _Float16 sf_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 sf_h4 __attribute__((ext_vector_type(4)));
typedef float sf_f32x16 __attribute__((ext_vector_type(16)));
typedef float sf_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSfRows = 256;   // output rows per workgroup
constexpr int kSfMaxK = 27;

struct SpGemmF16Args {
  const _Float16* in;        // [n_in, cin]
  const int32_t* nbr;        // [n_out, K]
  const _Float16* wpk;       // packed weights, see pd3_sparse_pack_weight_f16
  const float *bias, *scale, *shift;  // [cout] or null
  const _Float16* residual;  // [n_out, cout] or null
  void* out;                 // [n_out, cout] fp16, or fp32 with out_f32
  const int* n_out_dev;
  int n_out_cap, K, cin, cout, relu, out_f32;
  const int32_t* order;
  int out_ld, out_off;       // row stride and first column of `out` (a channel slice of a wider row-major matrix)
};

template <int NC, int KC>
__global__ __launch_bounds__(256, 2) void sp_gemm_rows_f16_kernel(SpGemmF16Args a) {
  constexpr int S = KC / 16;               // K-steps (MFMAs along K) per chunk
  constexpr int COUT = NC * 32;
  constexpr int LINE = KC + 8;             // halfs per output channel in LDS (16 bytes of padding)
  constexpr int WSZ = COUT * LINE;         // halfs of one staged W chunk
  constexpr int WQ = COUT * KC / 8;        // 16-byte pieces of a W chunk in global memory
  constexpr int WPT = (WQ + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[];
  _Float16* Ws = reinterpret_cast<_Float16*>(sf_smem);                       // [2][WSZ]
  int* nbs = reinterpret_cast<int*>(Ws + 2 * WSZ);                           // [256][K]
  uint32_t* masks = reinterpret_cast<uint32_t*>(nbs + kSfRows * a.K);        // [0] workgroup, [1 + 32-row block]
  int* rows = reinterpret_cast<int*>(masks + 16);                            // [256]
  const int lane = lane_id(), wave = wave_id();
  const int l31 = lane & 31, kh = lane >> 5;
  const int K = a.K, cin = a.cin;
  const int n_out = a.n_out_dev ? min(*a.n_out_dev, a.n_out_cap) : a.n_out_cap;
  const int tile = sp_window_tile(blockIdx.x, (n_out + kSfRows - 1) / kSfRows, 8192 / kSfRows);  // (a window per XCD)
  if (tile < 0) return;
  const int row0 = tile * kSfRows;
  {
    int r = row0 + (int)threadIdx.x;
    if (a.order) r = a.order[row0 + threadIdx.x];
    rows[threadIdx.x] = r >= 0 && r < n_out ? r : -1;
    if (threadIdx.x < 9) masks[threadIdx.x] = 0u;
  }
  __syncthreads();
{  // rulebook rows of the tile -> LDS, eight loads per thread in flight (see sparse_conv_x3.hip)
    const int total = kSfRows * K;
    for (int e0 = threadIdx.x; e0 < total; e0 += 8 * 256) {
      int v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = min(e0 + u * 256, total - 1);
        const int i = e / K, k = e - i * K;
        const int r = rows[i];
        v[u] = a.nbr[(int64_t)max(r, 0) * K + k];
        v[u] = r >= 0 ? v[u] : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (e0 + u * 256 < total) nbs[e0 + u * 256] = v[u];
    }
  }
  __syncthreads();
  {  // which offsets does each block of 32 rows need (thread = row)
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
  const uint32_t blk_mask[2] = {masks[1 + 2 * wave], masks[2 + 2 * wave]};
  const uint32_t wave_mask = blk_mask[0] | blk_mask[1];
  const int nchunks = cin / KC;

  sf_f32x16 acc[NC][2];
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][rb][r] = 0.f;
  sf_h8 wreg[WPT], bcur[2][S], bnext[2][S];
  const sf_h8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int s = 0; s < S; ++s) bcur[rb][s] = bnext[rb][s] = hz;

  auto fetch_w = [&](int k, int c) {
    const sf_h8* wk = reinterpret_cast<const sf_h8*>(a.wpk) + ((int64_t)k * nchunks + c) * WQ;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = threadIdx.x + i * 256;
      wreg[i] = hz;
      if (WQ % 256 == 0 || e < WQ) wreg[i] = wk[e];
    }
  };
  auto stash_w = [&](_Float16* dst) {  // piece e = (co, 8 halfs q of its KC): a plain copy into padded lines
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = threadIdx.x + i * 256;
      if (WQ % 256 != 0 && e >= WQ) break;
      const int co = e / (KC / 8), q = e - co * (KC / 8);
      *reinterpret_cast<sf_h8*>(dst + co * LINE + q * 8) = wreg[i];
    }
  };
  auto fetch_b = [&](int k, int c, sf_h8 (&dst)[2][S]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      if (!((blk_mask[rb] >> k) & 1u)) continue;  // block-uniform
      const int j = nbs[(wave * 64 + rb * 32 + l31) * K + k];
#pragma unroll
      for (int s = 0; s < S; ++s) dst[rb][s] = hz;
      if (j >= 0) {
        const sf_h8* src = reinterpret_cast<const sf_h8*>(a.in + (int64_t)j * cin + c * KC + kh * (KC / 2));
#pragma unroll
        for (int s = 0; s < S; ++s) dst[rb][s] = src[s];
      }
    }
  };
  auto next_step = [&](int& k, int& c) {
    if (++c < nchunks) return;
    c = 0;
    const uint32_t rest = k + 1 < 32 ? wg_mask >> (k + 1) : 0u;
    k = rest ? k + 1 + __builtin_ctz(rest) : -1;
  };

  int k = wg_mask ? __builtin_ctz(wg_mask) : -1, c = 0, buf = 0;
  if (k >= 0) {
    fetch_w(k, c);
    if ((wave_mask >> k) & 1u) fetch_b(k, c, bcur);
    stash_w(Ws);
  }
  __syncthreads();
  while (k >= 0) {
    int k2 = k, c2 = c;
    next_step(k2, c2);
    const bool more = k2 >= 0;
    const bool need = (wave_mask >> k) & 1u, need2 = more && ((wave_mask >> k2) & 1u);
    if (more) fetch_w(k2, c2);
    if (need2) fetch_b(k2, c2, bnext);
    if (need) {
      const bool n0 = (blk_mask[0] >> k) & 1u, n1 = (blk_mask[1] >> k) & 1u;
      // A: lane (m = l31, kh) of channel block i, K-step s: 8 halfs at line (32 i + l31), piece 2 s + kh
      const _Float16* wl = Ws + buf * WSZ + l31 * LINE + kh * 8;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        sf_h8 av[S];
#pragma unroll
        for (int s = 0; s < S; ++s) av[s] = *reinterpret_cast<const sf_h8*>(wl + i * 32 * LINE + s * 16);
        if (n0) {
#pragma unroll
          for (int s = 0; s < S; ++s)
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bcur[0][s], acc[i][0], 0, 0, 0);
        }
        if (n1) {
#pragma unroll
          for (int s = 0; s < S; ++s)
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bcur[1][s], acc[i][1], 0, 0, 0);
        }
      }
    }
    if (more) stash_w(Ws + (buf ^ 1) * WSZ);
    if (need2) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int s = 0; s < S; ++s) bcur[rb][s] = bnext[rb][s];
    }
    __syncthreads();
    buf ^= 1;
    k = k2;
    c = c2;
  }
  // epilogue: D[m = (reg & 3) + 8 (reg >> 2) + 4 kh][n = l31] of block (i, rb): channel 32 i + m, output row n of the
  // wave's block rb; four consecutive channels (regs 4 q .. 4 q + 3) leave as one 8-byte (fp16) / 16-byte (fp32) store
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int row = rows[wave * 64 + rb * 32 + l31];
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
          const sf_f32x4 b4 = *reinterpret_cast<const sf_f32x4*>(a.bias + co);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += b4[e];
        }
        if (a.scale) {
          const sf_f32x4 s4 = *reinterpret_cast<const sf_f32x4*>(a.scale + co);
          const sf_f32x4 h4 = *reinterpret_cast<const sf_f32x4*>(a.shift + co);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], s4[e], h4[e]);
        }
        if (a.residual) {
          const sf_h4 r4 = *reinterpret_cast<const sf_h4*>(a.residual + (int64_t)row * COUT + co);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
        }
        if (a.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (a.out_f32) {
          *reinterpret_cast<sf_f32x4*>(reinterpret_cast<float*>(a.out) + (int64_t)row * a.out_ld + a.out_off + co) =
              sf_f32x4{v[0], v[1], v[2], v[3]};
        } else {
          const sf_h4 pk = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
          *reinterpret_cast<sf_h4*>(reinterpret_cast<_Float16*>(a.out) + (int64_t)row * a.out_ld + a.out_off + co) = pk;
        }
      }
    }
  }
}

// weight [K, cin, cout] fp32 (Paddle layout, kd kh kw flattened) -> packed fp16 [K][cin / KC][cout][KC / 16][2][8]:
// element (s, kh, e) of output channel co in chunk c is W[k][c KC + kh KC / 2 + 8 s + e][co]
__global__ __launch_bounds__(256) void sp_pack_weight_f16_kernel(const float* __restrict__ w, int K, int cin, int cout,
                                                                 int kc, _Float16* __restrict__ out) {
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
  out[t] = (_Float16)w[((int64_t)k * cin + ci) * cout + co];
}

static inline int sf_chunk(int cin) { return cin % 64 == 0 ? 64 : (cin % 32 == 0 ? 32 : 16); }

}  // namespace pd3

using namespace pd3;

extern "C" int pd3_sparse_pack_weight_f16(const float* weight, int kernel_volume, int cin, int cout, void* packed,
                                          void* stream) {
  if (!weight || !packed || kernel_volume <= 0 || cin <= 0 || cout <= 0) return PD3_EINVAL;
  if (cin % 16 != 0 || (cout != 32 && cout != 64 && cout != 128)) return PD3_EUNSUPPORTED;
  const int64_t total = (int64_t)kernel_volume * cin * cout;
  sp_pack_weight_f16_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      weight, kernel_volume, cin, cout, sf_chunk(cin), static_cast<_Float16*>(packed));
  return launch_status();
}

extern "C" int pd3_gather_gemm_f16(const void* in_feats_f16, const int32_t* nbr, const int32_t* n_out, int n_out_cap,
                                   int kernel_volume, int cin, int cout, const void* weight_packed_f16,
                                   const float* bias, const float* scale, const float* shift, const void* residual_f16,
                                   int relu, const int32_t* order, void* out, int out_f32, int out_ld, int out_off,
                                   void* stream);

extern "C" int pd3_sparse_conv3d_features_f16(const void* in_feats_f16, const int32_t* nbr, const int32_t* n_out,
                                              int n_out_cap, int kernel_volume, int cin, int cout,
                                              const void* weight_packed_f16, const float* bias, const float* scale,
                                              const float* shift, const void* residual_f16, int relu,
                                              const int32_t* order, void* out, int out_f32, void* stream) {
  return pd3_gather_gemm_f16(in_feats_f16, nbr, n_out, n_out_cap, kernel_volume, cin, cout, weight_packed_f16, bias,
                             scale, shift, residual_f16, relu, order, out, out_f32, cout, 0, stream);
}

extern "C" int pd3_gather_gemm_f16(const void* in_feats_f16, const int32_t* nbr, const int32_t* n_out, int n_out_cap,
                                   int kernel_volume, int cin, int cout, const void* weight_packed_f16,
                                   const float* bias, const float* scale, const float* shift, const void* residual_f16,
                                   int relu, const int32_t* order, void* out, int out_f32, int out_ld, int out_off,
                                   void* stream) {
  if (!in_feats_f16 || !nbr || !weight_packed_f16 || !out || n_out_cap <= 0 || kernel_volume <= 0 || cin <= 0 ||
      cout <= 0)
    return PD3_EINVAL;
  if (out_ld < cout || out_off < 0 || out_off + cout > out_ld || out_off % 4 != 0 || out_ld % 4 != 0) return PD3_EINVAL;
  if (residual_f16 && (out_ld != cout || out_off != 0)) return PD3_EUNSUPPORTED;
  if ((scale == nullptr) != (shift == nullptr)) return PD3_EINVAL;
  if (cin % 16 != 0 || (cout != 32 && cout != 64 && cout != 128) || kernel_volume > kSfMaxK) return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(in_feats_f16) % 16 != 0 || reinterpret_cast<uintptr_t>(weight_packed_f16) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out) % 16 != 0 || (residual_f16 && reinterpret_cast<uintptr_t>(residual_f16) % 8 != 0) ||
      (bias && reinterpret_cast<uintptr_t>(bias) % 16 != 0) || (scale && reinterpret_cast<uintptr_t>(scale) % 16 != 0) ||
      (shift && reinterpret_cast<uintptr_t>(shift) % 16 != 0))
    return PD3_EINVAL;
  SpGemmF16Args a{static_cast<const _Float16*>(in_feats_f16), nbr, static_cast<const _Float16*>(weight_packed_f16),
                  bias, scale, shift, static_cast<const _Float16*>(residual_f16), out, n_out, n_out_cap, kernel_volume,
                  cin, cout, relu ? 1 : 0, out_f32 ? 1 : 0, order, out_ld, out_off};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int kc = sf_chunk(cin), nc = cout / 32;
  const size_t lds = (size_t)2 * cout * (kc + 8) * sizeof(_Float16) +
                     ((size_t)kSfRows * kernel_volume + 16 + kSfRows) * sizeof(int);
  const unsigned grid = sp_window_grid(ceil_div(n_out_cap, kSfRows), 8192 / kSfRows);
  hipError_t e;
#define PD3_SF(NCV, KCV)                                                                           \
  do {                                                                                             \
    if (lds > 48 * 1024) {                                                                         \
      e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(sp_gemm_rows_f16_kernel<NCV, KCV>), (int)lds);               \
      if (e != hipSuccess) return (int)e;                                                          \
    }                                                                                              \
    sp_gemm_rows_f16_kernel<NCV, KCV><<<grid, 256, lds, s>>>(a);                                   \
  } while (0)
  switch (nc * 100 + kc) {
    case 116: PD3_SF(1, 16); break;
    case 132: PD3_SF(1, 32); break;
    case 164: PD3_SF(1, 64); break;
    case 216: PD3_SF(2, 16); break;
    case 232: PD3_SF(2, 32); break;
    case 264: PD3_SF(2, 64); break;
    case 416: PD3_SF(4, 16); break;
    case 432: PD3_SF(4, 32); break;
    case 464: PD3_SF(4, 64); break;
    default: return PD3_EUNSUPPORTED;
  }
#undef PD3_SF
  return launch_status();
}
