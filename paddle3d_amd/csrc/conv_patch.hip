// Non-overlapping-patch convolutions of the SECOND FPN as one fp32-MFMA GEMM kernel with bias + ReLU fused
// and a channel offset on the output, so the three FPN branches write straight into the concatenated
// [N, 384, H, W] map the CenterHead reads (no torch.cat pass).
// (reference: paddle3d/models/necks/second_fpn.py:99-157 -- per level a Conv2D (kernel = stride = 1/upsample
//  stride), or a Conv2DTranspose (kernel = stride = upsample stride), + BatchNorm + ReLU, then concat.)
//
//   D[m][p] = sum_k A[m][k] * B[k][p]      on v_mfma_f32_32x32x2_f32, tile 64 rows x 256 pixels, 4 waves,
//                                          each wave 2x2 MFMA tiles, K walked 16 rows per trip
//   mode 0  Conv2D k2 s2:        m = co,            k = (ci, py, px),  B = in[ci][2y+py][2x+px]
//   mode 1  1x1 (conv or deconv): m = co,            k = ci,            B = in[ci][y][x]
//   mode 2  Conv2DTranspose k2 s2: m = (co, dy, dx),  k = ci,            B = in[ci][y][x],  D -> out[co][2y+dy][2x+dx]
//   mode 3  Conv2DTranspose k4 s4: m = (co, dy, dx),  k = ci,            B = in[ci][y][x],  D -> out[co][4y+dy][4x+dx]
//           (the third level of PointPillars' SECOND FPN, configs/pointpillars/*.yml: upsample_strides [1, 2, 4])
// Mode 1 also takes a row count that is not a multiple of 64 (the SSD head's 1x1 convolutions, 20 maps): the packed
// A matrix is zero-padded to 64 rows by the host packer and rows >= m_valid are never stored.
// A comes pre-packed from the host as [M/64][K/16][16][64]; B rows are staged with aligned float4 loads
// (mode 0: the 4 x 256 input window of 2 x 128 output pixels, read back at stride 2).  Same software
// pipeline as conv3x3.hip: trip c+1 travels global -> registers while trip c is multiplied out of
// double-buffered LDS, one barrier per trip, every LDS address in the unrolled loop is base + constant.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"

namespace pd3 {

typedef float pg_f32x16 __attribute__((ext_vector_type(16)));
typedef float pg_f32x4 __attribute__((ext_vector_type(4)));
typedef float pg_f32x2 __attribute__((ext_vector_type(2)));

constexpr int kPgM = 64;    // rows per tile
constexpr int kPgP = 256;   // pixels per tile
constexpr int kPgK = 16;    // K rows per trip
constexpr int kPgXsz = kPgK * kPgP;  // 4096 floats
constexpr int kPgWsz = kPgK * kPgM;  // 1024 floats
constexpr size_t kPgLds = (size_t)2 * (kPgXsz + kPgWsz) * sizeof(float);  // 40 KB

struct PgArgs {
  const float* x;
  const float* wp;
  const float* bias;
  float* out;
  int cin, m_rows;      // GEMM K (in rows of B per pixel: cin, or 4 cin for mode 0 handled below) and M
  int m_valid;          // rows that exist (modes 0, 1: output channels; the packed matrix is padded to m_rows)
  int hi, wi;           // input map (wi = row pitch)
  int wv;               // valid input columns (mode 2 only: columns >= wv are padding and produce no output)
  int ho, wo;           // output map
  int ctot, coff;       // channels of the output tensor, first channel written
  int relu, ptiles;
};

// HALF (mode 1 with <= 32 output rows, e.g. the 20 maps of the fused SSD head): the second 32-row MFMA block of the tile
// would only multiply padding, so it is not issued -- the layer is bound by reading its input once.
template <int MODE, bool HALF = false>
__global__ __launch_bounds__(256) void patch_gemm_kernel(PgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float pg_smem[];  // X[2][4096] then W[2][1024]
  const int lane = lane_id(), wave = wave_id();
  const int nmt = a.m_rows / kPgM;
  // XCD-aware order: pixel tile pt lives on XCD pt % 8, its row tiles follow each other there
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = slot % nmt, pt = (slot / nmt) * 8 + xcd;
  if (pt >= a.ptiles) return;
  const int64_t iplane = (int64_t)a.hi * a.wi, oplane = (int64_t)a.ho * a.wo;
  // pixel tile: mode 0 -> 2 output rows x 128 columns; modes 1, 2 -> 256 consecutive pixels of the input plane
  int n, p0 = 0, oy0 = 0, ox0 = 0;
  if (MODE == 0) {
    const int tiles_x = (a.wo + 127) / 128, tiles_y = a.ho / 2;  // the last tile of a row may be partial
    n = pt / (tiles_x * tiles_y);
    oy0 = ((pt / tiles_x) % tiles_y) * 2;
    ox0 = (pt % tiles_x) * 128;
  } else {
    const int tiles = (int)((iplane + kPgP - 1) / kPgP);  // the last tile of a plane may be partial
    n = pt / tiles;
    p0 = (pt % tiles) * kPgP;
  }
  const int kk = lane >> 5;
  const int ktot = MODE == 0 ? a.cin * 4 : a.cin;
  const int chunks = ktot / kPgK;
  // per-lane LDS bases: B operand of pixel block t (2 per wave), A operand of row block 0
  int xb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int pj = (wave * 2 + t) * 32 + (lane & 31);
    // mode 0: X buffer = [4 ci][4 rows][256 cols]; K row k = (ci, py, px) with px = kk
    xb[t] = MODE == 0 ? (pj / 128) * 2 * 256 + (pj % 128) * 2 + kk : kk * kPgP + pj;
  }
  const int wb = 2 * kPgXsz + kk * kPgM + (lane & 31);
  pg_f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
  // staging: 1024 float4 of X per trip (4 per thread), 256 float4 of W (1 per thread)
  const float* xin = a.x + (int64_t)n * a.cin * iplane;
  int gofs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = threadIdx.x + i * 256;
    if (MODE == 0) {  // e -> (ci 0..3, row 0..3, c4 0..63) of the 4-channel window
      const int ci = e >> 8, r = (e >> 6) & 3, c4 = e & 63;
      // columns past the row are clamped to its last float4: they only feed pixels that are never stored
      gofs[i] = (int)(ci * iplane + (int64_t)(2 * oy0 + r) * a.wi + min(2 * ox0 + c4 * 4, a.wi - 4));
    } else {          // e -> (ci 0..15, c4 0..63)
      const int ci = e >> 6, c4 = e & 63;
      // columns past the plane are clamped to its last float4: they only feed pixels that are never stored
      gofs[i] = (int)(ci * iplane + min((int64_t)p0 + c4 * 4, iplane - 4));
    }
  }
  const int cpt = MODE == 0 ? 4 : 16;  // input channels per trip
  const pg_f32x4* wsrc = reinterpret_cast<const pg_f32x4*>(a.wp) + (int64_t)mt * chunks * (kPgWsz / 4);
  pg_f32x4 xr[4], wr;

#define PG_FETCH(cc)                                                                     \
  {                                                                                      \
    const float* xc_ = xin + (int64_t)(cc) * cpt * iplane;                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                        \
        xr[i] = *reinterpret_cast<const pg_f32x4*>(xc_ + gofs[i]);                       \
    wr = wsrc[(int64_t)(cc) * (kPgWsz / 4) + threadIdx.x];                               \
  }
#define PG_STASH(buf)                                                                    \
  {                                                                                      \
    float* xd_ = pg_smem + (buf) * kPgXsz;                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                        \
        *reinterpret_cast<pg_f32x4*>(xd_ + (threadIdx.x + i * 256) * 4) = xr[i];         \
    *reinterpret_cast<pg_f32x4*>(pg_smem + 2 * kPgXsz + (buf) * kPgWsz + threadIdx.x * 4) = wr; \
  }

  PG_FETCH(0)
  PG_STASH(0)
  __syncthreads();
  for (int cc = 0; cc < chunks; ++cc) {
    const int nx = min(cc + 1, chunks - 1);  // the last trip re-fetches its own chunk into the idle buffer
    PG_FETCH(nx)
    __builtin_amdgcn_sched_barrier(0);  // keep the loads in flight ahead of the MFMA block
    const float* Xs = pg_smem + (cc & 1) * kPgXsz;
    const float* Ws = pg_smem + (cc & 1) * kPgWsz + wb;
#pragma unroll
    for (int s = 0; s < kPgK / 2; ++s) {
      // K rows 2s, 2s+1 (lane half kk): mode 0 -> (ci = s >> 1, py = s & 1, px = kk)
      const int xo = MODE == 0 ? (s >> 1) * 1024 + (s & 1) * 256 : 2 * s * kPgP;
      const float a0 = Ws[s * 2 * kPgM], a1 = Ws[s * 2 * kPgM + 32];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float b = Xs[xb[t] + xo];
        acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][t], 0, 0, 0);
        if (!HALF) acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    PG_STASH((cc + 1) & 1)
    __syncthreads();
  }
#undef PG_FETCH
#undef PG_STASH

  // epilogue: D layout col = lane & 31 (pixel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* obase = a.out + ((int64_t)n * a.ctot + a.coff) * oplane;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int pj = (wave * 2 + t) * 32 + (lane & 31);
    if (MODE != 0 && (int64_t)p0 + pj >= iplane) continue;  // pixel of a partial tile
    if (MODE == 0 && ox0 + (pj % 128) >= a.wo) continue;
#pragma unroll
    for (int m = 0; m < (HALF ? 1 : 2); ++m) {
      if (MODE == 3) {
        // a register quad = the four dx of one (co, dy): row = r + 8 q + 4 kk of the 32-row block, 16 rows per co
        const int p = p0 + pj, y = p / a.wi, xx = p - y * a.wi;
        if (xx >= a.wv) continue;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int co = mt * (kPgM / 16) + m * 2 + (qd >> 1);
          const int dy = (qd & 1) * 2 + kk;
          const float bv = a.bias ? a.bias[co] : 0.f;
          pg_f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[m][t][qd * 4 + r] + bv;
            if (a.relu) v[r] = fmaxf(v[r], 0.f);
          }
          *reinterpret_cast<pg_f32x4*>(obase + (int64_t)co * oplane + (int64_t)(4 * y + dy) * a.wo + 4 * xx) = v;
        }
      } else if (MODE == 2) {
        // rows 4q .. 4q+3 of a register quad = (dy, dx) of one output channel
        const int p = p0 + pj, y = p / a.wi, xx = p - y * a.wi;
        if (xx >= a.wv) continue;  // padding column of a map whose width is not a multiple of 4
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int row = mt * kPgM + m * 32 + 8 * qd + 4 * kk;  // = 4 * co
          const int co = row >> 2;
          const float bv = a.bias ? a.bias[co] : 0.f;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[m][t][qd * 4 + r] + bv;
            if (a.relu) v[r] = fmaxf(v[r], 0.f);
          }
          float* o = obase + (int64_t)co * oplane + (int64_t)(2 * y) * a.wo + 2 * xx;
          *reinterpret_cast<pg_f32x2*>(o) = (pg_f32x2){v[0], v[1]};
          *reinterpret_cast<pg_f32x2*>(o + a.wo) = (pg_f32x2){v[2], v[3]};
        }
      } else {
        const int64_t opix = MODE == 0 ? (int64_t)(oy0 + pj / 128) * a.wo + ox0 + (pj % 128) : (int64_t)p0 + pj;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int co = mt * kPgM + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
          if (co >= a.m_valid) continue;
          float v = acc[m][t][reg] + (a.bias ? a.bias[co] : 0.f);
          if (a.relu) v = fmaxf(v, 0.f);
          obase[(int64_t)co * oplane + opix] = v;
        }
      }
    }
  }
}

template <int MODE, bool HALF = false>
static int launch_patch_gemm(const PgArgs& a, int64_t ptiles, hipStream_t s) {
  const int64_t nwg = (ptiles + 7) / 8 * 8 * (a.m_rows / kPgM);
  if (nwg >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  patch_gemm_kernel<MODE, HALF><<<(unsigned)nwg, 256, kPgLds, s>>>(a);
  return launch_status();
}

}  // namespace pd3

using namespace pd3;

extern "C" int pd3_patch_conv_bias_relu(const float* x, const float* w_packed, const float* bias, int mode,
                                        int batch, int cin, int cout, int h, int w, int w_valid, int relu,
                                        float* out, int out_channels_total, int out_channel_offset, void* stream) {
  if (!x || !w_packed || !out || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || w_valid <= 0 ||
      w_valid > w || (mode < 2 && w_valid != w))
    return PD3_EINVAL;
  if (mode < 0 || mode > 3 || out_channel_offset < 0 || out_channel_offset + cout > out_channels_total)
    return PD3_EINVAL;
  if (reinterpret_cast<uintptr_t>(w_packed) % 16 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out) % (mode == 3 ? 16 : 8) != 0)
    return PD3_EINVAL;
  if ((int64_t)cin * h * w >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;  // 32-bit staging offsets
  PgArgs a;
  a.x = x;
  a.wp = w_packed;
  a.bias = bias;
  a.out = out;
  a.cin = cin;
  a.hi = h;
  a.wi = w;
  a.wv = w_valid;
  a.ctot = out_channels_total;
  a.coff = out_channel_offset;
  a.relu = relu;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int64_t ptiles;
  if (mode == 0) {  // Conv2D kernel 2 stride 2
    if (h % 4 != 0 || w % 4 != 0 || (cin * 4) % kPgK != 0 || cout % kPgM != 0) return PD3_EUNSUPPORTED;
    a.m_rows = cout;
    a.m_valid = cout;
    a.ho = h / 2;
    a.wo = w / 2;
    ptiles = (int64_t)batch * (a.ho / 2) * ceil_div(a.wo, 128);
    a.ptiles = (int)ptiles;
    return launch_patch_gemm<0>(a, ptiles, s);
  }
  if (((int64_t)h * w) % 4 != 0 || cin % kPgK != 0) return PD3_EUNSUPPORTED;  // planes start float4-aligned
  ptiles = (int64_t)batch * ceil_div((int64_t)h * w, kPgP);
  a.ptiles = (int)ptiles;
  a.m_valid = INT32_MAX;
  if (mode == 1) {  // 1x1; w_packed holds ceil(cout / 64) * 64 rows
    a.m_rows = (int)ceil_div(cout, kPgM) * kPgM;
    a.m_valid = cout;
    a.ho = h;
    a.wo = w;
    return cout <= 32 ? launch_patch_gemm<1, true>(a, ptiles, s) : launch_patch_gemm<1>(a, ptiles, s);
  }
  if (mode == 3) {  // Conv2DTranspose kernel 4 stride 4
    if ((cout * 16) % kPgM != 0) return PD3_EUNSUPPORTED;
    a.m_rows = cout * 16;
    a.ho = 4 * h;
    a.wo = 4 * w_valid;
    return launch_patch_gemm<3>(a, ptiles, s);
  }
  if ((cout * 4) % kPgM != 0) return PD3_EUNSUPPORTED;  // Conv2DTranspose kernel 2 stride 2
  a.m_rows = cout * 4;
  a.ho = 2 * h;
  a.wo = 2 * w_valid;
  return launch_patch_gemm<2>(a, ptiles, s);
}
