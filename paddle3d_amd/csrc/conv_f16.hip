// Mixed-precision (AMP) form of the stride-1 3x3 convolutions of the dense BEV graph: fp16 activations and weights on
// the fp16 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate), bias + ReLU fused.  The reference ships an AMP
// configuration of the headline model (configs/centerpoint/centerpoint_pillars_02voxel_nuscenes_10sweep_ampO2_ultra.yml:
// 5-9, amp_cfg level O2: fp16 activations and weights in the convolutions) and publishes an FP16 figure next to the FP32
// one (docs/models/centerpoint/README.md:35); this is that path, reported as its own bench workload
// (`--workload centerpoint_pillars_amp`) with its error against the fp32 graph stated -- never as the fp32 headline.
//
// Direct implicit GEMM (no Winograd: F(4x4,3x3) amplifies fp16 rounding by its 4 .. 24x transform constants), laid out
// for what bounds an fp16 MFMA kernel on this machine -- LDS read bandwidth and the L2 -> CU ingest, not the matrix pipe:
//   * activations travel between the stride-1 layers as fp16 NHWC, so a pixel's 16 input channels of a K chunk are 32
//     contiguous bytes: the staged patch is a plain copy (no transpose, no conversion) and a lane's B operand of
//     v_mfma_f32_32x32x16_f16 (8 consecutive k of one column) is ONE aligned ds_read_b128; weights are packed on the host
//     as [cout tile][cin / 16][tap][co][16 ci] fp16, so the A operand is one ds_read_b128 too and a chunk is one linear
//     copy.  pd3_f32_nchw_to_f16_nhwc converts at the fp32 boundaries (after a stride-2 convolution, in front of the
//     head); the last layer of a chain writes fp32 NCHW for the fp32 kernels behind it.
//   * workgroup = 8 waves = M x N = (64 MB) output channels x (8 / MB slabs of 4 rows x 32 columns): a wave owns 64
//     channels x 128 pixels = 2 x 4 MFMA blocks (128 accumulators), reads 2 A + 4 B fragments (6 KB) per 8 MFMAs (256
//     matrix-pipe cycles): 24 B/clk per SIMD, 3/4 of the LDS's 128 B/clk per CU.  Per 16-channel chunk the workgroup ingests
//     9 x M x 32 B of weights + (rows + 2) x 34 x 32 B of patch = 56 KB (MB = 2) for 4608 pipe cycles: 12 B/clk, inside the
//     ~14 B/clk the L2 -> CU path sustains.  MB = 2 (128 channels x 16 rows) where cout % 128 == 0, MB = 1
//     (64 channels x 32 rows) for the 64-channel layers.
//   * LDS double-buffered (113 KB), chunk c + 1 travels global -> registers while chunk c is multiplied: one barrier per chunk.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"

namespace pd3 {

typedef _Float16 cf_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cf_h4 __attribute__((ext_vector_type(4)));
typedef float cf_f32x16 __attribute__((ext_vector_type(16)));
typedef float cf_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kCfCols = 32;               // output columns per workgroup (= N of one MFMA block)
constexpr int kCfPW = kCfCols + 2;        // staged patch width
constexpr int kCfKc = 16;                 // input channels per chunk (= K of the MFMA)
constexpr int kCfThreads = 512;

template <int MB>
struct CfShape {
  static constexpr int M = 64 * MB;                  // output channels per workgroup
  static constexpr int SLABS = 8 / MB;               // 4-row pixel slabs per workgroup
  static constexpr int R = 4 * SLABS;                // output rows per workgroup
  static constexpr int PATCH = (R + 2) * kCfPW * kCfKc;   // halfs
  static constexpr int WTS = 9 * M * kCfKc;               // halfs
  static constexpr int PPIECES = PATCH / 8;          // 16-byte pieces
  static constexpr int WPIECES = WTS / 8;
  static constexpr int PPT = (PPIECES + kCfThreads - 1) / kCfThreads;
  static constexpr int WPT = (WPIECES + kCfThreads - 1) / kCfThreads;
  static constexpr size_t LDS = (size_t)2 * (PATCH + WTS) * sizeof(_Float16);
};

// out_mode 0: fp16 NHWC (the next fp16 layer's input); 1: fp32 NCHW (what every fp32 kernel of the graph reads); 2: both
// (a block's last layer: fp16 NHWC for the next block's stride-2 convolution, fp32 NCHW for the FPN level -- out2)
template <int MB, int OUT_MODE>
__global__ __launch_bounds__(kCfThreads, 1) void conv3x3_f16_kernel(const _Float16* __restrict__ x,
                                                                    const _Float16* __restrict__ wp,
                                                                    const float* __restrict__ bias,
                                                                    void* __restrict__ out, int cin, int cout, int h,
                                                                    int w, int relu, int ptiles,
                                                                    float* __restrict__ out2 = nullptr) {
  using S = CfShape<MB>;
  extern __shared__ __attribute__((aligned(16))) _Float16 cf_smem[];
  const int lane = lane_id(), wave = wave_id();
  const int tiles_x = (w + kCfCols - 1) / kCfCols, tiles_y = (h + S::R - 1) / S::R;  // partial border tiles are masked
  // XCD-aware order (as the fp32 kernels): pixel tile pt lives on XCD pt % 8 with all its channel tiles
  const int nct = cout / S::M;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ct = slot % nct, pt = (slot / nct) * 8 + xcd;
  if (pt >= ptiles) return;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int y0 = ty * S::R, x0 = tx * kCfCols;
  const int chunks = cin / kCfKc;
  const _Float16* xin = x + (int64_t)n * h * w * cin;
  const cf_h8* wsrc = reinterpret_cast<const cf_h8*>(wp) + (int64_t)ct * chunks * S::WPIECES;

  // staging pattern (identical for every chunk): patch piece e = (pixel, half of its 16 channels)
  int pofs[S::PPT];
  unsigned plive = 0;
#pragma unroll
  for (int i = 0; i < S::PPT; ++i) {
    const int e = min((int)threadIdx.x + i * kCfThreads, S::PPIECES - 1);
    const int pix = e >> 1, hf = e & 1;
    const int pr = pix / kCfPW, pc = pix - pr * kCfPW;
    const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
    const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
    pofs[i] = ok ? (gy * w + gx) * cin + 8 * hf : 0;
    plive |= ok ? (1u << i) : 0u;
  }
  cf_h8 preg[S::PPT], wreg[S::WPT];
  auto fetch = [&](int c) {
    const _Float16* xc = xin + c * kCfKc;
#pragma unroll
    for (int i = 0; i < S::PPT; ++i) preg[i] = *reinterpret_cast<const cf_h8*>(xc + pofs[i]);
    const cf_h8* wc = wsrc + (int64_t)c * S::WPIECES;
#pragma unroll
    for (int i = 0; i < S::WPT; ++i) wreg[i] = wc[min((int)threadIdx.x + i * kCfThreads, S::WPIECES - 1)];
  };
  auto stash = [&](int buf) {
    _Float16* P = cf_smem + buf * (S::PATCH + S::WTS);
    _Float16* W = P + S::PATCH;
#pragma unroll
    for (int i = 0; i < S::PPT; ++i) {
      const int e = (int)threadIdx.x + i * kCfThreads;
      const cf_h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      if (S::PPIECES % kCfThreads == 0 || e < S::PPIECES)
        *reinterpret_cast<cf_h8*>(P + e * 8) = ((plive >> i) & 1u) ? preg[i] : z;
    }
#pragma unroll
    for (int i = 0; i < S::WPT; ++i) {
      const int e = (int)threadIdx.x + i * kCfThreads;
      if (S::WPIECES % kCfThreads == 0 || e < S::WPIECES) *reinterpret_cast<cf_h8*>(W + e * 8) = wreg[i];
    }
  };

  const int mw = wave % MB, nw = wave / MB;  // the wave's 64-channel block and its 4-row slab
  const int l31 = lane & 31, kh = lane >> 5;
  cf_f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  fetch(0);
  stash(0);
  __syncthreads();
  for (int c = 0; c < chunks; ++c) {
    const bool more = c + 1 < chunks;
    if (more) fetch(c + 1);
    const _Float16* P = cf_smem + (c & 1) * (S::PATCH + S::WTS);
    const _Float16* W = P + S::PATCH;
    // A: lane (m = l31, k = 8 kh ..) of channel block i, tap t;  B: lane (n = l31, k = 8 kh ..) of pixel row j
    const _Float16* wa = W + ((mw * 64 + l31) * kCfKc + 8 * kh);
    const _Float16* pb = P + (((nw * 4) * kCfPW + l31) * kCfKc + 8 * kh);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3, dx = t - 3 * dy;
      cf_h8 a[2], b[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const cf_h8*>(wa + (t * S::M + i * 32) * kCfKc);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const cf_h8*>(pb + ((j + dy) * kCfPW + dx) * kCfKc);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) stash((c + 1) & 1);
    __syncthreads();
  }

  // epilogue: D[row = (reg & 3) + 8 (reg >> 2) + 4 kh][col = l31] of block (i, j): channel co0 + 32 i + row, pixel
  // (y0 + 4 nw + j, x0 + l31)
  const int co0 = ct * S::M + mw * 64;
  const int xg = x0 + l31;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = bias ? bias[co0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yg = y0 + nw * 4 + j;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] = acc[i][j][r] + bv[r];
        if (relu) v[r] = fmaxf(v[r], 0.f);
      }
      if (yg >= h || xg >= w) continue;  // a border tile's pixels outside the map (round 5: 180 x 180 maps of config 4)
      if (OUT_MODE == 0 || OUT_MODE == 2) {
        _Float16* o = reinterpret_cast<_Float16*>(out) + (((int64_t)n * h + yg) * w + xg) * cout + co0 + 32 * i + 4 * kh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const cf_h4 pk = {(_Float16)v[4 * q], (_Float16)v[4 * q + 1], (_Float16)v[4 * q + 2], (_Float16)v[4 * q + 3]};
          *reinterpret_cast<cf_h4*>(o + 8 * q) = pk;
        }
      }
      if (OUT_MODE == 1 || OUT_MODE == 2) {
        float* o = (OUT_MODE == 1 ? reinterpret_cast<float*>(out) : out2) +
                   (((int64_t)n * cout + co0 + 32 * i + 4 * kh) * h + yg) * w + xg;
        const int64_t plane = (int64_t)h * w;
#pragma unroll
        for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(v[r], o + ((r & 3) + 8 * (r >> 2)) * plane);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The stride-2 3x3 / pad 1 convolutions that open SecondBackbone's blocks 1 and 2 (second_backbone.py:84-113) under AMP:
// the same direct implicit GEMM on fp16 NHWC, output pixel (y, x) reading input pixels (2 y - 1 + dy, 2 x - 1 + dx).
// Workgroup = 8 waves = 128 output channels x 8 output rows x 32 output columns; wave (mw, nw) owns 64 channels x 2 rows =
// 2 x 2 MFMA blocks.  The staged patch is 17 input rows x 65 input columns x 16 channels with the COLUMNS DE-INTERLEAVED
// (even columns, then odd columns of a row): output column l reads input column 2 l + dx = entry l + (dx >> 1) of plane
// dx & 1, so the 32 lanes of a B operand read 32 consecutive 32-byte pixels exactly as in the stride-1 kernel (with the
// columns interleaved their stride would be 64 bytes: a four-way bank conflict on every ds_read_b128).
// fp16 NHWC out (the block's stride-1 layers follow).  cin % 16 == 0, cout % 128 == 0.
// Round 5, later: MW = 64-channel blocks per workgroup (2: 128 channels, 8 waves; 1: 64 channels, 4 waves -- the 64 -> 64
// layer that opens block 0) and GATHER = PointPillarsScatter fused in: the "image" is the pillar features [M, cin] fp16
// behind the inverse map (cell -> pillar row or -1, pd3_pointpillars_inverse_map), so the first layer of the backbone
// runs on the fp16 matrix cores as well and the canvas is never written (as in the fp32 pd3_scatter_conv3x3_bias_relu).
constexpr int kCs2R = 8;                         // output rows per workgroup
constexpr int kCs2PR = 2 * kCs2R + 1;            // staged input rows
constexpr int kCs2Half = kCfCols + 1;            // entries per parity plane of a staged row (33)
constexpr int kCs2Patch = kCs2PR * 2 * kCs2Half * kCfKc;  // halfs
constexpr int kCs2PPieces = kCs2Patch / 8;
template <int MW>
struct Cs2Shape {
  static constexpr int THREADS = 256 * MW;
  static constexpr int M = 64 * MW;
  static constexpr int WTS = 9 * M * kCfKc;
  static constexpr int WPIECES = WTS / 8;
  static constexpr int PPT = (kCs2PPieces + THREADS - 1) / THREADS;
  static constexpr int WPT = (WPIECES + THREADS - 1) / THREADS;
  static constexpr size_t LDS = (size_t)2 * (kCs2Patch + WTS) * sizeof(_Float16);
};

template <int MW, bool GATHER>
__global__ __launch_bounds__(256 * MW, 1) void conv3x3_s2_f16_kernel(const _Float16* __restrict__ x,
                                                                      const int32_t* __restrict__ inv,
                                                                      const _Float16* __restrict__ wp,
                                                                      const float* __restrict__ bias,
                                                                      _Float16* __restrict__ out, int cin, int cout,
                                                                      int h, int w, int ho, int wo, int relu,
                                                                      int ptiles) {
  using S = Cs2Shape<MW>;
  extern __shared__ __attribute__((aligned(16))) _Float16 cf_smem[];
  const int lane = lane_id(), wave = wave_id();
  const int tiles_x = (wo + kCfCols - 1) / kCfCols, tiles_y = (ho + kCs2R - 1) / kCs2R;
  const int nct = cout / S::M;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ct = slot % nct, pt = (slot / nct) * 8 + xcd;
  if (pt >= ptiles) return;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int y0 = ty * kCs2R, x0 = tx * kCfCols;  // output coordinates of the tile
  const int chunks = cin / kCfKc;
  const _Float16* xin = GATHER ? x : x + (int64_t)n * h * w * cin;
  const cf_h8* wsrc = reinterpret_cast<const cf_h8*>(wp) + (int64_t)ct * chunks * S::WPIECES;

  // staging pattern: piece e = (staged pixel, half of its 16 channels); staged pixel = (row pr, parity, entry)
  int64_t pofs[S::PPT];
  unsigned plive = 0;
#pragma unroll
  for (int i = 0; i < S::PPT; ++i) {
    const int e = min((int)threadIdx.x + i * S::THREADS, kCs2PPieces - 1);
    const int pix = e >> 1, hf = e & 1;
    const int pr = pix / (2 * kCs2Half), rem = pix - pr * (2 * kCs2Half);
    const int par = rem / kCs2Half, ent = rem - par * kCs2Half;
    const int pc = 2 * ent + par;  // input column of the patch, 0 .. 65 (65 = the odd plane's unused last entry)
    const int gy = 2 * y0 - 1 + pr, gx = 2 * x0 - 1 + pc;
    bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w && pc <= 2 * kCfCols;
    int64_t pixel = (int64_t)gy * w + gx;
    if (GATHER) {  // the cell's pillar row (or none)
      const int row = ok ? inv[(int64_t)n * h * w + pixel] : -1;
      ok = row >= 0;
      pixel = row;
    }
    pofs[i] = ok ? pixel * cin + 8 * hf : 0;
    plive |= ok ? (1u << i) : 0u;
  }
  cf_h8 preg[S::PPT], wreg[S::WPT];
  auto fetch = [&](int c) {
    const _Float16* xc = xin + c * kCfKc;
#pragma unroll
    for (int i = 0; i < S::PPT; ++i) preg[i] = *reinterpret_cast<const cf_h8*>(xc + pofs[i]);
    const cf_h8* wc = wsrc + (int64_t)c * S::WPIECES;
#pragma unroll
    for (int i = 0; i < S::WPT; ++i) wreg[i] = wc[min((int)threadIdx.x + i * S::THREADS, S::WPIECES - 1)];
  };
  auto stash = [&](int buf) {
    _Float16* P = cf_smem + buf * (kCs2Patch + S::WTS);
    _Float16* W = P + kCs2Patch;
#pragma unroll
    for (int i = 0; i < S::PPT; ++i) {
      const int e = (int)threadIdx.x + i * S::THREADS;
      const cf_h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      if (kCs2PPieces % S::THREADS == 0 || e < kCs2PPieces)
        *reinterpret_cast<cf_h8*>(P + e * 8) = ((plive >> i) & 1u) ? preg[i] : z;
    }
#pragma unroll
    for (int i = 0; i < S::WPT; ++i) {
      const int e = (int)threadIdx.x + i * S::THREADS;
      if (S::WPIECES % S::THREADS == 0 || e < S::WPIECES) *reinterpret_cast<cf_h8*>(W + e * 8) = wreg[i];
    }
  };

  const int mw = wave % MW, nw = wave / MW;  // the wave's 64-channel block and its pair of output rows
  const int l31 = lane & 31, kh = lane >> 5;
  cf_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  fetch(0);
  stash(0);
  __syncthreads();
  for (int c = 0; c < chunks; ++c) {
    const bool more = c + 1 < chunks;
    if (more) fetch(c + 1);
    const _Float16* P = cf_smem + (c & 1) * (kCs2Patch + S::WTS);
    const _Float16* W = P + kCs2Patch;
    const _Float16* wa = W + ((mw * 64 + l31) * kCfKc + 8 * kh);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3, dx = t - 3 * dy;
      cf_h8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const cf_h8*>(wa + (t * S::M + i * 32) * kCfKc);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int pr = 2 * (nw * 2 + j) + dy;
        b[j] = *reinterpret_cast<const cf_h8*>(P + (((pr * 2 + (dx & 1)) * kCs2Half + l31 + (dx >> 1)) * kCfKc + 8 * kh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) stash((c + 1) & 1);
    __syncthreads();
  }
  const int co0 = ct * S::M + mw * 64;
  const int xg = x0 + l31;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = bias ? bias[co0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh] : 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int yg = y0 + nw * 2 + j;
      if (yg >= ho || xg >= wo) continue;
      _Float16* o = out + (((int64_t)n * ho + yg) * wo + xg) * cout + co0 + 32 * i + 4 * kh;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][j][4 * q + e] + bv[4 * q + e];
          if (relu) v[e] = fmaxf(v[e], 0.f);
        }
        const cf_h4 pk = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *reinterpret_cast<cf_h4*>(o + 8 * q) = pk;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The final SeparateHead convolutions under AMP (center_head.py:99-118: per head a 3x3 convolution from the 64 channels
// of its first stage to 1 .. 3 maps): a grouped 3x3 convolution reading the first stage's fp16 NHWC output -- group g
// = channels 64 g .. 64 g + 63 of every pixel, 128 contiguous bytes -- and writing the fp32 NCHW maps the post-processing
// reads.  The fp32 form of this pair wrote the 2304-channel first-stage map as fp32 NCHW (2.4 GB per 16 frames, 1.02 ms)
// and fetched it back (0.53 ms); in fp16 NHWC it is half of that in each direction and every fetched line is used whole.
// Workgroup = (8 x 32-pixel tile, group, frame): the tile's (10 x 34) x 64 halfs are staged once (pixel lines padded to
// 68 halfs: conflict-free ds_read_b128); wave = two tile rows, v_mfma_f32_32x32x16_f16 with the weights as the A operand.
template <int CO>
__global__ __launch_bounds__(256) void grouped_conv3x3_small_f16_kernel(
    const _Float16* __restrict__ x, const _Float16* __restrict__ wg, const float* __restrict__ bias, int groups, int h,
    int w, float* __restrict__ out, int out_groups, int out_group0) {
  constexpr int TR = 8, TC = 32, PW = TC + 2, PS = 68;  // tile rows / columns, patch width, halfs per staged pixel
  __shared__ __attribute__((aligned(16))) _Float16 patch[(TR + 2) * PW * PS];
  const int tiles_x = (w + TC - 1) / TC;
  const int tx0 = (blockIdx.x % tiles_x) * TC, ty0 = (blockIdx.x / tiles_x) * TR;
  const int g = blockIdx.y, n = blockIdx.z;
  const int c = groups * 64;
  const _Float16* xin = x + (int64_t)n * h * w * c + g * 64;
  const int l31 = threadIdx.x & 31, kh = (threadIdx.x >> 5) & 1;
  cf_h8 aw[36];  // A fragments: W[tap][m = l31][16 s + 8 kh ..] for m < CO, zero rows above
  {
    const _Float16* wgrp = wg + (int64_t)g * 9 * CO * 64 + (l31 < CO ? l31 : 0) * 64 + 8 * kh;
    const cf_h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int sk = 0; sk < 4; ++sk) {
        const cf_h8 v = *reinterpret_cast<const cf_h8*>(wgrp + t * CO * 64 + 16 * sk);
        aw[t * 4 + sk] = l31 < CO ? v : z;
      }
  }
  {
    // all of a thread's pieces are requested before the first one is parked (one load per loop trip made the staging a
    // chain of eleven memory round trips: 0.59 ms per call, most of it here)
    constexpr int PIECES = (TR + 2) * PW * 8, PPT = (PIECES + 255) / 256;
    cf_h8 reg[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int e = min((int)threadIdx.x + i * 256, PIECES - 1);
      const int pix = e >> 3, q = e & 7;
      const int pr = pix / PW, pc = pix - pr * PW;
      const int gy = ty0 - 1 + pr, gx = tx0 - 1 + pc;
      const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
      const cf_h8 v = *reinterpret_cast<const cf_h8*>(xin + (ok ? ((int64_t)gy * w + gx) * c : 0) + q * 8);
      const cf_h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      reg[i] = ok ? v : z;
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int e = (int)threadIdx.x + i * 256;
      if (e < PIECES) *reinterpret_cast<cf_h8*>(patch + (e >> 3) * PS + (e & 7) * 8) = reg[i];
    }
  }
  __syncthreads();
  // The products run on the matrix cores (v_mfma_f32_32x32x16_f16): A = the group's weights, output channel m = lane &
  // 31 (rows CO .. 31 of the block are zero), B = the 32 pixels of a tile row, K = 16 channels per step -- 36 steps per
  // tap-and-chunk, two tile rows per wave.  (v_dot2c_f32_f16 with one pixel per thread issued at ~22 cycles per
  // instruction here: 864 of them per thread, 0.58 ms per call; the 32-wide M block wastes 29 of its 32 rows and is
  // still four times faster.)  The 36 A fragments were requested before the staging loads, so they have landed.
  const int ty = 2 * wave_id();
  cf_f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int dy = t / 3, dx = t - 3 * dy;
#pragma unroll
    for (int sk = 0; sk < 4; ++sk) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const cf_h8 bv = *reinterpret_cast<const cf_h8*>(patch + ((ty + j + dy) * PW + l31 + dx) * PS + 16 * sk + 8 * kh);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw[t * 4 + sk], bv, acc[j], 0, 0, 0);
      }
    }
  }
  // D[m = (reg & 3) + 8 (reg >> 2) + 4 kh][n = l31]: channels 0 .. CO - 1 are registers 0 .. CO - 1 of the kh = 0 lanes
  const int xg = tx0 + l31;
  if (kh != 0 || xg >= w) return;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int y = ty0 + ty + j;
    if (y >= h) continue;
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      const int ch = (out_group0 + g) * CO + o;
      out[(((int64_t)n * out_groups * CO + ch) * h + y) * w + xg] = acc[j][o] + (bias ? bias[g * CO + o] : 0.f);
    }
  }
}

// fp32 NCHW -> fp16 NHWC (the boundary in front of a chain of fp16 layers): one workgroup per (n, y, 64 columns),
// channels in chunks of 64 through an LDS tile (reads coalesced along x, writes along c)
__global__ __launch_bounds__(256) void f32_nchw_to_f16_nhwc_kernel(const float* __restrict__ x, int c, int h, int w,
                                                                   _Float16* __restrict__ out) {
  __shared__ float tile[64][65];
  const int tiles_x = (w + 63) / 64;
  const int bx = blockIdx.x % tiles_x, y = (blockIdx.x / tiles_x) % h, n = blockIdx.x / (tiles_x * h);
  const int x0 = bx * 64;
  const int tx = threadIdx.x & 63, tq = threadIdx.x >> 6;
  for (int c0 = 0; c0 < c; c0 += 64) {
    for (int r = tq; r < 64; r += 4) {
      const int ch = c0 + r, xx = x0 + tx;
      tile[r][tx] = (ch < c && xx < w) ? x[(((int64_t)n * c + ch) * h + y) * w + xx] : 0.f;
    }
    __syncthreads();
    for (int r = tq; r < 64; r += 4) {  // r = pixel, tx = channel
      const int xx = x0 + r, ch = c0 + tx;
      if (xx < w && ch < c) out[(((int64_t)n * h + y) * w + xx) * c + ch] = (_Float16)tile[tx][r];
    }
    __syncthreads();
  }
}

}  // namespace pd3

using namespace pd3;

template <int MB, int OUT_MODE>
static int launch_conv_f16(const void* x, const void* wp, const float* bias, int batch, int cin, int cout, int h, int w,
                           int relu, void* out, hipStream_t s, float* out2 = nullptr) {
  using S = CfShape<MB>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_f16_kernel<MB, OUT_MODE>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS);
  if (e != hipSuccess) return (int)e;
  const int64_t ptiles = (int64_t)batch * ceil_div(h, S::R) * ceil_div(w, kCfCols);
  const int64_t nwg = (ptiles + 7) / 8 * 8 * (cout / S::M);
  if (nwg >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  conv3x3_f16_kernel<MB, OUT_MODE><<<(unsigned)nwg, kCfThreads, S::LDS, s>>>(
      static_cast<const _Float16*>(x), static_cast<const _Float16*>(wp), bias, out, cin, cout, h, w, relu, (int)ptiles,
      out2);
  return launch_status();
}

extern "C" int pd3_conv3x3_f16_bias_relu(const void* x_f16_nhwc, const void* w_packed_f16, const float* bias, int batch,
                                         int cin, int cout, int h, int w, int relu, void* out, int out_mode,
                                         int channels_per_tile, void* stream) {
  if (!x_f16_nhwc || !w_packed_f16 || !out || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return PD3_EINVAL;
  if ((out_mode != 0 && out_mode != 1) || (channels_per_tile != 64 && channels_per_tile != 128)) return PD3_EINVAL;
  if (reinterpret_cast<uintptr_t>(x_f16_nhwc) % 16 != 0 || reinterpret_cast<uintptr_t>(w_packed_f16) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out) % 16 != 0)
    return PD3_EINVAL;
  if (cin % kCfKc != 0 || cout % channels_per_tile != 0) return PD3_EUNSUPPORTED;
  if ((int64_t)h * w * cin >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;  // 32-bit staging offsets
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (channels_per_tile == 128)
    return out_mode == 0 ? launch_conv_f16<2, 0>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out, s)
                         : launch_conv_f16<2, 1>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out, s);
  return out_mode == 0 ? launch_conv_f16<1, 0>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out, s)
                       : launch_conv_f16<1, 1>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out, s);
}

extern "C" int pd3_conv3x3_f16_bias_relu_dual(const void* x_f16_nhwc, const void* w_packed_f16, const float* bias,
                                              int batch, int cin, int cout, int h, int w, int relu, void* out_f16_nhwc,
                                              float* out_f32_nchw, int channels_per_tile, void* stream) {
  if (!x_f16_nhwc || !w_packed_f16 || !out_f16_nhwc || !out_f32_nchw || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 ||
      w <= 0)
    return PD3_EINVAL;
  if (channels_per_tile != 64 && channels_per_tile != 128) return PD3_EINVAL;
  if (reinterpret_cast<uintptr_t>(x_f16_nhwc) % 16 != 0 || reinterpret_cast<uintptr_t>(w_packed_f16) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out_f16_nhwc) % 16 != 0 || reinterpret_cast<uintptr_t>(out_f32_nchw) % 16 != 0)
    return PD3_EINVAL;
  if (cin % kCfKc != 0 || cout % channels_per_tile != 0) return PD3_EUNSUPPORTED;
  if ((int64_t)h * w * cin >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (channels_per_tile == 128)
    return launch_conv_f16<2, 2>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out_f16_nhwc, s, out_f32_nchw);
  return launch_conv_f16<1, 2>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out_f16_nhwc, s, out_f32_nchw);
}

template <int MW, bool GATHER>
static int launch_conv_s2_f16(const void* x, const int32_t* inv, const void* wp, const float* bias, int batch, int cin,
                              int cout, int h, int w, int relu, void* out, hipStream_t s) {
  using S = Cs2Shape<MW>;
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_s2_f16_kernel<MW, GATHER>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS);
  if (e != hipSuccess) return (int)e;
  const int64_t ptiles = (int64_t)batch * ceil_div(ho, kCs2R) * ceil_div(wo, kCfCols);
  const int64_t nwg = (ptiles + 7) / 8 * 8 * (cout / S::M);
  if (nwg >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  conv3x3_s2_f16_kernel<MW, GATHER><<<(unsigned)nwg, S::THREADS, S::LDS, s>>>(
      static_cast<const _Float16*>(x), inv, static_cast<const _Float16*>(wp), bias, static_cast<_Float16*>(out), cin, cout,
      h, w, ho, wo, relu, (int)ptiles);
  return launch_status();
}

extern "C" int pd3_conv3x3_s2_f16_bias_relu(const void* x_f16_nhwc, const void* w_packed_f16, const float* bias, int batch,
                                            int cin, int cout, int h, int w, int relu, void* out_f16_nhwc, void* stream) {
  if (!x_f16_nhwc || !w_packed_f16 || !out_f16_nhwc || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0)
    return PD3_EINVAL;
  if (reinterpret_cast<uintptr_t>(x_f16_nhwc) % 16 != 0 || reinterpret_cast<uintptr_t>(w_packed_f16) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out_f16_nhwc) % 8 != 0)
    return PD3_EINVAL;
  if (cin % kCfKc != 0 || cout % 128 != 0) return PD3_EUNSUPPORTED;
  if ((int64_t)h * w * cin >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  return launch_conv_s2_f16<2, false>(x_f16_nhwc, nullptr, w_packed_f16, bias, batch, cin, cout, h, w, relu,
                                      out_f16_nhwc, static_cast<hipStream_t>(stream));
}

extern "C" int pd3_scatter_conv3x3_s2_f16_bias_relu(const void* features_f16, const int32_t* inverse_map,
                                                    const void* w_packed_f16, const float* bias, int batch, int cin,
                                                    int cout, int ny, int nx, int relu, void* out_f16_nhwc,
                                                    int channels_per_tile, void* stream) {
  if (!features_f16 || !inverse_map || !w_packed_f16 || !out_f16_nhwc || batch <= 0 || cin <= 0 || cout <= 0 || ny <= 0 ||
      nx <= 0)
    return PD3_EINVAL;
  if (reinterpret_cast<uintptr_t>(features_f16) % 16 != 0 || reinterpret_cast<uintptr_t>(w_packed_f16) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out_f16_nhwc) % 8 != 0)
    return PD3_EINVAL;
  if ((channels_per_tile != 64 && channels_per_tile != 128) || cin % kCfKc != 0 || cout % channels_per_tile != 0)
    return PD3_EUNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (channels_per_tile == 128)
    return launch_conv_s2_f16<2, true>(features_f16, inverse_map, w_packed_f16, bias, batch, cin, cout, ny, nx, relu,
                                       out_f16_nhwc, s);
  return launch_conv_s2_f16<1, true>(features_f16, inverse_map, w_packed_f16, bias, batch, cin, cout, ny, nx, relu,
                                     out_f16_nhwc, s);
}

extern "C" int pd3_grouped_conv3x3_small_f16(const void* x_f16_nhwc, const void* w_f16, const float* bias, int batch,
                                             int groups, int channels_per_group, int out_per_group, int h, int w,
                                             float* out, int out_groups, int out_group0, void* stream) {
  if (!x_f16_nhwc || !w_f16 || !out || batch <= 0 || groups <= 0 || h <= 0 || w <= 0 || out_groups < groups ||
      out_group0 < 0 || out_group0 + groups > out_groups)
    return PD3_EINVAL;
  if (channels_per_group != 64 || out_per_group < 1 || out_per_group > 4 || groups > 65535 || batch > 65535)
    return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(x_f16_nhwc) % 16 != 0 || reinterpret_cast<uintptr_t>(w_f16) % 16 != 0) return PD3_EINVAL;
  const dim3 grid((unsigned)(ceil_div(w, 32) * ceil_div(h, 8)), (unsigned)groups, (unsigned)batch);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const _Float16* x = static_cast<const _Float16*>(x_f16_nhwc);
  const _Float16* wg = static_cast<const _Float16*>(w_f16);
  switch (out_per_group) {
    case 1: grouped_conv3x3_small_f16_kernel<1><<<grid, 256, 0, s>>>(x, wg, bias, groups, h, w, out, out_groups, out_group0); break;
    case 2: grouped_conv3x3_small_f16_kernel<2><<<grid, 256, 0, s>>>(x, wg, bias, groups, h, w, out, out_groups, out_group0); break;
    case 3: grouped_conv3x3_small_f16_kernel<3><<<grid, 256, 0, s>>>(x, wg, bias, groups, h, w, out, out_groups, out_group0); break;
    default: grouped_conv3x3_small_f16_kernel<4><<<grid, 256, 0, s>>>(x, wg, bias, groups, h, w, out, out_groups, out_group0); break;
  }
  return launch_status();
}

extern "C" int pd3_f32_nchw_to_f16_nhwc(const float* x, int batch, int channels, int h, int w, void* out, void* stream) {
  if (!x || !out || batch <= 0 || channels <= 0 || h <= 0 || w <= 0) return PD3_EINVAL;
  const int64_t blocks = (int64_t)batch * h * ((w + 63) / 64);
  if (blocks >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  f32_nchw_to_f16_nhwc_kernel<<<(unsigned)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(
      x, channels, h, w, static_cast<_Float16*>(out));
  return launch_status();
}
