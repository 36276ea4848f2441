// Sparse 3-D convolution (submanifold + regular/strided) for gfx950.
// Reference: the CenterPoint-Voxel middle encoder builds on Paddle-core `paddle.sparse.nn.SubmConv3D /
// Conv3D / BatchNorm / ReLU` and `paddle.sparse.add` (call sites: paddle3d/models/middle_encoders/
// sparse_resnet.py:31-59, :115-206; sparsenet.py:31-64).  The arithmetic lives in the paddlepaddle wheel
// (>= 2.4.0, not vendored, not installable here), so parity is pinned on the PUBLIC definition instead:
//   submanifold conv: output index set = input index set, out[p] = sum_k W[k] . in[p + k - pad]
//   regular conv:     output index set = every q reached by some active input, out[q] = sum_k W[k] .
//                     in[q * stride - pad + k]        (weight layout [kd, kh, kw, Cin, Cout])
// and checked against torch.nn.functional.conv3d on densified inputs (tests/test_sparse_conv_gpu.py).
//
// Design: coordinates are linearised to uint32 keys ((b*D+z)*H+y)*W+x.  An open-addressing hash table in
// global memory (atomicCAS insert, linear probing) answers "which row holds coordinate c".  A convolution
// is (1) its output key set -- the input set itself (submanifold), or the radix-sorted unique set of
// reachable outputs (regular; sorted => deterministic row order), (2) a neighbour table nbr[row][K] of
// input rows (-1 = absent), (3) ONE gather-GEMM kernel that walks the K offsets in order for a tile of
// output rows, staging W[k] and the gathered input rows in LDS -- a fixed summation order, so results are
// run-to-run identical -- with bias / folded BatchNorm / residual add / ReLU fused into the epilogue.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"

#include <algorithm>

namespace pd3 {

constexpr uint32_t kSpEmpty = 0xFFFFFFFFu;

struct SpShape {
  int batch, d, h, w;
};

__device__ __forceinline__ uint32_t sp_hash(uint32_t k) {
  k ^= k >> 16;
  k *= 0x7feb352dU;
  k ^= k >> 15;
  k *= 0x846ca68bU;
  k ^= k >> 16;
  return k;
}

__device__ __forceinline__ uint32_t sp_key(int b, int z, int y, int x, const SpShape& s) {
  return (uint32_t)(((b * s.d + z) * s.h + y) * s.w + x);
}

// coords [n,4] (b,z,y,x) -> keys; rows with b < 0 (padding) get kSpEmpty
__global__ __launch_bounds__(256) void sp_keys_kernel(const int32_t* __restrict__ coords, int n,
                                                      SpShape s, uint32_t* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = coords[i * 4], z = coords[i * 4 + 1], y = coords[i * 4 + 2], x = coords[i * 4 + 3];
  const bool ok = b >= 0 && b < s.batch && z >= 0 && z < s.d && y >= 0 && y < s.h && x >= 0 && x < s.w;
  keys[i] = ok ? sp_key(b, z, y, x, s) : kSpEmpty;
}

__global__ __launch_bounds__(256) void sp_hash_insert_kernel(const uint32_t* __restrict__ keys, int n,
                                                             uint32_t* __restrict__ tkeys,
                                                             int* __restrict__ tvals, uint32_t mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = keys[i];
  if (k == kSpEmpty) return;
  uint32_t slot = sp_hash(k) & mask;
  while (true) {
    const uint32_t prev = atomicCAS(&tkeys[slot], kSpEmpty, k);
    if (prev == kSpEmpty || prev == k) {
      // duplicate coordinates (never produced by the voxelizer): the highest row wins, deterministically
      atomicMax(&tvals[slot], i);
      return;
    }
    slot = (slot + 1) & mask;
  }
}

__device__ __forceinline__ int sp_lookup(uint32_t k, const uint32_t* __restrict__ tkeys,
                                         const int* __restrict__ tvals, uint32_t mask) {
  uint32_t slot = sp_hash(k) & mask;
  while (true) {
    const uint32_t cur = tkeys[slot];
    if (cur == k) return tvals[slot];
    if (cur == kSpEmpty) return -1;
    slot = (slot + 1) & mask;
  }
}

struct SpConv {
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
};

// neighbour table: thread per (output row, kernel offset).  out coordinates come from out_keys.
__global__ __launch_bounds__(256) void sp_rulebook_kernel(const uint32_t* __restrict__ out_keys,
                                                          const int* __restrict__ n_out_dev,
                                                          int n_out_cap, SpShape in_s, SpShape out_s,
                                                          SpConv c, const uint32_t* __restrict__ tkeys,
                                                          const int* __restrict__ tvals, uint32_t mask,
                                                          int32_t* __restrict__ nbr) {
  const int K = c.kd * c.kh * c.kw;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int n_out = n_out_dev ? min(*n_out_dev, n_out_cap) : n_out_cap;
  if (t >= (int64_t)n_out * K) return;
  const int row = (int)(t / K), k = (int)(t - (int64_t)row * K);
  const uint32_t key = out_keys[row];
  int j = -1;
  if (key != kSpEmpty) {
    uint32_t r = key;
    const int x = (int)(r % (uint32_t)out_s.w);
    r /= (uint32_t)out_s.w;
    const int y = (int)(r % (uint32_t)out_s.h);
    r /= (uint32_t)out_s.h;
    const int z = (int)(r % (uint32_t)out_s.d);
    const int b = (int)(r / (uint32_t)out_s.d);
    const int kz = k / (c.kh * c.kw), ky = (k / c.kw) % c.kh, kx = k % c.kw;
    const int iz = z * c.sd - c.pd + kz, iy = y * c.sh - c.ph + ky, ix = x * c.sw - c.pw + kx;
    if (iz >= 0 && iz < in_s.d && iy >= 0 && iy < in_s.h && ix >= 0 && ix < in_s.w)
      j = sp_lookup(sp_key(b, iz, iy, ix, in_s), tkeys, tvals, mask);
  }
  nbr[t] = j;
}

// regular conv: every (input row, offset) proposes the output it contributes to (or kSpEmpty)
__global__ __launch_bounds__(256) void sp_candidates_kernel(const uint32_t* __restrict__ in_keys, int n_in,
                                                            SpShape in_s, SpShape out_s, SpConv c,
                                                            uint32_t* __restrict__ cand) {
  const int K = c.kd * c.kh * c.kw;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n_in * K) return;
  const int row = (int)(t / K), k = (int)(t - (int64_t)row * K);
  const uint32_t key = in_keys[row];
  uint32_t out = kSpEmpty;
  if (key != kSpEmpty) {
    uint32_t r = key;
    const int x = (int)(r % (uint32_t)in_s.w);
    r /= (uint32_t)in_s.w;
    const int y = (int)(r % (uint32_t)in_s.h);
    r /= (uint32_t)in_s.h;
    const int z = (int)(r % (uint32_t)in_s.d);
    const int b = (int)(r / (uint32_t)in_s.d);
    const int kz = k / (c.kh * c.kw), ky = (k / c.kw) % c.kh, kx = k % c.kw;
    const int nz = z + c.pd - kz, ny = y + c.ph - ky, nx = x + c.pw - kx;  // = q * stride
    if (nz >= 0 && ny >= 0 && nx >= 0 && nz % c.sd == 0 && ny % c.sh == 0 && nx % c.sw == 0) {
      const int qz = nz / c.sd, qy = ny / c.sh, qx = nx / c.sw;
      if (qz < out_s.d && qy < out_s.h && qx < out_s.w) out = sp_key(b, qz, qy, qx, out_s);
    }
  }
  cand[t] = out;
}

// sorted candidates -> head flags (first of each run of equal valid keys), as ints for the scan
__global__ __launch_bounds__(256) void sp_heads_kernel(const uint32_t* __restrict__ sorted, int64_t n,
                                                       int* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = sorted[i];
  flags[i] = (k != kSpEmpty && (i == 0 || sorted[i - 1] != k)) ? 1 : 0;
}

struct EpiUniqueKeys {
  const uint32_t* sorted;
  uint32_t* out_keys;
  int32_t* out_coords;
  SpShape s;
  int cap;
  __device__ __forceinline__ void operator()(int, int64_t i, int flag, int prefix, int) const {
    if (flag && prefix < cap) {
      uint32_t r = sorted[i];
      out_keys[prefix] = r;
      const int x = (int)(r % (uint32_t)s.w);
      r /= (uint32_t)s.w;
      const int y = (int)(r % (uint32_t)s.h);
      r /= (uint32_t)s.h;
      out_coords[prefix * 4 + 3] = x;
      out_coords[prefix * 4 + 2] = y;
      out_coords[prefix * 4 + 1] = (int)(r % (uint32_t)s.d);
      out_coords[prefix * 4 + 0] = (int)(r / (uint32_t)s.d);
    }
  }
};

// ---------------------------------------------------------------------------------------------------
// gather-GEMM: tile of kSpRows output rows per workgroup (4 waves x 8 rows), lanes = output channels.
// ---------------------------------------------------------------------------------------------------
constexpr int kSpRows = 32;
constexpr int kSpRowsPerWave = 8;

struct SpGemmArgs {
  const float* in;       // [n_in, cin]
  const int32_t* nbr;    // [n_out, K]
  const float* weight;   // [K, cin, cout]
  const float* bias;     // [cout] or null
  const float* scale;    // [cout] or null (folded BatchNorm)
  const float* shift;    // [cout] or null
  const float* residual; // [n_out, cout] or null
  float* out;            // [n_out, cout]
  const int* n_out_dev;  // device row count or null
  int n_out_cap, K, cin, cout, relu;
  const int32_t* order;  // optional tile order (sp_tile_order_kernel): slot -> output row or -1; a scheduling
                         // hint only -- which rows share a tile never changes a row's result
};

template <int COUT_PER_LANE>
__global__ __launch_bounds__(256) void sp_gather_gemm_kernel(SpGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sp_smem[];
  float* W = sp_smem;                       // [cin][cout]
  float* xs = W + (size_t)a.cin * a.cout;   // [kSpRows][cin]
  int* nb = reinterpret_cast<int*>(xs + (size_t)kSpRows * a.cin);  // [kSpRows]
  const int lane = lane_id(), wave = wave_id();
  const int n_out = a.n_out_dev ? min(*a.n_out_dev, a.n_out_cap) : a.n_out_cap;
  const int row0 = blockIdx.x * kSpRows;
  if (row0 >= n_out) return;
  float acc[kSpRowsPerWave][COUT_PER_LANE];
#pragma unroll
  for (int r = 0; r < kSpRowsPerWave; ++r)
#pragma unroll
    for (int u = 0; u < COUT_PER_LANE; ++u) acc[r][u] = 0.f;

  for (int k = 0; k < a.K; ++k) {
    __syncthreads();  // previous offset's W / xs fully consumed
    if (threadIdx.x < kSpRows) {
      const int row = row0 + threadIdx.x;
      nb[threadIdx.x] = row < n_out ? a.nbr[(int64_t)row * a.K + k] : -1;
    }
    __syncthreads();
    // does any row of the tile have this neighbour?  (uniform: every thread reads the same 32 ints)
    int any = 0;
    for (int r = 0; r < kSpRows; ++r) any |= (nb[r] >= 0);
    if (!any) continue;
    const float* wk = a.weight + (int64_t)k * a.cin * a.cout;
    for (int e = threadIdx.x; e < a.cin * a.cout; e += blockDim.x) W[e] = wk[e];
    for (int e = threadIdx.x; e < kSpRows * a.cin; e += blockDim.x) {
      const int r = e / a.cin, ci = e - r * a.cin;
      const int j = nb[r];
      xs[e] = j >= 0 ? a.in[(int64_t)j * a.cin + ci] : 0.f;
    }
    __syncthreads();
    bool mine = false;
#pragma unroll
    for (int r = 0; r < kSpRowsPerWave; ++r) mine |= nb[wave * kSpRowsPerWave + r] >= 0;
    if (!mine) continue;  // wave-uniform
    for (int ci = 0; ci < a.cin; ++ci) {
      float w[COUT_PER_LANE];
#pragma unroll
      for (int u = 0; u < COUT_PER_LANE; ++u) {
        const int co = lane + u * kWave;
        w[u] = co < a.cout ? W[ci * a.cout + co] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < kSpRowsPerWave; ++r) {
        const float x = xs[(wave * kSpRowsPerWave + r) * a.cin + ci];
#pragma unroll
        for (int u = 0; u < COUT_PER_LANE; ++u) acc[r][u] = fmaf(x, w[u], acc[r][u]);
      }
    }
  }
  // epilogue: bias, folded BN, residual, ReLU
#pragma unroll
  for (int r = 0; r < kSpRowsPerWave; ++r) {
    const int row = row0 + wave * kSpRowsPerWave + r;
    if (row >= n_out) continue;
#pragma unroll
    for (int u = 0; u < COUT_PER_LANE; ++u) {
      const int co = lane + u * kWave;
      if (co >= a.cout) continue;
      float v = acc[r][u];
      if (a.bias) v += a.bias[co];
      if (a.scale) v = fmaf(v, a.scale[co], a.shift[co]);
      if (a.residual) v += a.residual[(int64_t)row * a.cout + co];
      if (a.relu) v = fmaxf(v, 0.f);
      a.out[(int64_t)row * a.cout + co] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// gather-GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, one fma chain per output).
// Tile = 64 output rows per workgroup, 16 per wave; a wave owns its rows for all NB = Cout/16 column
// blocks (NB independent accumulators keep the 40-cycle dependent latency hidden).  Per kernel offset k:
// the wave gathers ITS 16 input rows into LDS (wave-local, no workgroup barrier), the workgroup stages
// W[k] in chunks of <= 32 input channels, and every lane feeds the MFMA one A and one B value from LDS:
//   A[i = lane & 15][kk = lane >> 4] = x[row i][ci0 + kk],  B[kk][j = lane & 15] = W[k][ci0 + kk][nb*16 + j],
//   D: col = lane & 15, row = (lane >> 4) * 4 + reg          (cdna_hip_programming.md section 3).
// Offsets no row of the tile needs are skipped for the workgroup, offsets none of a wave's 16 rows needs
// are skipped for that wave; the summation order over k is fixed => bit-reproducible results.
// LDS strides: A rows are padded to stride == 2 (mod 16) floats and W rows to Cout + 16 floats so that
// the 32-lane groups of a ds_read_b32 hit 32 distinct banks.
// ---------------------------------------------------------------------------------------------------
constexpr int kSpTile = 64;
constexpr int kSpChunk = 32;  // input channels of W staged per step
typedef float sp_f32x4 __attribute__((ext_vector_type(4)));

template <int NB>
__global__ __launch_bounds__(256) void sp_gemm_mfma_kernel(SpGemmArgs a, int cin_pad, int astride,
                                                           int wstride) {
  extern __shared__ __attribute__((aligned(16))) float sp_smem[];
  float* As_all = sp_smem;                                      // [4][16][astride]
  float* Ws = As_all + (size_t)4 * 16 * astride;                // [kSpChunk][wstride]
  int* nb = reinterpret_cast<int*>(Ws + (size_t)kSpChunk * wstride);  // [kSpTile]
  const int lane = lane_id(), wave = wave_id();
  float* As = As_all + (size_t)wave * 16 * astride;
  const int n_out = a.n_out_dev ? min(*a.n_out_dev, a.n_out_cap) : a.n_out_cap;
  const int row0 = blockIdx.x * kSpTile;
  if (row0 >= n_out) return;
  const int cout = NB * 16;
  sp_f32x4 acc[NB];
#pragma unroll
  for (int u = 0; u < NB; ++u) acc[u] = sp_f32x4{0.f, 0.f, 0.f, 0.f};
  const int ar = lane & 15, akk = lane >> 4;
  const int nchunks = (cin_pad + kSpChunk - 1) / kSpChunk;
  const bool vec_rows = (a.cin % 4) == 0;

  for (int k = 0; k < a.K; ++k) {
    int mine = -1;
    if (threadIdx.x < kSpTile) {
      const int row = row0 + threadIdx.x;
      mine = row < n_out ? a.nbr[(int64_t)row * a.K + k] : -1;
      nb[threadIdx.x] = mine;
    }
    // barrier: nb visible, previous offset's Ws consumed; OR-reduce "someone needs this offset"
    if (!__syncthreads_or(mine >= 0)) continue;
    const int myj = nb[wave * 16 + ar];
    const bool wave_any = __ballot(myj >= 0) != 0ull;
    if (wave_any) {
      // gather this wave's 16 rows (zero rows where the neighbour is absent); float4 along channels
      const int q4 = cin_pad / 4;
      for (int e = lane; e < 16 * q4; e += kWave) {
        const int r = e / q4, c4 = e - r * q4;
        const int j = nb[wave * 16 + r];
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (j >= 0) {
          const float* src = a.in + (int64_t)j * a.cin + c4 * 4;
          if (vec_rows) {
            const float4 t = *reinterpret_cast<const float4*>(src);
            v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w;
          } else {
            const int left = a.cin - c4 * 4;
            if (left > 0) v0 = src[0];
            if (left > 1) v1 = src[1];
            if (left > 2) v2 = src[2];
            if (left > 3) v3 = src[3];
          }
        }
        float* dst = As + r * astride + c4 * 4;
        dst[0] = v0; dst[1] = v1; dst[2] = v2; dst[3] = v3;
      }
    }
    const float* wk = a.weight + (int64_t)k * a.cin * cout;
    // W[k] travels global -> registers one chunk ahead of its use (the loads of chunk c+1 are in flight while the
    // matrix cores work on chunk c), then registers -> LDS between the two barriers of the chunk
    constexpr int WPT = kSpChunk * NB * 16 / 4 / 256;  // float4 of a full chunk per thread (NB >= 2)
    sp_f32x4 wreg[WPT > 0 ? WPT : 1];
    const int cq = cout / 4;
    auto fetch_w = [&](int c) {
      const int ci0 = c * kSpChunk;
      const int rows = min(kSpChunk, cin_pad - ci0);
#pragma unroll
      for (int i = 0; i < (WPT > 0 ? WPT : 1); ++i) {
        const int e = threadIdx.x + i * 256;
        const int r = e / cq, c4 = e - r * cq;
        sp_f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (e < rows * cq && ci0 + r < a.cin) t = *reinterpret_cast<const sp_f32x4*>(wk + (int64_t)(ci0 + r) * cout + c4 * 4);
        wreg[i] = t;
      }
    };
    fetch_w(0);
    for (int c = 0; c < nchunks; ++c) {
      if (c > 0) __syncthreads();  // previous chunk of Ws consumed
      const int ci0 = c * kSpChunk;
      const int rows = min(kSpChunk, cin_pad - ci0);
#pragma unroll
      for (int i = 0; i < (WPT > 0 ? WPT : 1); ++i) {
        const int e = threadIdx.x + i * 256;
        const int r = e / cq, c4 = e - r * cq;
        if (e < rows * cq) {
          float* dst = Ws + r * wstride + c4 * 4;
          dst[0] = wreg[i][0]; dst[1] = wreg[i][1]; dst[2] = wreg[i][2]; dst[3] = wreg[i][3];
        }
      }
      if (c + 1 < nchunks) fetch_w(c + 1);
      __syncthreads();
      if (wave_any) {
        for (int s4 = 0; s4 < rows; s4 += 4) {
          const float av = As[ar * astride + ci0 + s4 + akk];
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const float bv = Ws[(s4 + akk) * wstride + u * 16 + ar];
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[u], 0, 0, 0);
          }
        }
      }
    }
  }
  // epilogue: D layout col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int co = u * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + wave * 16 + (lane >> 4) * 4 + r;
      if (row >= n_out) continue;
      float v = acc[u][r];
      if (a.bias) v += a.bias[co];
      if (a.scale) v = fmaf(v, a.scale[co], a.shift[co]);
      if (a.residual) v += a.residual[(int64_t)row * cout + co];
      if (a.relu) v = fmaxf(v, 0.f);
      a.out[(int64_t)row * cout + co] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// gather-GEMM, second form: the one the encoder's layers run (cin a multiple of 16, K <= 27).
// Tile = 128 output rows per workgroup (4 waves x 2 blocks of 16 rows) x all Cout = 16 NB columns.
// The A operand never touches LDS: with the MFMA's K index kk = lane >> 4 standing for the input channels
// ci0 + kk*T + t (t = K step 0..T-1; A and B only have to agree on the assignment), lane (i, kk) needs T
// CONSECUTIVE floats of input row nbr[row i][k] -- one or two 16-byte global loads straight into the operand
// registers (the four kk lanes of a row read one 64/128-byte piece of it), fetched one (offset, chunk) step
// ahead of their MFMAs.  W[k] chunks (4T input channels x Cout) go global -> registers (a step ahead) -> LDS,
// transposed on the way into "one line per lane": line (kk, j) holds W[ci0 + kk*T + t][16 u + j] as [u][t], so a
// lane's B values for a column block are T consecutive floats (ds_read_b128; lines are padded to NB*T + 4
// floats: conflict-free).  LDS is double-buffered: one barrier per step.  Per step and wave: 2*NB*T MFMAs
// behind 2*NB*T/8 LDS reads.
// The tile's [128][K] slice of the rulebook is loaded once (coalesced); offsets that no row of the workgroup /
// of a wave needs are skipped for the workgroup / the wave.  The summation order (k ascending, then the
// channel chunks, then t) is fixed => bit-reproducible results.
// ---------------------------------------------------------------------------------------------------
constexpr int kSg2Rows = 128;
constexpr int kSg2MaxK = 27;

template <int NB, int T>
__global__ __launch_bounds__(256, 2) void sp_gemm_rows_kernel(SpGemmArgs a) {
  constexpr int CH = 4 * T;                 // input channels per step
  constexpr int S = NB * T + 4;             // floats per W line in LDS
  constexpr int WSZ = 64 * S;               // one staged W chunk
  constexpr int WQ = CH * NB * 4;           // float4 of a W chunk in global memory
  constexpr int WPT = (WQ + 255) / 256;     // per thread
  constexpr int AQ = T / 4;                 // float4 of A per lane and row block
  extern __shared__ __attribute__((aligned(16))) float sp_smem[];
  float* Ws = sp_smem;                                        // [2][WSZ]
  int* nbs = reinterpret_cast<int*>(Ws + 2 * WSZ);            // [128][K]
  uint32_t* masks = reinterpret_cast<uint32_t*>(nbs + kSg2Rows * a.K);  // [0] workgroup, [1 + row block]
  int* rows = reinterpret_cast<int*>(masks + 16);             // [128] output row of a slot (-1: none)
  const int lane = lane_id(), wave = wave_id();
  const int K = a.K, cin = a.cin;
  const int n_out = a.n_out_dev ? min(*a.n_out_dev, a.n_out_cap) : a.n_out_cap;
  const int tile = sp_window_tile(blockIdx.x, (n_out + kSg2Rows - 1) / kSg2Rows, 8192 / kSg2Rows);  // (a window per XCD)
  if (tile < 0) return;  // (ordered: rows past n_out sort last in the last window, so whole tiles past it hold nothing)
  const int row0 = tile * kSg2Rows;
  if (a.order) {
    // the tile's rows come from the tile order: rows of one window with similar neighbour masks share a 16-row
    // block, so fewer (block, offset) steps run on rows that lack the neighbour
    if (threadIdx.x < kSg2Rows) {
      const int r = a.order[row0 + threadIdx.x];
      rows[threadIdx.x] = r >= 0 && r < n_out ? r : -1;
    }
    if (threadIdx.x < 9) masks[threadIdx.x] = 0u;
    __syncthreads();
{  // rulebook rows of the tile -> LDS, eight loads per thread in flight (see sparse_conv_x3.hip)
    const int total = kSg2Rows * K;
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
  } else {
    const int live = min(kSg2Rows, n_out - row0) * K;
    const int32_t* src = a.nbr + (int64_t)row0 * K;
    for (int e = threadIdx.x; e < kSg2Rows * K; e += 256) nbs[e] = e < live ? src[e] : -1;
    if (threadIdx.x < kSg2Rows) rows[threadIdx.x] = row0 + (int)threadIdx.x < n_out ? row0 + (int)threadIdx.x : -1;
    if (threadIdx.x < 9) masks[threadIdx.x] = 0u;
  }
  __syncthreads();
  if (threadIdx.x < kSg2Rows) {  // which offsets does each block of 16 rows need
    uint32_t m = 0;
    for (int k = 0; k < K; ++k) m |= nbs[threadIdx.x * K + k] >= 0 ? 1u << k : 0u;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) m |= (uint32_t)__shfl_xor((int)m, d, kWave);
    if ((lane & 15) == 0 && m) {
      atomicOr(&masks[1 + (threadIdx.x >> 4)], m);
      atomicOr(&masks[0], m);
    }
  }
  __syncthreads();
  const uint32_t wg_mask = masks[0];
  const uint32_t blk_mask[2] = {masks[1 + 2 * wave], masks[2 + 2 * wave]};  // per 16-row block of this wave
  const uint32_t wave_mask = blk_mask[0] | blk_mask[1];
  const int nchunks = cin / CH;
  const int cout = NB * 16;
  const int ai = lane & 15, akk = lane >> 4;

  sp_f32x4 acc[NB][2];
#pragma unroll
  for (int u = 0; u < NB; ++u) acc[u][0] = acc[u][1] = sp_f32x4{0.f, 0.f, 0.f, 0.f};
  sp_f32x4 wreg[WPT], acur[2][AQ], anext[2][AQ];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int q = 0; q < AQ; ++q) acur[rb][q] = anext[rb][q] = sp_f32x4{0.f, 0.f, 0.f, 0.f};

  auto fetch_w = [&](int k, int c) {
    const float* wk = a.weight + ((int64_t)k * cin + (int64_t)c * CH) * cout;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = threadIdx.x + i * 256;
      wreg[i] = sp_f32x4{0.f, 0.f, 0.f, 0.f};
      if (WQ % 256 == 0 || e < WQ) wreg[i] = *reinterpret_cast<const sp_f32x4*>(wk + (int64_t)e * 4);
    }
  };
  auto stash_w = [&](float* dst) {  // row r = kk*T + t of the chunk, columns 4 c4 .. 4 c4 + 3 = 16 u + j ..
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = threadIdx.x + i * 256;
      if (WQ % 256 != 0 && e >= WQ) break;
      const int r = e / (NB * 4), c4 = e - r * (NB * 4);
      const int kk = r / T, t = r - kk * T, u = c4 >> 2, j = (c4 & 3) * 4;
      float* d = dst + (kk * 16 + j) * S + u * T + t;
      d[0] = wreg[i][0];
      d[S] = wreg[i][1];
      d[2 * S] = wreg[i][2];
      d[3 * S] = wreg[i][3];
    }
  };
  auto fetch_a = [&](int k, int c, sp_f32x4 (&dst)[2][AQ]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      if (!((blk_mask[rb] >> k) & 1u)) continue;  // (block-uniform: no row of the block has this neighbour)
      const int j = nbs[(wave * 32 + rb * 16 + ai) * K + k];
#pragma unroll
      for (int q = 0; q < AQ; ++q) dst[rb][q] = sp_f32x4{0.f, 0.f, 0.f, 0.f};
      if (j >= 0) {
        const sp_f32x4* src = reinterpret_cast<const sp_f32x4*>(a.in + (int64_t)j * cin + c * CH + akk * T);
#pragma unroll
        for (int q = 0; q < AQ; ++q) dst[rb][q] = src[q];
      }
    }
  };
  auto next_step = [&](int& k, int& c) {  // (k, c) -> the following step, k = -1 at the end
    if (++c < nchunks) return;
    c = 0;
    const uint32_t rest = k + 1 < 32 ? wg_mask >> (k + 1) : 0u;
    k = rest ? k + 1 + __builtin_ctz(rest) : -1;
  };

  int k = wg_mask ? __builtin_ctz(wg_mask) : -1, c = 0, buf = 0;
  if (k >= 0) {
    fetch_w(k, c);
    if ((wave_mask >> k) & 1u) fetch_a(k, c, acur);
    stash_w(Ws);
  }
  __syncthreads();
  while (k >= 0) {
    int k2 = k, c2 = c;
    next_step(k2, c2);
    const bool more = k2 >= 0;
    const bool need = (wave_mask >> k) & 1u, need2 = more && ((wave_mask >> k2) & 1u);
    if (more) fetch_w(k2, c2);
    if (need2) fetch_a(k2, c2, anext);
    if (need) {
      const bool need_b0 = (blk_mask[0] >> k) & 1u, both = need_b0 && ((blk_mask[1] >> k) & 1u);
      const float* wl = Ws + buf * WSZ + lane * S;
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        float b[T];
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
          const sp_f32x4 v = *reinterpret_cast<const sp_f32x4*>(wl + u * T + q * 4);
          b[q * 4 + 0] = v[0];
          b[q * 4 + 1] = v[1];
          b[q * 4 + 2] = v[2];
          b[q * 4 + 3] = v[3];
        }
        // a 16-row block whose rows all lack this neighbour is skipped (its A registers hold stale values)
        if (both) {
#pragma unroll
          for (int t = 0; t < T; ++t) {
            acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[0][t >> 2][t & 3], b[t], acc[u][0], 0, 0, 0);
            acc[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[1][t >> 2][t & 3], b[t], acc[u][1], 0, 0, 0);
          }
        } else if (need_b0) {
#pragma unroll
          for (int t = 0; t < T; ++t)
            acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[0][t >> 2][t & 3], b[t], acc[u][0], 0, 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < T; ++t)
            acc[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[1][t >> 2][t & 3], b[t], acc[u][1], 0, 0, 0);
        }
      }
    }
    if (more) stash_w(Ws + (buf ^ 1) * WSZ);
    if (need2) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int q = 0; q < AQ; ++q) acur[rb][q] = anext[rb][q];
    }
    __syncthreads();
    buf ^= 1;
    k = k2;
    c = c2;
  }
  // epilogue: D layout col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int co = u * 16 + (lane & 15);
    const float bias = a.bias ? a.bias[co] : 0.f;
    const float sc = a.scale ? a.scale[co] : 1.f, sh = a.scale ? a.shift[co] : 0.f;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rows[wave * 32 + rb * 16 + (lane >> 4) * 4 + r];
        if (row < 0) continue;
        float v = acc[u][rb][r];
        if (a.bias) v += bias;
        if (a.scale) v = fmaf(v, sc, sh);
        if (a.residual) v += a.residual[(int64_t)row * cout + co];
        if (a.relu) v = fmaxf(v, 0.f);
        a.out[(int64_t)row * cout + co] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Tile order (round 5).  sp_gemm_rows_kernel runs a (16-row block, kernel offset) step whenever ANY row of the block
// has that neighbour; with rows in raster order a block's rows have different neighbour masks and 1.3x (128 -> 128)
// to 7x (the strided layers) more steps run than (row, offset) pairs exist.  Which rows share a tile is free -- a
// row's result does not depend on it -- so the rows of one WINDOW of kSpWindow consecutive rows are sorted by their
// 27-bit neighbour mask (ties by row: a deterministic order) and tiles take their rows from that order: rows with
// equal or similar masks end up in one block.  A window is still a piece of the raster order (a few neighbouring
// lines), so a tile's gathers keep their cache locality.  One workgroup per window: masks from the rulebook, a
// bitonic sort of (mask << 32 | index in window) in LDS, order[window slot] = row (or -1 past n_out).
// Measured on config 4 (two frames): steps / pairs 1.73 -> 1.31 (64 -> 64), 2.15 -> 1.41 (32 -> 32), 1.29 -> 1.17
// (128 -> 128), 4.8 - 7.0 -> 1.9 - 2.0 on the strided layers.
// ---------------------------------------------------------------------------------------------------
constexpr int kSpWindow = 8192;  // (= the window of sp_window_tile in the gather-GEMMs)
constexpr int kSpOrderThreads = 1024;
constexpr int kSpOrderEpt = kSpWindow / kSpOrderThreads;  // 8 consecutive rows of the window per thread

// One compare-exchange stage of the bitonic network over the window, element i = 8 * thread + r held in registers
// (key = neighbour mask, val = row in window).  Partners 1 / 2 / 4 apart are registers of the same thread, partners 8 ..
// 256 apart are the same register of another lane of the wave (ds_bpermute, no barrier), only partners 512 .. 4096 apart
// live in another wave and travel through LDS: 10 of the 91 stages pay a workgroup barrier (the first version ran all
// 91 through LDS: 160 us per rulebook, 1.3 ms per 8-scene step).  Equal keys stay where they are (both sides compare
// strictly), so the network is deterministic without unique keys.
template <int SIZE, int J>
__device__ __forceinline__ void sp_order_stage(uint32_t (&key)[kSpOrderEpt], uint32_t (&val)[kSpOrderEpt],
                                               uint32_t* lk, uint32_t* lv) {
  const int t = threadIdx.x;
  if constexpr (J < kSpOrderEpt) {
#pragma unroll
    for (int r = 0; r < kSpOrderEpt; ++r) {
      if ((r & J) != 0) continue;
      const bool up = ((t * kSpOrderEpt + r) & SIZE) == 0;
      const bool sw = up ? key[r] > key[r | J] : key[r] < key[r | J];
      const uint32_t k0 = key[r], v0 = val[r];
      key[r] = sw ? key[r | J] : k0;
      val[r] = sw ? val[r | J] : v0;
      key[r | J] = sw ? k0 : key[r | J];
      val[r | J] = sw ? v0 : val[r | J];
    }
  } else {
    constexpr int TJ = J / kSpOrderEpt;  // partner thread = t ^ TJ, same register
    const bool lower = (t & TJ) == 0;
    if constexpr (TJ >= kWave) {
#pragma unroll
      for (int r = 0; r < kSpOrderEpt; ++r) {
        lk[r * kSpOrderThreads + t] = key[r];
        lv[r * kSpOrderThreads + t] = val[r];
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < kSpOrderEpt; ++r) {
      uint32_t pk, pv;
      if constexpr (TJ >= kWave) {
        pk = lk[r * kSpOrderThreads + (t ^ TJ)];
        pv = lv[r * kSpOrderThreads + (t ^ TJ)];
      } else {
        pk = (uint32_t)__shfl_xor((int)key[r], TJ, kWave);
        pv = (uint32_t)__shfl_xor((int)val[r], TJ, kWave);
      }
      const bool up = ((t * kSpOrderEpt + r) & SIZE) == 0;
      const bool take = (lower == up) ? pk < key[r] : pk > key[r];  // this side keeps the smaller / the larger key
      key[r] = take ? pk : key[r];
      val[r] = take ? pv : val[r];
    }
    if constexpr (TJ >= kWave) __syncthreads();  // the next LDS stage overwrites the exchange area
  }
}
template <int SIZE, int J>
__device__ __forceinline__ void sp_order_merge(uint32_t (&key)[kSpOrderEpt], uint32_t (&val)[kSpOrderEpt],
                                               uint32_t* lk, uint32_t* lv) {
  sp_order_stage<SIZE, J>(key, val, lk, lv);
  if constexpr (J > 1) sp_order_merge<SIZE, J / 2>(key, val, lk, lv);
}
template <int SIZE>
__device__ __forceinline__ void sp_order_sort(uint32_t (&key)[kSpOrderEpt], uint32_t (&val)[kSpOrderEpt],
                                              uint32_t* lk, uint32_t* lv) {
  if constexpr (SIZE > 2) sp_order_sort<SIZE / 2>(key, val, lk, lv);
  sp_order_merge<SIZE, SIZE / 2>(key, val, lk, lv);
}

__global__ __launch_bounds__(kSpOrderThreads) void sp_tile_order_kernel(
    const int32_t* __restrict__ nbr, const int* __restrict__ n_out_dev, int n_out_cap, int K,
    int32_t* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_order_smem[];
  uint32_t* lk = reinterpret_cast<uint32_t*>(sp_order_smem);  // [8][1024] exchange area: keys
  uint32_t* lv = lk + kSpWindow;                               // values
  const int n_out = n_out_dev ? min(*n_out_dev, n_out_cap) : n_out_cap;
  const int win0 = blockIdx.x * kSpWindow;
  const int t = threadIdx.x;
  if (win0 >= n_out) {  // a window of padding only
#pragma unroll
    for (int r = 0; r < kSpOrderEpt; ++r) order[win0 + r * kSpOrderThreads + t] = -1;
    return;
  }
  uint32_t key[kSpOrderEpt], val[kSpOrderEpt];
#pragma unroll
  for (int r = 0; r < kSpOrderEpt; ++r) {
    const int i = t * kSpOrderEpt + r, row = win0 + i;
    uint32_t m = 0xFFFFFFFFu;  // rows past the count sort last (a real mask has at most 31 bits)
    if (row < n_out) {
      m = 0;
      const int32_t* src = nbr + (int64_t)row * K;
      if (K == 27) {  // the encoder's kernels: the row's 108 bytes as 16-byte pieces (4-byte aligned loads)
        struct Row27 { int32_t v[27]; } rw;
        __builtin_memcpy(&rw, src, sizeof(rw));
#pragma unroll
        for (int k = 0; k < 27; ++k) m |= rw.v[k] >= 0 ? 1u << k : 0u;
      } else {
        for (int k = 0; k < K; ++k) m |= src[k] >= 0 ? 1u << k : 0u;
      }
    }
    key[r] = m;
    val[r] = (uint32_t)i;
  }
  sp_order_sort<kSpWindow>(key, val, lk, lv);
  // element i of the sorted window leaves through LDS so that the stores are coalesced
#pragma unroll
  for (int r = 0; r < kSpOrderEpt; ++r) {
    lk[t * kSpOrderEpt + r] = key[r];
    lv[t * kSpOrderEpt + r] = val[r];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kSpOrderEpt; ++r) {
    const int i = r * kSpOrderThreads + t;
    order[win0 + i] = lk[i] == 0xFFFFFFFFu ? -1 : win0 + (int)lv[i];
  }
}

// values [n, c] at coords (b,z,y,x) -> dense [B, C*D, H, W]  (to_dense + transpose + reshape of
// sparse_resnet.py:202-205 in one pass; the destination is zero-filled first)
__global__ __launch_bounds__(256) void sp_to_dense_kernel(const float* __restrict__ feats,
                                                          const int32_t* __restrict__ coords,
                                                          const int* __restrict__ n_dev, int n_cap,
                                                          int c, SpShape s, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
  if (t >= (int64_t)n * c) return;
  const int row = (int)(t / c), ch = (int)(t - (int64_t)row * c);
  const int b = coords[row * 4], z = coords[row * 4 + 1], y = coords[row * 4 + 2], x = coords[row * 4 + 3];
  if (b < 0 || b >= s.batch) return;
  out[((((int64_t)b * c + ch) * s.d + z) * s.h + y) * s.w + x] = feats[t];
}

__global__ void sp_set_count_kernel(const int* __restrict__ total, int fixed, int cap,
                                    int32_t* __restrict__ n_out) {
  *n_out = total ? min(*total, cap) : min(fixed, cap);
}

static inline uint32_t table_size(int n) {
  uint32_t t = 1024;
  while (t < (uint32_t)n * 2u) t <<= 1;
  return t;
}

struct SpWorkspace {
  uint32_t *in_keys, *tkeys, *out_keys, *cand_a, *cand_b, *val_a, *val_b;
  int *tvals, *flags, *hist, *partial, *total;
  uint32_t tsize;
  size_t bytes;
};

static SpWorkspace sp_carve(void* base, int n_in, int K, bool subm, int out_cap) {
  Carver c(base);
  SpWorkspace w{};
  w.tsize = table_size(n_in);
  w.in_keys = c.take<uint32_t>((size_t)n_in);
  w.tkeys = c.take<uint32_t>(w.tsize);
  w.tvals = c.take<int>(w.tsize);
  w.total = c.take<int>(1);
  if (!subm) {
    const size_t nc = (size_t)n_in * K;
    const RadixPlan plan = radix_plan(0xFFFFFFFFu, (int64_t)nc);
    w.out_keys = c.take<uint32_t>((size_t)out_cap);
    w.cand_a = c.take<uint32_t>(nc);
    w.cand_b = c.take<uint32_t>(nc);
    w.val_a = c.take<uint32_t>(nc);
    w.val_b = c.take<uint32_t>(nc);
    w.flags = c.take<int>(nc);
    w.hist = c.take<int>(radix_hist_ints(plan));
    w.partial = c.take<int>((size_t)std::max(scan_num_tiles((int64_t)radix_hist_ints(plan)),
                                             scan_num_tiles((int64_t)nc)));
  }
  w.bytes = c.off;
  return w;
}

// ---------------------------------------------------------------------------------------------------
// Plan path (pd3_sparse_sort_coords / pd3_sparse_conv_outputs / pd3_sparse_rulebook): every index set of an
// encoder is a SORTED key array (raster order of (b, z, y, x)) with its length in device memory, so a chain of
// convolutions is enqueued without a host round trip.  Sortedness replaces the hash table: a (b, z, y) line is a
// contiguous piece of the array, the output set of a regular convolution is the compaction of a byte map of
// marked cells (sorted by construction), and a neighbour is found by a binary search inside one line.  Nothing
// on this path is atomic (global atomics run at a few G/s on this machine).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sp_plan_keys_kernel(const int32_t* __restrict__ coords, int n, SpShape s,
                                                           uint32_t cells, uint32_t* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = coords[i * 4], z = coords[i * 4 + 1], y = coords[i * 4 + 2], x = coords[i * 4 + 3];
  const bool ok = b >= 0 && b < s.batch && z >= 0 && z < s.d && y >= 0 && y < s.h && x >= 0 && x < s.w;
  keys[i] = ok ? sp_key(b, z, y, x, s) : cells;  // padding sorts last and needs no extra key bits
}

// sorted (keys [, order]) -> the caller's arrays, padding marked kSpEmpty; count = number of real keys
__global__ __launch_bounds__(256) void sp_plan_finish_kernel(const uint32_t* __restrict__ sorted,
                                                             const uint32_t* __restrict__ vals, int n,
                                                             uint32_t cells, const int* __restrict__ limit,
                                                             uint32_t* __restrict__ keys_out,
                                                             int32_t* __restrict__ order_out,
                                                             int* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = sorted[i];
  const bool real = k < cells;
  keys_out[i] = real ? k : kSpEmpty;
  if (order_out) order_out[i] = (int32_t)vals[i];
  if (real && (i == n - 1 || sorted[i + 1] >= cells)) *count = limit ? min(i + 1, *limit) : i + 1;
  if (i == 0 && !real) *count = 0;
}

// regular convolution: the set of outputs the inputs reach.  Every (input row, offset) marks its output cell in a
// byte map of the output grid (plain stores of the same value: no atomics); compacting the map in cell order
// yields the sorted key array directly -- no hash set, no sort.
__global__ __launch_bounds__(256) void sp_plan_mark_kernel(
    const uint32_t* __restrict__ in_keys, const int* __restrict__ n_in_dev, int n_in_cap, SpShape in_s,
    SpShape out_s, SpConv c, unsigned char* __restrict__ map) {
  const int K = c.kd * c.kh * c.kw;
  const int64_t total = (int64_t)min(*n_in_dev, n_in_cap) * K;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(t / K), k = (int)(t - (int64_t)row * K);
    const uint32_t key = in_keys[row];
    if (key == kSpEmpty) continue;
    uint32_t r = key;
    const int x = (int)(r % (uint32_t)in_s.w);
    r /= (uint32_t)in_s.w;
    const int y = (int)(r % (uint32_t)in_s.h);
    r /= (uint32_t)in_s.h;
    const int z = (int)(r % (uint32_t)in_s.d);
    const int b = (int)(r / (uint32_t)in_s.d);
    const int kz = k / (c.kh * c.kw), ky = (k / c.kw) % c.kh, kx = k % c.kw;
    const int nz = z + c.pd - kz, ny = y + c.ph - ky, nx = x + c.pw - kx;  // = q * stride
    if (nz < 0 || ny < 0 || nx < 0 || nz % c.sd != 0 || ny % c.sh != 0 || nx % c.sw != 0) continue;
    const int qz = nz / c.sd, qy = ny / c.sh, qx = nx / c.sw;
    if (qz >= out_s.d || qy >= out_s.h || qx >= out_s.w) continue;
    map[sp_key(b, qz, qy, qx, out_s)] = 1;
  }
}

constexpr int kMapTile = 16384;  // cells per workgroup: 256 threads x 4 x 16 bytes

__device__ __forceinline__ int sp_nonzero_bytes(uint4 v) {  // the map holds 0 / 1 only
  return __popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) +
         __popc(v.w & 0x01010101u);
}

// pass 1: marked cells per tile;  pass 2 (after the scan of the tile counts): keys = indices of the marked cells
template <bool EMIT>
__global__ __launch_bounds__(256) void sp_plan_compact_kernel(const unsigned char* __restrict__ map, int64_t cells,
                                                              int* __restrict__ partial,
                                                              uint32_t* __restrict__ out_keys, int out_cap) {
  __shared__ int smem[256 / kWave + 1];
  const int64_t base = (int64_t)blockIdx.x * kMapTile + (int64_t)threadIdx.x * 64;
  uint4 v[4];
  int mine = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = make_uint4(0u, 0u, 0u, 0u);
    if (base + i * 16 + 16 <= cells) {
      v[i] = *reinterpret_cast<const uint4*>(map + base + i * 16);
    } else {
      unsigned char tmp[16];
      for (int j = 0; j < 16; ++j) tmp[j] = base + i * 16 + j < cells ? map[base + i * 16 + j] : 0;
      __builtin_memcpy(&v[i], tmp, 16);
    }
    mine += sp_nonzero_bytes(v[i]);
  }
  int total;
  int at = block_exclusive_scan<256>(mine, smem, total);
  if (!EMIT) {
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
    return;
  }
  if (mine == 0) return;
  at += partial[blockIdx.x];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if ((w[j >> 2] >> (8 * (j & 3))) & 0xFFu) {
        if (at < out_cap) out_keys[at] = (uint32_t)(base + i * 16 + j);
        ++at;
      }
    }
  }
}

// tail of the key array = padding; n_out = min(total, cap)
__global__ __launch_bounds__(256) void sp_plan_pad_kernel(const int* __restrict__ total, int cap,
                                                          uint32_t* __restrict__ keys, int* __restrict__ n_out) {
  const int n = min(*total, cap);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *n_out = n;
  if (i >= n && i < cap) keys[i] = kSpEmpty;
}

// first row of every (b, z, y) line of a sorted key array (entry `lines` = row count): thread per line, a
// lower_bound over the array (the top levels of the search stay in cache)
__global__ __launch_bounds__(256) void sp_line_start_kernel(const uint32_t* __restrict__ keys,
                                                            const int* __restrict__ n_dev, int n_cap, int w,
                                                            int lines, int* __restrict__ line_start) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l > lines) return;
  const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
  const int64_t first = (int64_t)l * w;  // smallest key of the line
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)keys[mid] < first) lo = mid + 1;
    else hi = mid;
  }
  line_start[l] = lo;
}

// Rulebook by line search: thread per (output row, kz, ky).  The input line the row looks at is
// [line_start[l], line_start[l + 1]) of the sorted input array -- a few keys to a few hundred, ascending in x --
// so the first of the kw wanted keys is found by a binary search inside the line and the others are its
// successors.  Neighbouring threads search the same or neighbouring lines (rows are in raster order): the loads
// hit L1 / L2.  No table is built, nothing is atomic.
__global__ __launch_bounds__(256) void sp_rulebook_lines_kernel(
    const uint32_t* __restrict__ in_keys, const int* __restrict__ line_start, const uint32_t* __restrict__ out_keys,
    const int* __restrict__ n_out_dev, int n_out_cap, SpShape in_s, SpShape out_s, SpConv c,
    int32_t* __restrict__ nbr, int32_t* __restrict__ out_coords) {
  const int G = c.kd * c.kh, K = G * c.kw;
  const int n_out = n_out_dev ? min(*n_out_dev, n_out_cap) : n_out_cap;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n_out * G) return;
  const int row = (int)(t / G), g = (int)(t - (int64_t)row * G);
  uint32_t r = out_keys[row];
  const int x = (int)(r % (uint32_t)out_s.w);
  r /= (uint32_t)out_s.w;
  const int y = (int)(r % (uint32_t)out_s.h);
  r /= (uint32_t)out_s.h;
  const int z = (int)(r % (uint32_t)out_s.d);
  const int b = (int)(r / (uint32_t)out_s.d);
  if (out_coords && g == 0) *reinterpret_cast<int4*>(out_coords + (int64_t)row * 4) = make_int4(b, z, y, x);
  int32_t* dst = nbr + (int64_t)row * K + g * c.kw;
  const int iz = z * c.sd - c.pd + g / c.kh, iy = y * c.sh - c.ph + g % c.kh;
  int lo = 0, hi = 0;
  uint32_t line_key = 0;
  if (iz >= 0 && iz < in_s.d && iy >= 0 && iy < in_s.h) {
    const int l = (b * in_s.d + iz) * in_s.h + iy;
    lo = line_start[l];
    hi = line_start[l + 1];
    line_key = sp_key(b, iz, iy, 0, in_s);
  }
  const int ix0 = x * c.sw - c.pw;
  // first input row of the line with x >= ix0
  const int64_t want0 = (int64_t)line_key + ix0;
  int a = lo, e = hi;
  while (a < e) {
    const int mid = (a + e) >> 1;
    if ((int64_t)in_keys[mid] < want0) a = mid + 1;
    else e = mid;
  }
  for (int kx = 0; kx < c.kw; ++kx) {
    const int ix = ix0 + kx;
    int found = -1;
    if (ix >= 0 && ix < in_s.w) {
      const uint32_t want = line_key + (uint32_t)ix;
      while (a < hi && in_keys[a] < want) ++a;      // (at most kw - 1 steps in total)
      while (a < hi && in_keys[a] == want) found = a++;  // duplicate input coordinates: the highest row wins
    }
    dst[kx] = found;
  }
}

static bool out_shape(const SpShape& in, const SpConv& c, SpShape& out) {
  out.batch = in.batch;
  out.d = (in.d + 2 * c.pd - c.kd) / c.sd + 1;
  out.h = (in.h + 2 * c.ph - c.kh) / c.sh + 1;
  out.w = (in.w + 2 * c.pw - c.kw) / c.sw + 1;
  return out.d > 0 && out.h > 0 && out.w > 0 &&
         (int64_t)out.batch * out.d * out.h * out.w < (int64_t)0xFFFFFFFF;
}

}  // namespace pd3

using namespace pd3;

extern "C" size_t pd3_sparse_conv3d_workspace(int n_in, const int* kernel_size, int subm,
                                              int out_cap) {
  if (n_in <= 0 || !kernel_size || out_cap <= 0) return 0;
  const int K = kernel_size[0] * kernel_size[1] * kernel_size[2];
  if (K <= 0 || (int64_t)n_in * K >= ((int64_t)1 << 31)) return 0;
  return sp_carve(nullptr, n_in, K, subm != 0, out_cap).bytes;
}

extern "C" int pd3_sparse_conv3d_indices(const int32_t* in_coords, int n_in, int batch,
                                         const int* spatial_shape, const int* kernel_size,
                                         const int* stride, const int* padding, int subm,
                                         int32_t* out_coords, int32_t* nbr, int32_t* n_out,
                                         int out_cap, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  if (!in_coords || !spatial_shape || !kernel_size || !stride || !padding || !out_coords || !nbr ||
      !n_out || !workspace || n_in <= 0 || batch <= 0 || out_cap <= 0)
    return PD3_EINVAL;
  SpShape in_s{batch, spatial_shape[0], spatial_shape[1], spatial_shape[2]};
  SpConv c{kernel_size[0], kernel_size[1], kernel_size[2], stride[0], stride[1], stride[2],
           padding[0], padding[1], padding[2]};
  const int K = c.kd * c.kh * c.kw;
  if (K <= 0 || c.sd <= 0 || c.sh <= 0 || c.sw <= 0 || in_s.d <= 0 || in_s.h <= 0 || in_s.w <= 0)
    return PD3_EINVAL;
  if ((int64_t)in_s.batch * in_s.d * in_s.h * in_s.w >= (int64_t)0xFFFFFFFF) return PD3_EUNSUPPORTED;
  if ((int64_t)n_in * K >= ((int64_t)1 << 31)) return PD3_EUNSUPPORTED;
  SpShape out_s = in_s;
  if (subm) {
    // submanifold: stride 1 and "same" padding, so that the centre tap sits on the output coordinate
    if (c.sd != 1 || c.sh != 1 || c.sw != 1 || c.pd * 2 + 1 != c.kd || c.ph * 2 + 1 != c.kh ||
        c.pw * 2 + 1 != c.kw)
      return PD3_EUNSUPPORTED;
    if (out_cap < n_in) return PD3_EINVAL;
  } else if (!out_shape(in_s, c, out_s)) {
    return PD3_EUNSUPPORTED;
  }
  SpWorkspace w = sp_carve(workspace, n_in, K, subm != 0, out_cap);
  if (workspace_bytes < w.bytes) return PD3_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(w.tkeys, 0xFF, sizeof(uint32_t) * w.tsize, s);  // kSpEmpty
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(w.tvals, 0xFF, sizeof(int) * w.tsize, s);  // -1
  if (e != hipSuccess) return (int)e;
  const unsigned gb = (unsigned)ceil_div(n_in, 256);
  sp_keys_kernel<<<gb, 256, 0, s>>>(in_coords, n_in, in_s, w.in_keys);
  sp_hash_insert_kernel<<<gb, 256, 0, s>>>(w.in_keys, n_in, w.tkeys, w.tvals, w.tsize - 1);
  if (subm) {
    if (out_coords != in_coords) {
      e = hipMemcpyAsync(out_coords, in_coords, sizeof(int32_t) * 4 * (size_t)n_in,
                         hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) return (int)e;
    }
    sp_set_count_kernel<<<1, 1, 0, s>>>(nullptr, n_in, out_cap, n_out);
    sp_rulebook_kernel<<<(unsigned)ceil_div((int64_t)n_in * K, 256), 256, 0, s>>>(
        w.in_keys, nullptr, n_in, in_s, out_s, c, w.tkeys, w.tvals, w.tsize - 1, nbr);
    return launch_status();
  }
  const int64_t nc = (int64_t)n_in * K;
  sp_candidates_kernel<<<(unsigned)ceil_div(nc, 256), 256, 0, s>>>(w.in_keys, n_in, in_s, out_s, c,
                                                                   w.cand_a);
  const uint32_t max_key = (uint32_t)((int64_t)out_s.batch * out_s.d * out_s.h * out_s.w);  // < kSpEmpty
  (void)max_key;
  const RadixPlan plan = radix_plan(0xFFFFFFFFu, nc);
  const int where = enqueue_radix_sort(w.cand_a, w.val_a, w.cand_b, w.val_b, nc, nc, 1, plan,
                                       /*identity_vals=*/true, w.hist, w.partial, s);
  const uint32_t* sorted = where ? w.cand_b : w.cand_a;
  sp_heads_kernel<<<(unsigned)ceil_div(nc, 256), 256, 0, s>>>(sorted, nc, w.flags);
  EpiUniqueKeys epi{sorted, w.out_keys, out_coords, out_s, out_cap};
  enqueue_exclusive_scan(w.flags, nc, nc, 1, w.partial, w.total, (int*)nullptr, LoadIdentity{}, epi, s);
  sp_set_count_kernel<<<1, 1, 0, s>>>(w.total, 0, out_cap, n_out);
  sp_rulebook_kernel<<<(unsigned)ceil_div((int64_t)out_cap * K, 256), 256, 0, s>>>(
      w.out_keys, n_out, out_cap, in_s, out_s, c, w.tkeys, w.tvals, w.tsize - 1, nbr);
  return launch_status();
}


// ---- plan path entries ----------------------------------------------------------------------------------
namespace pd3 {
struct SpPlanWs {
  uint32_t *ka, *kb, *va, *vb, *table;
  int *hist, *partial, *counter;
  uint32_t tsize;
  size_t bytes;
};
static SpPlanWs sp_plan_carve(void* base, int64_t n, uint32_t max_key, bool with_table) {
  Carver c(base);
  SpPlanWs w{};
  (void)max_key;
  const size_t hist_ints = (size_t)kRsMaxBins * (size_t)ceil_div(n, kRsTile);  // whatever digit width the keys get
  w.ka = c.take<uint32_t>((size_t)n);
  w.kb = c.take<uint32_t>((size_t)n);
  w.va = c.take<uint32_t>((size_t)n);
  w.vb = c.take<uint32_t>((size_t)n);
  w.hist = c.take<int>(hist_ints);
  w.partial = c.take<int>((size_t)scan_num_tiles((int64_t)hist_ints));
  w.counter = c.take<int>(1);
  if (with_table) {
    w.tsize = table_size((int)std::min<int64_t>(n, (int64_t)1 << 30));
    w.table = c.take<uint32_t>(w.tsize);
  }
  w.bytes = c.off;
  return w;
}
static bool sp_shape_ok(const SpShape& s) {
  return s.batch > 0 && s.d > 0 && s.h > 0 && s.w > 0 &&
         (int64_t)s.batch * s.d * s.h * s.w < (int64_t)0xFFFFFFFE;
}
}  // namespace pd3

extern "C" size_t pd3_sparse_plan_workspace(int n_cap) {
  if (n_cap <= 0) return 0;
  return sp_plan_carve(nullptr, n_cap, 0xFFFFFFFEu, true).bytes;
}

extern "C" int pd3_sparse_sort_coords(const int32_t* coords, int n, int batch, const int* spatial_shape,
                                      uint32_t* keys_sorted, int32_t* order, int32_t* n_valid,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!coords || !spatial_shape || !keys_sorted || !order || !n_valid || !workspace || n <= 0) return PD3_EINVAL;
  SpShape sh{batch, spatial_shape[0], spatial_shape[1], spatial_shape[2]};
  if (!sp_shape_ok(sh)) return PD3_EUNSUPPORTED;
  const uint32_t cells = (uint32_t)((int64_t)sh.batch * sh.d * sh.h * sh.w);
  SpPlanWs w = sp_plan_carve(workspace, n, 0xFFFFFFFEu, true);
  if (workspace_bytes < w.bytes) return PD3_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned gb = (unsigned)ceil_div(n, 256);
  sp_plan_keys_kernel<<<gb, 256, 0, s>>>(coords, n, sh, cells, w.ka);
  const RadixPlan plan = radix_plan(cells, n);
  const int where = enqueue_radix_sort(w.ka, w.va, w.kb, w.vb, n, n, 1, plan, /*identity_vals=*/true, w.hist,
                                       w.partial, s);
  sp_plan_finish_kernel<<<gb, 256, 0, s>>>(where ? w.kb : w.ka, where ? w.vb : w.va, n, cells, nullptr,
                                           keys_sorted, order, n_valid);
  return launch_status();
}

extern "C" size_t pd3_sparse_conv_outputs_workspace(int batch, const int* spatial_shape, const int* kernel_size,
                                                    const int* stride, const int* padding) {
  if (!spatial_shape || !kernel_size || !stride || !padding) return 0;
  SpShape in_s{batch, spatial_shape[0], spatial_shape[1], spatial_shape[2]}, out_s{};
  SpConv c{kernel_size[0], kernel_size[1], kernel_size[2], stride[0], stride[1], stride[2],
           padding[0], padding[1], padding[2]};
  if (c.kd <= 0 || c.kh <= 0 || c.kw <= 0 || c.sd <= 0 || c.sh <= 0 || c.sw <= 0) return 0;
  if (!sp_shape_ok(in_s) || !out_shape(in_s, c, out_s) || !sp_shape_ok(out_s)) return 0;
  const int64_t cells = (int64_t)out_s.batch * out_s.d * out_s.h * out_s.w;
  Carver cv(nullptr);
  cv.take<unsigned char>((size_t)cells + 16);
  cv.take<int>((size_t)ceil_div(cells, kMapTile) + 1);
  cv.take<int>(1);
  return cv.off;
}

extern "C" int pd3_sparse_conv_outputs(const uint32_t* in_keys, const int32_t* n_in, int n_in_cap, int batch,
                                       const int* spatial_shape, const int* kernel_size, const int* stride,
                                       const int* padding, uint32_t* out_keys, int32_t* n_out, int out_cap,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  if (!in_keys || !n_in || !spatial_shape || !kernel_size || !stride || !padding || !out_keys || !n_out ||
      !workspace || n_in_cap <= 0 || out_cap <= 0)
    return PD3_EINVAL;
  SpShape in_s{batch, spatial_shape[0], spatial_shape[1], spatial_shape[2]}, out_s{};
  SpConv c{kernel_size[0], kernel_size[1], kernel_size[2], stride[0], stride[1], stride[2],
           padding[0], padding[1], padding[2]};
  if (c.kd <= 0 || c.kh <= 0 || c.kw <= 0 || c.sd <= 0 || c.sh <= 0 || c.sw <= 0) return PD3_EINVAL;
  if (!sp_shape_ok(in_s) || !out_shape(in_s, c, out_s) || !sp_shape_ok(out_s)) return PD3_EUNSUPPORTED;
  const int64_t cells = (int64_t)out_s.batch * out_s.d * out_s.h * out_s.w;
  const int tiles = (int)ceil_div(cells, kMapTile);
  Carver cv(workspace);
  unsigned char* map = cv.take<unsigned char>((size_t)cells + 16);
  int* partial = cv.take<int>((size_t)tiles + 1);
  int* total = cv.take<int>(1);
  if (workspace_bytes < cv.off) return PD3_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(map, 0, (size_t)cells, s);
  if (e != hipSuccess) return (int)e;
  const int K = c.kd * c.kh * c.kw;
  const int64_t work = (int64_t)n_in_cap * K;
  const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(work, 256), 256 * 64);
  sp_plan_mark_kernel<<<grid, 256, 0, s>>>(in_keys, n_in, n_in_cap, in_s, out_s, c, map);
  sp_plan_compact_kernel<false><<<(unsigned)tiles, 256, 0, s>>>(map, cells, partial, nullptr, 0);
  scan_partials_kernel<<<1, 1024, 0, s>>>(partial, tiles, total);
  sp_plan_compact_kernel<true><<<(unsigned)tiles, 256, 0, s>>>(map, cells, partial, out_keys, out_cap);
  sp_plan_pad_kernel<<<(unsigned)ceil_div(out_cap, 256), 256, 0, s>>>(total, out_cap, out_keys, n_out);
  return launch_status();
}

extern "C" size_t pd3_sparse_rulebook_workspace(int batch, const int* spatial_shape) {
  if (batch <= 0 || !spatial_shape || spatial_shape[0] <= 0 || spatial_shape[1] <= 0) return 0;
  const int64_t lines = (int64_t)batch * spatial_shape[0] * spatial_shape[1] + 1;
  Carver c(nullptr);
  c.take<int>((size_t)lines);
  return c.off;
}

extern "C" int pd3_sparse_rulebook(const uint32_t* in_keys, const int32_t* n_in, int n_in_cap,
                                   const uint32_t* out_keys, const int32_t* n_out, int n_out_cap, int batch,
                                   const int* spatial_shape, const int* kernel_size, const int* stride,
                                   const int* padding, int subm, int32_t* nbr, int32_t* out_coords,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!in_keys || !out_keys || !spatial_shape || !kernel_size || !stride || !padding || !nbr || !workspace ||
      n_in_cap <= 0 || n_out_cap <= 0)
    return PD3_EINVAL;
  SpShape in_s{batch, spatial_shape[0], spatial_shape[1], spatial_shape[2]}, out_s{};
  SpConv c{kernel_size[0], kernel_size[1], kernel_size[2], stride[0], stride[1], stride[2],
           padding[0], padding[1], padding[2]};
  if (c.kd <= 0 || c.kh <= 0 || c.kw <= 0 || c.sd <= 0 || c.sh <= 0 || c.sw <= 0) return PD3_EINVAL;
  if ((int64_t)n_out_cap * c.kd * c.kh >= ((int64_t)1 << 31)) return PD3_EUNSUPPORTED;
  if (!sp_shape_ok(in_s)) return PD3_EUNSUPPORTED;
  if (subm) {
    if (c.sd != 1 || c.sh != 1 || c.sw != 1 || c.pd * 2 + 1 != c.kd || c.ph * 2 + 1 != c.kh ||
        c.pw * 2 + 1 != c.kw)
      return PD3_EUNSUPPORTED;
    out_s = in_s;
  } else if (!out_shape(in_s, c, out_s)) {
    return PD3_EUNSUPPORTED;
  }
  const int64_t lines = (int64_t)in_s.batch * in_s.d * in_s.h + 1;
  if (lines >= ((int64_t)1 << 31)) return PD3_EUNSUPPORTED;
  Carver cv(workspace);
  int* line_start = cv.take<int>((size_t)lines);
  if (workspace_bytes < cv.off) return PD3_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  sp_line_start_kernel<<<(unsigned)ceil_div(lines, 256), 256, 0, s>>>(in_keys, n_in, n_in_cap, in_s.w,
                                                                      (int)lines - 1, line_start);
  sp_rulebook_lines_kernel<<<(unsigned)ceil_div((int64_t)n_out_cap * (c.kd * c.kh), 256), 256, 0, s>>>(
      in_keys, line_start, out_keys, n_out, n_out_cap, in_s, out_s, c, nbr, out_coords);
  return launch_status();
}

extern "C" int pd3_sparse_conv3d_features_ordered(const float* in_feats, const int32_t* nbr,
                                                  const int32_t* n_out, int n_out_cap, int kernel_volume,
                                                  int cin, int cout, const float* weight, const float* bias,
                                                  const float* scale, const float* shift,
                                                  const float* residual, int relu, const int32_t* order,
                                                  float* out, void* stream);

extern "C" int pd3_sparse_conv3d_features(const float* in_feats, const int32_t* nbr,
                                          const int32_t* n_out, int n_out_cap, int kernel_volume,
                                          int cin, int cout, const float* weight, const float* bias,
                                          const float* scale, const float* shift,
                                          const float* residual, int relu, float* out, void* stream) {
  return pd3_sparse_conv3d_features_ordered(in_feats, nbr, n_out, n_out_cap, kernel_volume, cin, cout, weight, bias,
                                            scale, shift, residual, relu, nullptr, out, stream);
}

extern "C" int pd3_sparse_conv3d_features_ordered(const float* in_feats, const int32_t* nbr,
                                                  const int32_t* n_out, int n_out_cap, int kernel_volume,
                                                  int cin, int cout, const float* weight, const float* bias,
                                                  const float* scale, const float* shift,
                                                  const float* residual, int relu, const int32_t* order,
                                                  float* out, void* stream) {
  if (!in_feats || !nbr || !weight || !out || n_out_cap <= 0 || kernel_volume <= 0 || cin <= 0 ||
      cout <= 0)
    return PD3_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return PD3_EINVAL;
  if (cout > 128) return PD3_EUNSUPPORTED;
  // `order` (pd3_sparse_tile_order) is a scheduling hint for the 128-row kernel; the other kernels ignore it
  SpGemmArgs a{in_feats, nbr, weight, bias, scale, shift, residual, out, n_out, n_out_cap,
               kernel_volume, cin, cout, relu ? 1 : 0, order};
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e;
  if (cin % 16 == 0 && kernel_volume <= kSg2MaxK && (cout == 16 || cout == 32 || cout == 64 || cout == 128) &&
      reinterpret_cast<uintptr_t>(weight) % 16 == 0 && reinterpret_cast<uintptr_t>(in_feats) % 16 == 0) {
    // the encoder's shapes: 128-row tiles, A operand straight from global memory
    const int t = cin % 32 == 0 ? 8 : 4, nb = cout / 16;
    const size_t lds = (size_t)2 * 64 * (nb * t + 4) * sizeof(float) +
                       ((size_t)kSg2Rows * kernel_volume + 16 + kSg2Rows) * sizeof(int);
    const unsigned grid = sp_window_grid(ceil_div(n_out_cap, kSg2Rows), 8192 / kSg2Rows);
#define PD3_SP_ROWS(NBV, TV)                                                                      \
  do {                                                                                            \
    if (lds > 48 * 1024) {                                                                        \
      e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(sp_gemm_rows_kernel<NBV, TV>), (int)lds);              \
      if (e != hipSuccess) return (int)e;                                                         \
    }                                                                                             \
    sp_gemm_rows_kernel<NBV, TV><<<grid, 256, lds, s>>>(a);                                       \
  } while (0)
    switch (nb * 16 + t) {
      case 1 * 16 + 4: PD3_SP_ROWS(1, 4); break;
      case 1 * 16 + 8: PD3_SP_ROWS(1, 8); break;
      case 2 * 16 + 4: PD3_SP_ROWS(2, 4); break;
      case 2 * 16 + 8: PD3_SP_ROWS(2, 8); break;
      case 4 * 16 + 4: PD3_SP_ROWS(4, 4); break;
      case 4 * 16 + 8: PD3_SP_ROWS(4, 8); break;
      case 8 * 16 + 4: PD3_SP_ROWS(8, 4); break;
      default: PD3_SP_ROWS(8, 8); break;
    }
#undef PD3_SP_ROWS
    return launch_status();
  }
  if (cout % 16 == 0 && (reinterpret_cast<uintptr_t>(weight) % 16 == 0) &&
      (cin % 4 != 0 || reinterpret_cast<uintptr_t>(in_feats) % 16 == 0)) {
    // matrix-core path, any cin (the encoder's input layer: cin = 5)
    const int cin_pad = (cin + 3) / 4 * 4;
    const int astride = ((cin_pad - 2 + 15) / 16) * 16 + 2;
    const int wstride = cout == 16 ? 16 : cout + 16;
    const size_t lds = ((size_t)4 * 16 * astride + (size_t)kSpChunk * wstride) * sizeof(float) +
                       kSpTile * sizeof(int);
    const unsigned grid = (unsigned)ceil_div(n_out_cap, kSpTile);
#define PD3_SP_MFMA(NBV)                                                                          \
  do {                                                                                            \
    if (lds > 48 * 1024) {                                                                        \
      e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(sp_gemm_mfma_kernel<NBV>), (int)lds);              \
      if (e != hipSuccess) return (int)e;                                                         \
    }                                                                                             \
    sp_gemm_mfma_kernel<NBV><<<grid, 256, lds, s>>>(a, cin_pad, astride, wstride);               \
  } while (0)
    switch (cout / 16) {
      case 1: PD3_SP_MFMA(1); break;
      case 2: PD3_SP_MFMA(2); break;
      case 4: PD3_SP_MFMA(4); break;
      case 8: PD3_SP_MFMA(8); break;
      default: goto valu_path;
    }
#undef PD3_SP_MFMA
    return launch_status();
  }
valu_path:
  {
    const size_t lds = ((size_t)cin * cout + (size_t)kSpRows * cin) * sizeof(float) + kSpRows * sizeof(int);
    if (lds > 160 * 1024) return PD3_EUNSUPPORTED;
    const unsigned grid = (unsigned)ceil_div(n_out_cap, kSpRows);
    if (cout <= 64) {
      if (lds > 48 * 1024) {
        e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(sp_gather_gemm_kernel<1>), (int)lds);
        if (e != hipSuccess) return (int)e;
      }
      sp_gather_gemm_kernel<1><<<grid, 256, lds, s>>>(a);
    } else {
      if (lds > 48 * 1024) {
        e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(sp_gather_gemm_kernel<2>), (int)lds);
        if (e != hipSuccess) return (int)e;
      }
      sp_gather_gemm_kernel<2><<<grid, 256, lds, s>>>(a);
    }
  }
  return launch_status();
}

extern "C" int64_t pd3_sparse_tile_order_entries(int n_out_cap) {
  if (n_out_cap <= 0) return 0;
  return ceil_div(n_out_cap, kSpWindow) * (int64_t)kSpWindow;
}

extern "C" int pd3_sparse_tile_order(const int32_t* nbr, const int32_t* n_out, int n_out_cap, int kernel_volume,
                                     int32_t* order, void* stream) {
  if (!nbr || !order || n_out_cap <= 0 || kernel_volume <= 0 || kernel_volume > 31) return PD3_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t lds = (size_t)2 * kSpWindow * sizeof(uint32_t);
  // raised on every launch like every other large-LDS kernel of the library (a host-side table write): a process-wide
  // flag would be unsynchronised and, if the attribute is kept per device, wrong for a second GPU in the process
  const hipError_t e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(sp_tile_order_kernel), (int)lds);
  if (e != hipSuccess) return (int)e;
  sp_tile_order_kernel<<<(unsigned)ceil_div(n_out_cap, kSpWindow), kSpOrderThreads, lds, s>>>(
      nbr, n_out, n_out_cap, kernel_volume, order);
  return launch_status();
}

extern "C" int pd3_sparse_to_dense(const float* feats, const int32_t* coords, const int32_t* n,
                                   int n_cap, int channels, int batch, const int* spatial_shape,
                                   float* dense, void* stream) {
  if (!feats || !coords || !spatial_shape || !dense || n_cap <= 0 || channels <= 0 || batch <= 0)
    return PD3_EINVAL;
  SpShape sh{batch, spatial_shape[0], spatial_shape[1], spatial_shape[2]};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t elems = (size_t)batch * channels * sh.d * sh.h * sh.w;
  hipError_t e = hipMemsetAsync(dense, 0, elems * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  sp_to_dense_kernel<<<(unsigned)ceil_div((int64_t)n_cap * channels, 256), 256, 0, s>>>(
      feats, coords, n, n_cap, channels, sh, dense);
  return launch_status();
}
