// PointPillarsScatter + the first (strided) convolution of SecondBackbone as a SPARSE convolution over the occupied
// pillars (round 6).  (reference: paddle3d/models/middle_encoders/pillar_scatter.py:57-93 followed by
// paddle3d/models/backbones/second_backbone.py:84-113 -- a dense cuDNN convolution over a canvas that is 11 % occupied.)
//
// A nuScenes canvas holds 30 k pillars on 262 k cells; a 3x3 / stride-2 output pixel sees 2.5 occupied cells of its nine
// on average and 60 % of the output pixels see none (their value is relu(bias), a constant per channel).  The dense
// implicit-GEMM kernel multiplies all nine taps of every pixel: 4.8 GFLOP per frame for 0.55 GFLOP of products that
// exist.  Here:
//   1. pc_flags_kernel + scan (EpiPillarRows) + pc_rulebook_order_kernel   output pixel -> row (raster order of the active pixels), its nine
//      pillar rows `nbr [row][9]` (read through the inverse map cell -> pillar of pd3_pointpillars_inverse_map) and the
//      inverse `cell_row [B, ho * wo]`; the row count stays on the device
//   2. the library's sparse gather-GEMM (pd3_sparse_tile_order + pd3_sparse_conv3d_features_bf16x3 / _f16) on that
//      rulebook: fp32 arithmetic on the bf16 matrix cores, bias + ReLU fused
//   3. rows_to_dense_fill_kernel   [rows, C] -> NCHW, a pixel without a row holds fill[c] = relu(bias[c])
// Steps 1 and 3 live here; the host glue is ops/conv.py:scatter_conv3x3_sparse.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"
#include "scan.hpp"

namespace pd3 {

struct PcGrid {
  int ny, nx, ho, wo, stride, pad;
};

__device__ __forceinline__ int pc_lookup(const int* __restrict__ inv, const PcGrid& g, int b, int oy, int ox, int k) {
  const int iy = oy * g.stride - g.pad + k / 3, ix = ox * g.stride - g.pad + k % 3;
  if (iy < 0 || iy >= g.ny || ix < 0 || ix >= g.nx) return -1;
  return inv[((int64_t)b * g.ny + iy) * g.nx + ix];
}

__global__ __launch_bounds__(256) void pc_flags_kernel(const int* __restrict__ inv, PcGrid g, int64_t cells,
                                                       int* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cells) return;
  const int plane = g.ho * g.wo;
  const int b = (int)(i / plane), r = (int)(i - (int64_t)b * plane);
  const int oy = r / g.wo, ox = r - oy * g.wo;
  int any = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) any |= pc_lookup(inv, g, b, oy, ox, k) >= 0 ? 1 : 0;
  flags[i] = any;
}

// the scan's epilogue only hands out the rows (a thread of the scan walks 16 consecutive pixels one after the other: the
// nine look-ups and ten scattered stores per active pixel made it 53 us); the rows' neighbours are written by the kernel
// that orders them, one thread per row
struct EpiPillarRows {
  int32_t *out_cell, *cell_row;
  int cap;
  __device__ __forceinline__ void operator()(int, int64_t i, int flag, int prefix, int) const {
    const bool on = flag && prefix < cap;
    cell_row[i] = on ? prefix : -1;
    if (on) out_cell[prefix] = (int32_t)i;
  }
};

// The tile order of that rulebook (the scheduling hint of pd3_sparse_tile_order: the rows of a window of 8192 grouped by
// their neighbour mask, so that a 32-row block of the gather-GEMM skips the taps none of its rows has).  Nine taps = a
// 9-bit mask: a counting sort over 512 bins in LDS instead of the general 31-bit bitonic network (76 us for this rulebook:
// 52 occupied windows = 52 workgroups of latency; this: one pass).  Within a bin the rows arrive in atomic order: which
// rows share a tile never changes a row's result.
// (kSpWindow = 8192 rows is the window of sp_window_tile in the gather-GEMMs.  Round 6 first grouped per 2048 rows with 256
// threads; grouping the whole window -- 1024 threads, still eight rows per thread -- makes a tile execute 4.9 taps instead
// of 6.8 and the feature kernel 13 % faster (190 -> 165 us per 16 frames) for 21 us more here (28 -> 49: 8192 LDS atomics
// on 512 bins per workgroup): tools/prof/prof_sparse_first_layer.py, profiles/r06_b16_launches.txt)
constexpr int kPcWindow = 8192, kPcThreads = 1024;
__global__ __launch_bounds__(kPcThreads) void pc_rulebook_order_kernel(const int* __restrict__ inv, PcGrid g,
                                                                 const int32_t* __restrict__ out_cell,
                                                                 const int* __restrict__ n_out_dev, int cap,
                                                                 int32_t* __restrict__ nbr, int32_t* __restrict__ order) {
  __shared__ int hist[512];
  __shared__ int scr[kPcThreads / kWave + 1];
  __shared__ uint16_t sorted[kPcWindow];
  const int n_out = min(*n_out_dev, cap);
  const int win0 = blockIdx.x * kPcWindow, t = threadIdx.x;
  const int nv = min(max(n_out - win0, 0), kPcWindow);  // rows of this window
  if (nv == 0) {
    if (order) {
#pragma unroll
      for (int r = 0; r < kPcWindow / kPcThreads; ++r) order[win0 + r * kPcThreads + t] = -1;
    }
    return;
  }
  for (int i = t; i < 512; i += kPcThreads) hist[i] = 0;
  __syncthreads();
  // A row is a chain of dependent reads (its cell -> nine pillar rows); one row after the other, eight chains per thread
  // were most of this kernel's time.  The cells of all eight rows are read first, then the look-ups four rows at a time
  // (36 reads in flight), the histogram last.
  constexpr int kRows = kPcWindow / kPcThreads;
  int m[kRows], cell[kRows];
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    const int i = r * kPcThreads + t;
    cell[r] = i < nv ? out_cell[win0 + i] : -1;
  }
  const int plane = g.ho * g.wo;
#pragma unroll
  for (int r0 = 0; r0 < kRows; r0 += 4) {
    int j[4][9];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = max(cell[r0 + q], 0);
      const int b = c / plane, rr = c - b * plane;
      const int oy = rr / g.wo, ox = rr - oy * g.wo;
#pragma unroll
      for (int k = 0; k < 9; ++k) j[q][k] = pc_lookup(inv, g, b, oy, ox, k);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + q, i = r * kPcThreads + t;
      m[r] = -1;
      if (cell[r] >= 0) {  // the row's nine pillar rows, written here, and their mask
        int mm = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          nbr[(int64_t)(win0 + i) * 9 + k] = j[q][k];
          mm |= j[q][k] >= 0 ? 1 << k : 0;
        }
        m[r] = mm;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kRows; ++r)
    if (m[r] >= 0) atomicAdd(&hist[m[r]], 1);
  if (!order) return;
  __syncthreads();
  int total;  // thread t < 512 owns bin t
  const int h0 = t < 512 ? hist[t] : 0;
  const int base = block_exclusive_scan<kPcThreads>(h0, scr, total);
  __syncthreads();
  if (t < 512) hist[t] = base;  // the bins' cursors
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kPcWindow / kPcThreads; ++r)
    if (m[r] >= 0) sorted[atomicAdd(&hist[m[r]], 1)] = (uint16_t)(r * kPcThreads + t);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kPcWindow / kPcThreads; ++r) {
    const int i = r * kPcThreads + t;
    order[win0 + i] = i < nv ? win0 + (int)sorted[i] : -1;
  }
}

// rows [n, C] -> out [B, C, plane] NCHW through cell_row, a cell without a row holds fill[c].  Workgroup = 256 consecutive
// cells x 32 channels through an LDS tile [channel][cell]: a lane fetches 16 bytes of its cell's row (the eight fetches of
// a cell's 128-byte half row come back to back), writes them along the cell axis (conflict-free) and the tile leaves as
// 1 KB channel segments.  Plain stores: the next layer reads the map at once, and what of it stays in the last-level
// cache is worth more than what streaming stores save here (128 x 64 tiles with streaming stores: 74 us, this: 63).
constexpr int kRdCells = 256, kRdCh = 32, kRdPitch = kRdCells + 4;
__global__ __launch_bounds__(256) void rows_to_dense_tile_kernel(const float* __restrict__ rows,
                                                                 const int* __restrict__ cell_row,
                                                                 const float* __restrict__ fill, int channels,
                                                                 int64_t plane, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float tile[kRdCh * kRdPitch];
  const int b = blockIdx.z, c0 = blockIdx.y * kRdCh;
  const int64_t cell0 = (int64_t)blockIdx.x * kRdCells;
  const int px = threadIdx.x & (kRdCells - 1), half = threadIdx.x / kRdCells;  // cell of the tile, 32-channel half
  const int64_t cell = cell0 + px;
  const int id = cell < plane ? cell_row[(int64_t)b * plane + cell] : -1;
  typedef float pc_f32x4 __attribute__((ext_vector_type(4)));
  pc_f32x4 v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int c = c0 + half * 32 + q * 4;
    v[q] = id >= 0 ? *reinterpret_cast<const pc_f32x4*>(rows + (int64_t)id * channels + c)
                   : (fill ? *reinterpret_cast<const pc_f32x4*>(fill + c) : pc_f32x4{0.f, 0.f, 0.f, 0.f});
  }
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[(half * 32 + q * 4 + e) * kRdPitch + px] = v[q][e];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kRdCh * kRdCells / 4 / 256; ++i) {
    const int e = i * 256 + threadIdx.x;
    const int c = e / (kRdCells / 4), p4 = e - c * (kRdCells / 4);
    if (cell0 + p4 * 4 < plane)
      *reinterpret_cast<pc_f32x4*>(out + ((int64_t)b * channels + c0 + c) * plane + cell0 + p4 * 4) =
          *reinterpret_cast<const pc_f32x4*>(tile + c * kRdPitch + p4 * 4);
  }
}

// (the general shape: 4 consecutive cells per lane x 4 channels per workgroup row, a 4 x 4 register transpose -- the form of
// canvas_write_vec4_kernel, scatter.hip)
__global__ __launch_bounds__(256) void rows_to_dense_fill_kernel(const float* __restrict__ rows,
                                                                 const int* __restrict__ cell_row,
                                                                 const float* __restrict__ fill, int channels,
                                                                 int64_t plane, float* __restrict__ out) {
  const int b = blockIdx.z;
  const int c = blockIdx.y * 4;
  const int64_t cell0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (cell0 >= plane) return;
  const int4 src = *reinterpret_cast<const int4*>(cell_row + (int64_t)b * plane + cell0);
  const int id[4] = {src.x, src.y, src.z, src.w};
  const float4 f = fill ? *reinterpret_cast<const float4*>(fill + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  float* o = out + ((int64_t)b * channels + c) * plane + cell0;
  float4 r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = id[k] >= 0 ? *reinterpret_cast<const float4*>(rows + (int64_t)id[k] * channels + c) : f;
  typedef float pc_f32x4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(pc_f32x4{r[0].x, r[1].x, r[2].x, r[3].x}, reinterpret_cast<pc_f32x4*>(o + 0 * plane));
  __builtin_nontemporal_store(pc_f32x4{r[0].y, r[1].y, r[2].y, r[3].y}, reinterpret_cast<pc_f32x4*>(o + 1 * plane));
  __builtin_nontemporal_store(pc_f32x4{r[0].z, r[1].z, r[2].z, r[3].z}, reinterpret_cast<pc_f32x4*>(o + 2 * plane));
  __builtin_nontemporal_store(pc_f32x4{r[0].w, r[1].w, r[2].w, r[3].w}, reinterpret_cast<pc_f32x4*>(o + 3 * plane));
}

}  // namespace pd3

using namespace pd3;

extern "C" size_t pd3_pillar_conv_rulebook_workspace(int batch, int ny, int nx, int stride) {
  if (batch <= 0 || ny <= 0 || nx <= 0 || stride <= 0) return 0;
  const int64_t cells = (int64_t)batch * ((ny + 2 - 3) / stride + 1) * ((nx + 2 - 3) / stride + 1);
  return align_up((size_t)cells * sizeof(int), 256) + align_up((size_t)scan_num_tiles(cells) * sizeof(int), 256);
}

extern "C" int pd3_pillar_conv_rulebook(const int32_t* inverse_map, int batch, int ny, int nx, int stride, int32_t* nbr,
                                        int32_t* out_cell, int32_t* cell_row, int32_t* n_out, int capacity,
                                        int32_t* order, void* workspace, size_t workspace_bytes, void* stream) {
  if (!inverse_map || !nbr || !out_cell || !cell_row || !n_out || !workspace || batch <= 0 || ny <= 0 || nx <= 0 ||
      capacity <= 0)
    return PD3_EINVAL;
  if (stride != 1 && stride != 2) return PD3_EUNSUPPORTED;
  PcGrid g{ny, nx, (ny + 2 - 3) / stride + 1, (nx + 2 - 3) / stride + 1, stride, 1};
  const int64_t cells = (int64_t)batch * g.ho * g.wo;
  if (cells >= ((int64_t)1 << 31)) return PD3_EUNSUPPORTED;
  if (workspace_bytes < pd3_pillar_conv_rulebook_workspace(batch, ny, nx, stride)) return PD3_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int* flags = static_cast<int*>(workspace);
  int* partial = reinterpret_cast<int*>(static_cast<char*>(workspace) + align_up((size_t)cells * sizeof(int), 256));
  pc_flags_kernel<<<(unsigned)ceil_div(cells, 256), 256, 0, s>>>(inverse_map, g, cells, flags);
  EpiPillarRows epi{out_cell, cell_row, capacity};
  enqueue_exclusive_scan(flags, cells, cells, 1, partial, n_out, (int*)nullptr, LoadIdentity{}, epi, s);
  // nbr, and (optional) order [ceil(capacity / 8192) * 8192] int32: slot -> row or -1, the `order` argument of the gather-GEMMs
  static_assert(kPcThreads >= 512 && 8192 % kPcWindow == 0, "a thread per bin; whole sub-windows per window");
  pc_rulebook_order_kernel<<<(unsigned)(ceil_div(capacity, 8192) * (8192 / kPcWindow)), kPcThreads, 0, s>>>(inverse_map, g, out_cell, n_out,
                                                                                     capacity, nbr, order);
  return launch_status();
}

extern "C" int pd3_rows_to_dense_fill(const float* rows, const int32_t* cell_row, const float* fill, int batch,
                                      int channels, int h, int w, float* out, void* stream) {
  if (!rows || !cell_row || !out || batch <= 0 || channels <= 0 || h <= 0 || w <= 0) return PD3_EINVAL;
  const int64_t plane = (int64_t)h * w;
  if (plane % 4 != 0 || channels % 4 != 0 || channels / 4 > 65535 || batch > 65535) return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(rows) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(cell_row) % 16 != 0 || (fill && reinterpret_cast<uintptr_t>(fill) % 16 != 0))
    return PD3_EINVAL;
  if (channels % kRdCh == 0) {
    dim3 grid((unsigned)ceil_div(plane, kRdCells), (unsigned)(channels / kRdCh), (unsigned)batch);
    rows_to_dense_tile_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(rows, cell_row, fill, channels, plane, out);
    return launch_status();
  }
  dim3 grid((unsigned)ceil_div(plane / 4, 256), (unsigned)(channels / 4), (unsigned)batch);
  rows_to_dense_fill_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(rows, cell_row, fill, channels, plane, out);
  return launch_status();
}
