// pointpillars_scatter for gfx950: fused zero-fill + NCHW scatter of pillar features.
// (reference: paddle3d/models/middle_encoders/pillar_scatter.py:57-93 -- zeros canvas,
//  paddle.scatter(overwrite=True) on [ny*nx, C], transpose to [C, ny*nx], concat over the batch.)
//
// The canvas (67 MB per nuScenes frame) dominates the traffic, so it is written exactly once, by a
// canvas-parallel kernel whose stores are full 16-byte lanes along the contiguous cell axis of each
// channel plane.  An inverse map cell -> pillar (int32 per cell, 1 MB per frame, L2 resident) is built
// first; the feature rows it points to are read as float4 and transposed in registers.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"

namespace pd3 {

__global__ __launch_bounds__(256) void fill_i32_kernel(int* __restrict__ p, int64_t n, int v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ __launch_bounds__(256) void inverse_map_kernel(const int32_t* __restrict__ coords,
                                                          int64_t m, int batch, int ny, int nx,
                                                          int* __restrict__ inv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int b = coords[i * 4 + 0], y = coords[i * 4 + 2], x = coords[i * 4 + 3];
  if (b < 0 || b >= batch || y < 0 || y >= ny || x < 0 || x >= nx) return;
  // overwrite semantics: the highest pillar index landing on a cell wins
  atomicMax(&inv[((int64_t)b * ny + y) * nx + x], (int)i);
}

// 4 consecutive cells per lane x 4 channels per workgroup row (4x4 register transpose).  grid.x runs along
// the cell axis, grid.y over channel groups: workgroups that are resident together write neighbouring 4 KB
// pieces of the same few channel planes (long sequential HBM streams) instead of 64 planes each; the inverse
// map (1 MB per frame) is re-read per channel group out of L2.
__global__ __launch_bounds__(256) void canvas_write_vec4_kernel(const float* __restrict__ feats,
                                                                const int* __restrict__ inv,
                                                                int channels, int64_t plane,
                                                                float* __restrict__ canvas) {
  const int b = blockIdx.z;
  const int c = blockIdx.y * 4;
  const int64_t cell0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (cell0 >= plane) return;
  const int4 src = *reinterpret_cast<const int4*>(inv + (int64_t)b * plane + cell0);
  const int id[4] = {src.x, src.y, src.z, src.w};
  float* out = canvas + ((int64_t)b * channels + c) * plane + cell0;
  float4 r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    r[k] = id[k] >= 0 ? *reinterpret_cast<const float4*>(feats + (int64_t)id[k] * channels + c)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
  // streaming stores: the 67 MB canvas of a frame is far beyond the caches, keeping it out of L2 leaves the
  // inverse map and the feature rows there
  typedef float sc_f32x4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(sc_f32x4{r[0].x, r[1].x, r[2].x, r[3].x}, reinterpret_cast<sc_f32x4*>(out + 0 * plane));
  __builtin_nontemporal_store(sc_f32x4{r[0].y, r[1].y, r[2].y, r[3].y}, reinterpret_cast<sc_f32x4*>(out + 1 * plane));
  __builtin_nontemporal_store(sc_f32x4{r[0].z, r[1].z, r[2].z, r[3].z}, reinterpret_cast<sc_f32x4*>(out + 2 * plane));
  __builtin_nontemporal_store(sc_f32x4{r[0].w, r[1].w, r[2].w, r[3].w}, reinterpret_cast<sc_f32x4*>(out + 3 * plane));
}

// Generic shape fallback: one thread per (channel, cell).
__global__ __launch_bounds__(256) void canvas_write_scalar_kernel(const float* __restrict__ feats,
                                                                  const int* __restrict__ inv,
                                                                  int channels, int64_t plane,
                                                                  float* __restrict__ canvas) {
  const int b = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)channels * plane) return;
  const int c = (int)(e / plane);
  const int64_t cell = e - (int64_t)c * plane;
  const int id = inv[(int64_t)b * plane + cell];
  canvas[(int64_t)b * channels * plane + e] = id >= 0 ? feats[(int64_t)id * channels + c] : 0.f;
}

}  // namespace pd3

using namespace pd3;

extern "C" size_t pd3_pointpillars_scatter_workspace(int batch, int ny, int nx) {
  if (batch <= 0 || ny <= 0 || nx <= 0) return 0;
  return align_up((size_t)batch * ny * nx * sizeof(int), 256);
}

// The inverse map alone (cell -> pillar row, -1 for an empty cell): what pd3_scatter_conv3x3_bias_relu reads instead of
// a materialised canvas.  inv [batch, ny * nx] int32.
extern "C" int pd3_pointpillars_inverse_map(const int32_t* coords, int64_t num_pillars, int batch, int ny, int nx,
                                            int32_t* inv, void* stream) {
  if (!inv || batch <= 0 || ny <= 0 || nx <= 0 || num_pillars < 0 || (num_pillars > 0 && !coords)) return PD3_EINVAL;
  if (num_pillars >= ((int64_t)1 << 31)) return PD3_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t cells = (int64_t)ny * nx * batch;
  fill_i32_kernel<<<(unsigned)ceil_div(cells, 256), 256, 0, s>>>(reinterpret_cast<int*>(inv), cells, -1);
  if (num_pillars > 0)
    inverse_map_kernel<<<(unsigned)ceil_div(num_pillars, 256), 256, 0, s>>>(coords, num_pillars, batch, ny, nx,
                                                                            reinterpret_cast<int*>(inv));
  return launch_status();
}

extern "C" int pd3_pointpillars_scatter(const float* voxel_features, const int32_t* coords,
                                        int64_t num_pillars, int channels, int batch, int ny,
                                        int nx, float* canvas, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  if (!canvas || !workspace || batch <= 0 || ny <= 0 || nx <= 0 || channels <= 0 || num_pillars < 0)
    return PD3_EINVAL;
  if (num_pillars > 0 && (!voxel_features || !coords)) return PD3_EINVAL;
  if (num_pillars >= ((int64_t)1 << 31)) return PD3_EINVAL;
  if (workspace_bytes < pd3_pointpillars_scatter_workspace(batch, ny, nx)) return PD3_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int* inv = static_cast<int*>(workspace);
  const int64_t plane = (int64_t)ny * nx;
  const int64_t cells = plane * batch;
  fill_i32_kernel<<<(unsigned)ceil_div(cells, 256), 256, 0, s>>>(inv, cells, -1);
  if (num_pillars > 0)
    inverse_map_kernel<<<(unsigned)ceil_div(num_pillars, 256), 256, 0, s>>>(coords, num_pillars,
                                                                            batch, ny, nx, inv);
  const bool vec = (plane % 4 == 0) && (channels % 4 == 0) && (channels / 4 <= 65535) && (batch <= 65535) &&
                   (reinterpret_cast<uintptr_t>(canvas) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(voxel_features) % 16 == 0);
  if (vec) {
    dim3 grid((unsigned)ceil_div(plane / 4, 256), (unsigned)(channels / 4), (unsigned)batch);
    canvas_write_vec4_kernel<<<grid, 256, 0, s>>>(voxel_features, inv, channels, plane, canvas);
  } else {
    dim3 grid((unsigned)ceil_div((int64_t)channels * plane, 256), batch);
    canvas_write_scalar_kernel<<<grid, 256, 0, s>>>(voxel_features, inv, channels, plane, canvas);
  }
  return launch_status();
}
