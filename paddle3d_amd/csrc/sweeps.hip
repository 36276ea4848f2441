// Multi-sweep merge for gfx950: the step immediately BEFORE hard_voxelize on the nuScenes path.
// (reference: LoadPointCloud.__call__, paddle3d/transforms/reader.py:118-164 -- per sweep: drop the points
//  with |x| < r and |y| < r (ego vehicle), apply the 4x4 `ref_from_curr` transform to xyz, append the sweep's
//  time lag as an extra column, concatenate after the key frame.)
//
// One flag pass + the batched exclusive scan (scan.hpp) + its fused epilogue that writes the surviving
// points in order: a stable compaction, so the merged cloud equals the reference's concatenation row for
// row.  The transform is evaluated in fp64 like NumPy's float64 `dot` (three unfused multiply-adds, left to
// right: -ffp-contract=off) and rounded to fp32 once.  The BLAS dgemm NumPy calls may fuse or reorder, which moves
// the fp64 value by an ulp in a quarter of the cases; the fp32 rounding hides that except on a double-rounding
// boundary (~1e-9 of the values): bit-equal to the reference's own code on every golden vector
// (tests/golden/make_reader_golden.py executes reader.py:91-170; 6 M random values on the CPU: 0 differences).
#include "../../include/paddle3d_amd.h"
#include "common.hpp"
#include "scan.hpp"

namespace pd3 {

constexpr int kMaxSweeps = 16;

struct SweepTable {
  int64_t begin[kMaxSweeps + 1];  // row range of sweep s in the concatenated input
  double m[kMaxSweeps][12];       // first three rows of ref_from_curr, row-major
  float time_lag[kMaxSweeps];
  int filter[kMaxSweeps];         // 0 for the key frame (kept as is, reader.py:121-127)
  int transform[kMaxSweeps];
  int count;
};

__device__ __forceinline__ int sweep_of(const SweepTable& t, int64_t i) {
  int s = 0;
  while (s + 1 < t.count && i >= t.begin[s + 1]) ++s;
  return s;
}

__global__ __launch_bounds__(256) void sweep_flags_kernel(const float* __restrict__ in, int64_t n,
                                                          int dim_in, SweepTable t, float radius,
                                                          int* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = sweep_of(t, i);
  int keep = 1;
  if (t.filter[s]) {
    const float x = in[i * dim_in], y = in[i * dim_in + 1];
    keep = !(fabsf(x) < radius && fabsf(y) < radius);  // reader.py:141-147
  }
  flags[i] = keep;
}

struct EpiMergeSweeps {
  const float* in;
  float* out;
  SweepTable t;
  int dim_in, use_dim, use_time_lag;
  __device__ __forceinline__ void operator()(int, int64_t i, int flag, int prefix, int) const {
    if (!flag) return;
    const int s = sweep_of(t, i);
    const float* p = in + i * dim_in;
    const int od = use_dim + (use_time_lag ? 1 : 0);
    float* o = out + (int64_t)prefix * od;
    float x = p[0], y = p[1], z = p[2];
    if (t.transform[s]) {  // reader.py:150-154, float64 homogeneous transform rounded to fp32
      const double* m = t.m[s];
      const double dx = x, dy = y, dz = z;
      x = (float)(m[0] * dx + m[1] * dy + m[2] * dz + m[3]);
      y = (float)(m[4] * dx + m[5] * dy + m[6] * dz + m[7]);
      z = (float)(m[8] * dx + m[9] * dy + m[10] * dz + m[11]);
    }
    o[0] = x;
    o[1] = y;
    o[2] = z;
    for (int c = 3; c < use_dim; ++c) o[c] = p[c];
    if (use_time_lag) o[use_dim] = t.time_lag[s];
  }
};

static __global__ void sweep_count_kernel(const int* __restrict__ total, int32_t* __restrict__ n_out) {
  *n_out = *total;
}

}  // namespace pd3

using namespace pd3;

extern "C" size_t pd3_merge_sweeps_workspace(int64_t num_points) {
  if (num_points <= 0) return 256;
  Carver c(nullptr);
  c.take<int>((size_t)num_points);
  c.take<int>((size_t)scan_num_tiles(num_points));
  c.take<int>(1);
  return c.off;
}

extern "C" int pd3_merge_sweeps(const float* points, const int64_t* sweep_offsets, int num_sweeps,
                                int dim_in, int use_dim, const double* ref_from_curr,
                                const int32_t* has_transform, const float* time_lag, int use_time_lag, float remove_radius,
                                float* out, int32_t* num_out, void* workspace, size_t workspace_bytes,
                                void* stream) {
  if (!points || !sweep_offsets || !out || !num_out || !workspace || num_sweeps <= 0 ||
      num_sweeps > kMaxSweeps || dim_in < 3 || use_dim < 3 || use_dim > dim_in)
    return PD3_EINVAL;
  const int64_t n = sweep_offsets[num_sweeps];
  if (n < 0 || n >= ((int64_t)1 << 31)) return PD3_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_out, 0, sizeof(int32_t), s);
    return e == hipSuccess ? 0 : (int)e;
  }
  if (workspace_bytes < pd3_merge_sweeps_workspace(n)) return PD3_EWORKSPACE;
  SweepTable t;
  t.count = num_sweeps;
  for (int i = 0; i <= num_sweeps; ++i) t.begin[i] = sweep_offsets[i];
  for (int i = 0; i < num_sweeps; ++i) {
    t.filter[i] = i > 0;  // sweep 0 is the key frame
    // a sweep whose `ref_from_curr` is None (reader.py:150; the key frame repeated as padding at a scene start,
    // nuscenes_pointcloud_det.py:93-100) keeps its coordinates bit for bit (an identity matrix would turn -0 into +0)
    t.transform[i] = (i > 0 && ref_from_curr && (!has_transform || has_transform[i])) ? 1 : 0;
    t.time_lag[i] = (time_lag && i > 0) ? time_lag[i] : 0.f;
    for (int k = 0; k < 12; ++k) t.m[i][k] = ref_from_curr ? ref_from_curr[(size_t)i * 16 + k] : 0.0;
  }
  Carver c(workspace);
  int* flags = c.take<int>((size_t)n);
  int* partial = c.take<int>((size_t)scan_num_tiles(n));
  int* total = c.take<int>(1);
  sweep_flags_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, s>>>(points, n, dim_in, t, remove_radius, flags);
  EpiMergeSweeps epi{points, out, t, dim_in, use_dim, use_time_lag ? 1 : 0};
  enqueue_exclusive_scan(flags, n, n, 1, partial, total, (int*)nullptr, LoadIdentity{}, epi, s);
  sweep_count_kernel<<<1, 1, 0, s>>>(total, num_out);
  return launch_status();
}
