// hard_voxelize for gfx950 -- deterministic, bit-exact with the reference CPU operator
// (reference: paddle3d/ops/voxel/voxelize_op.cc:19-82).
//
// The reference CPU kernel is a sequential scan: voxel ids are handed out in order of each cell's
// FIRST point, every voxel keeps its FIRST max_num_points_in_voxel points in input order, and once
// max_voxels voxels exist, points that would open a new voxel are dropped (:60-64).  Restated as
// order-independent facts:
//   voxel id of a cell  = rank of the cell's minimum point index among all occupied cells,
//                         dropped when rank >= max_voxels;
//   slot of a point     = number of earlier points in the same cell, dropped when >= max points.
// Both follow from ONE stable sort of the points by cell id (radix_sort.hpp): cells become contiguous
// segments whose elements stay in input order.  Pipeline per launch sequence (grid.y = frame):
//   1. cell_key_kernel     point -> uint32 cell key (INVALID = ncells for out-of-range points)
//   2. stable radix sort   (key, point index)
//   3. seg_head_kernel     mark[point] = sorted position if the point is its cell's first, else -1
//   4. exclusive scan of (mark >= 0) in ORIGINAL point order = voxel id; fused epilogue records the
//      segment start of every voxel id < max_voxels
//   5. gather_kernel       voxel-parallel: writes the complete fixed-shape outputs (data AND zero
//                          padding) exactly once, fully coalesced -- no separate memset of `voxels`.
// HBM traffic per frame ~= the algorithmic bytes (points read once, outputs written once); the sort
// scratch (a few MB per frame) lives in L2 / Infinity Cache.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"

#include <mutex>
#include <type_traits>
#include "radix_sort.hpp"
#include "scan.hpp"
#include "voxelize_wave3d.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace pd3 {

struct VoxGrid {
  float min_x, min_y, min_z;
  float size_x, size_y, size_z;
  int gx, gy, gz;
  uint32_t ncells;
};

// floor((p - min) / size) exactly as voxelize_op.cc:37-45 (fp32 subtract, correctly rounded fp32
// divide, floor), then the int conversion.  On the reference's x86 host a NaN / out-of-int-range
// value converts to INT_MIN (cvttss2si "integer indefinite") and is therefore rejected by the
// `coord < 0` test; v_cvt_i32_f32 would saturate / return 0 instead, so reject those explicitly.
__device__ __forceinline__ bool axis_cell(float p, float lo, float size, int extent, int& c) {
  const float q = floorf((p - lo) / size);
  if (!(q >= 0.0f && q < (float)extent)) return false;  // also false for NaN
  c = (int)q;
  return c < extent;
}

// The same for double points (PD_DISPATCH_FLOATING_TYPES, voxelize_op.cc:128, instantiates the CPU kernel for double
// too): `points[i] - range_min` and the division promote the float attributes to double (:37-45 with T = double).
__device__ __forceinline__ bool axis_cell(double p, float lo, float size, int extent, int& c) {
  const double q = floor((p - (double)lo) / (double)size);
  if (!(q >= 0.0 && q < (double)extent)) return false;  // also false for NaN
  c = (int)q;
  return c < extent;
}

template <typename T>
__global__ __launch_bounds__(256) void cell_key_kernel(const T* __restrict__ points,
                                                       const int32_t* __restrict__ num_points,
                                                       int64_t max_points, int dim, VoxGrid g,
                                                       uint32_t* __restrict__ keys) {
  const int frame = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= max_points) return;
  const int64_t n = num_points ? (int64_t)num_points[frame] : max_points;
  uint32_t key = g.ncells;
  if (i < n) {
    const T* p = points + ((int64_t)frame * max_points + i) * dim;
    int cx, cy, cz;
    if (axis_cell(p[0], g.min_x, g.size_x, g.gx, cx) && axis_cell(p[1], g.min_y, g.size_y, g.gy, cy) &&
        axis_cell(p[2], g.min_z, g.size_z, g.gz, cz)) {
      key = ((uint32_t)cz * (uint32_t)g.gy + (uint32_t)cy) * (uint32_t)g.gx + (uint32_t)cx;
    }
  }
  keys[(int64_t)frame * max_points + i] = key;
}

// dynamic_voxelize: the per-point half of voxelization (no per-voxel cap): coors[i] = (z, y, x) cell of point i by
// the same rule as hard_voxelize (voxelize_op.cc:37-45), (-1, -1, -1) if the point falls outside the range.
__global__ __launch_bounds__(256) void dynamic_voxelize_kernel(const float* __restrict__ points, int64_t n,
                                                               int dim, VoxGrid g, int32_t* __restrict__ coors) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = points + i * dim;
  int cx = 0, cy = 0, cz = 0;
  const bool in = axis_cell(p[0], g.min_x, g.size_x, g.gx, cx) && axis_cell(p[1], g.min_y, g.size_y, g.gy, cy) &&
                  axis_cell(p[2], g.min_z, g.size_z, g.gz, cz);
  coors[i * 3 + 0] = in ? cz : -1;
  coors[i * 3 + 1] = in ? cy : -1;
  coors[i * 3 + 2] = in ? cx : -1;
}

__global__ __launch_bounds__(256) void seg_head_kernel(const uint32_t* __restrict__ skey,
                                                       const uint32_t* __restrict__ sidx,
                                                       int64_t n, uint32_t ncells,
                                                       int* __restrict__ mark) {
  const int frame = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t* k = skey + (int64_t)frame * n;
  const uint32_t key = k[j];
  const bool head = key < ncells && (j == 0 || k[j - 1] != key);
  mark[(int64_t)frame * n + sidx[(int64_t)frame * n + j]] = head ? (int)j : -1;
}

// Epilogue of the flag scan: element i (original point order) is a cell's first point iff
// mark[i] >= 0; its exclusive prefix is the voxel id the sequential reference would hand out.
struct EpiVoxelStart {
  const int* mark;
  int* vox_start;
  int64_t n;
  int max_voxels;
  __device__ __forceinline__ void operator()(int frame, int64_t i, int flag, int prefix, int) const {
    if (flag && prefix < max_voxels)
      vox_start[(int64_t)frame * max_voxels + prefix] = mark[(int64_t)frame * n + i];
  }
};

// One thread per output float of `voxels` (coalesced 4-byte lanes over the contiguous [V,P,D] block).
template <typename T>
__global__ __launch_bounds__(256) void gather_voxels_kernel(
    const T* __restrict__ points, const uint32_t* __restrict__ skey,
    const uint32_t* __restrict__ sidx, const int* __restrict__ vox_start,
    const int* __restrict__ totals, int64_t n, int dim, int max_pts, int max_voxels,
    T* __restrict__ voxels) {
  const int frame = blockIdx.y;
  const int64_t per_frame = (int64_t)max_voxels * max_pts * dim;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= per_frame) return;
  const int row = max_pts * dim;
  const int v = (int)(e / row);
  const int r = (int)(e - (int64_t)v * row);
  const int k = r / dim, c = r - k * dim;
  const int nv = min(totals[frame], max_voxels);
  T out = (T)0;
  if (v < nv) {
    const int64_t s = vox_start[(int64_t)frame * max_voxels + v];
    const uint32_t* keyp = skey + (int64_t)frame * n;
    if (s + k < n && keyp[s + k] == keyp[s]) {
      const uint32_t pi = sidx[(int64_t)frame * n + s + k];
      out = points[((int64_t)frame * n + pi) * dim + c];
    }
  }
  voxels[(int64_t)frame * per_frame + e] = out;
}

// One thread per voxel slot: coords (z, y, x), num_points_per_voxel, and num_voxels.
__global__ __launch_bounds__(256) void voxel_meta_kernel(
    const uint32_t* __restrict__ skey, const int* __restrict__ vox_start,
    const int* __restrict__ totals, int64_t n, int max_pts, int max_voxels, VoxGrid g,
    int32_t* __restrict__ coords, int32_t* __restrict__ num_pts, int32_t* __restrict__ num_voxels,
    int32_t* __restrict__ coors4, uint2* __restrict__ vinfo) {
  const int frame = blockIdx.y;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const int nv = min(totals[frame], max_voxels);
  if (v == 0) num_voxels[frame] = nv;
  if (v >= max_voxels) return;
  int cz = 0, cy = 0, cx = 0, cnt = 0;
  int64_t s = 0;
  if (v < nv) {
    s = vox_start[(int64_t)frame * max_voxels + v];
    const uint32_t* keyp = skey + (int64_t)frame * n;
    const uint32_t key = keyp[s];
    cx = (int)(key % (uint32_t)g.gx);
    const uint32_t t = key / (uint32_t)g.gx;
    cy = (int)(t % (uint32_t)g.gy);
    cz = (int)(t / (uint32_t)g.gy);
    for (int k = 0; k < max_pts; ++k) {
      if (s + k < n && keyp[s + k] == key) ++cnt; else break;
    }
  }
  int32_t* co = coords + ((int64_t)frame * max_voxels + v) * 3;
  co[0] = cz;
  co[1] = cy;
  co[2] = cx;
  num_pts[(int64_t)frame * max_voxels + v] = cnt;
  // (place of the voxel's points in the sorted index list, points kept): what the slot-per-lane row writer of the
  // wave form (vw_rows_kernel) reads -- the sorted index list is its `clist`
  if (vinfo) vinfo[(int64_t)frame * max_voxels + v] = make_uint2((uint32_t)s, (uint32_t)cnt);
  if (coors4) {
    int32_t* c4 = coors4 + ((int64_t)frame * max_voxels + v) * 4;
    c4[0] = v < nv ? frame : -1;
    c4[1] = cz;
    c4[2] = cy;
    c4[3] = cx;
  }
}

static bool make_grid(const float* voxel_size, const float* range, VoxGrid& g) {
  // voxelize_op.cc:97-102: static_cast<int>(round((max - min) / size)) with fp32 operands
  g.min_x = range[0];
  g.min_y = range[1];
  g.min_z = range[2];
  g.size_x = voxel_size[0];
  g.size_y = voxel_size[1];
  g.size_z = voxel_size[2];
  g.gx = (int)std::round((double)((range[3] - range[0]) / voxel_size[0]));
  g.gy = (int)std::round((double)((range[4] - range[1]) / voxel_size[1]));
  g.gz = (int)std::round((double)((range[5] - range[2]) / voxel_size[2]));
  if (g.gx <= 0 || g.gy <= 0 || g.gz <= 0) return false;
  const int64_t cells = (int64_t)g.gx * g.gy * g.gz;
  if (cells >= (int64_t)1 << 31) return false;
  g.ncells = (uint32_t)cells;
  return true;
}

struct VoxWorkspace {
  uint32_t *keys_a, *vals_a, *keys_b, *vals_b;
  int *mark, *vox_start, *hist, *partial, *totals;
  uint2* vinfo;
  size_t bytes;
};

static VoxWorkspace carve(void* base, int batch, int64_t n, int max_voxels, const RadixPlan& plan) {
  Carver c(base);
  VoxWorkspace w;
  const size_t bn = (size_t)batch * n;
  w.keys_a = c.take<uint32_t>(bn);
  w.vals_a = c.take<uint32_t>(bn);
  w.keys_b = c.take<uint32_t>(bn);
  w.vals_b = c.take<uint32_t>(bn);
  w.mark = c.take<int>(bn);
  w.vox_start = c.take<int>((size_t)batch * max_voxels);
  w.hist = c.take<int>((size_t)batch * radix_hist_ints(plan));
  const size_t scan_tiles =
      (size_t)std::max(scan_num_tiles((int64_t)radix_hist_ints(plan)), scan_num_tiles(n));
  w.partial = c.take<int>((size_t)batch * scan_tiles);
  w.totals = c.take<int>((size_t)batch);
  w.vinfo = c.take<uint2>((size_t)batch * max_voxels);
  w.bytes = c.off;
  return w;
}


// ---------------------------------------------------------------------------------------------------
// tiled fast path (voxelize_tiled.hpp)
// ---------------------------------------------------------------------------------------------------
struct VtWorkspace {
  uint32_t *recs, *dir, *cposr, *tilecnt;
  unsigned short* pos16;
  uint32_t* gstart;
  uint2* vinfo;
  float* compact;
  int* totals;
  int64_t stride;  // tiles * kVtTile: per-frame length of the per-point arrays
  int64_t cap;     // per-frame capacity of the compact payload array, in points
  size_t bytes;
};

static VtWorkspace vt_carve(void* base, int batch, int64_t n, int dim, int max_pts, int max_voxels,
                            uint32_t ncells, const VtPlan& p) {
  (void)max_pts;
  (void)ncells;
  Carver c(base);
  VtWorkspace w;
  w.stride = (int64_t)p.tiles * kVtTile;
  w.cap = n;  // every in-range point of a frame at most once
  w.recs = c.take<uint32_t>((size_t)batch * w.stride);
  w.cposr = c.take<uint32_t>((size_t)batch * w.stride);
  w.pos16 = c.take<unsigned short>((size_t)batch * w.stride);
  w.dir = c.take<uint32_t>((size_t)batch * p.tiles * p.groups);
  w.gstart = c.take<uint32_t>((size_t)batch * p.groups * p.cpg);
  w.vinfo = c.take<uint2>((size_t)batch * max_voxels);
  w.tilecnt = c.take<uint32_t>((size_t)batch * p.tiles);
  w.compact = c.take<float>((size_t)batch * w.cap * dim + 4);
  w.totals = c.take<int>((size_t)batch);
  w.bytes = c.off;
  return w;
}

static bool tiled_applicable(const VoxGrid& g, int64_t n, int dim, int max_pts, int max_voxels,
                             VtPlan& plan) {
  plan = vt_plan(g.ncells, n, max_pts);
  const int64_t row = (int64_t)max_pts * dim;
  const int64_t rowq = (row % 4 == 0) ? row / 4 : row;
  return plan.ok && (int64_t)max_voxels * rowq < ((int64_t)1 << 24) - 4096;
}

static int run_tiled(const float* points, const int32_t* num_points, int batch, int64_t n, int dim,
                     const VoxGrid& g, int max_pts, int max_voxels, const VtPlan& plan,
                     float* voxels, int32_t* coords, int32_t* num_pts, int32_t* num_voxels,
                     int32_t* coors4, void* workspace, hipStream_t s, bool gather) {
  VtWorkspace w = vt_carve(workspace, batch, n, dim, max_pts, max_voxels, g.ncells, plan);
  VtGrid vg{g.min_x, g.min_y, g.min_z, g.size_x, g.size_y, g.size_z,
            (float)(1.0 / (double)g.size_x), (float)(1.0 / (double)g.size_y), (float)(1.0 / (double)g.size_z),
            g.gx, g.gy, g.gz, g.ncells};
  const size_t lds_a = ((size_t)kVtTile + (size_t)kVtRouteWaves * plan.groups + kVtRouteWaves + 2) * 4;
  const size_t lds_b = vt_group_lds(plan.cpg, plan.tiles);
  // the gather form keeps a list of point indices where the default form keeps the payload itself (same places)
  uint32_t* clist = reinterpret_cast<uint32_t*>(w.compact);
  const unsigned tile_grid = (unsigned)(plan.tiles * batch);
  vt_route_kernel<<<tile_grid, kVtRouteThreads, lds_a, s>>>(points, num_points, n, dim, vg, plan.low, plan.gbits,
                                                            plan.tiles, batch, max_voxels, w.recs, w.dir, w.pos16,
                                                            w.tilecnt, w.vinfo);
  vt_group_kernel<<<(unsigned)(plan.groups * batch), kVgThreads, lds_b, s>>>(
      w.recs, w.dir, plan.low, plan.gbits, plan.tiles, batch, max_pts, w.cposr, w.gstart, w.tilecnt,
      gather ? clist : nullptr, w.cap);
  // gather form: only the assign job of the C + D launch runs (the emit job's workgroups are not launched)
#define PD3_VT_EMIT(D)                                                                                         \
  vt_assign_emit_kernel<D><<<(gather ? 1 : 2) * tile_grid, kVtRouteThreads, 0, s>>>(                                          \
      points, n, dim, plan.tiles, batch, w.cposr, w.pos16, w.recs, w.dir, plan.low, plan.gbits, w.tilecnt,      \
      max_voxels, vg, w.vinfo, w.totals,                                                                       \
      coords, num_pts, coors4, w.cap, w.compact)
  switch (dim) {
    case 4: PD3_VT_EMIT(4); break;
    case 5: PD3_VT_EMIT(5); break;
    default: PD3_VT_EMIT(0); break;
  }
#undef PD3_VT_EMIT
  const int64_t row = (int64_t)max_pts * dim;
  const bool vec4 = (row % 4 == 0) && (reinterpret_cast<uintptr_t>(voxels) % 16 == 0);
  const int rowq = (int)(vec4 ? row / 4 : row);
  const int units = (int)ceil_div((int64_t)max_voxels * rowq, kVtRowsThreads * kVtRowsIlp);
  const int step_v = kVtRowsThreads / rowq, step_j = kVtRowsThreads % rowq;
  if (gather) {
    if (vec4)
      vt_rows_gather_kernel<4><<<(unsigned)(units * batch), kVtRowsThreads, 0, s>>>(
          points, n, clist, w.cap, w.vinfo, w.totals, batch, units, max_voxels, rowq, step_v, step_j, dim, voxels,
          coords, num_pts, num_voxels, coors4);
    else
      vt_rows_gather_kernel<1><<<(unsigned)(units * batch), kVtRowsThreads, 0, s>>>(
          points, n, clist, w.cap, w.vinfo, w.totals, batch, units, max_voxels, rowq, step_v, step_j, dim, voxels,
          coords, num_pts, num_voxels, coors4);
    return launch_status();
  }
  if (vec4)
    vt_rows_kernel<4><<<(unsigned)(units * batch), kVtRowsThreads, 0, s>>>(
        w.compact, w.cap, w.vinfo, w.totals, batch, units, max_voxels, rowq, step_v, step_j, dim, voxels, coords,
        num_pts, num_voxels, coors4);
  else
    vt_rows_kernel<1><<<(unsigned)(units * batch), kVtRowsThreads, 0, s>>>(
        w.compact, w.cap, w.vinfo, w.totals, batch, units, max_voxels, rowq, step_v, step_j, dim, voxels, coords,
        num_pts, num_voxels, coors4);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------
// wave form of the tiled path (voxelize_wave.hpp)
// ---------------------------------------------------------------------------------------------------
struct VwWorkspace {
  uint32_t *recs, *dir, *clist;
  uint2 *vinfo, *flist, *fcnt;
  int* totals;
  uint32_t *gregion, *aux;  // 3-D form only
  int64_t cap;
  size_t bytes;
};

// three_d (voxelize_wave3d.hpp): first-point lists sized by the points ([frame][N], in the groups' regions) instead
// of [group][cells per group], + the region starts and the per-record word of the multi-pass groups
static VwWorkspace vw_carve(void* base, int batch, int64_t n, int max_voxels, const VwPlan& p, bool three_d = false) {
  Carver c(base);
  VwWorkspace w;
  w.cap = n;
  w.recs = c.take<uint32_t>((size_t)batch * p.tiles * p.tile);
  w.dir = c.take<uint32_t>((size_t)batch * p.tiles * p.groups);
  w.clist = c.take<uint32_t>((size_t)batch * w.cap + 4);
  w.vinfo = c.take<uint2>((size_t)batch * max_voxels);
  w.totals = c.take<int>((size_t)batch);
  w.flist = c.take<uint2>(three_d ? (size_t)batch * w.cap + 4 : (size_t)batch * p.groups * p.cpg);
  w.fcnt = c.take<uint2>((size_t)batch * vw_pow2_above(p.tiles) * p.groups);
  w.gregion = three_d ? c.take<uint32_t>((size_t)batch * p.groups) : nullptr;
  w.aux = three_d ? c.take<uint32_t>((size_t)batch * w.cap + 4) : nullptr;
  w.bytes = c.off;
  return w;
}

static bool wave3d_applicable(const VoxGrid& g, int64_t n, int dim, int max_pts, int max_voxels, int batch, int shape,
                              VwPlan& plan) {
  plan = v3_plan(g.ncells, n, max_pts, batch, shape);
  // (the slot-per-lane row writer: D = 4 / 5, voxel * slot below 2^24)
  return plan.ok && (dim == 4 || dim == 5) && (int64_t)max_voxels * max_pts < ((int64_t)1 << 24) - 4096;
}

constexpr int kVwShapes = 5;

static bool wave_applicable(const VoxGrid& g, int64_t n, int dim, int max_pts, int max_voxels, int batch, int shape,
                            VwPlan& plan) {
  plan = vw_plan(g.ncells, n, max_pts, batch, shape);
  const int64_t row = (int64_t)max_pts * dim;
  const int64_t rowq = (row % 4 == 0) ? row / 4 : row;
  return plan.ok && (int64_t)max_voxels * rowq < ((int64_t)1 << 24) - 4096;
}

// variant bit 0: heavy waves of the group kernel raise their issue priority; bit 1 (run_wave_split): two half batches
// on two streams
// index_span / index_list (pd3_hard_voxelize_index): the per-voxel (start, count) words and the index list are left in
// the CALLER's arrays instead of the workspace, and the row writer does not run (voxels == nullptr): what remains of it
// is vw_finish_index_kernel (num_voxels, the padding rows of coords / counts / coors_batched).
__global__ __launch_bounds__(256) void vw_finish_index_kernel(const int* __restrict__ totals, int batch, int max_voxels,
                                                              int32_t* __restrict__ coords, int32_t* __restrict__ num_pts,
                                                              int32_t* __restrict__ num_voxels,
                                                              int32_t* __restrict__ coors4) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)batch * max_voxels) return;
  const int frame = (int)(t / max_voxels), v = (int)(t - (int64_t)frame * max_voxels);
  const int nv = min(totals[frame], max_voxels);
  if (v == 0) num_voxels[frame] = nv;
  if (v >= nv) {
    const VtInt3 z3{0, 0, 0};
    __builtin_memcpy(coords + t * 3, &z3, sizeof(z3));
    num_pts[t] = 0;
    if (coors4) *reinterpret_cast<int4*>(coors4 + t * 4) = make_int4(-1, 0, 0, 0);
  }
}

static int run_wave(const float* points, const int32_t* num_points, int batch, int64_t n, int dim, const VoxGrid& g,
                    int max_pts, int max_voxels, const VwPlan& plan, float* voxels, int32_t* coords,
                    int32_t* num_pts, int32_t* num_voxels, int32_t* coors4, void* workspace, hipStream_t s,
                    int variant = 0, int frame0 = 0, bool three_d = false, int32_t* index_span = nullptr,
                    int32_t* index_list = nullptr) {
  VwWorkspace w = vw_carve(workspace, batch, n, max_voxels, plan, three_d);
  if (index_span) {
    w.vinfo = reinterpret_cast<uint2*>(index_span);
    w.clist = reinterpret_cast<uint32_t*>(index_list);
  }
  VtGrid vg{g.min_x, g.min_y, g.min_z, g.size_x, g.size_y, g.size_z,
            (float)(1.0 / (double)g.size_x), (float)(1.0 / (double)g.size_y), (float)(1.0 / (double)g.size_z),
            g.gx, g.gy, g.gz, g.ncells};
  if (g.gz == 1 && !vt_single_cell_bounds(g.size_z, vg.z1_lo, vg.z1_hi)) vg.z1_lo = 0.f, vg.z1_hi = -1.f;
  const int waves = plan.threads / kWave;
  const bool pay = (variant & 4) != 0 && voxels != nullptr;  // path 17: payload carried through the route kernel
  const size_t lds_a = ((size_t)plan.tile + (size_t)waves * plan.groups + waves + 2) * 4 +
                       (pay ? (size_t)plan.tile * dim * 4 : 0);
  const unsigned tile_grid = (unsigned)(plan.tiles * batch);
  // up to ~104 KB of dynamic LDS (1024 x 10 tile): above the 48 KB a kernel may use without asking, so the cap is
  // raised per instantiation like every other large-LDS kernel of the library (a host-side table write per launch)
#define PD3_VW_ROUTE(T, R)                                                                                         \
  do {                                                                                                             \
    const hipError_t e_ = pd3_max_dynamic_lds(reinterpret_cast<const void*>(vw_route_kernel<T, R>), (int)lds_a);             \
    if (e_ != hipSuccess) return (int)e_;                                                                          \
    vw_route_kernel<T, R><<<tile_grid, T, lds_a, s>>>(points, num_points, n, dim, vg, plan.low, plan.gbits,        \
                                                      plan.tiles, batch, max_voxels, w.recs, w.dir, w.vinfo,       \
                                                      three_d ? 1 : 0);                                            \
  } while (0)
  if (pay) {
    // (measurement form) the carried payload goes to the head of the `voxels` buffer, which the row writer overwrites
    if (plan.threads != 512 || plan.rounds != 8 || lds_a > 160 * 1024 ||
        (int64_t)batch * plan.tiles * plan.tile * dim > (int64_t)batch * max_voxels * max_pts * dim)
      return PD3_EUNSUPPORTED;
    const hipError_t e_ = pd3_max_dynamic_lds(reinterpret_cast<const void*>(vw_route_kernel<512, 8, true>), (int)lds_a);
    if (e_ != hipSuccess) return (int)e_;
    vw_route_kernel<512, 8, true><<<tile_grid, 512, lds_a, s>>>(points, num_points, n, dim, vg, plan.low, plan.gbits,
                                                                plan.tiles, batch, max_voxels, w.recs, w.dir, w.vinfo,
                                                                three_d ? 1 : 0, voxels);
  } else
  if (plan.threads == 512 && plan.rounds == 8) PD3_VW_ROUTE(512, 8);
  else if (plan.threads == 1024 && plan.rounds == 8) PD3_VW_ROUTE(1024, 8);
  else if (plan.threads == 1024 && plan.rounds == 10) PD3_VW_ROUTE(1024, 10);
  else if (plan.threads == 512 && plan.rounds == 10) PD3_VW_ROUTE(512, 10);
  else if (plan.threads == 1024 && plan.rounds == 5) PD3_VW_ROUTE(1024, 5);
  else return PD3_EINVAL;
#undef PD3_VW_ROUTE
  const int tp = vw_pow2_above(plan.tiles);
  if (three_d)
    v3_group_kernel<<<(unsigned)(plan.groups * batch), kWave, v3_group_lds(plan.tiles), s>>>(
        w.recs, w.dir, plan.gbits, plan.tiles, plan.tile, tp, batch, max_pts, w.clist, w.cap, w.flist, w.fcnt,
        w.gregion, w.aux);
  else
    vw_group_kernel<<<(unsigned)(plan.groups * batch), kWave, vw_group_lds(plan.cpg, plan.tiles), s>>>(
        w.recs, w.dir, plan.low, plan.gbits, plan.tiles, plan.tile, tp, batch, max_pts, w.clist, w.cap, w.flist,
        w.fcnt, (variant & 1) ? (uint32_t)std::max<int64_t>(n / plan.groups, 1) : 0u);
  const size_t lds_c = (size_t)2 * (((size_t)plan.tile + 31) / 32) * 4 + (size_t)kVwAssignCap * 8;
  vw_assign_kernel<<<(unsigned)(plan.tiles * batch), kVwAssignThreads, lds_c, s>>>(
      w.flist, w.fcnt, plan.low, plan.gbits, plan.tiles, plan.tile, tp, batch, max_voxels, vg, w.vinfo, w.totals, coords,
      num_pts, coors4, frame0, three_d ? w.gregion : nullptr, w.cap);
  if (index_span) {
    vw_finish_index_kernel<<<(unsigned)ceil_div((int64_t)batch * max_voxels, 256), 256, 0, s>>>(
        w.totals, batch, max_voxels, coords, num_pts, num_voxels, coors4);
    return launch_status();
  }
  const int64_t row = (int64_t)max_pts * dim;
  const bool vec4 = (row % 4 == 0) && (reinterpret_cast<uintptr_t>(voxels) % 16 == 0);
  const int rowq = (int)(vec4 ? row / 4 : row);
  const int units = (int)ceil_div((int64_t)max_voxels * rowq, kVtRowsThreads * kVtRowsIlp);
  const int step_v = kVtRowsThreads / rowq, step_j = kVtRowsThreads % rowq;
  if (dim == 4 || dim == 5) {
    const int64_t slots = (int64_t)max_voxels * max_pts;
    const int sunits = (int)ceil_div(slots, kVwRowsThreads * kVwRowsIlp);
#define PD3_VW_ROWS(D)                                                                                            \
  vw_rows_kernel<D><<<(unsigned)(sunits * batch), kVwRowsThreads, 0, s>>>(                                   \
      points, n, w.clist, w.cap, w.vinfo, w.totals, batch, sunits, max_voxels, max_pts, voxels, coords, num_pts,   \
      num_voxels, coors4)
    if (dim == 4) PD3_VW_ROWS(4);
    else PD3_VW_ROWS(5);
#undef PD3_VW_ROWS
    return launch_status();
  }
  if (vec4)
    vt_rows_gather_kernel<4><<<(unsigned)(units * batch), kVtRowsThreads, 0, s>>>(
        points, n, w.clist, w.cap, w.vinfo, w.totals, batch, units, max_voxels, rowq, step_v, step_j, dim, voxels,
        coords, num_pts, num_voxels, coors4);
  else
    vt_rows_gather_kernel<1><<<(unsigned)(units * batch), kVtRowsThreads, 0, s>>>(
        points, n, w.clist, w.cap, w.vinfo, w.totals, batch, units, max_voxels, rowq, step_v, step_j, dim, voxels,
        coords, num_pts, num_voxels, coors4);
  return launch_status();
}

// Two half batches on two streams (measurement form, path 12 / 13): the four kernels of a half are a chain of
// differently bound launches (route: instruction + read stream, group / assign: latency, rows: write stream) with an
// idle tail at every boundary; two independent chains let the hardware run one half's latency-bound kernels beside
// the other half's streaming ones.  Fork / join by events on the caller's stream, so the op stays one unit of work
// on that stream (and stays capturable).  Same bytes out: every frame is processed by the same kernels.
static int run_wave_split(const float* points, const int32_t* num_points, int batch, int64_t n, int dim,
                          const VoxGrid& g, int max_pts, int max_voxels, const VwPlan& plan, float* voxels,
                          int32_t* coords, int32_t* num_pts, int32_t* num_voxels, int32_t* coors4, void* workspace,
                          hipStream_t s, int variant) {
  constexpr int kMaxDev = 16;
  static hipStream_t side[kMaxDev] = {};
  static hipEvent_t fork_ev[kMaxDev] = {}, join_ev[kMaxDev] = {};
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev || batch < 2)
    return run_wave(points, num_points, batch, n, dim, g, max_pts, max_voxels, plan, voxels, coords, num_pts,
                    num_voxels, coors4, workspace, s, variant);
  // One side stream and one fork / join event pair per device, shared by every caller: the lock is held from the fork
  // to the join so that two host threads on different streams of one device cannot interleave their records and waits
  // (this is a measurement form; the library's own choice never takes it).
  std::lock_guard<std::mutex> lock(mu);
  if (!side[dev]) {  // the slot counts as initialised only when all three objects exist
    hipStream_t st = nullptr;
    hipEvent_t ef = nullptr, ej = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ef, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ej, hipEventDisableTiming) != hipSuccess) {
      if (ej) (void)hipEventDestroy(ej);
      if (ef) (void)hipEventDestroy(ef);
      if (st) (void)hipStreamDestroy(st);
      return PD3_EINVAL;
    }
    fork_ev[dev] = ef;
    join_ev[dev] = ej;
    side[dev] = st;
  }
  const int b0 = batch / 2, b1 = batch - b0;
  const size_t half_bytes = align_up(vw_carve(nullptr, b0, n, max_voxels, plan).bytes, 256);
  const int64_t vrow = (int64_t)max_voxels * max_pts * dim;
  hipError_t e = hipEventRecord(fork_ev[dev], s);
  if (e == hipSuccess) e = hipStreamWaitEvent(side[dev], fork_ev[dev], 0);
  if (e != hipSuccess) return (int)e;
  int rc = run_wave(points, num_points, b0, n, dim, g, max_pts, max_voxels, plan, voxels, coords, num_pts, num_voxels,
                    coors4, workspace, s, variant);
  if (rc != 0) return rc;
  rc = run_wave(points + (int64_t)b0 * n * dim, num_points ? num_points + b0 : nullptr, b1, n, dim, g, max_pts,
                max_voxels, plan, voxels + (int64_t)b0 * vrow, coords + (int64_t)b0 * max_voxels * 3,
                num_pts + (int64_t)b0 * max_voxels, num_voxels + b0,
                coors4 ? coors4 + (int64_t)b0 * max_voxels * 4 : nullptr, static_cast<char*>(workspace) + half_bytes,
                side[dev], variant, b0);
  if (rc != 0) return rc;
  e = hipEventRecord(join_ev[dev], side[dev]);
  if (e == hipSuccess) e = hipStreamWaitEvent(s, join_ev[dev], 0);
  return e == hipSuccess ? 0 : (int)e;
}

// generic path: stable radix sort of (cell, index), segment heads, flag scan in point order, voxel-parallel gather
template <typename T>
static int run_sort_path(const T* points, const int32_t* num_points, int batch, int64_t n, int num_point_dim,
                         const VoxGrid& g, int max_num_points_in_voxel, int max_voxels, T* voxels, int32_t* coords,
                         int32_t* num_points_per_voxel, int32_t* num_voxels, int32_t* coors_batched, void* workspace,
                         hipStream_t s) {
  const int64_t max_points = n;
  const RadixPlan plan = radix_plan(g.ncells, max_points);
  VoxWorkspace w = carve(workspace, batch, max_points, max_voxels, plan);

  dim3 pgrid((unsigned)ceil_div(n, 256), batch);
  cell_key_kernel<T><<<pgrid, 256, 0, s>>>(points, num_points, n, num_point_dim, g, w.keys_a);
  const int where = enqueue_radix_sort(w.keys_a, w.vals_a, w.keys_b, w.vals_b, n, n, batch, plan,
                                       /*identity_vals=*/true, w.hist, w.partial, s);
  const uint32_t* skey = where ? w.keys_b : w.keys_a;
  const uint32_t* sidx = where ? w.vals_b : w.vals_a;
  seg_head_kernel<<<pgrid, 256, 0, s>>>(skey, sidx, n, g.ncells, w.mark);
  EpiVoxelStart epi{w.mark, w.vox_start, n, max_voxels};
  enqueue_exclusive_scan(w.mark, n, n, batch, w.partial, w.totals, (int*)nullptr, LoadNonNegative{},
                         epi, s);
  dim3 mgrid((unsigned)ceil_div(max_voxels, 256), batch);
  // fp32 points of 4 / 5 floats: the fixed-shape rows are written by the wave form's slot-per-lane row writer (one
  // index load, the point's 16 + 4 bytes, the same two streaming stores; a wave writes 64 * D * 4 contiguous bytes),
  // with the sorted index list as its `clist`.  The float-per-thread gather it replaces here spent 205 us per 8
  // frames of config 4 (1.25 TB/s) on three divisions and three dependent loads per output float.
  const int64_t slots = (int64_t)max_voxels * max_num_points_in_voxel;
  if constexpr (std::is_same<T, float>::value) {
    if ((num_point_dim == 4 || num_point_dim == 5) && slots < ((int64_t)1 << 24) - 4096) {
      voxel_meta_kernel<<<mgrid, 256, 0, s>>>(skey, w.vox_start, w.totals, n, max_num_points_in_voxel, max_voxels, g,
                                              coords, num_points_per_voxel, num_voxels, coors_batched, w.vinfo);
      const int sunits = (int)ceil_div(slots, kVwRowsThreads * kVwRowsIlp);
      if (num_point_dim == 4)
        vw_rows_kernel<4><<<(unsigned)(sunits * batch), kVwRowsThreads, 0, s>>>(
            points, n, sidx, n, w.vinfo, w.totals, batch, sunits, max_voxels, max_num_points_in_voxel, voxels, coords,
            num_points_per_voxel, num_voxels, coors_batched);
      else
        vw_rows_kernel<5><<<(unsigned)(sunits * batch), kVwRowsThreads, 0, s>>>(
            points, n, sidx, n, w.vinfo, w.totals, batch, sunits, max_voxels, max_num_points_in_voxel, voxels, coords,
            num_points_per_voxel, num_voxels, coors_batched);
      return launch_status();
    }
  }
  const int64_t per_frame = slots * num_point_dim;
  dim3 ggrid((unsigned)ceil_div(per_frame, 256), batch);
  gather_voxels_kernel<T><<<ggrid, 256, 0, s>>>(points, skey, sidx, w.vox_start, w.totals, n,
                                                num_point_dim, max_num_points_in_voxel, max_voxels,
                                                voxels);
  voxel_meta_kernel<<<mgrid, 256, 0, s>>>(skey, w.vox_start, w.totals, n, max_num_points_in_voxel,
                                          max_voxels, g, coords, num_points_per_voxel, num_voxels, coors_batched,
                                          nullptr);
  return launch_status();
}

}  // namespace pd3

using namespace pd3;

extern "C" size_t pd3_hard_voxelize_workspace(int batch, int64_t max_points, int num_point_dim,
                                              const float* voxel_size,
                                              const float* point_cloud_range,
                                              int max_num_points_in_voxel, int max_voxels) {
  VoxGrid g;
  if (batch <= 0 || max_points <= 0 || max_voxels <= 0 || !make_grid(voxel_size, point_cloud_range, g))
    return 0;
  const RadixPlan plan = radix_plan(g.ncells, max_points);
  size_t bytes = carve(nullptr, batch, max_points, max_voxels, plan).bytes;
  VtPlan vp;
  if (tiled_applicable(g, max_points, num_point_dim, max_num_points_in_voxel, max_voxels, vp))
    bytes = std::max(bytes, vt_carve(nullptr, batch, max_points, num_point_dim, max_num_points_in_voxel, max_voxels,
                                     g.ncells, vp).bytes);
  for (int shape = -1; shape < kVwShapes; ++shape) {
    VwPlan wp;
    if (wave_applicable(g, max_points, num_point_dim, max_num_points_in_voxel, max_voxels, batch, shape, wp)) {
      bytes = std::max(bytes, vw_carve(nullptr, batch, max_points, max_voxels, wp).bytes);
      if (batch >= 2)  // the two-stream form carves one region per half batch
        bytes = std::max(bytes, align_up(vw_carve(nullptr, batch / 2, max_points, max_voxels, wp).bytes, 256) +
                                    vw_carve(nullptr, batch - batch / 2, max_points, max_voxels, wp).bytes);
    }
  }
  for (int shape = -1; shape < 3; ++shape) {
    VwPlan wp;
    if (wave3d_applicable(g, max_points, num_point_dim, max_num_points_in_voxel, max_voxels, batch, shape, wp))
      bytes = std::max(bytes, vw_carve(nullptr, batch, max_points, max_voxels, wp, true).bytes);
  }
  return bytes;
}

extern "C" int pd3_hard_voxelize_path(const float* points, const int32_t* num_points, int batch,
                                      int64_t max_points, int num_point_dim, const float* voxel_size,
                                      const float* point_cloud_range, int max_num_points_in_voxel,
                                      int max_voxels, float* voxels, int32_t* coords,
                                      int32_t* num_points_per_voxel, int32_t* num_voxels,
                                      int32_t* coors_batched, void* workspace, size_t workspace_bytes,
                                      void* stream, int path) {
  VoxGrid g;
  if (!points || !voxels || !coords || !num_points_per_voxel || !num_voxels || !workspace)
    return PD3_EINVAL;
  if (batch <= 0 || max_points <= 0 || max_points >= ((int64_t)1 << 31) || num_point_dim < 3 ||
      max_num_points_in_voxel <= 0 || max_voxels <= 0)
    return PD3_EINVAL;
  // paths: 0 the library's choice; 1 sort; 2 / 3 tiled forms; 5 wave form; 6 .. 5 + kVwShapes wave form with a forced
  // route-tile shape; 11 wave form + wave priorities; 12 wave form as two half batches on two streams; 13 both;
  // 14 the wave form for 3-D grids (voxelize_wave3d.hpp; 15 / 16: its route tile forced to 8192 / 10240 points);
  // 17 measurement: path 6 (4096-point route tiles) + the points' payload carried through the route kernel's LDS slice
  if (!make_grid(voxel_size, point_cloud_range, g) || path < 0 || path > 17) return PD3_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t n = max_points;
  if (workspace_bytes < pd3_hard_voxelize_workspace(batch, max_points, num_point_dim, voxel_size,
                                                    point_cloud_range, max_num_points_in_voxel,
                                                    max_voxels))
    return PD3_EWORKSPACE;
  if (path == 17) {  // measurement: the wave form (4096-point route tiles) with the payload carried through the route
    VwPlan wp;
    if (!wave_applicable(g, n, num_point_dim, max_num_points_in_voxel, max_voxels, batch, 0, wp)) return PD3_EUNSUPPORTED;
    return run_wave(points, num_points, batch, n, num_point_dim, g, max_num_points_in_voxel, max_voxels, wp, voxels,
                    coords, num_points_per_voxel, num_voxels, coors_batched, workspace, s, 1 | 4);
  }
  if (path >= 14) {
    VwPlan wp;
    if (!wave3d_applicable(g, n, num_point_dim, max_num_points_in_voxel, max_voxels, batch, path == 14 ? -1 : path - 14, wp))
      return PD3_EUNSUPPORTED;
    return run_wave(points, num_points, batch, n, num_point_dim, g, max_num_points_in_voxel, max_voxels, wp, voxels,
                    coords, num_points_per_voxel, num_voxels, coors_batched, workspace, s, 0, 0, true);
  }
  if (path >= 11) {  // measurement variants of the wave form
    VwPlan wp;
    if (!wave_applicable(g, n, num_point_dim, max_num_points_in_voxel, max_voxels, batch, -1, wp))
      return PD3_EUNSUPPORTED;
    const int variant = path == 11 ? 1 : (path == 12 ? 2 : 3);
    if (variant & 2)
      return run_wave_split(points, num_points, batch, n, num_point_dim, g, max_num_points_in_voxel, max_voxels, wp,
                            voxels, coords, num_points_per_voxel, num_voxels, coors_batched, workspace, s, variant);
    return run_wave(points, num_points, batch, n, num_point_dim, g, max_num_points_in_voxel, max_voxels, wp, voxels,
                    coords, num_points_per_voxel, num_voxels, coors_batched, workspace, s, variant);
  }
  if (path >= 5) {  // wave form: 5 = shape chosen from the sizes, 6 .. = route-kernel shape forced (measurement)
    VwPlan wp;
    if (!wave_applicable(g, n, num_point_dim, max_num_points_in_voxel, max_voxels, batch, path - 6, wp))
      return PD3_EUNSUPPORTED;
    return run_wave(points, num_points, batch, n, num_point_dim, g, max_num_points_in_voxel, max_voxels, wp, voxels,
                    coords, num_points_per_voxel, num_voxels, coors_batched, workspace, s);
  }
  if (path == 0) {  // the library's choice: wave form (2-D, else 3-D), else the tiled gather form, else the sort path
    VwPlan wp;
    // (heavy group waves at raised issue priority: 112.8-116.2 against 113.1-122.1 us per 16 frames in four A/B pairs
    //  on two boxes, profiles/r04_vox_paths.txt; path 5 is the same without it)
    if (wave_applicable(g, n, num_point_dim, max_num_points_in_voxel, max_voxels, batch, -1, wp))
      return run_wave(points, num_points, batch, n, num_point_dim, g, max_num_points_in_voxel, max_voxels, wp, voxels,
                      coords, num_points_per_voxel, num_voxels, coors_batched, workspace, s, 1);
    if (wave3d_applicable(g, n, num_point_dim, max_num_points_in_voxel, max_voxels, batch, -1, wp))
      return run_wave(points, num_points, batch, n, num_point_dim, g, max_num_points_in_voxel, max_voxels, wp, voxels,
                      coords, num_points_per_voxel, num_voxels, coors_batched, workspace, s, 0, 0, true);
  }
  {
    VtPlan vp;
    const bool can = tiled_applicable(g, n, num_point_dim, max_num_points_in_voxel, max_voxels, vp);
    if (path >= 2 && !can) return PD3_EUNSUPPORTED;
    if (can && path != 1)
      return run_tiled(points, num_points, batch, n, num_point_dim, g, max_num_points_in_voxel,
                       max_voxels, vp, voxels, coords, num_points_per_voxel, num_voxels, coors_batched,
                       workspace, s, path != 2);  // the library's choice (path 0) is the gather form
  }
  return run_sort_path<float>(points, num_points, batch, max_points, num_point_dim, g, max_num_points_in_voxel,
                              max_voxels, voxels, coords, num_points_per_voxel, num_voxels, coors_batched, workspace, s);
}

extern "C" int64_t pd3_hard_voxelize_index_list_entries(int batch, int64_t max_points) {
  if (batch <= 0 || max_points <= 0) return 0;
  return (int64_t)batch * max_points + 4;
}

extern "C" int pd3_hard_voxelize_index(const float* points, const int32_t* num_points, int batch, int64_t max_points,
                                       int num_point_dim, const float* voxel_size, const float* point_cloud_range,
                                       int max_num_points_in_voxel, int max_voxels, int32_t* vox_span,
                                       int32_t* point_list, int32_t* coords, int32_t* num_points_per_voxel,
                                       int32_t* num_voxels, int32_t* coors_batched, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  VoxGrid g;
  if (!points || !vox_span || !point_list || !coords || !num_points_per_voxel || !num_voxels || !workspace)
    return PD3_EINVAL;
  if (batch <= 0 || max_points <= 0 || max_points >= ((int64_t)1 << 31) || num_point_dim < 3 ||
      max_num_points_in_voxel <= 0 || max_voxels <= 0)
    return PD3_EINVAL;
  if (!make_grid(voxel_size, point_cloud_range, g)) return PD3_EINVAL;
  if (workspace_bytes < pd3_hard_voxelize_workspace(batch, max_points, num_point_dim, voxel_size, point_cloud_range,
                                                    max_num_points_in_voxel, max_voxels))
    return PD3_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  VwPlan wp;
  if (wave_applicable(g, max_points, num_point_dim, max_num_points_in_voxel, max_voxels, batch, -1, wp))
    return run_wave(points, num_points, batch, max_points, num_point_dim, g, max_num_points_in_voxel, max_voxels, wp,
                    nullptr, coords, num_points_per_voxel, num_voxels, coors_batched, workspace, s, 1, 0, false, vox_span,
                    point_list);
  if (wave3d_applicable(g, max_points, num_point_dim, max_num_points_in_voxel, max_voxels, batch, -1, wp))
    return run_wave(points, num_points, batch, max_points, num_point_dim, g, max_num_points_in_voxel, max_voxels, wp,
                    nullptr, coords, num_points_per_voxel, num_voxels, coors_batched, workspace, s, 0, 0, true, vox_span,
                    point_list);
  return PD3_EUNSUPPORTED;  // (the tiled and sort forms keep no per-voxel index list: run pd3_hard_voxelize)
}

extern "C" int pd3_hard_voxelize(const float* points, const int32_t* num_points, int batch,
                                 int64_t max_points, int num_point_dim, const float* voxel_size,
                                 const float* point_cloud_range, int max_num_points_in_voxel,
                                 int max_voxels, float* voxels, int32_t* coords,
                                 int32_t* num_points_per_voxel, int32_t* num_voxels,
                                 int32_t* coors_batched, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  return pd3_hard_voxelize_path(points, num_points, batch, max_points, num_point_dim, voxel_size,
                                point_cloud_range, max_num_points_in_voxel, max_voxels, voxels, coords,
                                num_points_per_voxel, num_voxels, coors_batched, workspace, workspace_bytes,
                                stream, 0);
}

// double points (the reference's CPU kernel instantiated for double): always the generic sort path
extern "C" int pd3_hard_voxelize_f64(const double* points, const int32_t* num_points, int batch,
                                     int64_t max_points, int num_point_dim, const float* voxel_size,
                                     const float* point_cloud_range, int max_num_points_in_voxel,
                                     int max_voxels, double* voxels, int32_t* coords,
                                     int32_t* num_points_per_voxel, int32_t* num_voxels,
                                     int32_t* coors_batched, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  VoxGrid g;
  if (!points || !voxels || !coords || !num_points_per_voxel || !num_voxels || !workspace) return PD3_EINVAL;
  if (batch <= 0 || max_points <= 0 || max_points >= ((int64_t)1 << 31) || num_point_dim < 3 ||
      max_num_points_in_voxel <= 0 || max_voxels <= 0)
    return PD3_EINVAL;
  if (!make_grid(voxel_size, point_cloud_range, g)) return PD3_EINVAL;
  if (workspace_bytes < pd3_hard_voxelize_workspace(batch, max_points, num_point_dim, voxel_size, point_cloud_range,
                                                    max_num_points_in_voxel, max_voxels))
    return PD3_EWORKSPACE;
  return run_sort_path<double>(points, num_points, batch, max_points, num_point_dim, g, max_num_points_in_voxel,
                               max_voxels, voxels, coords, num_points_per_voxel, num_voxels, coors_batched, workspace,
                               static_cast<hipStream_t>(stream));
}

extern "C" int pd3_version(void) { return 200; }
extern "C" const char* pd3_target_arch(void) { return "gfx950"; }

extern "C" int pd3_dynamic_voxelize(const float* points, int64_t num_points, int num_point_dim,
                                    const float* voxel_size, const float* point_cloud_range, int32_t* coors,
                                    void* stream) {
  if (num_points < 0 || num_point_dim < 3 || !voxel_size || !point_cloud_range) return PD3_EINVAL;
  if (num_points == 0) return 0;
  if (!points || !coors) return PD3_EINVAL;
  VoxGrid g;
  if (!make_grid(voxel_size, point_cloud_range, g)) return PD3_EINVAL;
  dynamic_voxelize_kernel<<<(unsigned)ceil_div(num_points, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      points, num_points, num_point_dim, g, coors);
  return launch_status();
}
