// hard_voxelize fast path for BEV-sized grids ("tiled" path): no global sort, no global gather, five kernels.
//
// Same order-independent restatement of the reference's sequential scan as voxelize.hip
// (voxel id of a cell = rank of its first point among all first points; slot of a point = number of earlier
// points in its cell), organised so that every byte that crosses HBM moves in coalesced runs:
//
//   A  route_kernel   (tile of 4096 consecutive points, 512 threads)
//        point -> cell key -> (group, cell-in-group); a group is a diagonal set of 2^LOW cells (see kVtSkew).
//        The tile's 4-byte records are written SORTED BY GROUP, stable in point order, into the tile's own
//        16 KB slice, plus one directory entry dir[group][tile] = (offset, count).  The in-order rank of a
//        point among the wave's points of the same group is the value a returning LDS atomic add hands back
//        (lanes of one ds_add_rtn are served in ascending lane order, instructions of a wave in order --
//        checked on gfx950 by tools/hwcheck/lds_atomic_order.hip and, implicitly, by every bit-exact test).
//   B  group_kernel   (one wave per group)
//        walks the directory row of its group in tile order -> its points in INPUT ORDER; the in-cell slot
//        of a point is again what a returning LDS atomic add on the cell's counter hands back.  Every in-range
//        point gets slot8[point] = min(slot, 255); the cell's first point gets a flag, (cell key, kept count)
//        is parked at that point's index, and the per-64-point chunk counters the next kernel sums are bumped.
//   C  assign_kernel  prefix over the first-point flags in point order = voxel id (the reference's hand-out
//        order) and, in the same scan, the prefix of the kept counts = the voxel's base in the compact
//        payload array.  Writes vinfo[voxel] = (base, count), cellbase[cell key] = base (0xFFFFFFFF for cells
//        past max_voxels) and the voxel's coords / count rows.
//   D  emit_kernel    streams the points a second time (coalesced; they are still in the Infinity Cache),
//        recomputes the cell, and stores each kept point at compact[cellbase[cell] + slot]: the payload in
//        DESTINATION order, densely packed (2.7 MB per nuScenes frame; the 20-byte stores merge in L2).
//   E  rows_kernel    voxel-parallel, one 16-byte store per lane: a row's valid floats are one contiguous run
//        of the compact array; the complete fixed-shape outputs (rows, zero padding, coords, counts) are
//        written exactly once.
//
// HBM traffic per frame: points read (A) and re-read (D), outputs written once (E); everything between is a few
// MB of scratch.  Workgroups are mapped XCD-aware (vt_unit): with batch % 8 == 0 every frame's workgroups of
// every kernel run on one XCD, so the scattered small stores of a frame (records, directory, slots, compact
// payload) merge in that XCD's L2 instead of leaving it as partial lines.
// Preconditions (else the generic sort path of voxelize.hip runs): cells <= 2^20, N < 2^(32-LOW) - 1,
// N <= 4096 * 1024, max points per voxel <= 254.
#pragma once
#include "common.hpp"

#include <algorithm>

namespace pd3 {

constexpr int kVtTile = 4096;
constexpr int kVtRouteThreads = 512;
constexpr int kVtRounds = kVtTile / kVtRouteThreads;  // 8
constexpr int kVtRouteWaves = kVtRouteThreads / kWave;
constexpr int kVtMaxGroups = 1024;
constexpr int kVtMaxTiles = 1024;
constexpr int kVtMaxPts = 254;                         // slot8 keeps 255 for "dropped"
constexpr int kVtAssignThreads = 256;
constexpr int kVtAssignPoints = kVtAssignThreads * 8;  // 2048 points per scan block (divides kVtTile)
constexpr int kVtChunk = 64;                           // points per first-flag counter
constexpr int kVtChunksPerTile = kVtTile / kVtChunk;
constexpr int kVtChunksPerBlock = kVtAssignPoints / kVtChunk;

struct VtGrid {  // mirror of VoxGrid (kept separate so this header stands alone)
  float min_x, min_y, min_z, size_x, size_y, size_z;
  float inv_x, inv_y, inv_z;  // fp32(1 / size): the fast path of vt_axis_cell
  int gx, gy, gz;
  uint32_t ncells;
};

struct VtPlan {
  int low;      // log2(cells per group)
  int cpg;      // cells per group
  int groups;   // per frame, a power of two
  int gbits;    // log2(groups)
  int tiles;    // per frame
  bool ok;
};

static inline VtPlan vt_plan(uint32_t ncells, int64_t n, int max_pts) {
  VtPlan p{};
  int gbits = 10;  // as many groups (= waves of the group kernel) as the route kernel's LDS table allows
  while (gbits > 4 && (int64_t)ncells < ((int64_t)64 << gbits)) --gbits;  // small grids: >= 64 cells per group
  p.gbits = gbits;
  p.groups = 1 << gbits;
  const int64_t per = ceil_div((int64_t)ncells, (int64_t)p.groups);
  int low = 0;
  while (((int64_t)1 << low) < per) ++low;
  p.low = low;
  p.cpg = 1 << low;
  p.tiles = (int)ceil_div(n, kVtTile);
  p.ok = low <= 10 && p.tiles <= kVtMaxTiles && n < ((int64_t)1 << (32 - low)) - 1 && max_pts <= kVtMaxPts;
  return p;
}

// x / d for x < 2^24 and small d, without the integer-division sequence: float estimate + correction.
__device__ __forceinline__ uint32_t vt_div(uint32_t x, uint32_t d, float inv_d) {
  uint32_t q = (uint32_t)((float)x * inv_d);
  if (q * d > x) --q;
  else if ((q + 1u) * d <= x) ++q;
  return q;
}

// Cells are dealt to groups DIAGONALLY: cell key = local * G + lo  ->  group = (lo + kVtSkew * local) mod G
// (G a power of two).  A plain "consecutive cells" or "every G-th cell" assignment makes a group a BEV row or
// column, and the rows/columns through the sensor carry ~16x the average number of points (LiDAR density
// ~ 1/r); the skew spreads every dense neighbourhood over hundreds of groups.  (group, local) <-> key is a
// bijection.
constexpr uint32_t kVtSkew = 7;

__device__ __forceinline__ void vt_key_to_group(uint32_t key, int gbits, uint32_t& grp, uint32_t& local) {
  const uint32_t gm = (1u << gbits) - 1u;
  local = key >> gbits;
  grp = ((key & gm) + kVtSkew * local) & gm;
}

__device__ __forceinline__ uint32_t vt_group_to_key(uint32_t grp, uint32_t local, int gbits) {
  const uint32_t gm = (1u << gbits) - 1u;
  return (local << gbits) | ((grp - kVtSkew * local) & gm);
}

// x, y, z of a point as ONE 12-byte load (global_load_dwordx3 needs only dword alignment).
struct __attribute__((packed, aligned(4))) VtXyz {
  float x, y, z;
};
typedef float vt_f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// Cell index along one axis: floor((p - lo) / size) exactly as voxelize_op.cc:37-45 evaluates it (fp32
// subtract, correctly rounded fp32 divide, floor; see axis_cell in voxelize.hip).  The divide is ~11
// instructions, so the quotient is first estimated as m = (p - lo) * fp32(1 / size): |m - RN((p - lo) / size)|
// < |m| * 2^-22, hence floor(m) is the reference's value whenever m keeps a distance of |m| * 2^-21 from the
// two neighbouring integers; only lanes closer than that (points on cell boundaries), huge or non-finite
// values take the divide.
__device__ __forceinline__ bool vt_axis_cell(float p, float lo, float size, float inv, int extent, int& c) {
  const float t = p - lo;
  const float m = t * inv;
  float q = floorf(m);
  const float frac = m - q;
  const float tol = fabsf(m) * 4.76837158203125e-07f + 9.313225746154785e-10f;  // 2^-21, 2^-30
  if (!(frac >= tol && frac <= 1.0f - tol && fabsf(m) < 1048576.0f)) q = floorf(t / size);
  if (!(q >= 0.0f && q < (float)extent)) return false;  // also false for NaN
  c = (int)q;
  return c < extent;
}

__device__ __forceinline__ bool vt_cell_key(float x, float y, float z, const VtGrid& g, uint32_t& key) {
  int cx, cy, cz;
  if (!(vt_axis_cell(x, g.min_x, g.size_x, g.inv_x, g.gx, cx) &&
        vt_axis_cell(y, g.min_y, g.size_y, g.inv_y, g.gy, cy) &&
        vt_axis_cell(z, g.min_z, g.size_z, g.inv_z, g.gz, cz)))
    return false;
  key = ((uint32_t)cz * (uint32_t)g.gy + (uint32_t)cy) * (uint32_t)g.gx + (uint32_t)cx;
  return true;
}

// LDS written by some lanes of a wave and read by others: DS ops of one wave execute in order, so only the
// compiler has to be kept from reordering across this point (no s_barrier: waves run independently).
__device__ __forceinline__ void vt_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Workgroup -> (frame, unit).  With batch % 8 == 0, workgroup b (which the hardware places on XCD b % 8) works
// on a frame f with f % 8 == b % 8: all workgroups of a frame share one L2.  Speed only -- nothing depends on
// the placement.  The grid is 1-D with batch * units workgroups.
__device__ __forceinline__ void vt_unit(uint32_t b, uint32_t units, uint32_t batch, int& frame, int& unit) {
  if ((batch & 7u) == 0u) {
    const uint32_t j = b >> 3, m = j / units;
    unit = (int)(j - m * units);
    frame = (int)((b & 7u) + 8u * m);
  } else {
    const uint32_t f = b / units;
    frame = (int)f;
    unit = (int)(b - f * units);
  }
}

// ------------------------------------------------------------------------------------------------ A
// Wave w of the workgroup owns the CONTIGUOUS 512 points [w*512, (w+1)*512) of the tile (8 steps of 64),
// so "stable in point order" = (wave, step, lane) order and almost everything is wave-synchronous:
//   phase 1 (no barrier)  keys; ord = returning atomic add on the wave's private count of the point's group
//                         = number of earlier points of this wave in that group
//   barrier, phase 2      thread t: tile histogram of groups 2t, 2t+1, exclusive scan over groups, per-wave
//                         start offsets; directory entries written
//   barrier, phase 3 (no barrier)  record position = the wave's start offset of the group + ord
__global__ __launch_bounds__(kVtRouteThreads) void vt_route_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ num_points, int64_t n, int dim,
    VtGrid g, int low, int gbits, int tiles, int batch, uint32_t* __restrict__ recs,
    uint32_t* __restrict__ dir, unsigned char* __restrict__ isfirst,
    unsigned long long* __restrict__ cnt64) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
  const int groups = 1 << gbits;
  uint32_t* cnt_all = reinterpret_cast<uint32_t*>(vt_smem);                  // [waves][groups]
  int* scan_tmp = reinterpret_cast<int*>(cnt_all + (size_t)kVtRouteWaves * groups);  // [waves + 1]
  int frame, tile;
  vt_unit(blockIdx.x, (uint32_t)tiles, (uint32_t)batch, frame, tile);
  const int lane = lane_id(), wave = wave_id();
  uint32_t* cnt = cnt_all + (size_t)wave * groups;
  const int64_t nf = num_points ? min((int64_t)num_points[frame], n) : n;

  for (int d = lane; d < groups; d += kWave) cnt[d] = 0u;
  vt_wave_sync();
  // this tile's "is the first point of its cell" flags and chunk counters start out clear
  reinterpret_cast<unsigned long long*>(isfirst + ((int64_t)frame * tiles + tile) * kVtTile)[threadIdx.x] = 0ull;
  if (threadIdx.x < kVtChunksPerTile)
    cnt64[((int64_t)frame * tiles + tile) * kVtChunksPerTile + threadIdx.x] = 0ull;

  // phase 1
  const float* pf = points + (int64_t)frame * n * dim;
  const int64_t wave_base = (int64_t)tile * kVtTile + (int64_t)wave * (kVtRounds * kWave);
  VtXyz p[kVtRounds];
#pragma unroll
  for (int r = 0; r < kVtRounds; ++r) {
    const int64_t i = wave_base + r * kWave + lane;
    p[r].x = p[r].y = p[r].z = __builtin_nanf("");
    if (i < nf) __builtin_memcpy(&p[r], pf + i * dim, sizeof(VtXyz));  // 4-byte aligned 12-byte load
  }
  uint32_t key[kVtRounds];   // (group << low) | cell-in-group, all ones = not routed
  uint32_t ord[kVtRounds];   // earlier points of this wave in the same group
#pragma unroll
  for (int r = 0; r < kVtRounds; ++r) {
    uint32_t cellkey = 0, grp = 0, local = 0;
    key[r] = 0xFFFFFFFFu;
    ord[r] = 0;
    if (vt_cell_key(p[r].x, p[r].y, p[r].z, g, cellkey)) {  // NaN (beyond nf) is never valid
      vt_key_to_group(cellkey, gbits, grp, local);
      key[r] = (grp << low) | local;
      ord[r] = atomicAdd(&cnt[grp], 1u);  // ds_add_rtn_u32: lane order within the step, steps in order
    }
  }
  __syncthreads();
  // phase 2: tile-level offsets.  Groups are spread over the threads, two per thread (groups <= 1024).
  {
    const int d0 = threadIdx.x * 2;
    int c0 = 0, c1 = 0;
    if (d0 < groups) {
      for (int w = 0; w < kVtRouteWaves; ++w) c0 += (int)cnt_all[(size_t)w * groups + d0];
      for (int w = 0; w < kVtRouteWaves; ++w) c1 += (int)cnt_all[(size_t)w * groups + d0 + 1];
    }
    int total;
    const int ex = block_exclusive_scan<kVtRouteThreads>(c0 + c1, scan_tmp, total);
    // directory is group-major (dir[frame][group][tile]) so that the group kernel reads its row as one
    // contiguous run; these strided 4-byte stores are fire-and-forget and merge in L2
    if (d0 < groups) {
      uint32_t* dcol0 = dir + ((int64_t)frame * groups + d0) * tiles + tile;
      dcol0[0] = (uint32_t)ex | ((uint32_t)c0 << 16);
      dcol0[tiles] = (uint32_t)(ex + c0) | ((uint32_t)c1 << 16);
      uint32_t acc = (uint32_t)ex;
      for (int w = 0; w < kVtRouteWaves; ++w) {  // per-wave start of group d0 inside the tile
        const uint32_t c = cnt_all[(size_t)w * groups + d0];
        cnt_all[(size_t)w * groups + d0] = acc;
        acc += c;
      }
      for (int w = 0; w < kVtRouteWaves; ++w) {
        const uint32_t c = cnt_all[(size_t)w * groups + d0 + 1];
        cnt_all[(size_t)w * groups + d0 + 1] = acc;
        acc += c;
      }
    }
  }
  __syncthreads();
  // phase 3: records written grouped, stable
  uint32_t* out = recs + ((int64_t)frame * tiles + tile) * kVtTile;
  const uint32_t low_mask = (1u << low) - 1u;
#pragma unroll
  for (int r = 0; r < kVtRounds; ++r) {
    const uint32_t k = key[r];
    if (k != 0xFFFFFFFFu) {
      const uint32_t idx = (uint32_t)(wave_base + r * kWave + lane);
      out[cnt[k >> low] + ord[r]] = (idx << low) | (k & low_mask);
    }
  }
}

// ------------------------------------------------------------------------------------------------ B
constexpr int kVtGroupSteps = 8;                           // 64-record steps per pass
constexpr int kVtGroupPass = kWave * kVtGroupSteps;        // 512 records per pass

static inline size_t vt_group_lds(int cpg, int tiles) {
  return (size_t)cpg * 8 + (size_t)(tiles + 1) * 4 + (size_t)tiles * 4 + (size_t)kVtGroupPass * 4;
}

// One WAVE per group, one wave per workgroup: fully wave-synchronous (no barrier anywhere), ~4 KB of LDS, so
// all groups of a batch are resident at once.  The group's record stream (its points in input order) is cut
// into passes of 512 records; ALL records of a pass are fetched with independent loads up front, then handed
// 64 at a time to the cell counters: slot of a point = what the returning LDS atomic add on its cell's counter
// hands back = number of earlier points of the cell (lanes in ascending order, steps in order).
__global__ __launch_bounds__(kWave) void vt_group_kernel(
    const uint32_t* __restrict__ recs, const uint32_t* __restrict__ dir, int low, int gbits, int tiles,
    int batch, int max_pts, unsigned char* __restrict__ slot8, uint2* __restrict__ owner,
    unsigned char* __restrict__ isfirst, unsigned long long* __restrict__ cnt64) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
  const int cpg = 1 << low, groups = 1 << gbits;
  uint32_t* first = reinterpret_cast<uint32_t*>(vt_smem);      // [cpg] first point of the cell
  uint32_t* run = first + cpg;                                 // [cpg] points of the cell so far
  int* tpre = reinterpret_cast<int*>(run + cpg);               // [tiles + 1] exclusive prefix of per-tile counts
  int* toff = tpre + tiles + 1;                                // [tiles] offset of the segment inside its tile
  uint32_t* srcpos = reinterpret_cast<uint32_t*>(toff + tiles);  // [kVtGroupPass] routed position per record
  int frame, grp;
  vt_unit(blockIdx.x, (uint32_t)groups, (uint32_t)batch, frame, grp);
  const int lane = threadIdx.x;

  // directory row of this group (contiguous) -> per-tile (offset, count) and the exclusive scan
  const uint32_t* drow = dir + ((int64_t)frame * groups + grp) * tiles;
  int running = 0;
  for (int t0 = 0; t0 < tiles; t0 += kWave) {
    const int t = t0 + lane;
    int c = 0;
    if (t < tiles) {
      const uint32_t d = drow[t];
      toff[t] = (int)(d & 0xFFFFu);
      c = (int)(d >> 16);
    }
    const int inc = wave_inclusive_scan(c);
    if (t < tiles) tpre[t] = running + inc - c;
    running += __shfl(inc, kWave - 1, kWave);
  }
  const int n_g = running;
  if (lane == 0) tpre[tiles] = n_g;
  for (int c = lane; c < cpg; c += kWave) run[c] = 0u;
  vt_wave_sync();
  if (n_g == 0) return;

  const int64_t stride = (int64_t)tiles * kVtTile;
  const uint32_t* rf = recs + (int64_t)frame * stride;
  unsigned char* slot_f = slot8 + (int64_t)frame * stride;
  const uint32_t cell_mask = (uint32_t)cpg - 1u;

  for (int p0 = 0; p0 < n_g; p0 += kVtGroupPass) {
    // source address of every record of this pass: lanes = tiles expand their segments into LDS
    const int p1 = min(p0 + kVtGroupPass, n_g);
    for (int t = lane; t < tiles; t += kWave) {
      const int lo = max(tpre[t], p0), hi = min(tpre[t + 1], p1);
      const uint32_t src = (uint32_t)t * kVtTile + (uint32_t)toff[t] - (uint32_t)tpre[t];
      for (int j = lo; j < hi; ++j) srcpos[j - p0] = src + (uint32_t)j;
    }
    vt_wave_sync();
    uint32_t rec[kVtGroupSteps];
#pragma unroll
    for (int u = 0; u < kVtGroupSteps; ++u) {
      const int j = p0 + u * kWave + lane;
      rec[u] = 0xFFFFFFFFu;  // idx field all ones never occurs (N < 2^(32-low) - 1 is enforced by the plan)
      if (j < p1) rec[u] = rf[srcpos[j - p0]];
    }
#pragma unroll
    for (int u = 0; u < kVtGroupSteps; ++u) {
      if (rec[u] != 0xFFFFFFFFu) {
        const uint32_t cell = rec[u] & cell_mask;
        const uint32_t idx = rec[u] >> low;
        const uint32_t slot = atomicAdd(&run[cell], 1u);  // number of earlier points in this cell
        if (slot == 0) first[cell] = idx;                 // the cell's first point: its index orders the voxels
        slot_f[idx] = (unsigned char)min(slot, 255u);
      }
    }
    vt_wave_sync();
  }
  // per occupied cell: raise the flag of its first point, park (cell key, kept count) there, bump the
  // counter of the 64-point chunk the first point lies in (flags in the high word, kept points in the low)
  const int64_t nchunk = stride / kVtChunk;
  for (int c = lane; c < cpg; c += kWave) {
    const uint32_t k = run[c];
    if (k > 0) {
      const uint32_t at = first[c];
      const uint32_t key = vt_group_to_key((uint32_t)grp, (uint32_t)c, gbits);
      const uint32_t kept = min(k, (uint32_t)max_pts);
      owner[(int64_t)frame * stride + at] = make_uint2(key, kept);
      isfirst[(int64_t)frame * stride + at] = 1;
      atomicAdd(&cnt64[(int64_t)frame * nchunk + at / kVtChunk], (1ull << 32) | (unsigned long long)kept);
    }
  }
}

// ------------------------------------------------------------------------------------------------ C
// voxel id = number of first-point flags before the cell's first point; base = kept points of the voxels
// before it.  Both ride one 64-bit sum (flags in the high word, kept points in the low word).  A thread owns
// 8 consecutive points (one 64-bit load of their flags); a workgroup 2048 points = 32 chunk counters.
__device__ __forceinline__ unsigned long long vt_shfl_up64(unsigned long long v, int d) {
  const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, kWave);
  const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, kWave);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long vt_shfl_xor64(unsigned long long v, int d) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d, kWave);
  const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, kWave);
  return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(kVtAssignThreads) void vt_assign_kernel(
    const unsigned char* __restrict__ isfirst, int64_t stride, const uint2* __restrict__ owner,
    const unsigned long long* __restrict__ cnt64, int nblk, int batch, int max_voxels, VtGrid g,
    uint2* __restrict__ vinfo, uint32_t* __restrict__ cellbase, int* __restrict__ totals,
    int32_t* __restrict__ coords, int32_t* __restrict__ num_pts, int32_t* __restrict__ coors4) {
  constexpr int W = kVtAssignThreads / kWave;
  __shared__ unsigned long long s_inc[W], s_before[W], s_all[W];
  int frame, blk;
  vt_unit(blockIdx.x, (uint32_t)nblk, (uint32_t)batch, frame, blk);
  const int64_t i = (int64_t)frame * stride + ((int64_t)blk * kVtAssignThreads + threadIdx.x) * 8;
  const unsigned long long x = *reinterpret_cast<const unsigned long long*>(isfirst + i);
  uint2 o[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    o[b] = make_uint2(0u, 0u);
    if ((x >> (8 * b)) & 1ull) o[b] = owner[i + b];
  }
  // sum of the chunk counters before this block (and, for block 0, of all of them -> totals)
  const int nchunk = nblk * kVtChunksPerBlock;
  const unsigned long long* cc = cnt64 + (int64_t)frame * nchunk;
  unsigned long long before = 0, all = 0;
  const int lim = blk == 0 ? nchunk : blk * kVtChunksPerBlock;
  for (int c = threadIdx.x; c < lim; c += kVtAssignThreads) {
    const unsigned long long v = cc[c];
    all += v;
    if (c < blk * kVtChunksPerBlock) before += v;
  }
  unsigned long long mine = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b)
    if ((x >> (8 * b)) & 1ull) mine += (1ull << 32) | (unsigned long long)o[b].y;
  // one barrier: wave-level inclusive scan of `mine`, wave-level sums of `before` / `all`
  unsigned long long inc = mine;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const unsigned long long nb = vt_shfl_up64(inc, d);
    if (lane_id() >= d) inc += nb;
    before += vt_shfl_xor64(before, d);
    all += vt_shfl_xor64(all, d);
  }
  if (lane_id() == kWave - 1) {
    s_inc[wave_id()] = inc;
    s_before[wave_id()] = before;
    s_all[wave_id()] = all;
  }
  __syncthreads();
  unsigned long long ex = inc - mine, tot_all = 0;
#pragma unroll
  for (int w = 0; w < W; ++w) {
    ex += s_before[w];
    tot_all += s_all[w];
    if (w < wave_id()) ex += s_inc[w];
  }
  if (blk == 0 && threadIdx.x == 0) totals[frame] = (int)(tot_all >> 32);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    if ((x >> (8 * b)) & 1ull) {
      const uint32_t vid = (uint32_t)(ex >> 32), base = (uint32_t)ex;
      const uint32_t key = o[b].x;
      uint32_t cb = 0xFFFFFFFFu;
      if (vid < (uint32_t)max_voxels) {
        const int64_t row = (int64_t)frame * max_voxels + vid;
        vinfo[row] = make_uint2(base, o[b].y);
        cb = base;
        const int cx = (int)(key % (uint32_t)g.gx);
        const uint32_t t = key / (uint32_t)g.gx;
        const int cy = (int)(t % (uint32_t)g.gy), cz = (int)(t / (uint32_t)g.gy);
        int32_t* co = coords + row * 3;  // coords (z, y, x)
        co[0] = cz;
        co[1] = cy;
        co[2] = cx;
        num_pts[row] = (int)o[b].y;
        if (coors4) *reinterpret_cast<int4*>(coors4 + row * 4) = make_int4(frame, cz, cy, cx);
      }
      cellbase[(int64_t)frame * g.ncells + key] = cb;
      ex += (1ull << 32) | (unsigned long long)o[b].y;
    }
  }
}

// ------------------------------------------------------------------------------------------------ D
// Second pass over the points: a kept point (slot < max points, cell within the voxel cap) is stored at its
// destination-ordered place in the compact payload array.  Thread t of the workgroup takes points
// t, t + 512, ... of the tile: fully coalesced loads; the stores are DIM*4-byte pieces that merge in L2.
template <int DIM>
__global__ __launch_bounds__(kVtRouteThreads) void vt_emit_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ num_points, int64_t n, int dim_rt, VtGrid g,
    int tiles, int batch, int max_pts, const unsigned char* __restrict__ slot8,
    const uint32_t* __restrict__ cellbase, int64_t cap, float* __restrict__ compact) {
  const int dim = DIM > 0 ? DIM : dim_rt;
  int frame, tile;
  vt_unit(blockIdx.x, (uint32_t)tiles, (uint32_t)batch, frame, tile);
  const int64_t nf = num_points ? min((int64_t)num_points[frame], n) : n;
  const float* pf = points + (int64_t)frame * n * dim;
  const unsigned char* sf = slot8 + (int64_t)frame * tiles * kVtTile;
  const uint32_t* cb = cellbase + (int64_t)frame * g.ncells;
  float* cf = compact + (int64_t)frame * cap * dim;
  const int64_t base_i = (int64_t)tile * kVtTile + threadIdx.x;
  if (DIM == 4 || DIM == 5) {
    vt_f32x4u a[kVtRounds];
    float e[kVtRounds];
    uint32_t s[kVtRounds];
#pragma unroll
    for (int r = 0; r < kVtRounds; ++r) {
      const int64_t i = base_i + r * kVtRouteThreads;
      a[r] = vt_f32x4u{__builtin_nanf(""), 0.f, 0.f, 0.f};
      e[r] = 0.f;
      s[r] = 255u;
      if (i < nf) {
        a[r] = *reinterpret_cast<const vt_f32x4u*>(pf + i * DIM);
        if (DIM == 5) e[r] = pf[i * DIM + 4];
        s[r] = sf[i];
      }
    }
    uint32_t dst[kVtRounds];
#pragma unroll
    for (int r = 0; r < kVtRounds; ++r) {
      uint32_t key = 0;
      dst[r] = 0xFFFFFFFFu;
      if (vt_cell_key(a[r].x, a[r].y, a[r].z, g, key) && s[r] < (uint32_t)max_pts) {
        const uint32_t b = cb[key];
        if (b != 0xFFFFFFFFu) dst[r] = b + s[r];
      }
    }
#pragma unroll
    for (int r = 0; r < kVtRounds; ++r) {
      if (dst[r] != 0xFFFFFFFFu) {
        float* d = cf + (int64_t)dst[r] * DIM;
        *reinterpret_cast<vt_f32x4u*>(d) = a[r];
        if (DIM == 5) d[4] = e[r];
      }
    }
  } else {
    for (int r = 0; r < kVtRounds; ++r) {
      const int64_t i = base_i + r * kVtRouteThreads;
      if (i >= nf) continue;
      const float* src = pf + i * dim;
      uint32_t key = 0;
      const uint32_t s = sf[i];
      if (!vt_cell_key(src[0], src[1], src[2], g, key) || s >= (uint32_t)max_pts) continue;
      const uint32_t b = cb[key];
      if (b == 0xFFFFFFFFu) continue;
      float* d = cf + ((int64_t)b + s) * dim;
      for (int c = 0; c < dim; ++c) d[c] = src[c];
    }
  }
}

// ------------------------------------------------------------------------------------------------ E
// Output writer.  The frame's voxels tensor is one flat array of V * P * D floats; a lane owns VEC consecutive
// floats of it (VEC = 4 when a row is a whole number of float4: one 16-byte store).  Row v's valid floats are
// the run compact[base(v) * D ... + count(v) * D): one (unaligned) 16-byte load, everything behind is padding.
// Rows >= num_voxels are zeros, and so are their coords / count rows (the live ones were written by C).
constexpr int kVtRowsThreads = 256;
constexpr int kVtRowsIlp = 4;

template <int VEC>
__global__ __launch_bounds__(kVtRowsThreads) void vt_rows_kernel(
    const float* __restrict__ compact, int64_t cap, const uint2* __restrict__ vinfo,
    const int* __restrict__ totals, int batch, int units, int max_voxels, int rowq, int step_v, int step_j,
    int dim, float* __restrict__ voxels, int32_t* __restrict__ coords, int32_t* __restrict__ num_pts,
    int32_t* __restrict__ num_voxels, int32_t* __restrict__ coors4) {
  int frame, unit;
  vt_unit(blockIdx.x, (uint32_t)units, (uint32_t)batch, frame, unit);
  const int nv = min(totals[frame], max_voxels);
  if (unit == 0 && threadIdx.x == 0) num_voxels[frame] = nv;
  const uint32_t total_q = (uint32_t)max_voxels * (uint32_t)rowq;
  const float* cf = compact + (int64_t)frame * cap * dim;
  const uint2* vi = vinfo + (int64_t)frame * max_voxels;
  float* vf = voxels + (int64_t)frame * max_voxels * ((int64_t)rowq * VEC);
  // element u of this thread is q0 + u * 256: its (row, offset in row) follows from the previous one
  const uint32_t q0 = (uint32_t)unit * (kVtRowsIlp * kVtRowsThreads) + threadIdx.x;
  uint32_t v[kVtRowsIlp], j[kVtRowsIlp];
  v[0] = vt_div(min(q0, total_q), (uint32_t)rowq, 1.0f / (float)rowq);
  j[0] = min(q0, total_q) - v[0] * (uint32_t)rowq;
#pragma unroll
  for (int u = 1; u < kVtRowsIlp; ++u) {
    v[u] = v[u - 1] + (uint32_t)step_v;
    j[u] = j[u - 1] + (uint32_t)step_j;
    if (j[u] >= (uint32_t)rowq) {
      j[u] -= (uint32_t)rowq;
      ++v[u];
    }
  }
  uint2 info[kVtRowsIlp];
#pragma unroll
  for (int u = 0; u < kVtRowsIlp; ++u) {
    info[u] = make_uint2(0u, 0u);
    if ((int)v[u] < nv) info[u] = vi[v[u]];
  }
  float val[kVtRowsIlp][VEC];
#pragma unroll
  for (int u = 0; u < kVtRowsIlp; ++u) {
    const uint32_t nfl = info[u].y * (uint32_t)dim;  // valid floats of the row
    const uint32_t r0 = j[u] * VEC;
    const float* src = cf + (int64_t)info[u].x * dim + r0;
#pragma unroll
    for (int c = 0; c < VEC; ++c) val[u][c] = 0.f;
    if (VEC == 4 && r0 + 4 <= nfl) {
      const vt_f32x4u a = *reinterpret_cast<const vt_f32x4u*>(src);
      val[u][0] = a.x;
      val[u][1] = a.y;
      val[u][2] = a.z;
      val[u][3] = a.w;
    } else {
#pragma unroll
      for (int c = 0; c < VEC; ++c)
        if (r0 + c < nfl) val[u][c] = src[c];
    }
  }
#pragma unroll
  for (int u = 0; u < kVtRowsIlp; ++u) {
    const uint32_t q = q0 + (uint32_t)u * kVtRowsThreads;
    if (q >= total_q) continue;
    float* dst = vf + (int64_t)q * VEC;
    if (VEC == 4) {
      *reinterpret_cast<float4*>(dst) = make_float4(val[u][0], val[u][1], val[u][2], val[u][3]);
    } else {
#pragma unroll
      for (int c = 0; c < VEC; ++c) dst[c] = val[u][c];
    }
    if (j[u] == 0 && (int)v[u] >= nv) {  // padding rows of coords / count / coors4 (batch = -1: coors_pad)
      const int64_t row = (int64_t)frame * max_voxels + v[u];
      int32_t* co = coords + row * 3;
      co[0] = 0;
      co[1] = 0;
      co[2] = 0;
      num_pts[row] = 0;
      if (coors4) *reinterpret_cast<int4*>(coors4 + row * 4) = make_int4(-1, 0, 0, 0);
    }
  }
}

}  // namespace pd3
