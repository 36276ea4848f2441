// hard_voxelize fast path for BEV-sized grids ("tiled" path): no global sort, four kernels.
//
// Same order-independent restatement of the reference's sequential scan as voxelize.hip, but the
// grouping-by-cell is done hierarchically so that every point record (4 bytes) moves exactly once
// through L2 and all exact in-order ranking happens in LDS:
//
//   A  route_kernel   (tile of 4096 consecutive points, 512 threads)
//        point -> cell key -> (group, cell-in-group); a group is a diagonal set of 2^LOW cells (see kVtSkew).
//        The tile's records are written SORTED BY GROUP, stable in point order (LDS bitmask ranking),
//        into the tile's own 16 KB slice, plus one directory row dir[tile][group] = (offset, count).
//        No global scan / no inter-workgroup dependency: a tile only needs its own histogram.
//   B  group_kernel   (one workgroup per group, 256 threads)
//        walks the directory column of its group in tile order -> its points in INPUT ORDER, and keeps the
//        running per-cell count of its 2^LOW cells in LDS.  Exact in-cell ranks come from the same bitmask
//        trick; a point with rank < P drops its index into the cell's list (plist[cell][rank], a dense
//        per-frame array that only occupied cells ever touch).  The cell's first point raises a byte flag
//        at its own point index and parks the cell key (and later the final count) there.
//   C  count + assign kernels
//        prefix count over the flags = voxel id in first-point order (the reference's hand-out order);
//        voxel id < max_voxels -> (cell key, count).
//   D  write_kernel   voxel-parallel, float4 lanes: the complete fixed-shape outputs (rows, zero padding,
//        coords, counts) are written exactly once, coalesced; points are gathered from L2.
//
// HBM traffic per frame: points read once (A) + outputs written once (D); everything between is a few MB
// of L2-resident scratch.  Preconditions (else the generic sort path of voxelize.hip runs):
// groups = ceil(ncells / 2^LOW) <= 1024 and N < 2^(32-LOW), N <= 4096*1024.
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

struct VtGrid {  // mirror of VoxGrid (kept separate so this header stands alone)
  float min_x, min_y, min_z, size_x, size_y, size_z;
  int gx, gy, gz;
  uint32_t ncells;
};

struct VtPlan {
  int low;      // log2(cells per group)
  int cpg;      // cells per group
  int groups;   // per frame
  int tiles;    // per frame
  int64_t slots;  // slot capacity per frame
  bool ok;
};

static inline VtPlan vt_plan(uint32_t ncells, int64_t n, int max_pts) {
  (void)max_pts;
  VtPlan p{};
  int low = 9;
  if (ceil_div((int64_t)ncells, 1 << low) > kVtMaxGroups) low = 10;
  p.low = low;
  p.cpg = 1 << low;
  p.groups = (int)ceil_div((int64_t)ncells, p.cpg);
  p.tiles = (int)ceil_div(n, kVtTile);
  p.slots = 0;
  p.ok = p.groups <= kVtMaxGroups && p.tiles <= kVtMaxTiles && n < ((int64_t)1 << (32 - low)) - 1;
  return p;
}

// x / d for x < 2^24 and small d, without the integer-division sequence: float estimate + correction.
__device__ __forceinline__ uint32_t vt_div(uint32_t x, uint32_t d, float inv_d) {
  uint32_t q = (uint32_t)((float)x * inv_d);
  if (q * d > x) --q;
  else if ((q + 1u) * d <= x) ++q;
  return q;
}

// Cells are dealt to groups DIAGONALLY: cell key = local * G + lo  ->  group = (lo + kVtSkew * local) mod G.
// A plain "consecutive cells" or "every G-th cell" assignment makes a group a BEV row or column, and the
// rows/columns through the sensor carry ~16x the average number of points (LiDAR density ~ 1/r); the
// skew spreads every dense neighbourhood over hundreds of groups.  (group, local) <-> key is a bijection.
constexpr uint32_t kVtSkew = 7;

__device__ __forceinline__ void vt_key_to_group(uint32_t key, uint32_t G, float inv_g, uint32_t& grp,
                                                uint32_t& local) {
  local = vt_div(key, G, inv_g);
  const uint32_t t = key - local * G + kVtSkew * local;
  grp = t - vt_div(t, G, inv_g) * G;
}

__device__ __forceinline__ uint32_t vt_group_to_key(uint32_t grp, uint32_t local, uint32_t G, float inv_g) {
  const uint32_t s = kVtSkew * local;
  const uint32_t sm = s - vt_div(s, G, inv_g) * G;     // (skew * local) mod G
  const uint32_t lo = grp >= sm ? grp - sm : grp + G - sm;
  return local * G + lo;
}

// x, y, z of a point as ONE 12-byte load (global_load_dwordx3 needs only dword alignment); the separate
// p[0], p[1], p[2] loads cost three address-coalescing passes over the same cache lines.
struct __attribute__((packed, aligned(4))) VtXyz {
  float x, y, z;
};

__device__ __forceinline__ bool vt_axis_cell(float p, float lo, float size, int extent, int& c) {
  const float q = floorf((p - lo) / size);  // voxelize_op.cc:37-45; see axis_cell in voxelize.hip
  if (!(q >= 0.0f && q < (float)extent)) return false;
  c = (int)q;
  return c < extent;
}

// LDS written by some lanes of a wave and read by others: DS ops of one wave execute in order, so only the
// compiler has to be kept from reordering across this point (no s_barrier: waves run independently).
__device__ __forceinline__ void vt_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------ A
// Wave w of the workgroup owns the CONTIGUOUS 512 points [w*512, (w+1)*512) of the tile (8 steps of 64),
// so "stable in point order" = (wave, step, lane) order and almost everything is wave-synchronous:
//   phase 1 (no barrier)  keys + per-wave histogram over groups (private LDS table)
//   barrier, phase 2      thread g: tile histogram of group g, exclusive scan over groups, per-wave start
//                         offsets; directory row written
//   barrier, phase 3 (no barrier)  per-wave stable ranking with the bitmask table, records written
// Three workgroup barriers per 4096 points in total.
__global__ __launch_bounds__(kVtRouteThreads) void vt_route_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ num_points, int64_t n, int dim,
    VtGrid g, int low, int groups, int tiles, uint32_t* __restrict__ recs,
    uint32_t* __restrict__ dir, unsigned char* __restrict__ isfirst) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
  unsigned long long* mask_all = reinterpret_cast<unsigned long long*>(vt_smem);  // [waves][groups]
  int* run_all = reinterpret_cast<int*>(mask_all + (size_t)kVtRouteWaves * groups);  // [waves][groups]
  int* scan_tmp = run_all + (size_t)kVtRouteWaves * groups;                          // [waves + 1]
  const int frame = blockIdx.y, tile = blockIdx.x;
  const int lane = lane_id(), wave = wave_id();
  unsigned long long* mask = mask_all + (size_t)wave * groups;
  int* run = run_all + (size_t)wave * groups;
  const int64_t nf = num_points ? min((int64_t)num_points[frame], n) : n;
  const float inv_g = 1.0f / (float)groups;

  for (int d = lane; d < groups; d += kWave) {
    run[d] = 0;
    mask[d] = 0ull;
  }
  vt_wave_sync();
  // the "is the first point of its cell" flags of this tile start out clear (8 bytes per thread)
  reinterpret_cast<unsigned long long*>(isfirst + ((int64_t)frame * tiles + tile) * kVtTile)[threadIdx.x] = 0ull;
  // phase 1: keys + per-wave histogram over groups
  uint32_t key[kVtRounds];
  const float* pf = points + (int64_t)frame * n * dim;
  const int64_t wave_base = (int64_t)tile * kVtTile + (int64_t)wave * (kVtRounds * kWave);
#pragma unroll
  for (int r = 0; r < kVtRounds; ++r) {
    const int64_t i = wave_base + r * kWave + lane;
    uint32_t k = 0xFFFFFFFFu;
    if (i < nf) {
      VtXyz p;
      __builtin_memcpy(&p, pf + i * dim, sizeof(VtXyz));  // 4-byte aligned 12-byte load
      int cx, cy, cz;
      if (vt_axis_cell(p.x, g.min_x, g.size_x, g.gx, cx) && vt_axis_cell(p.y, g.min_y, g.size_y, g.gy, cy) &&
          vt_axis_cell(p.z, g.min_z, g.size_z, g.gz, cz)) {
        const uint32_t cellkey = ((uint32_t)cz * (uint32_t)g.gy + (uint32_t)cy) * (uint32_t)g.gx + (uint32_t)cx;
        uint32_t grp, local;
        vt_key_to_group(cellkey, (uint32_t)groups, inv_g, grp, local);
        k = (grp << low) | local;  // routed key: group in the high bits, cell-in-group in the low bits
        atomicAdd(&run[grp], 1);
      }
    }
    key[r] = k;
  }
  __syncthreads();
  // phase 2: tile-level offsets.  Groups are spread over the threads, two per thread (groups <= 1024).
  {
    const int d0 = threadIdx.x * 2;
    int c0 = 0, c1 = 0;
    if (d0 < groups)
      for (int w = 0; w < kVtRouteWaves; ++w) c0 += run_all[(size_t)w * groups + d0];
    if (d0 + 1 < groups)
      for (int w = 0; w < kVtRouteWaves; ++w) c1 += run_all[(size_t)w * groups + d0 + 1];
    int total;
    const int ex = block_exclusive_scan<kVtRouteThreads>(c0 + c1, scan_tmp, total);
    // directory is stored group-major (dir[frame][group][tile]) so that the group kernel reads its column
    // as one contiguous run; these strided 4-byte stores are fire-and-forget
    uint32_t* dcol0 = dir + (int64_t)frame * groups * tiles + tile;
    if (d0 < groups) {
      dcol0[(int64_t)d0 * tiles] = (uint32_t)ex | ((uint32_t)c0 << 16);
      int acc = ex;
      for (int w = 0; w < kVtRouteWaves; ++w) {  // per-wave start of group d0 inside the tile
        const int c = run_all[(size_t)w * groups + d0];
        run_all[(size_t)w * groups + d0] = acc;
        acc += c;
      }
    }
    if (d0 + 1 < groups) {
      dcol0[(int64_t)(d0 + 1) * tiles] = (uint32_t)(ex + c0) | ((uint32_t)c1 << 16);
      int acc = ex + c0;
      for (int w = 0; w < kVtRouteWaves; ++w) {
        const int c = run_all[(size_t)w * groups + d0 + 1];
        run_all[(size_t)w * groups + d0 + 1] = acc;
        acc += c;
      }
    }
  }
  __syncthreads();
  // phase 3: per-wave stable ranking, records written grouped
  uint32_t* out = recs + (int64_t)frame * tiles * kVtTile + (int64_t)tile * kVtTile;
  const unsigned long long below_me = (1ull << lane) - 1ull;
  const uint32_t low_mask = (1u << low) - 1u;
#pragma unroll
  for (int r = 0; r < kVtRounds; ++r) {
    const uint32_t k = key[r];
    const bool valid = k != 0xFFFFFFFFu;
    const int grp = valid ? (int)(k >> low) : 0;
    if (valid) atomicOr(&mask[grp], 1ull << lane);
    vt_wave_sync();
    int rank = 0, total = 0, pos = 0;
    if (valid) {
      const unsigned long long m = mask[grp];
      rank = __popcll(m & below_me);
      total = __popcll(m);
      pos = run[grp] + rank;
    }
    vt_wave_sync();
    if (valid) {
      if (rank == 0) {
        run[grp] += total;
        mask[grp] = 0ull;
      }
      const uint32_t idx = (uint32_t)(wave_base + r * kWave + lane);
      out[pos] = (idx << low) | (k & low_mask);
    }
    vt_wave_sync();
  }
}

// ------------------------------------------------------------------------------------------------ B
// Per-cell point lists live in a DENSE per-frame array indexed by cell key; only occupied cells are ever
// touched, so nothing needs initialising: plist[key][k] = index of the cell's k-th point (k < P).
struct VtCells {
  uint32_t* plist;  // [frames][ncells][P]
};

constexpr int kVtGroupSteps = 16;                          // 64-record steps per pass
constexpr int kVtGroupPass = kWave * kVtGroupSteps;        // 1024 records per pass

// One WAVE per group, one wave per workgroup: fully wave-synchronous (no barrier anywhere), ~9 KB of LDS,
// so every group of a batch is resident at once and the kernel lasts as long as its slowest wave.  The
// group's record stream (its points in input order) is cut into passes of 1024 records; ALL records of a
// pass are fetched with independent loads up front (one global round trip per pass, and the diagonal
// group assignment keeps almost every group within one pass), then ranked 64 at a time with the LDS
// bitmask table: rank of a point = points of its cell seen so far + lower lanes of its step with the
// same cell.
__global__ __launch_bounds__(kWave) void vt_group_kernel(
    const uint32_t* __restrict__ recs, const uint32_t* __restrict__ dir, int low, int groups,
    int tiles, int max_pts, uint32_t ncells, VtCells s, uint2* __restrict__ owner,
    unsigned char* __restrict__ isfirst) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
  const int cpg = 1 << low;
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(vt_smem);  // [cpg]
  int* run = reinterpret_cast<int*>(mask + cpg);                               // [cpg] points so far
  int* first = run + cpg;                                                      // [cpg] first point idx
  int* tpre = first + cpg;       // [tiles + 1] exclusive prefix of this group's per-tile counts
  int* toff = tpre + tiles + 1;  // [tiles] offset of the group's segment inside each tile
  uint32_t* srcpos = reinterpret_cast<uint32_t*>(toff + tiles);  // [kVtGroupPass] routed position per record
  const int grp = blockIdx.x, frame = blockIdx.y;
  const int lane = threadIdx.x;

  // directory row of this group (contiguous) -> per-tile (offset, count) and the exclusive scan
  const uint32_t* dcol = dir + ((int64_t)frame * groups + grp) * tiles;
  int running = 0;
  for (int t0 = 0; t0 < tiles; t0 += kWave) {
    const int t = t0 + lane;
    int c = 0;
    if (t < tiles) {
      const uint32_t d = dcol[t];
      toff[t] = (int)(d & 0xFFFFu);
      c = (int)(d >> 16);
    }
    const int inc = wave_inclusive_scan(c);
    if (t < tiles) tpre[t] = running + inc - c;
    running += __shfl(inc, kWave - 1, kWave);
  }
  const int n_g = running;
  if (lane == 0) tpre[tiles] = n_g;
  for (int c = lane; c < cpg; c += kWave) {
    run[c] = 0;
    mask[c] = 0ull;
  }
  vt_wave_sync();
  if (n_g == 0) return;

  const uint32_t* rf = recs + (int64_t)frame * tiles * kVtTile;
  const unsigned long long below_me = (1ull << lane) - 1ull;
  const uint32_t cell_mask = (uint32_t)cpg - 1u;
  const float inv_g = 1.0f / (float)groups;
  uint32_t* plist_f = s.plist + (int64_t)frame * ncells * max_pts;
  const int64_t own_base = (int64_t)frame * tiles * kVtTile;

  for (int p0 = 0; p0 < n_g; p0 += kVtGroupPass) {
    // source address of every record of this pass: lanes = tiles expand their segments into LDS
    // (load-balanced "expand": ~n_g / tiles stores per lane instead of a binary search per record)
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
      if (p0 + u * kWave >= n_g) break;  // uniform
      const bool valid = rec[u] != 0xFFFFFFFFu;
      const int cell = (int)(rec[u] & cell_mask);
      const uint32_t idx = rec[u] >> low;
      if (valid) atomicOr(&mask[cell], 1ull << lane);
      vt_wave_sync();
      int rank = 0, total = 0, b0 = 0;
      if (valid) {
        const unsigned long long m = mask[cell];
        rank = __popcll(m & below_me);
        total = __popcll(m);
        b0 = run[cell];
      }
      vt_wave_sync();
      if (valid) {
        const int slot = b0 + rank;  // number of earlier points in this cell
        if (slot < max_pts) {
          const uint32_t key = vt_group_to_key((uint32_t)grp, (uint32_t)cell, (uint32_t)groups, inv_g);
          plist_f[(int64_t)key * max_pts + slot] = idx;
          if (slot == 0) first[cell] = (int)idx;  // the cell's first point: its index orders the voxels
        }
        if (rank == 0) {
          run[cell] = b0 + total;
          mask[cell] = 0ull;
        }
      }
      vt_wave_sync();
    }
  }
  // per occupied cell: raise the flag of its first point and park (cell key, final count) there
  for (int c = lane; c < cpg; c += kWave) {
    const int k = run[c];
    if (k > 0) {
      const int64_t at = own_base + first[c];
      const uint32_t key = vt_group_to_key((uint32_t)grp, (uint32_t)c, (uint32_t)groups, inv_g);
      owner[at] = make_uint2(key, (uint32_t)min(k, max_pts));
      isfirst[at] = 1;
    }
  }
}

// ------------------------------------------------------------------------------------------------ C
// voxel id = number of first-point flags before the cell's first point.  Two tiny kernels: per-block
// counts, then every block sums the blocks before it and hands out ids.  A thread owns 8 consecutive
// points (one 64-bit load of their flags); the gathers of a voxel's (cell key, count) are independent.
constexpr int kVtAssignThreads = 256;
constexpr int kVtAssignPoints = kVtAssignThreads * 8;  // 2048 points per block (divides kVtTile)

__device__ __forceinline__ int vt_flag_count(unsigned long long x) {
  return __popcll(x & 0x0101010101010101ull);
}

__global__ __launch_bounds__(kVtAssignThreads) void vt_count_kernel(
    const unsigned char* __restrict__ isfirst, int64_t stride, int* __restrict__ wsum) {
  __shared__ int scan_tmp[kVtAssignThreads / kWave + 1];
  const int frame = blockIdx.y;
  const int64_t i = (int64_t)frame * stride + ((int64_t)blockIdx.x * kVtAssignThreads + threadIdx.x) * 8;
  const unsigned long long x = *reinterpret_cast<const unsigned long long*>(isfirst + i);
  int total;
  (void)block_exclusive_scan<kVtAssignThreads>(vt_flag_count(x), scan_tmp, total);
  if (threadIdx.x == 0) wsum[(int64_t)frame * gridDim.x + blockIdx.x] = total;
}

__global__ __launch_bounds__(kVtAssignThreads) void vt_assign_kernel(
    const unsigned char* __restrict__ isfirst, int64_t stride, const uint2* __restrict__ owner,
    const int* __restrict__ wsum, int max_voxels, uint32_t* __restrict__ vid2key,
    int* __restrict__ vid_npts, int* __restrict__ totals) {
  __shared__ int scan_tmp[kVtAssignThreads / kWave + 1];
  const int frame = blockIdx.y, nblk = gridDim.x;
  // sum of the blocks before this one (and, for block 0, of all blocks -> totals)
  const int* ws = wsum + (int64_t)frame * nblk;
  int before = 0, all = 0;
  for (int b = threadIdx.x; b < nblk; b += kVtAssignThreads) {
    const int v = ws[b];
    all += v;
    if (b < (int)blockIdx.x) before += v;
  }
  int tot_before, tot_all;
  (void)block_exclusive_scan<kVtAssignThreads>(before, scan_tmp, tot_before);
  if (blockIdx.x == 0) {  // uniform
    (void)block_exclusive_scan<kVtAssignThreads>(all, scan_tmp, tot_all);
    if (threadIdx.x == 0) totals[frame] = tot_all;
  }
  const int64_t i = (int64_t)frame * stride + ((int64_t)blockIdx.x * kVtAssignThreads + threadIdx.x) * 8;
  const unsigned long long x = *reinterpret_cast<const unsigned long long*>(isfirst + i);
  int blk_total;
  int vid = tot_before + block_exclusive_scan<kVtAssignThreads>(vt_flag_count(x), scan_tmp, blk_total);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    if ((x >> (8 * b)) & 1ull) {
      if (vid < max_voxels) {
        const uint2 o = owner[i + b];
        vid2key[(int64_t)frame * max_voxels + vid] = o.x;
        vid_npts[(int64_t)frame * max_voxels + vid] = (int)o.y;
      }
      ++vid;
    }
  }
}

// ------------------------------------------------------------------------------------------------ D
// Output writer.  Thread (x, y) of a 32 x 8 block owns chunk column x (VEC consecutive floats at a fixed
// offset inside a voxel row) of kVtWriteIlp voxel rows; a wave therefore covers two adjacent rows, so its
// stores are two contiguous row segments.  The column's (point k, channel c) split is computed once; per
// row the dependent chain (voxel -> cell key -> point list -> point) is walked for all kVtWriteIlp rows
// in lock step, so that many independent loads are in flight per level.  Rows >= num_voxels and slots
// >= num_points are written as zeros by the same stores: no separate memset of the outputs.
constexpr int kVtWriteIlp = 4;
constexpr int kVtWriteRows = 8;  // blockDim.y

template <int VEC, int DIM>
__global__ __launch_bounds__(256) void vt_write_kernel(
    const float* __restrict__ points, VtCells s, const uint32_t* __restrict__ vid2key,
    const int* __restrict__ vid_npts, const int* __restrict__ totals, int64_t n, uint32_t ncells,
    int dim_rt, int max_pts, int max_voxels, int rowq, VtGrid g, float* __restrict__ voxels,
    int32_t* __restrict__ coords, int32_t* __restrict__ num_pts, int32_t* __restrict__ num_voxels,
    int32_t* __restrict__ coors4) {
  const int dim = DIM > 0 ? DIM : dim_rt;
  const int frame = blockIdx.y;
  const int nv = min(totals[frame], max_voxels);
  if (blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) num_voxels[frame] = nv;
  const float* pf = points + (int64_t)frame * n * dim;
  const int vbase = blockIdx.x * (kVtWriteRows * kVtWriteIlp) + threadIdx.y;
  for (int q = threadIdx.x; q < rowq; q += 32) {
    const int k0 = (q * VEC) / dim, c0 = (q * VEC) - k0 * dim;
    const bool two = VEC > 1 && c0 + VEC > dim;  // the chunk straddles points k0 and k0 + 1
    int v[kVtWriteIlp], np[kVtWriteIlp];
    uint32_t key[kVtWriteIlp], pa[kVtWriteIlp], pb[kVtWriteIlp];
#pragma unroll
    for (int j = 0; j < kVtWriteIlp; ++j) {  // level 1: voxel -> (cell key, count)
      v[j] = vbase + j * kVtWriteRows;
      key[j] = 0;
      np[j] = 0;
      if (v[j] < nv) {
        key[j] = vid2key[(int64_t)frame * max_voxels + v[j]];
        np[j] = vid_npts[(int64_t)frame * max_voxels + v[j]];
      }
    }
#pragma unroll
    for (int j = 0; j < kVtWriteIlp; ++j) {  // level 2: the (<= 2) point-list entries of this chunk
      const uint32_t* pl = s.plist + ((int64_t)frame * ncells + key[j]) * max_pts;
      pa[j] = (k0 < np[j]) ? pl[k0] : 0u;
      pb[j] = (two && k0 + 1 < np[j]) ? pl[k0 + 1] : 0u;
    }
    float val[kVtWriteIlp][VEC];
#pragma unroll
    for (int j = 0; j < kVtWriteIlp; ++j) {  // level 3: the floats
      int k = k0, c = c0;
      uint32_t pi = pa[j];
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        val[j][u] = (k < np[j]) ? pf[(int64_t)pi * dim + c] : 0.f;
        if (++c == dim) {
          c = 0;
          ++k;
          pi = pb[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kVtWriteIlp; ++j) {
      if (v[j] >= max_voxels) continue;
      float* dst = voxels + ((int64_t)frame * max_voxels + v[j]) * ((int64_t)rowq * VEC) + (int64_t)q * VEC;
      if (VEC == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(val[j][0], val[j][1], val[j][2], val[j][3]);
      } else {
#pragma unroll
        for (int u = 0; u < VEC; ++u) dst[u] = val[j][u];
      }
      if (q == 0) {  // voxel meta: coords (z, y, x) and count, zero padded
        int cz = 0, cy = 0, cx = 0;
        if (v[j] < nv) {
          cx = (int)(key[j] % (uint32_t)g.gx);
          const uint32_t t = key[j] / (uint32_t)g.gx;
          cy = (int)(t % (uint32_t)g.gy);
          cz = (int)(t / (uint32_t)g.gy);
        }
        int32_t* co = coords + ((int64_t)frame * max_voxels + v[j]) * 3;
        co[0] = cz;
        co[1] = cy;
        co[2] = cx;
        num_pts[(int64_t)frame * max_voxels + v[j]] = np[j];
        if (coors4) {  // (batch, z, y, x), batch = -1 on padding rows (HardVoxelizer's coors_pad)
          *reinterpret_cast<int4*>(coors4 + ((int64_t)frame * max_voxels + v[j]) * 4) =
              make_int4(v[j] < nv ? frame : -1, cz, cy, cx);
        }
      }
    }
  }
}

// Output writer, second form: one LANE per (voxel row, point slot).  The lane copies its point as one 16-byte
// (+ one 4-byte for D = 5) load / store pair -- consecutive lanes write consecutive D*4-byte slots, so a wave's
// stores are one contiguous run across adjacent voxel rows -- or writes the slot's zero padding.  A fifth of
// the instructions of the chunk-column form (whose 4-byte gathers cost ~370 instructions per wave trip).
typedef float vt_f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <int DIM>
__global__ __launch_bounds__(256) void vt_write_points_kernel(
    const float* __restrict__ points, VtCells s, const uint32_t* __restrict__ vid2key,
    const int* __restrict__ vid_npts, const int* __restrict__ totals, int64_t n, uint32_t ncells, int max_pts,
    int max_voxels, int rows_per_block, VtGrid g, float* __restrict__ voxels, int32_t* __restrict__ coords,
    int32_t* __restrict__ num_pts, int32_t* __restrict__ num_voxels, int32_t* __restrict__ coors4) {
  static_assert(DIM == 4 || DIM == 5, "point rows of 4 or 5 floats");
  const int frame = blockIdx.y;
  const int nv = min(totals[frame], max_voxels);
  if (blockIdx.x == 0 && threadIdx.x == 0) num_voxels[frame] = nv;
  const int r = (int)threadIdx.x / max_pts, k = (int)threadIdx.x - r * max_pts;
  const int v = blockIdx.x * rows_per_block + r;
  if (r >= rows_per_block || v >= max_voxels) return;
  uint32_t key = 0;
  int np = 0;
  if (v < nv) {
    key = vid2key[(int64_t)frame * max_voxels + v];
    np = vid_npts[(int64_t)frame * max_voxels + v];
  }
  vt_f32x4u a = {0.f, 0.f, 0.f, 0.f};
  float b = 0.f;
  if (k < np) {
    const uint32_t pi = s.plist[((int64_t)frame * ncells + key) * max_pts + k];
    const float* src = points + ((int64_t)frame * n + pi) * DIM;
    a = *reinterpret_cast<const vt_f32x4u*>(src);
    if (DIM == 5) b = src[4];
  }
  float* dst = voxels + (((int64_t)frame * max_voxels + v) * max_pts + k) * DIM;
  *reinterpret_cast<vt_f32x4u*>(dst) = a;
  if (DIM == 5) dst[4] = b;
  if (k == 0) {  // voxel meta: coords (z, y, x) and count, zero padded
    int cz = 0, cy = 0, cx = 0;
    if (v < nv) {
      cx = (int)(key % (uint32_t)g.gx);
      const uint32_t t = key / (uint32_t)g.gx;
      cy = (int)(t % (uint32_t)g.gy);
      cz = (int)(t / (uint32_t)g.gy);
    }
    int32_t* co = coords + ((int64_t)frame * max_voxels + v) * 3;
    co[0] = cz;
    co[1] = cy;
    co[2] = cx;
    num_pts[(int64_t)frame * max_voxels + v] = np;
    if (coors4)  // (batch, z, y, x), batch = -1 on padding rows (HardVoxelizer's coors_pad)
      *reinterpret_cast<int4*>(coors4 + ((int64_t)frame * max_voxels + v) * 4) =
          make_int4(v < nv ? frame : -1, cz, cy, cx);
  }
}

}  // namespace pd3
