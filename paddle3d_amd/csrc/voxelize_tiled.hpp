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
//        trick, 256 points a step; a point with rank < P drops its index into the cell's list
//        (plist[cell][rank], a dense per-frame array that only occupied cells ever touch).  The cell's
//        first point sets a bit in a per-frame bitmap over point indices and records its cell there.
//   C  count + assign kernels (one bitmap word per thread)
//        prefix popcount over the bitmap = voxel id in first-point order (the reference's hand-out
//        order); voxel id < max_voxels -> cell key.
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
constexpr int kVtGroupThreads = 256;
constexpr int kVtGroupWaves = kVtGroupThreads / kWave;
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
  p.ok = p.groups <= kVtMaxGroups && p.tiles <= kVtMaxTiles && n < ((int64_t)1 << (32 - low));
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

__device__ __forceinline__ bool vt_axis_cell(float p, float lo, float size, int extent, int& c) {
  const float q = floorf((p - lo) / size);  // voxelize_op.cc:37-45; see axis_cell in voxelize.hip
  if (!(q >= 0.0f && q < (float)extent)) return false;
  c = (int)q;
  return c < extent;
}

// ------------------------------------------------------------------------------------------------ A
__global__ __launch_bounds__(kVtRouteThreads) void vt_route_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ num_points, int64_t n, int dim,
    VtGrid g, int low, int groups, int tiles, uint32_t* __restrict__ recs,
    uint32_t* __restrict__ dir) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(vt_smem);  // [waves][groups]
  int* run = reinterpret_cast<int*>(mask + (size_t)kVtRouteWaves * groups);    // [groups]
  int* scan_tmp = run + groups;                                                // [waves + 1]
  const int frame = blockIdx.y, tile = blockIdx.x;
  const int lane = lane_id(), wave = wave_id();
  const int64_t nf = num_points ? min((int64_t)num_points[frame], n) : n;
  const float inv_g = 1.0f / (float)groups;

  for (int d = threadIdx.x; d < groups; d += kVtRouteThreads) {
    run[d] = 0;
#pragma unroll
    for (int w = 0; w < kVtRouteWaves; ++w) mask[(size_t)w * groups + d] = 0ull;
  }
  __syncthreads();
  // phase 1: keys + tile histogram over groups
  uint32_t key[kVtRounds];
  const float* pf = points + (int64_t)frame * n * dim;
#pragma unroll
  for (int r = 0; r < kVtRounds; ++r) {
    const int64_t i = (int64_t)tile * kVtTile + r * kVtRouteThreads + threadIdx.x;
    uint32_t k = 0xFFFFFFFFu;
    if (i < nf) {
      const float* p = pf + i * dim;
      int cx, cy, cz;
      if (vt_axis_cell(p[0], g.min_x, g.size_x, g.gx, cx) && vt_axis_cell(p[1], g.min_y, g.size_y, g.gy, cy) &&
          vt_axis_cell(p[2], g.min_z, g.size_z, g.gz, cz)) {
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
  // phase 1b: exclusive scan of the histogram (<= 1024 bins, 2 per thread) -> tile-local offsets
  {
    const int d0 = threadIdx.x * 2;
    const int c0 = d0 < groups ? run[d0] : 0;
    const int c1 = d0 + 1 < groups ? run[d0 + 1] : 0;
    int total;
    const int ex = block_exclusive_scan<kVtRouteThreads>(c0 + c1, scan_tmp, total);
    uint32_t* drow = dir + ((int64_t)frame * tiles + tile) * groups;
    if (d0 < groups) {
      run[d0] = ex;
      drow[d0] = (uint32_t)ex | ((uint32_t)c0 << 16);
    }
    if (d0 + 1 < groups) {
      run[d0 + 1] = ex + c0;
      drow[d0 + 1] = (uint32_t)(ex + c0) | ((uint32_t)c1 << 16);
    }
  }
  __syncthreads();
  // phase 2: stable rank inside the tile, write records grouped
  uint32_t* out = recs + (int64_t)frame * tiles * kVtTile + (int64_t)tile * kVtTile;
  const unsigned long long below_me = (1ull << lane) - 1ull;
  const uint32_t low_mask = (1u << low) - 1u;
#pragma unroll
  for (int r = 0; r < kVtRounds; ++r) {
    const uint32_t k = key[r];
    const bool valid = k != 0xFFFFFFFFu;
    const int grp = valid ? (int)(k >> low) : 0;
    if (valid) atomicOr(&mask[(size_t)wave * groups + grp], 1ull << lane);
    __syncthreads();
    int rank = 0, total = 0, pos = 0;
    if (valid) {
#pragma unroll
      for (int w = 0; w < kVtRouteWaves; ++w) {
        const unsigned long long m = mask[(size_t)w * groups + grp];
        const int c = __popcll(m);
        total += c;
        if (w < wave) rank += c;
        if (w == wave) rank += __popcll(m & below_me);
      }
      pos = run[grp] + rank;
    }
    __syncthreads();
    if (valid) {
      if (rank == 0) {
        run[grp] += total;
#pragma unroll
        for (int w = 0; w < kVtRouteWaves; ++w) mask[(size_t)w * groups + grp] = 0ull;
      }
      const uint32_t idx = (uint32_t)(tile * kVtTile + r * kVtRouteThreads + threadIdx.x);
      out[pos] = (idx << low) | (k & low_mask);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ B
// Per-cell results live in DENSE per-frame arrays indexed by cell key; only occupied cells are ever
// touched, so nothing needs initialising: cell_npts[key] = min(count, P), plist[key][k] = k-th point.
struct VtCells {
  int* npts;        // [frames][ncells]
  uint32_t* plist;  // [frames][ncells][P]
};

// LDS written by some lanes of a wave and read by others: DS ops of one wave execute in order, so only the
// compiler has to be kept from reordering across this point (no s_barrier: waves run independently).
__device__ __forceinline__ void vt_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// One WAVE per group (4 groups per workgroup), fully wave-synchronous: no workgroup barrier anywhere, so a
// heavy group (a BEV row through the sensor) never stalls its neighbours, and 64-record steps keep the
// per-step latency at one LDS round trip.  The next step's records are fetched before the current step is
// ranked (software prefetch).
__global__ __launch_bounds__(kVtGroupThreads) void vt_group_kernel(
    const uint32_t* __restrict__ recs, const uint32_t* __restrict__ dir, int low, int groups,
    int tiles, int max_pts, int64_t n, uint32_t ncells, VtCells s, uint32_t* __restrict__ owner,
    uint32_t* __restrict__ bitmap, int64_t bitmap_words) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
  const int cpg = 1 << low;
  const int lane = lane_id(), wave = wave_id();
  const size_t per_wave = (size_t)cpg * 8 + (size_t)cpg * 4 + (size_t)(2 * tiles + 2) * 4;
  unsigned char* my = vt_smem + (size_t)wave * ((per_wave + 15) / 16 * 16);
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(my);  // [cpg]
  int* cnt = reinterpret_cast<int*>(mask + cpg);                          // [cpg]
  int* tpre = cnt + cpg;         // [tiles + 1] exclusive prefix of this group's per-tile counts
  int* toff = tpre + tiles + 1;  // [tiles] offset of the group's segment inside each tile
  const int grp = blockIdx.x * kVtGroupWaves + wave, frame = blockIdx.y;
  if (grp >= groups) return;  // whole wave

  for (int c = lane; c < cpg; c += kWave) {
    cnt[c] = 0;
    mask[c] = 0ull;
  }
  // directory column -> per-tile (offset, count), then an exclusive scan across the tiles
  const uint32_t* dcol = dir + (int64_t)frame * tiles * groups + grp;
  int running = 0;
  for (int t0 = 0; t0 < tiles; t0 += kWave) {
    const int t = t0 + lane;
    int c = 0;
    if (t < tiles) {
      const uint32_t d = dcol[(int64_t)t * groups];
      toff[t] = (int)(d & 0xFFFFu);
      c = (int)(d >> 16);
    }
    const int inc = wave_inclusive_scan(c);
    if (t < tiles) tpre[t] = running + inc - c;
    running += __shfl(inc, kWave - 1, kWave);
  }
  const int n_g = running;
  if (lane == 0) tpre[tiles] = n_g;
  vt_wave_sync();
  if (n_g == 0) return;

  const uint32_t* rf = recs + (int64_t)frame * tiles * kVtTile;
  const unsigned long long below_me = (1ull << lane) - 1ull;
  const uint32_t cell_mask = (uint32_t)cpg - 1u;
  const float inv_g = 1.0f / (float)groups;
  uint32_t* plist_f = s.plist + (int64_t)frame * ncells * max_pts;

  auto fetch = [&](int j) -> uint32_t {
    // largest t with tpre[t] <= j  (tpre is non-decreasing, tpre[tiles] = n_g > j)
    int lo = 0, hi = tiles;  // invariant: tpre[lo] <= j < tpre[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tpre[mid] <= j) lo = mid; else hi = mid;
    }
    return rf[(int64_t)lo * kVtTile + toff[lo] + (j - tpre[lo])];
  };

  uint32_t rec_next = lane < n_g ? fetch(lane) : 0u;
  for (int j0 = 0; j0 < n_g; j0 += kWave) {
    const uint32_t rec = rec_next;
    const bool valid = j0 + lane < n_g;
    const int jn = j0 + kWave + lane;
    rec_next = jn < n_g ? fetch(jn) : 0u;  // in flight while this step is ranked
    const int cell = (int)(rec & cell_mask);
    const uint32_t idx = rec >> low;
    if (valid) atomicOr(&mask[cell], 1ull << lane);
    vt_wave_sync();
    int rank = 0, total = 0, base = 0;
    if (valid) {
      const unsigned long long m = mask[cell];
      rank = __popcll(m & below_me);
      total = __popcll(m);
      base = cnt[cell];
    }
    vt_wave_sync();
    if (valid) {
      const int slot = base + rank;  // number of earlier points in this cell
      const uint32_t key = vt_group_to_key((uint32_t)grp, (uint32_t)cell, (uint32_t)groups, inv_g);
      if (slot < max_pts) plist_f[(int64_t)key * max_pts + slot] = idx;
      if (slot == 0) {  // the cell's first point: its index orders the voxels
        owner[(int64_t)frame * bitmap_words * 32 + idx] = key;
        atomicOr(&bitmap[(int64_t)frame * bitmap_words + (idx >> 5)], 1u << (idx & 31));
      }
      if (rank == 0) {
        cnt[cell] = base + total;
        mask[cell] = 0ull;
      }
    }
    vt_wave_sync();
  }
  int* npts_f = s.npts + (int64_t)frame * ncells;
  for (int c = lane; c < cpg; c += kWave) {
    const int k = cnt[c];
    if (k > 0) {
      const uint32_t key = vt_group_to_key((uint32_t)grp, (uint32_t)c, (uint32_t)groups, inv_g);
      if (key < ncells) npts_f[key] = min(k, max_pts);
    }
  }
}

// ------------------------------------------------------------------------------------------------ C
// voxel id = number of set bits before the cell's first point.  Two tiny kernels: per-block popcounts,
// then every block sums the blocks before it and hands out ids.  One bitmap word per thread.
constexpr int kVtAssignThreads = 256;

__global__ __launch_bounds__(kVtAssignThreads) void vt_count_kernel(const uint32_t* __restrict__ bitmap,
                                                                    int64_t bitmap_words,
                                                                    int* __restrict__ wsum) {
  __shared__ int scan_tmp[kVtAssignThreads / kWave + 1];
  const int frame = blockIdx.y;
  const int64_t w = (int64_t)blockIdx.x * kVtAssignThreads + threadIdx.x;
  const int c = w < bitmap_words ? __popc(bitmap[(int64_t)frame * bitmap_words + w]) : 0;
  int total;
  (void)block_exclusive_scan<kVtAssignThreads>(c, scan_tmp, total);
  if (threadIdx.x == 0) wsum[(int64_t)frame * gridDim.x + blockIdx.x] = total;
}

__global__ __launch_bounds__(kVtAssignThreads) void vt_assign_kernel(
    const uint32_t* __restrict__ bitmap, int64_t bitmap_words, const uint32_t* __restrict__ owner,
    const int* __restrict__ wsum, int64_t n, int max_voxels, uint32_t* __restrict__ vid2key,
    int* __restrict__ totals) {
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
  (void)block_exclusive_scan<kVtAssignThreads>(all, scan_tmp, tot_all);
  if (blockIdx.x == 0 && threadIdx.x == 0) totals[frame] = tot_all;
  const int64_t w = (int64_t)blockIdx.x * kVtAssignThreads + threadIdx.x;
  uint32_t bits = w < bitmap_words ? bitmap[(int64_t)frame * bitmap_words + w] : 0u;
  int blk_total;
  int vid = tot_before + block_exclusive_scan<kVtAssignThreads>(__popc(bits), scan_tmp, blk_total);
  const uint32_t* own = owner + (int64_t)frame * n + w * 32;
  while (bits && vid < max_voxels) {
    const int b = __ffs((int)bits) - 1;
    bits &= bits - 1u;
    vid2key[(int64_t)frame * max_voxels + vid] = own[b];
    ++vid;
  }
}

// ------------------------------------------------------------------------------------------------ D
// VEC consecutive floats of one voxel row per thread.  rowq = P*D/VEC chunks per row.
template <int VEC, int DIM>
__global__ __launch_bounds__(256) void vt_write_kernel(
    const float* __restrict__ points, VtCells s, const uint32_t* __restrict__ vid2key,
    const int* __restrict__ totals, int64_t n, uint32_t ncells, int dim_rt, int max_pts,
    int max_voxels, int rowq, float inv_rowq, VtGrid g, float* __restrict__ voxels,
    int32_t* __restrict__ coords, int32_t* __restrict__ num_pts, int32_t* __restrict__ num_voxels) {
  const int dim = DIM > 0 ? DIM : dim_rt;
  const int frame = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;  // chunk index inside the frame
  const int nv = min(totals[frame], max_voxels);
  if (e == 0) num_voxels[frame] = nv;
  if (e >= max_voxels * rowq) return;
  int v = (int)((float)e * inv_rowq);
  if (v * rowq > e) --v;
  else if ((v + 1) * rowq <= e) ++v;
  const int q = e - v * rowq;
  float val[VEC];
#pragma unroll
  for (int u = 0; u < VEC; ++u) val[u] = 0.f;
  int np = 0;
  uint32_t key = 0;
  if (v < nv) {
    key = vid2key[(int64_t)frame * max_voxels + v];
    np = s.npts[(int64_t)frame * ncells + key];
    const uint32_t* pl = s.plist + ((int64_t)frame * ncells + key) * max_pts;
    const float* pf = points + (int64_t)frame * n * dim;
    int k = (q * VEC) / dim, c = (q * VEC) - k * dim;
    uint32_t pi = k < np ? pl[k] : 0u;
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      if (k < np) val[u] = pf[(int64_t)pi * dim + c];
      if (++c == dim) {
        c = 0;
        ++k;
        if (u + 1 < VEC) pi = k < np ? pl[k] : 0u;
      }
    }
  }
  float* dst = voxels + ((int64_t)frame * max_voxels + v) * ((int64_t)rowq * VEC) + (int64_t)q * VEC;
  if (VEC == 4) {
    *reinterpret_cast<float4*>(dst) = make_float4(val[0], val[1], val[2], val[3]);
  } else {
#pragma unroll
    for (int u = 0; u < VEC; ++u) dst[u] = val[u];
  }
  if (q == 0) {  // voxel meta: coords (z, y, x) and count, zero padded
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
  }
}

}  // namespace pd3
