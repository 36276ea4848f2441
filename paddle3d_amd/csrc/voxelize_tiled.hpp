// hard_voxelize fast path for BEV-sized grids ("tiled" path): no global sort, no global gather, five kernels.
//
// Same order-independent restatement of the reference's sequential scan as voxelize.hip
// (voxel id of a cell = rank of its first point among all first points; slot of a point = number of earlier
// points in its cell).  What this machine charges for is memory TRANSACTIONS, not bytes (tools/hwcheck/
// memrates.hip on MI355X: ~190 G scattered 4-byte stores/s and ~80 G scattered 20-byte stores/s chip-wide,
// i.e. 25 / 60 us for one access per point of a 16-frame batch, against 15 us to stream the batch), so every
// array is laid out such that it is read and written in coalesced runs; the only scattered accesses left are
// one 16-byte record per occupied cell (B -> C) and the 20-byte payload store of the kept points (D).
//
//   A  route_kernel   (tile of 4096 consecutive points, 512 threads)
//        point -> cell key -> (group, cell-in-group); a group is a diagonal set of 2^LOW (<= 512) cells
//        (see kVtSkew).  The tile's 4-byte records are sorted BY GROUP, stable in point order, in LDS and
//        written as one coalesced 16 KB slice, plus the directory row dir[tile][group] = (offset, count) and
//        pos16[point] = the point's position in the slice.  The in-order rank of a point among the wave's
//        points of the same group is the value a returning LDS atomic add hands back (lanes of one ds_add_rtn
//        are served in ascending lane order, instructions of a wave in order -- checked on gfx950 by
//        tools/hwcheck/lds_atomic_order.hip and, implicitly, by every bit-exact test).
//   B  group_kernel   (one wave per group)
//        walks the directory column of its group in tile order -> its points in INPUT ORDER (contiguous runs
//        of the routed slices).  Phase 1: slot of a point = what a returning LDS atomic add on its cell's
//        counter hands back.  Then a scan over the group's cells of min(count, P) places every cell in the
//        group's region of the compact payload array (the region is sized by the group's record count, so its
//        start is the sum of the group's offsets inside the tiles' slices: no global counter).  Phase 2: every
//        record gets cposr[routed position] = compact position (or "dropped"), bit 31 marking a cell's first
//        point; the first point also parks (cell key, compact start, kept count) at its point index and bumps
//        its tile's first-point counter.
//   C  assign_kernel  (tile)  prefix over the first-point flags in point order = voxel id (the reference's
//        hand-out order); writes vinfo[voxel] = (compact start, count) and the voxel's coords / count rows
//        (staged in LDS, coalesced).
//   D  emit_kernel    (tile)  streams the points a second time (coalesced; still in the Infinity Cache) and
//        stores each kept point at compact[cposr[pos16[point]]]: the payload grouped by cell, densely packed
//        (2.7 MB per nuScenes frame; the 20-byte stores merge in L2).  Independent of C.
//   E  rows_kernel    voxel-parallel, one 16-byte store per lane: a row's valid floats are one contiguous run
//        of the compact array; the complete fixed-shape voxels tensor (rows and zero padding) and the padding
//        of the coords / count rows are written exactly once.
//
// HBM traffic per frame: points read (A) and re-read (D), outputs written once (E); everything between is a few
// MB of scratch.  Workgroups are mapped XCD-aware (vt_unit): with batch % 8 == 0 every frame's workgroups of
// every kernel run on one XCD, so the small stores of a frame merge in that XCD's L2.
// Preconditions (else the generic sort path of voxelize.hip runs): cells <= 2^19, N < 2^(32-LOW) - 1,
// N <= 4096 * 1024, max points per voxel <= 254.
#pragma once
#include "common.hpp"

#include <algorithm>

namespace pd3 {

constexpr int kVtTile = 4096;
constexpr int kVtRouteThreads = 512;
constexpr int kVtRounds = kVtTile / kVtRouteThreads;  // 8
constexpr int kVtRouteWaves = kVtRouteThreads / kWave;
constexpr int kVtMaxLow = 9;       // cells per group <= 512: short record streams, one wave per group
constexpr int kVtMaxGbits = 10;    // groups <= 1024: two per thread in the route kernel's scan
constexpr int kVtMaxTiles = 1024;
constexpr int kVtMaxPts = 254;
constexpr uint32_t kVtDropped = 0x7FFFFFFFu;  // cposr: the point is not stored
constexpr uint32_t kVtFirstBit = 0x80000000u;  // cposr: the point is the first of its cell

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
  int bits = 0;
  while (((int64_t)1 << bits) < (int64_t)ncells) ++bits;
  p.gbits = std::min(std::max(bits - kVtMaxLow, 2), kVtMaxGbits);
  p.low = std::max(bits - p.gbits, 0);
  p.groups = 1 << p.gbits;
  p.cpg = 1 << p.low;
  p.tiles = (int)ceil_div(n, kVtTile);
  p.ok = p.low <= kVtMaxLow && p.tiles <= kVtMaxTiles && n < ((int64_t)1 << (32 - p.low)) - 1 &&
         n < ((int64_t)1 << 31) - 1 && max_pts <= kVtMaxPts;
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
// ~ 1/r); the skew spreads every dense neighbourhood over all groups.  (group, local) <-> key is a bijection.
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
struct __attribute__((packed, aligned(4))) VtInt3 {
  int32_t a, b, c;
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
//                         start offsets; directory row written (coalesced)
//   barrier, phase 3      record -> LDS slice at (wave's start offset of the group + ord); pos16 written
//   barrier, phase 4      the slice leaves as one coalesced run
// Side jobs: the tile's first-point counter and its share of vinfo start out clear.
__global__ __launch_bounds__(kVtRouteThreads, 8) void vt_route_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ num_points, int64_t n, int dim,
    VtGrid g, int low, int gbits, int tiles, int batch, int max_voxels, uint32_t* __restrict__ recs,
    uint32_t* __restrict__ dir, unsigned short* __restrict__ pos16, uint32_t* __restrict__ tilecnt,
    uint2* __restrict__ vinfo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
  const int groups = 1 << gbits;
  uint32_t* stage = reinterpret_cast<uint32_t*>(vt_smem);                     // [kVtTile]
  uint32_t* cnt_all = stage + kVtTile;                                        // [waves][groups]
  int* scan_tmp = reinterpret_cast<int*>(cnt_all + (size_t)kVtRouteWaves * groups);  // [waves + 1]
  int frame, tile;
  vt_unit(blockIdx.x, (uint32_t)tiles, (uint32_t)batch, frame, tile);
  const int lane = lane_id(), wave = wave_id();
  uint32_t* cnt = cnt_all + (size_t)wave * groups;
  const int64_t nf = num_points ? min((int64_t)num_points[frame], n) : n;

  for (int d = lane; d < groups; d += kWave) cnt[d] = 0u;
  vt_wave_sync();
  if (threadIdx.x == 0) tilecnt[(int64_t)frame * tiles + tile] = 0u;
  {  // vinfo rows of voxels that never come to life must read (0, 0)
    const int per = (int)ceil_div(max_voxels, tiles);
    const int v1 = min((tile + 1) * per, max_voxels);
    for (int v = tile * per + (int)threadIdx.x; v < v1; v += kVtRouteThreads)
      vinfo[(int64_t)frame * max_voxels + v] = make_uint2(0u, 0u);
  }

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
  int tile_total;
  {
    const int d0 = threadIdx.x * 2;
    int c0 = 0, c1 = 0;
    if (d0 < groups) {
      for (int w = 0; w < kVtRouteWaves; ++w) c0 += (int)cnt_all[(size_t)w * groups + d0];
      for (int w = 0; w < kVtRouteWaves; ++w) c1 += (int)cnt_all[(size_t)w * groups + d0 + 1];
    }
    const int ex = block_exclusive_scan<kVtRouteThreads>(c0 + c1, scan_tmp, tile_total);
    if (d0 < groups) {
      // (offset, count) fit 13 + 13 bits; one 8-byte store per thread: the row leaves coalesced
      *reinterpret_cast<uint2*>(dir + ((int64_t)frame * tiles + tile) * groups + d0) =
          make_uint2((uint32_t)ex | ((uint32_t)c0 << 16), (uint32_t)(ex + c0) | ((uint32_t)c1 << 16));
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
  // phase 3: records into the LDS slice, grouped and stable; position of every point to pos16
  unsigned short* pos_f = pos16 + (int64_t)frame * tiles * kVtTile;
  const uint32_t low_mask = (1u << low) - 1u;
#pragma unroll
  for (int r = 0; r < kVtRounds; ++r) {
    const uint32_t k = key[r];
    const int64_t i = wave_base + r * kWave + lane;
    uint32_t pos = 0xFFFFu;
    if (k != 0xFFFFFFFFu) {
      pos = cnt[k >> low] + ord[r];
      stage[pos] = ((uint32_t)i << low) | (k & low_mask);
    }
    pos_f[i] = (unsigned short)pos;
  }
  __syncthreads();
  // phase 4
  uint32_t* out = recs + ((int64_t)frame * tiles + tile) * kVtTile;
  for (int j = threadIdx.x; j < tile_total; j += kVtRouteThreads) out[j] = stage[j];
}

// ------------------------------------------------------------------------------------------------ B
constexpr int kVtGroupSteps = 24;                          // 64-record steps per pass
constexpr int kVtGroupPass = kWave * kVtGroupSteps;        // 1536 records per pass

static inline size_t vt_group_lds(int cpg, int tiles) {
  return (size_t)cpg * 4 + (size_t)kVtGroupPass * 4 + (size_t)(tiles + 1) * 4 + (size_t)tiles * 4 * 2;
}

// One WAVE per group, one wave per workgroup: fully wave-synchronous (no barrier anywhere).  The group's
// record stream (its points in input order) is cut into passes of 1536 records (a nuScenes group has ~530;
// a group that fits one pass never leaves the registers); ALL records of a pass are fetched with independent
// loads up front.
__global__ __launch_bounds__(kWave, 4) void vt_group_kernel(
    const uint32_t* __restrict__ recs, const uint32_t* __restrict__ dir, int low, int gbits, int tiles,
    int batch, int max_pts, unsigned char* __restrict__ slotr, uint32_t* __restrict__ cposr,
    uint4* __restrict__ owner, uint32_t* __restrict__ tilecnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
  const int cpg = 1 << low, groups = 1 << gbits;
  uint32_t* run = reinterpret_cast<uint32_t*>(vt_smem);          // [cpg] points of the cell so far; later
                                                                 //       (start in region << 8) | kept count
  uint32_t* srcpos = run + cpg;                                  // [kVtGroupPass] routed position per record
  int* tpre = reinterpret_cast<int*>(srcpos + kVtGroupPass);     // [tiles + 1] exclusive prefix of per-tile counts
  int* toff = tpre + tiles + 1;                                  // [tiles] offset of the segment inside its tile
  uint32_t* tfirst = reinterpret_cast<uint32_t*>(toff + tiles);  // [tiles] first points seen per tile
  int frame, grp;
  vt_unit(blockIdx.x, (uint32_t)groups, (uint32_t)batch, frame, grp);
  const int lane = threadIdx.x;

  // directory column of this group -> per-tile (offset, count) and the exclusive scan
  const uint32_t* dcol = dir + (int64_t)frame * tiles * groups + grp;
  int running = 0, offsum = 0;
  for (int t0 = 0; t0 < tiles; t0 += kWave) {
    const int t = t0 + lane;
    int c = 0;
    if (t < tiles) {
      const uint32_t d = dcol[(int64_t)t * groups];
      toff[t] = (int)(d & 0xFFFFu);
      offsum += (int)(d & 0xFFFFu);
      c = (int)(d >> 16);
      tfirst[t] = 0u;
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
  uint32_t* cpos_f = cposr + (int64_t)frame * stride;
  const uint32_t cell_mask = (uint32_t)cpg - 1u;

  // the group's region of the compact array is sized by its record count: it starts where the records of the
  // groups before it would end = the sum over the tiles of this group's offset inside the tile's slice
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) offsum += __shfl_xor(offsum, d, kWave);
  const uint32_t region = (uint32_t)offsum;
  uint4* owner_f = owner + (int64_t)frame * stride;

  // a record's compact position (or "dropped"); a cell's first point parks the cell's facts at its index
#define PD3_VT_PLACE(REC, SP, SLOT)                                                                            \
  {                                                                                                            \
    const uint32_t cell_ = (REC) & cell_mask, sp_ = (SP);                                                      \
    const uint32_t packed_ = run[cell_];                                                                       \
    const uint32_t start_ = region + (packed_ >> 8);                                                           \
    uint32_t cp_ = (SLOT) < (uint32_t)max_pts ? start_ + (SLOT) : kVtDropped;                                  \
    if ((SLOT) == 0u) {                                                                                        \
      cp_ |= kVtFirstBit;                                                                                      \
      owner_f[(REC) >> low] =                                                                                  \
          make_uint4(vt_group_to_key((uint32_t)grp, cell_, gbits), start_, packed_ & 0xFFu, 0u);              \
      atomicAdd(&tfirst[sp_ / kVtTile], 1u);                                                                   \
    }                                                                                                          \
    cpos_f[sp_] = cp_;                                                                                         \
  }
  // cells -> places in the group's region.  Any order of the cells will do (the array is scratch), so lane l
  // takes cells l, l + 64, ...: conflict-free LDS walks and one scan across the lanes.
#define PD3_VT_SCAN_CELLS()                                                                                    \
  {                                                                                                            \
    uint32_t mine_ = 0;                                                                                        \
    for (int c = lane; c < cpg; c += kWave) mine_ += min(run[c], (uint32_t)max_pts);                           \
    uint32_t at_ = (uint32_t)wave_inclusive_scan((int)mine_) - mine_;                                          \
    for (int c = lane; c < cpg; c += kWave) {                                                                  \
      const uint32_t kept_ = min(run[c], (uint32_t)max_pts);                                                   \
      run[c] = (at_ << 8) | kept_;                                                                             \
      at_ += kept_;                                                                                            \
    }                                                                                                          \
    vt_wave_sync();                                                                                            \
  }
#define PD3_VT_EXPAND(P0, P1)                                                                                  \
  {                                                                                                            \
    for (int t = lane; t < tiles; t += kWave) {                                                                \
      const int lo_ = max(tpre[t], (P0)), hi_ = min(tpre[t + 1], (P1));                                        \
      const uint32_t src_ = (uint32_t)t * kVtTile + (uint32_t)toff[t] - (uint32_t)tpre[t];                     \
      for (int j = lo_; j < hi_; ++j) srcpos[j - (P0)] = src_ + (uint32_t)j;                                   \
    }                                                                                                          \
    vt_wave_sync();                                                                                            \
  }

  // all records of a pass are fetched with independent loads up front (their routed positions stay in LDS)
#define PD3_VT_LOAD(P0, P1)                                                                                    \
  uint32_t rec[kVtGroupSteps], sl[kVtGroupSteps];                                                              \
  _Pragma("unroll") for (int u = 0; u < kVtGroupSteps; ++u) {                                                  \
    const int j = (P0) + u * kWave + lane;                                                                     \
    rec[u] = 0xFFFFFFFFu; /* idx field all ones never occurs (N < 2^(32-low) - 1: the plan) */                 \
    sl[u] = 0;                                                                                                 \
    if (j < (P1)) rec[u] = rf[srcpos[j - (P0)]];                                                               \
  }

  if (n_g <= kVtGroupPass) {
    // the whole group in registers: records are fetched once, slots never leave the wave
    PD3_VT_EXPAND(0, n_g)
    PD3_VT_LOAD(0, n_g)
#pragma unroll
    for (int u = 0; u < kVtGroupSteps; ++u)
      if (rec[u] != 0xFFFFFFFFu) sl[u] = atomicAdd(&run[rec[u] & cell_mask], 1u);  // earlier points in this cell
    vt_wave_sync();
    PD3_VT_SCAN_CELLS()
#pragma unroll
    for (int u = 0; u < kVtGroupSteps; ++u)
      if (rec[u] != 0xFFFFFFFFu) PD3_VT_PLACE(rec[u], srcpos[u * kWave + lane], sl[u])
  } else {
    // long record streams (one cell hammered by thousands of points): passes; slots travel through slotr
    unsigned char* slot_f = slotr + (int64_t)frame * stride;
    for (int p0 = 0; p0 < n_g; p0 += kVtGroupPass) {
      const int p1 = min(p0 + kVtGroupPass, n_g);
      PD3_VT_EXPAND(p0, p1)
      PD3_VT_LOAD(p0, p1)
#pragma unroll
      for (int u = 0; u < kVtGroupSteps; ++u)
        if (rec[u] != 0xFFFFFFFFu) {
          sl[u] = atomicAdd(&run[rec[u] & cell_mask], 1u);
          slot_f[srcpos[u * kWave + lane]] = (unsigned char)min(sl[u], 255u);
        }
      vt_wave_sync();
    }
    PD3_VT_SCAN_CELLS()
    for (int p0 = 0; p0 < n_g; p0 += kVtGroupPass) {
      const int p1 = min(p0 + kVtGroupPass, n_g);
      PD3_VT_EXPAND(p0, p1)
      PD3_VT_LOAD(p0, p1)
#pragma unroll
      for (int u = 0; u < kVtGroupSteps; ++u)
        if (rec[u] != 0xFFFFFFFFu) sl[u] = slot_f[srcpos[u * kWave + lane]];
#pragma unroll
      for (int u = 0; u < kVtGroupSteps; ++u)
        if (rec[u] != 0xFFFFFFFFu) PD3_VT_PLACE(rec[u], srcpos[u * kWave + lane], sl[u])
      vt_wave_sync();
    }
  }
#undef PD3_VT_LOAD
#undef PD3_VT_PLACE
#undef PD3_VT_SCAN_CELLS
#undef PD3_VT_EXPAND
  vt_wave_sync();
  for (int t = lane; t < tiles; t += kWave)
    if (tfirst[t]) atomicAdd(&tilecnt[(int64_t)frame * tiles + t], tfirst[t]);
}

// ------------------------------------------------------------------------------------------------ C + D
// Two independent per-tile jobs share ONE launch (workgroups [0, tiles*batch) run C, the rest D): one kernel
// boundary less, and C's dependent-latency chain overlaps D's store traffic.
struct VtAssignLds {
  uint32_t slice[kVtTile];        // the tile's cposr; later key of the tile's j-th new voxel
  uint32_t st_info[kVtTile];      // compact start of the tile's j-th new voxel
  unsigned char st_kept[kVtTile];  // its kept count
  int s_inc[kVtRouteWaves], s_before[kVtRouteWaves], s_all[kVtRouteWaves];
};

// C: voxel id = number of first-point flags before the cell's first point.  One workgroup per tile; thread t
// owns the 8 consecutive points 8t .. 8t+7.  The voxels a tile opens have consecutive ids, so their rows of
// vinfo / coords / num_points / coors4 are staged in LDS and leave coalesced.
__device__ __forceinline__ void vt_assign_tile(
    VtAssignLds& L, int frame, int tile, const uint32_t* __restrict__ cposr,
    const unsigned short* __restrict__ pos16, const uint4* __restrict__ owner,
    const uint32_t* __restrict__ tilecnt, int tiles, int max_voxels, const VtGrid& g,
    uint2* __restrict__ vinfo, int* __restrict__ totals, int32_t* __restrict__ coords,
    int32_t* __restrict__ num_pts, int32_t* __restrict__ coors4) {
  const int64_t tbase = ((int64_t)frame * tiles + tile) * kVtTile;
  {
    const uint4* src = reinterpret_cast<const uint4*>(cposr + tbase);
    uint4* dst = reinterpret_cast<uint4*>(L.slice);
    for (int j = threadIdx.x; j < kVtTile / 4; j += kVtRouteThreads) dst[j] = src[j];
  }
  // first-point counts of the tiles before this one (and of all tiles -> totals)
  int before = 0, all = 0;
  for (int t = threadIdx.x; t < tiles; t += kVtRouteThreads) {
    const int v = (int)tilecnt[(int64_t)frame * tiles + t];
    all += v;
    if (t < tile) before += v;
  }
  const uint4 pw = *reinterpret_cast<const uint4*>(pos16 + tbase + (int64_t)threadIdx.x * 8);
  __syncthreads();
  const uint32_t pword[4] = {pw.x, pw.y, pw.z, pw.w};
  uint32_t flags = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const uint32_t pos = (pword[b >> 1] >> (16 * (b & 1))) & 0xFFFFu;
    if (pos != 0xFFFFu && (L.slice[pos] & kVtFirstBit)) flags |= 1u << b;
  }
  uint4 o[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    o[b] = make_uint4(0u, 0u, 0u, 0u);
    if (flags & (1u << b)) o[b] = owner[tbase + (int64_t)threadIdx.x * 8 + b];
  }
  const int mine = __popc(flags);
  int inc = mine;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int nb = __shfl_up(inc, d, kWave);
    if (lane_id() >= d) inc += nb;
    before += __shfl_xor(before, d, kWave);
    all += __shfl_xor(all, d, kWave);
  }
  if (lane_id() == kWave - 1) {
    L.s_inc[wave_id()] = inc;
    L.s_before[wave_id()] = before;
    L.s_all[wave_id()] = all;
  }
  __syncthreads();  // also: every thread is done reading `slice`
  int vid0 = 0, tot_all = 0, local = inc - mine, tile_new = 0;
#pragma unroll
  for (int w = 0; w < kVtRouteWaves; ++w) {
    vid0 += L.s_before[w];
    tot_all += L.s_all[w];
    tile_new += L.s_inc[w];
    if (w < wave_id()) local += L.s_inc[w];
  }
  if (tile == 0 && threadIdx.x == 0) totals[frame] = tot_all;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    if (flags & (1u << b)) {
      L.slice[local] = o[b].x;
      L.st_info[local] = o[b].y;
      L.st_kept[local] = (unsigned char)o[b].z;
      ++local;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < tile_new; j += kVtRouteThreads) {
    const int vid = vid0 + j;
    if (vid >= max_voxels) break;
    const int64_t row = (int64_t)frame * max_voxels + vid;
    const uint32_t key = L.slice[j], kept = L.st_kept[j];
    vinfo[row] = make_uint2(L.st_info[j], kept);
    const int cx = (int)(key % (uint32_t)g.gx);
    const uint32_t t = key / (uint32_t)g.gx;
    const int cy = (int)(t % (uint32_t)g.gy), cz = (int)(t / (uint32_t)g.gy);
    const VtInt3 c3{cz, cy, cx};  // coords (z, y, x)
    __builtin_memcpy(coords + row * 3, &c3, sizeof(c3));
    num_pts[row] = (int)kept;
    if (coors4) *reinterpret_cast<int4*>(coors4 + row * 4) = make_int4(frame, cz, cy, cx);
  }
}

// D: second pass over the points: a kept point is stored at its place in the compact payload array.  Thread t
// of the workgroup takes points t, t + 512, ... of the tile: fully coalesced loads; the destination comes from
// the tile's cposr slice (staged in LDS) through pos16; the stores are DIM*4-byte pieces that merge in L2.
template <int DIM>
__device__ __forceinline__ void vt_emit_tile(uint32_t* __restrict__ slice, int frame, int tile,
                                             const float* __restrict__ points, int64_t n, int dim_rt, int tiles,
                                             const uint32_t* __restrict__ cposr,
                                             const unsigned short* __restrict__ pos16, int64_t cap,
                                             float* __restrict__ compact) {
  const int dim = DIM > 0 ? DIM : dim_rt;
  const int64_t tbase = ((int64_t)frame * tiles + tile) * kVtTile;
  const float* pf = points + (int64_t)frame * n * dim;
  float* cf = compact + (int64_t)frame * cap * dim;
  const int64_t base_i = (int64_t)tile * kVtTile + threadIdx.x;
  {
    const uint4* src = reinterpret_cast<const uint4*>(cposr + tbase);
    uint4* dst = reinterpret_cast<uint4*>(slice);
    for (int j = threadIdx.x; j < kVtTile / 4; j += kVtRouteThreads) dst[j] = src[j];
  }
  uint32_t pos[kVtRounds];
#pragma unroll
  for (int r = 0; r < kVtRounds; ++r) pos[r] = pos16[tbase + threadIdx.x + r * kVtRouteThreads];
  if (DIM == 4 || DIM == 5) {
    vt_f32x4u a[kVtRounds];
    float e[kVtRounds];
#pragma unroll
    for (int r = 0; r < kVtRounds; ++r) {
      const int64_t i = base_i + r * kVtRouteThreads;
      a[r] = vt_f32x4u{0.f, 0.f, 0.f, 0.f};
      e[r] = 0.f;
      if (pos[r] != 0xFFFFu) {  // routed points lie inside the frame
        a[r] = *reinterpret_cast<const vt_f32x4u*>(pf + i * DIM);
        if (DIM == 5) e[r] = pf[i * DIM + 4];
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kVtRounds; ++r) {
      if (pos[r] == 0xFFFFu) continue;
      const uint32_t cp = slice[pos[r]] & ~kVtFirstBit;
      if (cp == kVtDropped) continue;
      float* d = cf + (int64_t)cp * DIM;
      *reinterpret_cast<vt_f32x4u*>(d) = a[r];
      if (DIM == 5) d[4] = e[r];
    }
  } else {
    __syncthreads();
    for (int r = 0; r < kVtRounds; ++r) {
      if (pos[r] == 0xFFFFu) continue;
      const uint32_t cp = slice[pos[r]] & ~kVtFirstBit;
      if (cp == kVtDropped) continue;
      const float* src = pf + (base_i + r * kVtRouteThreads) * dim;
      float* d = cf + (int64_t)cp * dim;
      for (int c = 0; c < dim; ++c) d[c] = src[c];
    }
  }
}

template <int DIM>
__global__ __launch_bounds__(kVtRouteThreads, 4) void vt_assign_emit_kernel(
    const float* __restrict__ points, int64_t n, int dim_rt, int tiles, int batch,
    const uint32_t* __restrict__ cposr, const unsigned short* __restrict__ pos16,
    const uint4* __restrict__ owner, const uint32_t* __restrict__ tilecnt, int max_voxels, VtGrid g,
    uint2* __restrict__ vinfo, int* __restrict__ totals, int32_t* __restrict__ coords,
    int32_t* __restrict__ num_pts, int32_t* __restrict__ coors4, int64_t cap, float* __restrict__ compact) {
  __shared__ VtAssignLds L;
  const uint32_t per_job = (uint32_t)tiles * (uint32_t)batch;
  int frame, tile;
  if (blockIdx.x < per_job) {
    vt_unit(blockIdx.x, (uint32_t)tiles, (uint32_t)batch, frame, tile);
    vt_assign_tile(L, frame, tile, cposr, pos16, owner, tilecnt, tiles, max_voxels, g, vinfo, totals, coords,
                   num_pts, coors4);
  } else {
    vt_unit(blockIdx.x - per_job, (uint32_t)tiles, (uint32_t)batch, frame, tile);
    vt_emit_tile<DIM>(L.slice, frame, tile, points, n, dim_rt, tiles, cposr, pos16, cap, compact);
  }
}

// ------------------------------------------------------------------------------------------------ E
// Output writer.  The frame's voxels tensor is one flat array of V * P * D floats; a lane owns VEC consecutive
// floats of it (VEC = 4 when a row is a whole number of float4: one 16-byte store).  Row v's valid floats are
// the run compact[start(v) * D ... + count(v) * D): one (unaligned) 16-byte load, everything behind is padding.
// vinfo rows of voxels that never came to life read (0, 0) (route kernel), so nothing here waits for the voxel
// count except the padding rows of coords / count / coors4 (the live ones were written by C).
constexpr int kVtRowsThreads = 256;
constexpr int kVtRowsIlp = 8;

template <int VEC>
__global__ __launch_bounds__(kVtRowsThreads) void vt_rows_kernel(
    const float* __restrict__ compact, int64_t cap, const uint2* __restrict__ vinfo,
    const int* __restrict__ totals, int batch, int units, int max_voxels, int rowq, int step_v, int step_j,
    int dim, float* __restrict__ voxels, int32_t* __restrict__ coords, int32_t* __restrict__ num_pts,
    int32_t* __restrict__ num_voxels, int32_t* __restrict__ coors4) {
  int frame, unit;
  vt_unit(blockIdx.x, (uint32_t)units, (uint32_t)batch, frame, unit);
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
  // the rows this workgroup touches are consecutive: their vinfo entries come in through LDS, one coalesced load
  __shared__ uint2 st_vi[kVtRowsIlp * kVtRowsThreads + 2];
  const uint32_t qb = (uint32_t)unit * (kVtRowsIlp * kVtRowsThreads);
  const uint32_t vb = vt_div(min(qb, total_q), (uint32_t)rowq, 1.0f / (float)rowq);
  const uint32_t ve = vt_div(min(qb + kVtRowsIlp * kVtRowsThreads - 1u, total_q), (uint32_t)rowq, 1.0f / (float)rowq);
  for (uint32_t r = threadIdx.x; r <= ve - vb; r += kVtRowsThreads)
    st_vi[r] = vb + r < (uint32_t)max_voxels ? vi[vb + r] : make_uint2(0u, 0u);
  __syncthreads();
  uint2 info[kVtRowsIlp];
#pragma unroll
  for (int u = 0; u < kVtRowsIlp; ++u) info[u] = st_vi[min(v[u], ve) - vb];
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
  const int nv = min(totals[frame], max_voxels);
  if (unit == 0 && threadIdx.x == 0) num_voxels[frame] = nv;
#pragma unroll
  for (int u = 0; u < kVtRowsIlp; ++u) {
    const uint32_t q = q0 + (uint32_t)u * kVtRowsThreads;
    if (q >= total_q) break;
    float* dst = vf + (int64_t)q * VEC;
    if (VEC == 4) {
      *reinterpret_cast<float4*>(dst) = make_float4(val[u][0], val[u][1], val[u][2], val[u][3]);
    } else {
#pragma unroll
      for (int c = 0; c < VEC; ++c) dst[c] = val[u][c];
    }
    if (j[u] == 0 && (int)v[u] >= nv) {  // padding rows of coords / count / coors4 (batch = -1: coors_pad)
      const int64_t row = (int64_t)frame * max_voxels + v[u];
      const VtInt3 z3{0, 0, 0};
      __builtin_memcpy(coords + row * 3, &z3, sizeof(z3));
      num_pts[row] = 0;
      if (coors4) *reinterpret_cast<int4*>(coors4 + row * 4) = make_int4(-1, 0, 0, 0);
    }
  }
}

}  // namespace pd3
