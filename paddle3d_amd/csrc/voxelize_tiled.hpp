// hard_voxelize fast path for BEV-sized grids ("tiled" path): no global sort, no global gather, four launches.
//
// Same order-independent restatement of the reference's sequential scan as voxelize.hip
// (voxel id of a cell = rank of its first point among all first points; slot of a point = number of earlier
// points in its cell).  What this machine charges for is memory TRANSACTIONS, not bytes (tools/hwcheck/
// memrates.hip on MI355X: ~190 G scattered 4-byte stores/s and ~60 G scattered 20-byte stores/s chip-wide,
// i.e. 25 / 36 us for one access per kept point of a 16-frame batch, against 15 us to stream the batch), so
// every array is laid out such that it is read and written in coalesced runs; the only scattered accesses left
// are the 20-byte payload store of the kept points (D) and one 4-byte record read per occupied cell (C).
//
//   A  route_kernel   (tile of 4096 consecutive points, 512 threads)
//        point -> cell key -> (group, cell-in-group); a group is a diagonal set of 2^LOW (<= 4096) cells
//        (see kVtSkew).  The tile's 4-byte records are sorted BY GROUP, stable in point order, in LDS and
//        written as one coalesced 16 KB slice, plus the directory row dir[tile][group] = (offset, count) and
//        pos16[point] = the point's position in the slice.  The in-order rank of a point among the wave's
//        points of the same group is the value a returning LDS atomic add hands back (lanes of one ds_add_rtn
//        are served in ascending lane order, instructions of a wave in order -- checked on gfx950 by
//        tools/hwcheck/lds_atomic_order.hip and, implicitly, by every bit-exact test).
//   B  group_kernel   (one 256-thread workgroup per group)
//        walks the directory column of its group in tile order -> its points in INPUT ORDER as contiguous runs
//        of the routed slices (a nuScenes run is ~58 records: one wave-wide load).  Slot of a point = earlier
//        points of its cell (LDS counters, see the kernel).  Then a scan over the group's cells of
//        min(count, P) places every cell in the group's region of the compact payload array (the region is
//        sized by the group's record count, so its start is the sum of the group's offsets inside the tiles'
//        slices: no global counter), and every record gets cposr[routed position] = compact position (or
//        "dropped"); a cell's first point also carries the cell's kept count and bumps its tile's first-point
//        counter.  All global accesses are coalesced runs.
//   C  assign job     (tile)  prefix over the first-point flags in point order = voxel id (the reference's
//        hand-out order); the cell of a first point = (group owning its slice position, cell field of its
//        routed record); writes vinfo[voxel] = (compact start, count) and the voxel's coords / count rows
//        (staged in LDS, coalesced).
//   D  emit job       (tile)  streams the points a second time (coalesced; still in the Infinity Cache) and
//        stores each kept point at compact[cposr[pos16[point]]]: the payload grouped by cell (the 20-byte
//        stores merge in L2).  Independent of C; C and D share one launch.
//   E  rows_kernel    voxel-parallel, one 16-byte store per lane: a row's valid floats are one contiguous run
//        of the compact array; the complete fixed-shape voxels tensor (rows and zero padding) and the padding
//        of the coords / count rows are written exactly once (streaming stores: nothing re-reads them here).
// The library's default is the GATHER FORM of the same pipeline (path 3): B also leaves clist[compact place] =
// point index (the list the compact array would have been filled from), D is not launched at all, and E'
// (rows_gather_kernel) reads a row's points straight from the input through clist -- no compact payload array,
// no second pass over the points: 163 -> 135 us per 16 nuScenes frames.  Path 2 keeps D + E as described.
//
// HBM traffic per frame: points read (A) and re-read (D), outputs written once (E); everything between is a few
// MB of scratch.  Workgroups are mapped XCD-aware (vt_unit): with batch % 8 == 0 every frame's workgroups of
// every kernel run on one XCD, so the small stores of a frame merge in that XCD's L2.
// Preconditions (else the generic sort path of voxelize.hip runs): cells <= 2^22 with <= 1024 groups,
// N <= 2^22 - 3, max points per voxel <= 254.
#pragma once
#include "common.hpp"

#include <algorithm>
#include <cmath>

namespace pd3 {

constexpr int kVtTile = 4096;
constexpr int kVtRouteThreads = 512;
constexpr int kVtRounds = kVtTile / kVtRouteThreads;  // 8
constexpr int kVtRouteWaves = kVtRouteThreads / kWave;
constexpr int kVtMaxLow = 12;      // cells per group <= 4096: a tile's run of one group is ~a wave of records
constexpr int kVtMaxGbits = 10;    // groups <= 1024: two per thread in the route kernel's scan
constexpr int kVtMaxTiles = 1024;
constexpr int kVtMaxPts = 254;
// cposr word of a routed record: bits 0..21 compact position (all ones: the point is not stored), and for the
// first point of a cell bit 31 set and bits 22..29 = points kept for the cell
constexpr uint32_t kVtCpMask = 0x3FFFFFu;
constexpr uint32_t kVtDropped = kVtCpMask;
constexpr int kVtKeptShift = 22;
constexpr uint32_t kVtFirstBit = 0x80000000u;

struct VtGrid {  // mirror of VoxGrid (kept separate so this header stands alone)
  float min_x, min_y, min_z, size_x, size_y, size_z;
  float inv_x, inv_y, inv_z;  // fp32(1 / size): the fast path of vt_axis_cell
  int gx, gy, gz;
  uint32_t ncells;
  // pillar grids (gz == 1): z - min_z lies in cell 0 exactly when it lies in [z1_lo, z1_hi], the fp32 values whose
  // correctly rounded quotient by size_z floors to 0 (found on the host with the same fp32 divide, vt_single_cell_bounds);
  // z1_lo > z1_hi: not used
  float z1_lo = 0.f, z1_hi = -1.f;
};

// [lo, hi] = { t : floor(RN(t / size)) == 0 } for fp32 t, by walking the neighbours of -0 and of `size` with the
// host's correctly rounded fp32 divide (the operation the reference performs, voxelize_op.cc:43-45)
static inline bool vt_single_cell_bounds(float size, float& lo, float& hi) {
  if (!(size > 0.f) || !std::isfinite(size)) return false;
  volatile float s = size;
  float h = size;
  int it = 0;
  while (!((float)(h / s) < 1.0f)) {
    h = std::nextafterf(h, -INFINITY);
    if (++it > 64) return false;
  }
  float l = -0.0f;
  for (it = 0; it < 64; ++it) {
    const float n = std::nextafterf(l, -INFINITY);
    if ((float)(n / s) == 0.0f) l = n;
    else break;
  }
  if (it == 64) return false;
  lo = l;
  hi = h;
  return true;
}

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
  p.low = std::max(std::min(bits - 2, kVtMaxLow), 0);
  while (p.low > 0 && n >= ((int64_t)1 << (32 - p.low)) - 1) --p.low;  // a record = (point index << low) | cell
  p.gbits = std::max(bits - p.low, 2);
  p.groups = 1 << p.gbits;
  p.cpg = 1 << p.low;
  p.tiles = (int)ceil_div(n, kVtTile);
  p.ok = p.gbits <= kVtMaxGbits && p.tiles <= kVtMaxTiles && n < (int64_t)kVtCpMask - 1 &&
         max_pts <= kVtMaxPts;
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
// ~ 1/r); the skew spreads every dense neighbourhood over all groups.  (group, local) <-> key is a bijection for any
// skew.  Which one: counted on 16 synthetic nuScenes sweeps, the fullest group holds 2.0x the mean with skew 7 (round 2)
// and 1.52-1.56x with the best odd skews (37, 25, 95 ...; the floor: single cells hold up to 70 % of a group's mean);
// 37 measured 1.8 us per 16 frames faster than 7 on one box (tools/prof/prof_vox_ab.py).  -DPD3_VT_SKEW=k builds another.
#ifndef PD3_VT_SKEW
#define PD3_VT_SKEW 37
#endif
constexpr uint32_t kVtSkew = PD3_VT_SKEW;

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
  int cx, cy, cz = 0;
  if (!(vt_axis_cell(x, g.min_x, g.size_x, g.inv_x, g.gx, cx) &&
        vt_axis_cell(y, g.min_y, g.size_y, g.inv_y, g.gy, cy)))
    return false;
  if (g.z1_hi >= g.z1_lo) {  // one cell along z: two comparisons instead of the exact cell index
    const float t = z - g.min_z;
    if (!(t >= g.z1_lo && t <= g.z1_hi)) return false;  // also false for NaN
  } else if (!vt_axis_cell(z, g.min_z, g.size_z, g.inv_z, g.gz, cz)) {
    return false;
  }
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
constexpr int kVgThreads = 256;
constexpr int kVgWaves = kVgThreads / kWave;        // 4
constexpr int kVgSteps = 40;                        // 64-record steps per wave per pass (<= 64: step facts
                                                    // live one per lane)
constexpr int kVgChunk = 4;                         // steps whose LDS traffic is issued back to back
constexpr int kVgPassSteps = kVgWaves * kVgSteps;   // 160

static inline size_t vt_group_lds(int cpg, int tiles) {
  const size_t cw = (size_t)std::max(cpg >> 1, 1), bw = (size_t)std::max(cpg >> 2, 1);
  return kVgWaves * cw * 4 + bw * 4 + (size_t)kVgPassSteps * 8 + (size_t)(3 * tiles + 1) * 4 +
         (kVgWaves + 2) * 4 + 16 + (size_t)kVgThreads * 4;
}

// One 256-thread workgroup per group (<= 4096 cells; a nuScenes group receives ~4200 records, ~58 from each
// tile's slice).  The group's record stream in INPUT ORDER is the concatenation, in tile order, of its runs in
// the routed slices; it is cut into STEPS of <= 64 records of one run (so every load and store of a step is one
// contiguous piece), and a PASS of <= 160 steps is dealt to the four waves in contiguous, balanced shares.
// Slot of a record = records of its cell in earlier passes (`base`, one saturating byte per cell: only
//                    min(count, P) matters)
//                  + in earlier waves of this pass (prefix over the per-wave count tables)
//                  + earlier in this wave (what the returning LDS atomic add hands back).
// A group that fits one pass (the normal case) keeps records and slots in registers; longer streams are walked
// twice (count, then place), cell starts parked in `gstart`.  Every global access is a coalesced run; there is
// no scattered access in this kernel.  The LDS work is latency-bound (one wave per SIMD when the batch is
// small), so it is written branch-free in chunks of 8 steps whose LDS operations are all in flight together.
__global__ __launch_bounds__(kVgThreads, 4) void vt_group_kernel(
    const uint32_t* __restrict__ recs, const uint32_t* __restrict__ dir, int low, int gbits, int tiles,
    int batch, int max_pts, uint32_t* __restrict__ cposr, uint32_t* __restrict__ gstart,
    uint32_t* __restrict__ tilecnt, uint32_t* __restrict__ clist, int64_t cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
  const int cpg = 1 << low, groups = 1 << gbits;
  const int cw = max(cpg >> 1, 1);                                   // two 16-bit counters per word
  const int bw = max(cpg >> 2, 1);                                   // four byte counters per word
  uint32_t* cntw = reinterpret_cast<uint32_t*>(vt_smem);             // [waves][cw]; later the cell start table
  uint32_t* base = cntw + kVgWaves * cw;                             // [bw] records so far, a byte per cell
  uint32_t* stepsrc = base + bw;                                     // [pass] routed position of the step
  uint32_t* stepcnt = stepsrc + kVgPassSteps;                        // [pass] records of the step
  uint32_t* tdir = stepcnt + kVgPassSteps;                           // [tiles] directory entries of the group
  uint32_t* tstep = tdir + tiles;                                    // [tiles + 1] exclusive prefix of steps
  uint32_t* tfirst = tstep + tiles + 1;                              // [tiles] first points seen per tile
  int* scan_tmp = reinterpret_cast<int*>(tfirst + tiles);            // [waves + 2]
  uint32_t* sink = reinterpret_cast<uint32_t*>(scan_tmp + kVgWaves + 2 + 4) + threadIdx.x;  // a word of its own per lane
  int frame, grp;
  vt_unit(blockIdx.x, (uint32_t)groups, (uint32_t)batch, frame, grp);
  const int lane = lane_id(), wave = wave_id();
#ifdef PD3_VT_TRACE
  long long tr[12];
  int trn = 0;
#define PD3_MARK() tr[trn++] = __builtin_readcyclecounter()
#else
#define PD3_MARK()
#endif
  PD3_MARK();

  // directory column: thread t takes the tiles t*tpt .. (steps scanned across the workgroup)
  const uint32_t* dcol = dir + (int64_t)frame * tiles * groups + grp;
  const int tpt = (int)ceil_div(tiles, kVgThreads);
  int my_steps = 0, my_off = 0;
  if (threadIdx.x == 0) scan_tmp[kVgWaves + 1] = 0;
  for (int k = 0; k < tpt; ++k) {
    const int t = (int)threadIdx.x * tpt + k;
    if (t < tiles) {
      const uint32_t d = dcol[(int64_t)t * groups];
      tdir[t] = d;
      tfirst[t] = 0u;
      my_steps += (int)((d >> 16) + 63u) >> 6;
      my_off += (int)(d & 0xFFFFu);
    }
  }
  for (int i = threadIdx.x; i < bw; i += kVgThreads) base[i] = 0u;
  int total_steps;
  int at_step = block_exclusive_scan<kVgThreads>(my_steps, scan_tmp, total_steps);
  for (int k = 0; k < tpt; ++k) {
    const int t = (int)threadIdx.x * tpt + k;
    if (t < tiles) {
      tstep[t] = (uint32_t)at_step;
      at_step += (int)((tdir[t] >> 16) + 63u) >> 6;
    }
  }
  if (threadIdx.x == 0) tstep[tiles] = (uint32_t)total_steps;
  // the group's region of the compact array is sized by its record count: it starts where the records of the
  // groups before it would end = the sum over the tiles of this group's offset inside the tile's slice.
  // (Packing the regions densely -- exclusive scan of the groups' kept totals, applied per point in the emit
  // job -- was tried: 2.7 instead of 5.4 MB per frame, but the lookups cost more than the locality gives.)
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) my_off += __shfl_xor(my_off, d, kWave);
  if (lane == 0 && my_off) atomicAdd(&scan_tmp[kVgWaves + 1], my_off);
  __syncthreads();
  PD3_MARK();
  if (total_steps == 0) return;
  const uint32_t region = (uint32_t)scan_tmp[kVgWaves + 1];

  const int64_t stride = (int64_t)tiles * kVtTile;
  const uint32_t* rf = recs + (int64_t)frame * stride;
  uint32_t* cpos_f = cposr + (int64_t)frame * stride;
  uint32_t* clist_f = clist ? clist + (int64_t)frame * cap : nullptr;  // gather form: point index per compact place
  uint32_t* gs = gstart + ((int64_t)frame * groups + grp) * cpg;
  const uint32_t cell_mask = (uint32_t)cpg - 1u;
  const int npass = (int)ceil_div(total_steps, kVgPassSteps);
  uint32_t* cnt_mine = cntw + wave * cw;

  // per record ONE register: bits 0..11 cell in group, bits 12..27 slot (<= 2048 per pass + 255), bit 31 = the
  // lane holds no record (its cell field then names some cell of the group, a harmless LDS address)
  uint32_t rec[kVgSteps];
  constexpr uint32_t kNoRec = 0x80000000u;
  int s_n = 0;                       // steps of this wave in the current pass
  uint32_t my_src = 0, my_cnt = 0;   // lane u: routed position / record count of the wave's step u

  // Sweep 0 counts (and, when the group is one pass, ranks); the cells are then placed; sweep 1 writes the cposr
  // words -- from the registers of sweep 0 when the group is one pass, else by walking the passes again.
  const bool one_pass = npass == 1;
  for (int sweep = 0; sweep < 2; ++sweep) {
    for (int p = 0; p < npass; ++p) {
      if (!(sweep == 1 && one_pass)) {
        __syncthreads();  // the previous pass is done with the step table and the count tables
        const int s0 = p * kVgPassSteps, s1 = min(s0 + kVgPassSteps, total_steps);
        for (int t = threadIdx.x; t < tiles; t += kVgThreads) {
          const uint32_t d = tdir[t];
          const int a = (int)tstep[t], b = (int)tstep[t + 1];
          for (int s = max(a, s0); s < min(b, s1); ++s) {
            const uint32_t chunk = (uint32_t)(s - a);
            stepsrc[s - s0] = (uint32_t)t * kVtTile + (d & 0xFFFFu) + chunk * kWave;
            stepcnt[s - s0] = min((d >> 16) - chunk * kWave, (uint32_t)kWave);
          }
        }
        for (int i = threadIdx.x; i < kVgWaves * cw; i += kVgThreads) cntw[i] = 0u;
        __syncthreads();
        {
          const int spw = (int)ceil_div(s1 - s0, kVgWaves);
          const int s_lo = min(wave * spw, s1 - s0);
          s_n = min(spw, s1 - s0 - s_lo);
          my_src = lane < s_n ? stepsrc[s_lo + lane] : 0u;
          my_cnt = lane < s_n ? stepcnt[s_lo + lane] : 0u;
        }
        // every lane loads (lanes past the end of a step re-read its first record): no divergent branch, so
        // the loads of all steps are in flight together
#pragma unroll
        for (int c0 = 0; c0 < kVgSteps; c0 += kVgChunk) {
#pragma unroll
          for (int k = 0; k < kVgChunk; ++k) rec[c0 + k] = 0u;
          if (c0 < s_n) {
#pragma unroll
            for (int k = 0; k < kVgChunk; ++k) {
              const int c = __builtin_amdgcn_readlane((int)my_cnt, c0 + k);
              const uint32_t src = (uint32_t)__builtin_amdgcn_readlane((int)my_src, c0 + k);
              rec[c0 + k] = (rf + src)[lane < c ? lane : 0];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < kVgSteps; ++u) {  // (the loaded word is used either way: it stays a plain load)
          const int c = __builtin_amdgcn_readlane((int)my_cnt, u);
          rec[u] = (rec[u] & cell_mask) | (lane < c ? 0u : kNoRec);
        }
        PD3_MARK();
#pragma unroll
        for (int c0 = 0; c0 < kVgSteps; c0 += kVgChunk) {
          if (c0 < s_n) {
            // One returning LDS add per RUN of equal cells, not per record: firing-ordered points put runs of
            // neighbouring lanes into one cell, and lanes that hit one address are served one after the other
            // (tools/hwcheck/lds_atomic_rate: 4.8 G instr/s chip-wide for one cell against 64 G for distinct ones).
            // The head lane of a run adds the run's length and hands the returned count to its followers, who add
            // their distance from the head; lanes without a record and followers add 0 to a word of their own.  A
            // cell that comes back later in the step is a second run: two adds to one address, served in lane order
            // like before.
            uint32_t old[kVgChunk], behind[kVgChunk];
            int headl[kVgChunk];
#pragma unroll
            for (int k = 0; k < kVgChunk; ++k) {
              const uint32_t r = rec[c0 + k], cell = r & cell_mask;
              const bool valid = !(r & kNoRec);
              const uint32_t prev = (uint32_t)__shfl_up((int)cell, 1, kWave);  // records of a step are lanes 0 .. c-1
              const bool head = valid && (lane == 0 || prev != cell);
              const unsigned long long hm = __ballot(head);
              const unsigned long long upto = hm & ((2ull << lane) - 1ull);        // heads at or before this lane
              headl[k] = valid ? 63 - __builtin_clzll(upto | 1ull) : lane;
              const unsigned long long after = hm & ~((2ull << lane) - 1ull);      // heads behind this lane
              const int cnt_step = __builtin_amdgcn_readlane((int)my_cnt, c0 + k);
              const int next = after ? __builtin_ctzll(after) : cnt_step;
              behind[k] = (uint32_t)(lane - headl[k]);
              uint32_t* addr = head ? &cnt_mine[cell >> 1] : sink;
              old[k] = atomicAdd(addr, head ? (uint32_t)(next - lane) << ((cell & 1u) * 16u) : 0u);
            }
#pragma unroll
            for (int k = 0; k < kVgChunk; ++k) {
              const uint32_t from_head = (uint32_t)__shfl((int)old[k], headl[k], kWave);
              rec[c0 + k] |= ((((from_head >> ((rec[c0 + k] & cell_mask & 1u) * 16u)) & 0xFFFFu) + behind[k]) & 0xFFFFu) << 12;
            }
          }
        }
        PD3_MARK();
        __syncthreads();
        // per cell: counts of the four waves -> each wave's base (carry from earlier passes included).  A
        // thread takes base words j, j + 256, ...: four cells = two words of every wave's table.
        for (int j0 = threadIdx.x; j0 < bw; j0 += 2 * kVgThreads) {
          uint32_t v[2][kVgWaves][2], bb[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int j = min(j0 + q * kVgThreads, bw - 1);
            bb[q] = base[j];
#pragma unroll
            for (int w = 0; w < kVgWaves; ++w) {
              v[q][w][0] = cntw[w * cw + min(2 * j, cw - 1)];
              v[q][w][1] = cntw[w * cw + min(2 * j + 1, cw - 1)];
            }
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int j = j0 + q * kVgThreads;
            if (j >= bw) break;
            uint32_t b0 = bb[q] & 0xFFu, b1 = (bb[q] >> 8) & 0xFFu, b2 = (bb[q] >> 16) & 0xFFu, b3 = bb[q] >> 24;
#pragma unroll
            for (int w = 0; w < kVgWaves; ++w) {
              cntw[w * cw + 2 * j] = b0 | (b1 << 16);
              if (2 * j + 1 < cw) cntw[w * cw + 2 * j + 1] = b2 | (b3 << 16);
              b0 = min(b0 + (v[q][w][0] & 0xFFFFu), 255u);
              b1 = min(b1 + (v[q][w][0] >> 16), 255u);
              if (2 * j + 1 < cw) {
                b2 = min(b2 + (v[q][w][1] & 0xFFFFu), 255u);
                b3 = min(b3 + (v[q][w][1] >> 16), 255u);
              }
            }
            base[j] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
          }
        }
        if (sweep == 1 || one_pass) {
          __syncthreads();
          PD3_MARK();
#pragma unroll
          for (int c0 = 0; c0 < kVgSteps; c0 += kVgChunk) {
            if (c0 < s_n) {
              uint32_t b[kVgChunk];
#pragma unroll
              for (int k = 0; k < kVgChunk; ++k) b[k] = cnt_mine[(rec[c0 + k] & cell_mask) >> 1];
#pragma unroll
              for (int k = 0; k < kVgChunk; ++k)  // a saturated base (255) is past every limit (P <= 254)
                rec[c0 + k] += ((b[k] >> ((rec[c0 + k] & cell_mask & 1u) * 16u)) & 0xFFFFu) << 12;
            }
          }
        }
      }
      if (sweep == 1) {
        // cposr words of the pass; cell starts (kept << 24 | start) come from the table in LDS (one pass) or gs
        uint32_t firsts = 0;  // lane u: first points seen in the wave's step u
#pragma unroll
        for (int c0 = 0; c0 < kVgSteps; c0 += kVgChunk) {
          if (c0 < s_n) {
            uint32_t packed[kVgChunk], full[kVgChunk];
#pragma unroll
            for (int k = 0; k < kVgChunk; ++k) full[k] = 0u;
            if (clist_f) {  // the records again (coalesced, L2): their upper bits are the point indices
#pragma unroll
              for (int k = 0; k < kVgChunk; ++k) {
                const int c = __builtin_amdgcn_readlane((int)my_cnt, c0 + k);
                const uint32_t src = (uint32_t)__builtin_amdgcn_readlane((int)my_src, c0 + k);
                full[k] = (rf + src)[lane < c ? lane : 0];
              }
            }
            if (one_pass) {
#pragma unroll
              for (int k = 0; k < kVgChunk; ++k) packed[k] = cntw[rec[c0 + k] & cell_mask];
            } else {
#pragma unroll
              for (int k = 0; k < kVgChunk; ++k) packed[k] = gs[rec[c0 + k] & cell_mask];
            }
#pragma unroll
            for (int k = 0; k < kVgChunk; ++k) {
              const int u = c0 + k;
              const bool valid = !(rec[u] & kNoRec);
              const uint32_t slot = (rec[u] >> 12) & 0xFFFFu;
              const bool first = valid && slot == 0u;
              const uint32_t src = (uint32_t)__builtin_amdgcn_readlane((int)my_src, u);
              uint32_t w = slot < (uint32_t)max_pts ? region + (packed[k] & 0xFFFFFFu) + slot : kVtDropped;
              if (first) w |= kVtFirstBit | ((packed[k] >> 24) << kVtKeptShift);
              if (valid) (cpos_f + src)[lane] = w;
              // gather form: the list the row writer walks (a 4-byte store inside the group's ~17 KB region)
              if (clist_f && valid && slot < (uint32_t)max_pts) clist_f[w & kVtCpMask] = full[k] >> low;
              const uint32_t nfirst = (uint32_t)__popcll(__ballot(first));
              firsts = lane == u ? nfirst : firsts;
            }
          }
        }
        if (lane < s_n && firsts) atomicAdd(&tfirst[my_src / kVtTile], firsts);
      }
    }
    if (sweep == 0) {
      // cells -> places in the group's region.  Any order of the cells will do (the array is scratch): thread
      // t takes the cells of base words t, t + 256, ...  Entry = (kept << 24) | start.  With one pass the count
      // tables are dead and the start table takes their place; else it goes to gs and `base` starts over.
      PD3_MARK();
      __syncthreads();
      uint32_t mine = 0;
      for (int j = threadIdx.x; j < bw; j += kVgThreads) {
        const uint32_t b = base[j];
        mine += min(b & 0xFFu, (uint32_t)max_pts) + min((b >> 8) & 0xFFu, (uint32_t)max_pts) +
                min((b >> 16) & 0xFFu, (uint32_t)max_pts) + min(b >> 24, (uint32_t)max_pts);
      }
      int all;
      uint32_t at = (uint32_t)block_exclusive_scan<kVgThreads>((int)mine, scan_tmp, all);
      for (int j = threadIdx.x; j < bw; j += kVgThreads) {
        const uint32_t b = base[j];
        uint32_t e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t kept = min((b >> (8 * q)) & 0xFFu, (uint32_t)max_pts);
          e[q] = (kept << 24) | at;
          at += kept;
        }
        if (one_pass) {
          if (cpg >= 4) {
            *reinterpret_cast<uint4*>(cntw + 4 * j) = make_uint4(e[0], e[1], e[2], e[3]);
          } else {
            for (int q = 0; q < cpg; ++q) cntw[q] = e[q];
          }
        } else {
          for (int q = 0; q < min(cpg, 4); ++q) gs[4 * j + q] = e[q];
          base[j] = 0u;
        }
      }
      __threadfence_block();
      __syncthreads();
      PD3_MARK();
    }
  }
  PD3_MARK();
  __syncthreads();
  for (int t = threadIdx.x; t < tiles; t += kVgThreads)
    if (tfirst[t]) atomicAdd(&tilecnt[(int64_t)frame * tiles + t], tfirst[t]);
#ifdef PD3_VT_TRACE
  PD3_MARK();
  if (lane == 0 && wave == 0 && gridDim.x <= 64)
    printf("WG %d steps %d npass %d total %lld\n", (int)blockIdx.x, total_steps, npass, tr[trn - 1] - tr[0]);
  if ((blockIdx.x == 0 || blockIdx.x == 37) && lane == 0)
    printf("wg %d wave %d steps %d npass %d: %lld %lld %lld %lld %lld %lld %lld %lld %lld\n", (int)blockIdx.x, wave,
           total_steps, npass, tr[1] - tr[0], tr[2] - tr[1], tr[3] - tr[2], tr[4] - tr[3], tr[5] - tr[4],
           tr[6] - tr[5], tr[7] - tr[6], tr[8] - tr[7], 0ll);
#endif
#undef PD3_MARK
}

// ------------------------------------------------------------------------------------------------ C + D
// Two independent per-tile jobs share ONE launch (workgroups [0, tiles*batch) run C, the rest D): one kernel
// boundary less, and C's dependent-latency chain overlaps D's store traffic.
struct VtAssignLds {
  uint32_t slice[kVtTile];        // the tile's cposr; later key of the tile's j-th new voxel
  uint32_t st_info[kVtTile];      // compact start of the tile's j-th new voxel
  unsigned char st_kept[kVtTile];  // its kept count
  unsigned short goff[1 << kVtMaxGbits];  // the tile's directory row: start of every group's run in the slice
  int s_inc[kVtRouteWaves], s_before[kVtRouteWaves], s_all[kVtRouteWaves];
};

// Both jobs start from the tile's cposr slice and directory row in LDS.  Ends with a barrier.
__device__ __forceinline__ void vt_load_slice(VtAssignLds& L, int frame, int tile, int tiles, int gbits,
                                              const uint32_t* __restrict__ cposr,
                                              const uint32_t* __restrict__ dir) {
  const int groups = 1 << gbits;
  const int64_t tbase = ((int64_t)frame * tiles + tile) * kVtTile;
  const uint4* src = reinterpret_cast<const uint4*>(cposr + tbase);
  uint4* dst = reinterpret_cast<uint4*>(L.slice);
  for (int j = threadIdx.x; j < kVtTile / 4; j += kVtRouteThreads) dst[j] = src[j];
  const uint32_t* drow = dir + ((int64_t)frame * tiles + tile) * groups;
  for (int j = threadIdx.x; j < groups; j += kVtRouteThreads) L.goff[j] = (unsigned short)drow[j];
  __syncthreads();
}

// the last group whose run in the slice starts at or before pos (empty groups share their successor's start)
__device__ __forceinline__ uint32_t vt_group_of(const VtAssignLds& L, uint32_t pos, int gbits) {
  uint32_t grp = 0;
  for (int step = 1 << (gbits - 1); step > 0; step >>= 1)
    if ((uint32_t)L.goff[grp + step] <= pos) grp += step;
  return grp;
}

// C: voxel id = number of first-point flags before the cell's first point.  One workgroup per tile; thread t
// owns the 8 consecutive points 8t .. 8t+7.  The voxels a tile opens have consecutive ids, so their rows of
// vinfo / coords / num_points / coors4 are staged in LDS and leave coalesced.
__device__ __forceinline__ void vt_assign_tile(
    VtAssignLds& L, int frame, int tile, const uint32_t* __restrict__ cposr,
    const unsigned short* __restrict__ pos16, const uint32_t* __restrict__ recs,
    const uint32_t* __restrict__ dir, int low, int gbits,
    const uint32_t* __restrict__ tilecnt, int tiles, int max_voxels, const VtGrid& g,
    uint2* __restrict__ vinfo, int* __restrict__ totals, int32_t* __restrict__ coords,
    int32_t* __restrict__ num_pts, int32_t* __restrict__ coors4) {
  const int64_t tbase = ((int64_t)frame * tiles + tile) * kVtTile;
  // first-point counts of the tiles before this one (and of all tiles -> totals)
  int before = 0, all = 0;
  for (int t = threadIdx.x; t < tiles; t += kVtRouteThreads) {
    const int v = (int)tilecnt[(int64_t)frame * tiles + t];
    all += v;
    if (t < tile) before += v;
  }
  const uint4 pw = *reinterpret_cast<const uint4*>(pos16 + tbase + (int64_t)threadIdx.x * 8);
  vt_load_slice(L, frame, tile, tiles, gbits, cposr, dir);
  const uint32_t pword[4] = {pw.x, pw.y, pw.z, pw.w};
  // a first point's cposr word carries (compact start, kept count); its cell = (group, cell in group): the
  // group is the run of the slice its position lies in, the cell in the group sits in its routed record
  uint32_t flags = 0;
  uint32_t o_word[8], o_key[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const uint32_t pos = (pword[b >> 1] >> (16 * (b & 1))) & 0xFFFFu;
    o_word[b] = 0u;
    o_key[b] = 0u;
    if (pos != 0xFFFFu && (L.slice[pos] & kVtFirstBit)) {
      flags |= 1u << b;
      o_key[b] = recs[tbase + pos];
      const uint32_t grp = vt_group_of(L, pos, gbits);
      o_word[b] = L.slice[pos];
      o_key[b] = vt_group_to_key(grp, o_key[b] & ((1u << low) - 1u), gbits);
    }
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
      L.slice[local] = o_key[b];
      L.st_info[local] = o_word[b] & kVtCpMask;
      L.st_kept[local] = (unsigned char)((o_word[b] >> kVtKeptShift) & 0xFFu);
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
__device__ __forceinline__ void vt_emit_tile(VtAssignLds& L, int frame, int tile,
                                             const float* __restrict__ points, int64_t n, int dim_rt, int tiles,
                                             int gbits, const uint32_t* __restrict__ cposr,
                                             const uint32_t* __restrict__ dir,
                                             const unsigned short* __restrict__ pos16, int64_t cap,
                                             float* __restrict__ compact) {
  const uint32_t* slice = L.slice;
  const int dim = DIM > 0 ? DIM : dim_rt;
  const int64_t tbase = ((int64_t)frame * tiles + tile) * kVtTile;
  const float* pf = points + (int64_t)frame * n * dim;
  float* cf = compact + (int64_t)frame * cap * dim;
  const int64_t base_i = (int64_t)tile * kVtTile + threadIdx.x;
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
    vt_load_slice(L, frame, tile, tiles, gbits, cposr, dir);
#pragma unroll
    for (int r = 0; r < kVtRounds; ++r) {
      if (pos[r] == 0xFFFFu) continue;
      const uint32_t cp = slice[pos[r]] & kVtCpMask;
      if (cp == kVtDropped) continue;
      float* d = cf + (int64_t)cp * DIM;
      *reinterpret_cast<vt_f32x4u*>(d) = a[r];
      if (DIM == 5) d[4] = e[r];
    }
  } else {
    vt_load_slice(L, frame, tile, tiles, gbits, cposr, dir);
    for (int r = 0; r < kVtRounds; ++r) {
      if (pos[r] == 0xFFFFu) continue;
      const uint32_t cp = slice[pos[r]] & kVtCpMask;
      if (cp == kVtDropped) continue;
      const float* src = pf + (base_i + r * kVtRouteThreads) * dim;
      float* d = cf + (int64_t)cp * dim;
      for (int c = 0; c < dim; ++c) d[c] = src[c];
    }
  }
}

template <int DIM>
__global__ __launch_bounds__(kVtRouteThreads, 8) void vt_assign_emit_kernel(
    const float* __restrict__ points, int64_t n, int dim_rt, int tiles, int batch,
    const uint32_t* __restrict__ cposr, const unsigned short* __restrict__ pos16,
    const uint32_t* __restrict__ recs, const uint32_t* __restrict__ dir, int low, int gbits,
    const uint32_t* __restrict__ tilecnt, int max_voxels, VtGrid g,
    uint2* __restrict__ vinfo, int* __restrict__ totals, int32_t* __restrict__ coords,
    int32_t* __restrict__ num_pts, int32_t* __restrict__ coors4, int64_t cap, float* __restrict__ compact) {
  __shared__ VtAssignLds L;
  const uint32_t per_job = (uint32_t)tiles * (uint32_t)batch;
  int frame, tile;
  if (blockIdx.x < per_job) {
    vt_unit(blockIdx.x, (uint32_t)tiles, (uint32_t)batch, frame, tile);
    vt_assign_tile(L, frame, tile, cposr, pos16, recs, dir, low, gbits, tilecnt, tiles, max_voxels, g, vinfo,
                   totals, coords, num_pts, coors4);
  } else {
    vt_unit(blockIdx.x - per_job, (uint32_t)tiles, (uint32_t)batch, frame, tile);
    vt_emit_tile<DIM>(L, frame, tile, points, n, dim_rt, tiles, gbits, cposr, dir, pos16, cap, compact);
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
      typedef float f32x4a __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(f32x4a{val[u][0], val[u][1], val[u][2], val[u][3]},
                                  reinterpret_cast<f32x4a*>(dst));  // written once, read by the next op
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

// ------------------------------------------------------------------------------------------------ E'
// Gather form of the output writer (hard_voxelize path 3): no compact payload array and no second pass over the
// points.  The group kernel leaves clist[compact place] = point index; a row's j-th point is
// points[clist[start(v) + j]].  A lane still owns VEC consecutive floats of the flat voxels array; they belong to
// at most two points (dim >= 3), whose indices come from clist (neighbouring lanes read neighbouring entries) and
// whose floats are single dword loads (lanes of a row hit the same 20-byte records: the texture path merges them).
// Every load is issued for every lane with a clamped address (no divergent branch in front of a load), so the index
// loads of all eight elements and then the point loads of all eight elements are in flight together.
template <int VEC>
__global__ __launch_bounds__(kVtRowsThreads) void vt_rows_gather_kernel(
    const float* __restrict__ points, int64_t n, const uint32_t* __restrict__ clist, int64_t cap,
    const uint2* __restrict__ vinfo, const int* __restrict__ totals, int batch, int units, int max_voxels, int rowq,
    int step_v, int step_j, int dim, float* __restrict__ voxels, int32_t* __restrict__ coords,
    int32_t* __restrict__ num_pts, int32_t* __restrict__ num_voxels, int32_t* __restrict__ coors4) {
  int frame, unit;
  vt_unit(blockIdx.x, (uint32_t)units, (uint32_t)batch, frame, unit);
  const uint32_t total_q = (uint32_t)max_voxels * (uint32_t)rowq;
  const float* pf = points + (int64_t)frame * n * dim;
  const uint32_t* cl = clist + (int64_t)frame * cap;
  const uint2* vi = vinfo + (int64_t)frame * max_voxels;
  float* vf = voxels + (int64_t)frame * max_voxels * ((int64_t)rowq * VEC);
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
  __shared__ uint2 st_vi[kVtRowsIlp * kVtRowsThreads + 2];
  const uint32_t qb = (uint32_t)unit * (kVtRowsIlp * kVtRowsThreads);
  const uint32_t vb = vt_div(min(qb, total_q), (uint32_t)rowq, 1.0f / (float)rowq);
  const uint32_t ve = vt_div(min(qb + kVtRowsIlp * kVtRowsThreads - 1u, total_q), (uint32_t)rowq, 1.0f / (float)rowq);
  for (uint32_t r = threadIdx.x; r <= ve - vb; r += kVtRowsThreads)
    st_vi[r] = vb + r < (uint32_t)max_voxels ? vi[vb + r] : make_uint2(0u, 0u);
  __syncthreads();
  const float inv_dim = 1.0f / (float)dim;
  uint32_t k0[kVtRowsIlp], nfl[kVtRowsIlp], r0[kVtRowsIlp], i0[kVtRowsIlp], i1[kVtRowsIlp];
#pragma unroll
  for (int u = 0; u < kVtRowsIlp; ++u) {
    const uint2 info = st_vi[min(v[u], ve) - vb];
    nfl[u] = info.y * (uint32_t)dim;  // valid floats of the row
    r0[u] = j[u] * VEC;
    const uint32_t s0 = vt_div(r0[u], (uint32_t)dim, inv_dim);  // first point the lane's floats belong to
    k0[u] = r0[u] - s0 * (uint32_t)dim;
    // clamped to the row's last point (to entry 0 of the frame's list for an empty row): always a mapped address
    const uint32_t last = info.x + (info.y ? info.y - 1u : 0u);
    i0[u] = cl[min(info.x + s0, last)];
    i1[u] = cl[min(info.x + s0 + 1u, last)];
  }
  float val[kVtRowsIlp][VEC];
#pragma unroll
  for (int u = 0; u < kVtRowsIlp; ++u) {
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      uint32_t k = k0[u] + (uint32_t)c;
      const bool second = k >= (uint32_t)dim;
      k -= second ? (uint32_t)dim : 0u;
      const bool live = r0[u] + (uint32_t)c < nfl[u];
      const uint32_t idx = live ? (second ? i1[u] : i0[u]) : 0u;  // a dead lane reads point 0 (in bounds) and drops it
      const float x = pf[(int64_t)idx * dim + k];
      val[u][c] = live ? x : 0.f;
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
      typedef float f32x4a __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(f32x4a{val[u][0], val[u][1], val[u][2], val[u][3]},
                                  reinterpret_cast<f32x4a*>(dst));
    } else {
#pragma unroll
      for (int c = 0; c < VEC; ++c) dst[c] = val[u][c];
    }
    if (j[u] == 0 && (int)v[u] >= nv) {  // padding rows of coords / count / coors4
      const int64_t row = (int64_t)frame * max_voxels + v[u];
      const VtInt3 z3{0, 0, 0};
      __builtin_memcpy(coords + row * 3, &z3, sizeof(z3));
      num_pts[row] = 0;
      if (coors4) *reinterpret_cast<int4*>(coors4 + row * 4) = make_int4(-1, 0, 0, 0);
    }
  }
}

}  // namespace pd3
