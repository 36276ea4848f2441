// hard_voxelize, "wave" form of the tiled path (path 5): the same order-independent restatement as
// voxelize_tiled.hpp (voxel id of a cell = rank of its first point among all first points; slot of a point =
// number of earlier points in its cell), restructured around what the round-2 measurements charged for:
//
//   * the group kernel of the tiled path (one 256-thread workgroup per 4096 cells) was one round of workgroups
//     whose time was a workgroup's latency chain: barriers, a prefix over the four waves' count tables, a step
//     table in which a step never crossed a run boundary.  Here a group is 1024 cells (or fewer) and belongs to
//     ONE WAVE: no barrier, no cross-wave prefix, the wave's record stream cut into exact 64-record steps
//     (the tile a record comes from is found by a binary search over the group's directory column in LDS),
//     sixteen independent waves per CU hide each other's latency.
//   * pos16 / cposr (6 bytes per point, written and re-read) are gone: a cell's first point leaves an 8-byte
//     record in its group's list; voxel ids come from merging the lists tile by tile through an LDS bitmap.
//   * the row writer maps a lane to a (voxel, point slot) instead of to four floats of the flat output: ~170
//     instructions per float4 went into finding out whose floats they were.
// What sets the pace of these kernels is instruction issue (a wave64 instruction occupies its SIMD for several
// cycles: 4.8 M points x ~150 instructions are 20+ us whatever the memory does) and the NUMBER of scattered memory
// accesses, not bytes and rarely latency; DESIGN.md section 4.1 has the measurements each choice below rests on.
//
//   A  vw_route_kernel    as vt_route_kernel (tile of THREADS * R points sorted by group in LDS, one coalesced
//                         slice + directory row), minus pos16; clears vinfo.
//   B  vw_group_kernel    one wave per group: directory column -> prefix; sweep 1 hands out slots (returning
//                         LDS adds, lane order = point order) and counts; cells are placed in the group's region
//                         of the index list; sweep 2 puts the list together in LDS (clist[place] = point index of
//                         the kept points, written as one coalesced run) and appends the first-point records.
//                         Groups of up to kVwRegSteps * 64 records keep (record, slot) in registers between the
//                         sweeps; longer ones walk their records twice.
//   C  vw_assign_kernel   per route tile: bitmap of its first points = voxel ids; vinfo / coords / counts of the
//                         voxels the tile opens.
//   E  vw_rows_kernel     the fixed-shape output written once, a lane per (voxel, point slot) (D = 4 / 5; other
//                         point widths take vt_rows_gather_kernel of voxelize_tiled.hpp).
// Preconditions (else the other paths run): cells <= 2^20, groups <= 1024 per frame, N < 2^22 - 3, P <= 254.
#pragma once
#include "voxelize_tiled.hpp"

namespace pd3 {

constexpr int kVwMaxLow = 10;       // cells per group <= 1024: two 4 KB tables per wave, sixteen waves per CU

struct VwPlan {
  int low, cpg, groups, gbits;
  int threads, rounds, tile, tiles;  // route kernel shape
  bool ok;
};

// Route-kernel shape: the tile is THREADS * R points.  Longer tiles mean longer runs per (tile, group) and a
// shorter directory column.
static inline VwPlan vw_plan(uint32_t ncells, int64_t n, int max_pts, int batch, int shape) {
  VwPlan p{};
  int bits = 0;
  while (((int64_t)1 << bits) < (int64_t)ncells) ++bits;
  p.low = std::max(std::min(bits - 2, kVwMaxLow), 0);
  while (p.low > 0 && n >= ((int64_t)1 << (32 - p.low)) - 1) --p.low;
  p.gbits = std::max(bits - p.low, 2);
  p.groups = 1 << p.gbits;
  p.cpg = 1 << p.low;
  struct Shape { int threads, rounds; };
  static const Shape shapes[] = {{512, 8}, {1024, 8}, {1024, 10}, {512, 10}, {1024, 5}};
  const int nshapes = (int)(sizeof(shapes) / sizeof(shapes[0]));
  int pick = shape;
  if (pick < 0 || pick >= nshapes) {
    // the longest tile that still leaves a workgroup and a half per CU: long tiles mean long runs per (tile,
    // group), which is what the group kernel's 64-record steps and the directory search want (measured on 16
    // nuScenes frames: 10240-point tiles 119 us, 4096-point tiles 127 us for the whole operator); small batches
    // keep the short tile so that the route kernel still covers the chip
    pick = 0;
    for (int k = 1; k < 3; ++k)
      if (ceil_div(n, (int64_t)shapes[k].threads * shapes[k].rounds) * batch >= 384) pick = k;
  }
  p.threads = shapes[pick].threads;
  p.rounds = shapes[pick].rounds;
  p.tile = p.threads * p.rounds;
  p.tiles = (int)ceil_div(n, p.tile);
  p.ok = bits <= 20 && p.gbits <= kVtMaxGbits && p.tiles <= kVtMaxTiles &&
         n < (int64_t)kVtCpMask - 1 && max_pts <= kVtMaxPts;
  return p;
}

// ------------------------------------------------------------------------------------------------ A
// PAY (path 17, a MEASUREMENT form: round 5's answer to "carry the payload through the route pass"): the point's D
// floats travel with its record -- staged in the same LDS slice order (20 more bytes per point: the tile shrinks to
// 4096 points, one 512-thread workgroup per CU) and written as one coalesced run per tile into `pay`.  Nothing reads
// `pay` (the row writer still gathers): the path measures what the carried payload costs the route kernel, the other
// half of the form -- a row writer that streams a compact payload -- is the tiled form's path 2 (vt_rows_kernel).
template <int THREADS, int R, bool PAY = false>
__global__ __launch_bounds__(THREADS, 8) void vw_route_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ num_points, int64_t n, int dim, VtGrid g, int low,
    int gbits, int tiles, int batch, int max_voxels, uint32_t* __restrict__ recs, uint32_t* __restrict__ dir,
    uint2* __restrict__ vinfo, int in_tile_index, float* __restrict__ pay = nullptr) {
  // in_tile_index (the 3-D form, voxelize_wave3d.hpp): the record carries the point's index INSIDE its tile (< 2^14)
  // above `low` = 18 cell bits; the tile is what the group kernel's directory search finds anyway
  constexpr int kTile = THREADS * R;
  constexpr int kWaves = THREADS / kWave;
  extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
  const int groups = 1 << gbits;
  uint32_t* stage = reinterpret_cast<uint32_t*>(vt_smem);                      // [kTile]
  uint32_t* cnt_all = stage + kTile;                                           // [waves][groups]
  int* scan_tmp = reinterpret_cast<int*>(cnt_all + (size_t)kWaves * groups);   // [waves + 1]
  float* pstage = reinterpret_cast<float*>(scan_tmp + kWaves + 2);             // PAY: [kTile][dim]
  int frame, tile;
  vt_unit(blockIdx.x, (uint32_t)tiles, (uint32_t)batch, frame, tile);
  const int lane = lane_id(), wave = wave_id();
  uint32_t* cnt = cnt_all + (size_t)wave * groups;
  const int64_t nf = num_points ? min((int64_t)num_points[frame], n) : n;

  for (int d = lane; d < groups; d += kWave) cnt[d] = 0u;
  vt_wave_sync();
  {  // this tile's share of the arrays the later kernels expect clear
    const int per = (int)ceil_div(max_voxels, tiles);
    const int v1 = min((tile + 1) * per, max_voxels);
    for (int v = tile * per + (int)threadIdx.x; v < v1; v += THREADS)
      vinfo[(int64_t)frame * max_voxels + v] = make_uint2(0u, 0u);
  }

  // phase 1: keys; rank of a point among the wave's earlier points of its group (returning LDS add, lane order)
  const float* pf = points + (int64_t)frame * n * dim;
  const int64_t wave_base = (int64_t)tile * kTile + (int64_t)wave * (R * kWave);
  VtXyz p[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t i = wave_base + r * kWave + lane;
    p[r].x = p[r].y = p[r].z = __builtin_nanf("");
    if (i < nf) __builtin_memcpy(&p[r], pf + i * dim, sizeof(VtXyz));
  }
  uint32_t key[R], ord[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    uint32_t cellkey = 0, grp = 0, local = 0;
    key[r] = 0xFFFFFFFFu;
    ord[r] = 0;
    if (vt_cell_key(p[r].x, p[r].y, p[r].z, g, cellkey)) {
      vt_key_to_group(cellkey, gbits, grp, local);
      key[r] = (grp << low) | local;
      ord[r] = atomicAdd(&cnt[grp], 1u);
    }
  }
  __syncthreads();
  // phase 2: tile-level offsets, groups spread over the threads (two per thread: groups <= 2 * THREADS)
  int tile_total;
  {
    const int d0 = threadIdx.x * 2;
    int c0 = 0, c1 = 0;
    if (d0 < groups) {
      for (int w = 0; w < kWaves; ++w) c0 += (int)cnt_all[(size_t)w * groups + d0];
      for (int w = 0; w < kWaves; ++w) c1 += (int)cnt_all[(size_t)w * groups + d0 + 1];
    }
    const int ex = block_exclusive_scan<THREADS>(c0 + c1, scan_tmp, tile_total);
    if (d0 < groups) {
      *reinterpret_cast<uint2*>(dir + ((int64_t)frame * tiles + tile) * groups + d0) =
          make_uint2((uint32_t)ex | ((uint32_t)c0 << 16), (uint32_t)(ex + c0) | ((uint32_t)c1 << 16));
      uint32_t acc = (uint32_t)ex;
      for (int w = 0; w < kWaves; ++w) {
        const uint32_t c = cnt_all[(size_t)w * groups + d0];
        cnt_all[(size_t)w * groups + d0] = acc;
        acc += c;
      }
      for (int w = 0; w < kWaves; ++w) {
        const uint32_t c = cnt_all[(size_t)w * groups + d0 + 1];
        cnt_all[(size_t)w * groups + d0 + 1] = acc;
        acc += c;
      }
    }
  }
  __syncthreads();
  // phase 3: records into the LDS slice, grouped and stable
  const uint32_t low_mask = (1u << low) - 1u;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t k = key[r];
    const int64_t i = wave_base + r * kWave + lane;
    const uint32_t iw = in_tile_index ? (uint32_t)(wave * (R * kWave) + r * kWave + lane) : (uint32_t)i;
    if (k != 0xFFFFFFFFu) {
      const uint32_t at = cnt[k >> low] + ord[r];
      stage[at] = (iw << low) | (k & low_mask);
      if (PAY) {  // the point's floats behind its record (x, y, z are in registers, the rest is read again)
        float* d = pstage + (size_t)at * dim;
        d[0] = p[r].x;
        d[1] = p[r].y;
        d[2] = p[r].z;
        for (int j = 3; j < dim; ++j) d[j] = pf[i * dim + j];
      }
    }
  }
  __syncthreads();
  // phase 4: the slice leaves as one coalesced run
  uint32_t* out = recs + ((int64_t)frame * tiles + tile) * kTile;
  for (int j = threadIdx.x; j < tile_total; j += THREADS) out[j] = stage[j];
  if (PAY) {
    float* po = pay + ((int64_t)frame * tiles + tile) * kTile * dim;
    for (int j = threadIdx.x; j < tile_total * dim; j += THREADS) po[j] = pstage[j];
  }
}

// ------------------------------------------------------------------------------------------------ B
constexpr int kVwRegSteps = 40;  // groups of up to 2560 records keep (record, slot) in registers between the sweeps
constexpr int kVwChunk = 8;      // steps whose searches, loads and LDS adds are issued back to back

static inline int vw_pow2_above(int tiles) {  // entries of the padded directory prefix: a power of two > tiles
  int tp = 2;
  while (tp < tiles + 1) tp <<= 1;
  return tp;
}
static inline size_t vw_group_lds(int cpg, int tiles) { return ((size_t)2 * cpg + (size_t)3 * vw_pow2_above(tiles)) * 4; }

// the tile whose run holds record r of the group's stream: the last t with pre[t] <= r (pre is padded with
// 0xFFFFFFFF up to a power of two; empty tiles repeat their successor's value and are skipped by "last")
__device__ __forceinline__ uint32_t vw_tile_of(const uint32_t* pre, int tp, uint32_t r) {
  uint32_t t = 0;
  for (int st = tp >> 1; st > 0; st >>= 1) {
    const uint32_t c = t + (uint32_t)st;
    if (pre[c] <= r) t = c;
  }
  return t;
}

// A wave's time here is a chain of latencies (LDS search -> global load -> returning LDS add), and the kernel lasts
// as long as its heaviest group (about twice the mean on a nuScenes frame), so everything that does not depend on
// each other is issued together: the searches of eight steps, then their loads, then their adds (which the LDS
// executes in program order: that order is the slot order).
// How a cell's first point reaches the kernel that hands out voxel ids: the wave appends a record (index inside its
// tile, cell, place of the cell's points in the index list, kept count) to its OWN list in stream order -- first points
// come in increasing index, so the records of one route tile are a contiguous piece of the list -- and counts them
// per tile; vw_assign_kernel merges the lists of a tile through an LDS bitmap.  (The first form of this round stored
// one byte in a per-point map + an 8-byte record at the point's own index and read both back in point order: two
// scattered stores and a scattered load per first point, a count kernel and a 17 us assign kernel; 122 -> 117 us.)
__global__ __launch_bounds__(kWave) void vw_group_kernel(
    const uint32_t* __restrict__ recs, const uint32_t* __restrict__ dir, int low, int gbits, int tiles, int tile_len,
    int tp, int batch, int max_pts, uint32_t* __restrict__ clist, int64_t cap, uint2* __restrict__ flist,
    uint2* __restrict__ fcnt, uint32_t prio_unit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
  const int cpg = 1 << low, groups = 1 << gbits;
  uint32_t* A = reinterpret_cast<uint32_t*>(vt_smem);  // [cpg] (kept << 24) | place of the cell in the region
  uint32_t* B = A + cpg;                               // [cpg] records of the cell so far
  uint32_t* pre = B + cpg;                             // [tp]  records of the group before tile t
  uint32_t* tsrc = pre + tp;                           // [tp]  routed position of the group's run in tile t
  uint32_t* cntT = tsrc + tp;                          // [tp]  first points of the group per tile
  int frame, grp;
  vt_unit(blockIdx.x, (uint32_t)groups, (uint32_t)batch, frame, grp);
  const int lane = threadIdx.x;
  // this group's column of [tp][groups]: (first points of the group in tile t, those in the tiles before it)
  uint2* fcol = fcnt + (int64_t)frame * tp * groups + grp;

  // directory column -> prefix of the run lengths; the group's region of the index list is sized by its record
  // count and starts at the sum of its offsets inside the tiles' slices (no global counter)
  const uint32_t* dcol = dir + (int64_t)frame * tiles * groups + grp;
  uint32_t d0 = lane < tiles ? dcol[(int64_t)lane * groups] : 0u;  // in flight while the tables are cleared
  for (int c = lane; c < cpg; c += kWave) {
    A[c] = 0u;
    B[c] = 0u;
  }
  for (int t = lane; t < tp; t += kWave) cntT[t] = 0u;
  uint32_t total = 0, region = 0;
  for (int t0 = 0; t0 < tp; t0 += kWave) {
    const int t = t0 + lane;
    const uint32_t d = t0 == 0 ? d0 : (t < tiles ? dcol[(int64_t)t * groups] : 0u);
    const uint32_t c = d >> 16, off = d & 0xFFFFu;
    const uint32_t inc = (uint32_t)wave_inclusive_scan((int)c);
    if (t < tp) {
      pre[t] = t <= tiles ? total + inc - c : 0xFFFFFFFFu;
      tsrc[t] = (uint32_t)t * (uint32_t)tile_len + off;
    }
    total += (uint32_t)__shfl((int)inc, kWave - 1, kWave);
    uint32_t o = off;
#pragma unroll
    for (int dd = 1; dd < kWave; dd <<= 1) o += (uint32_t)__shfl_xor((int)o, dd, kWave);
    region += o;
  }
  vt_wave_sync();
  if (total == 0u) {
    for (int t = lane; t < tiles; t += kWave) fcol[(int64_t)t * groups] = make_uint2(0u, 0u);
    return;
  }
  // The kernel lasts as long as its heaviest wave (1.5x the mean records) and all waves are resident at once, four
  // per SIMD: a wave that knows it is heavy asks the issue arbiter for more than its share while its lighter
  // neighbours are still running (prio_unit = mean records per group; 0 = off, the round-3 behaviour).
  if (prio_unit) {
    if (total > prio_unit + (prio_unit >> 1)) __builtin_amdgcn_s_setprio(3);
    else if (total > prio_unit) __builtin_amdgcn_s_setprio(2);
    else if (total > (prio_unit >> 1)) __builtin_amdgcn_s_setprio(1);
  }

  const uint32_t* rf = recs + (int64_t)frame * tiles * tile_len;
  uint32_t* cl = clist + (int64_t)frame * cap + region;
  const uint32_t cell_mask = (uint32_t)cpg - 1u;
  const int nsteps = (int)((total + 63u) >> 6);
  const bool in_regs = nsteps <= kVwRegSteps;
  const uint32_t P = (uint32_t)max_pts;
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  uint2* fl = flist + ((int64_t)frame * groups + grp) * cpg;  // <= one first point per cell
  const float inv_tile = 1.0f / (float)tile_len;
  uint32_t nfirst = 0;  // first points written so far (wave-uniform)
  // a first point's record, appended in stream order; x = index inside its tile | cell in group << 14
  auto announce = [&](bool first, uint32_t idx, uint32_t cell, uint32_t word) {
    const unsigned long long m = __ballot(first);
    if (first) {
      const uint32_t t = vt_div(idx, (uint32_t)tile_len, inv_tile);
      atomicAdd(&cntT[t], 1u);
      fl[nfirst + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] =
          make_uint2((idx - t * (uint32_t)tile_len) | (cell << 14), word);
    }
    nfirst += (uint32_t)__popcll(m);
  };

  // records 64 s .. 64 s + 63 of the group's stream (input order), one per lane: where they lie in the routed
  // slices.  Lanes past the end name record 0 of the frame (a mapped address; they are masked by `slot`).
  auto source = [&](int s) -> uint32_t {
    const uint32_t r = (uint32_t)s * kWave + (uint32_t)lane;
    const uint32_t t = vw_tile_of(pre, tp, r);
    return r < total ? tsrc[t] + (r - pre[t]) : 0u;
  };
  // what a kept / first point leaves behind (direct form: the walk of a group too long for the registers)
  auto emit = [&](uint32_t rec, uint32_t slot) {
    const uint32_t cell = rec & cell_mask, info = A[cell];
    const uint32_t place = info & 0xFFFFFFu, idx = rec >> low;
    if (slot < P) cl[place + slot] = idx;  // (a lane without a record carries slot = kNone)
    announce(slot == 0u, idx, cell, (region + place) | (info & 0xFF000000u));
  };
  // eight steps: sources, records, slots (or only the counts) -- the walk of a group too long for the registers
  auto chunk = [&](int s0, uint32_t* rec, uint32_t* slot, bool want_slots) {
#pragma unroll
    for (int k = 0; k < kVwChunk; ++k) rec[k] = source(s0 + k);
#pragma unroll
    for (int k = 0; k < kVwChunk; ++k) rec[k] = rf[rec[k]];
#pragma unroll
    for (int k = 0; k < kVwChunk; ++k) {
      const bool valid = (uint32_t)(s0 + k) * kWave + (uint32_t)lane < total;
      // lanes without a record add 0 to the cell of the record they re-read (a word of the table: harmless)
      if (want_slots) {
        const uint32_t old = atomicAdd(&B[rec[k] & cell_mask], valid ? 1u : 0u);
        slot[k] = valid ? old : kNone;
      } else {
        atomicAdd(&B[rec[k] & cell_mask], valid ? 1u : 0u);
      }
    }
  };

  uint32_t rec[kVwRegSteps], slot[kVwRegSteps];
  // sweep 1: slots and counts.  In registers: ALL searches (level by level across the steps: a search loop per
  // step would be forty chains of dependent LDS reads one after the other), then ALL loads, then ALL adds.
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < kVwRegSteps; ++k) rec[k] = 0u;
    for (int st = tp >> 1; st > 0; st >>= 1) {
#pragma unroll
      for (int s0 = 0; s0 < kVwRegSteps; s0 += kVwChunk) {
        if (s0 < nsteps) {
          uint32_t pv[kVwChunk];
#pragma unroll
          for (int k = 0; k < kVwChunk; ++k) pv[k] = pre[rec[s0 + k] + (uint32_t)st];
#pragma unroll
          for (int k = 0; k < kVwChunk; ++k)
            if (pv[k] <= (uint32_t)(s0 + k) * kWave + (uint32_t)lane) rec[s0 + k] += (uint32_t)st;
        }
      }
    }
#pragma unroll
    for (int s0 = 0; s0 < kVwRegSteps; s0 += kVwChunk) {
      if (s0 < nsteps) {
        uint32_t ts[kVwChunk], pt[kVwChunk];
#pragma unroll
        for (int k = 0; k < kVwChunk; ++k) {
          ts[k] = tsrc[rec[s0 + k]];
          pt[k] = pre[rec[s0 + k]];
        }
#pragma unroll
        for (int k = 0; k < kVwChunk; ++k) {
          const uint32_t r = (uint32_t)(s0 + k) * kWave + (uint32_t)lane;
          rec[s0 + k] = rf[r < total ? ts[k] + (r - pt[k]) : 0u];
        }
      }
    }
#pragma unroll
    for (int s0 = 0; s0 < kVwRegSteps; s0 += kVwChunk) {
      if (s0 < nsteps) {
#pragma unroll
        for (int k = 0; k < kVwChunk; ++k) {
          const bool valid = (uint32_t)(s0 + k) * kWave + (uint32_t)lane < total;
          const uint32_t old = atomicAdd(&B[rec[s0 + k] & cell_mask], valid ? 1u : 0u);
          slot[s0 + k] = valid ? old : kNone;
        }
      } else {
#pragma unroll
        for (int k = 0; k < kVwChunk; ++k) slot[s0 + k] = kNone;
      }
    }
  } else {
    for (int s0 = 0; s0 < nsteps; s0 += kVwChunk) {
      uint32_t r8[kVwChunk], s8[kVwChunk];
      chunk(s0, r8, s8, false);
    }
  }
  vt_wave_sync();
  // cells -> places in the group's region (any order will do: the list is scratch).  Lane l takes cells l, l + 64, ...
  uint32_t kept_total;
  {
    uint32_t cnt[1 << (kVwMaxLow - 6)];
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < (1 << (kVwMaxLow - 6)); ++j) {
      const int c = lane + j * kWave;
      cnt[j] = c < cpg ? min(B[c], P) : 0u;
      sum += cnt[j];
    }
    const uint32_t inc = (uint32_t)wave_inclusive_scan((int)sum);
    kept_total = (uint32_t)__shfl((int)inc, kWave - 1, kWave);
    uint32_t at = inc - sum;
#pragma unroll
    for (int j = 0; j < (1 << (kVwMaxLow - 6)); ++j) {
      const int c = lane + j * kWave;
      if (c < cpg) {
        A[c] = (cnt[j] << 24) | at;
        B[c] = 0u;
      }
      at += cnt[j];
    }
  }
  vt_wave_sync();
  // sweep 2: the index list and the first-point records.  From registers, the list is first put together in LDS
  // (B is free now: a scattered 4-byte store to memory costs as much as streaming 40 bytes) and leaves coalesced;
  // what does not fit B goes to memory directly.
  if (in_regs) {
#pragma unroll
    for (int s0 = 0; s0 < kVwRegSteps; s0 += kVwChunk) {
      if (s0 < nsteps) {
        uint32_t info[kVwChunk];
#pragma unroll
        for (int k = 0; k < kVwChunk; ++k) info[k] = A[rec[s0 + k] & cell_mask];
#pragma unroll
        for (int k = 0; k < kVwChunk; ++k) {
          const uint32_t sl = slot[s0 + k], idx = rec[s0 + k] >> low;
          const uint32_t pos = (info[k] & 0xFFFFFFu) + sl;
          if (sl < P) {
            if (pos < (uint32_t)cpg) B[pos] = idx;
            else cl[pos] = idx;
          }
          announce(sl == 0u, idx, rec[s0 + k] & cell_mask, (region + (info[k] & 0xFFFFFFu)) | (info[k] & 0xFF000000u));
        }
      }
    }
    vt_wave_sync();
    const uint32_t staged = min(kept_total, (uint32_t)cpg);
    for (uint32_t i = (uint32_t)lane; i < staged; i += kWave) cl[i] = B[i];
  } else {
    for (int s0 = 0; s0 < nsteps; s0 += kVwChunk) {
      uint32_t r8[kVwChunk], s8[kVwChunk];
      chunk(s0, r8, s8, true);
#pragma unroll
      for (int k = 0; k < kVwChunk; ++k) emit(r8[k], s8[k]);
    }
  }
  vt_wave_sync();
  {  // per tile: the group's first points and their exclusive prefix over the tiles (the assign kernel's offset of
     // the tile's piece in the group's list: it used to sum the column itself, up to 74 loads per thread)
    uint32_t before = 0;
    for (int t0 = 0; t0 < tiles; t0 += kWave) {
      const int t = t0 + lane;
      const uint32_t c = t < tiles ? cntT[t] : 0u;
      const uint32_t inc = (uint32_t)wave_inclusive_scan((int)c);
      if (t < tiles) fcol[(int64_t)t * groups] = make_uint2(c, before + inc - c);
      before += (uint32_t)__shfl((int)inc, kWave - 1, kWave);
    }
  }
}

// ------------------------------------------------------------------------------------------------ C
// vw_assign_kernel: one workgroup per (frame, route tile).  The tile's first points lie in the groups' lists as one
// contiguous piece each (thread g: its offset = the group's counts of the earlier tiles, summed on the way to the
// tile's base = first points of the frame before it).  A first point sets its bit in an LDS bitmap of the tile;
// voxel id = base + bits before it; the records are put in voxel order in LDS and the rows leave coalesced.
// No per-point map, no scattered global access: ~0.4 M small contiguous reads instead of 2 M scattered accesses.
constexpr int kVwAssignThreads = 512;  // with 256 groups two threads share a group's piece of the list
constexpr int kVwAssignCap = 4096;  // records staged in voxel order (a tile that opens more voxels stores the rest directly)

__global__ __launch_bounds__(kVwAssignThreads) void vw_assign_kernel(
    const uint2* __restrict__ flist, const uint2* __restrict__ fcnt, int low, int gbits, int tiles, int tile_len,
    int tp, int batch, int max_voxels, VtGrid g, uint2* __restrict__ vinfo, int* __restrict__ totals,
    int32_t* __restrict__ coords, int32_t* __restrict__ num_pts, int32_t* __restrict__ coors4, int frame0,
    const uint32_t* __restrict__ gregion, int64_t cap) {
  // gregion != nullptr (the 3-D form): a group's list of first points lies at flist[frame * cap + gregion[frame][group]]
  // (sized by the group's records) instead of at a fixed [groups][cells per group] place, and cell keys may exceed
  // 2^24 (exact integer division instead of the float estimate)
  extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
  const int groups = 1 << gbits, cpg = 1 << low;

  const int words = (tile_len + 31) >> 5;
  uint32_t* bits = reinterpret_cast<uint32_t*>(vt_smem);     // [words]
  uint32_t* wpre = bits + words;                             // [words] first points of the tile before word w
  uint2* recv = reinterpret_cast<uint2*>(wpre + words);      // [kVwAssignCap] (place | kept << 24, cell key) by rank
  __shared__ int scan_tmp[kVwAssignThreads / kWave + 2];
  __shared__ int s_before[kVwAssignThreads / kWave];
  int frame, tile;
  vt_unit(blockIdx.x, (uint32_t)tiles, (uint32_t)batch, frame, tile);
  const uint2* fc = fcnt + (int64_t)frame * tp * groups;
  auto list_of = [&](int gi) -> const uint2* {
    return gregion ? flist + (int64_t)frame * cap + gregion[(int64_t)frame * groups + gi]
                   : flist + ((int64_t)frame * groups + gi) * cpg;
  };
  for (int w = threadIdx.x; w < words; w += kVwAssignThreads) bits[w] = 0u;
  __syncthreads();
  // per group (thread g, g + 512 with 1024 groups; with fewer groups than threads several threads share one):
  // offset of the tile's piece in the group's list = its counts of the earlier tiles, which the group's wave left
  // next to the tile's own count (round 4; the threads used to sum the column themselves)
  constexpr int kGpt = (1 << kVtMaxGbits) / kVwAssignThreads;  // groups per thread, at most
  uint32_t goff[kGpt], gcnt[kGpt];
  int before = 0;
  // fewer groups than threads: tpg threads share a group, each takes a contiguous share of its piece
  const int tpg = groups < kVwAssignThreads ? kVwAssignThreads / groups : 1;
  const int sub = tpg > 1 ? (int)threadIdx.x / groups : 0;
#pragma unroll
  for (int q = 0; q < kGpt; ++q) {
    const int gi = tpg > 1 ? (int)threadIdx.x % groups : (int)threadIdx.x + q * kVwAssignThreads;
    goff[q] = 0u;
    gcnt[q] = 0u;
    if (gi < groups && (tpg == 1 || q == 0)) {
      const uint2 ent = fc[(int64_t)tile * groups + gi];  // (count in this tile, count in the tiles before it)
      const uint32_t off = ent.y, c = ent.x;
      if (sub == 0) before += (int)off;
      const uint32_t lo = c * (uint32_t)sub / (uint32_t)tpg, hi = c * (uint32_t)(sub + 1) / (uint32_t)tpg;
      goff[q] = off + lo;
      gcnt[q] = hi - lo;
    }
  }
  auto group_of = [&](int q) { return tpg > 1 ? (int)threadIdx.x % groups : (int)threadIdx.x + q * kVwAssignThreads; };
  // pass A: bits of the tile's first points.  The first eight records of a group's piece (its mean is 5.5) stay in
  // registers for pass B; both passes issue their loads eight at a time.
  constexpr int kHold = 8;
  uint2 held[kGpt][kHold];
#pragma unroll
  for (int q = 0; q < kGpt; ++q) {
    const int gi = group_of(q);
    const uint32_t c = gi < groups ? gcnt[q] : 0u;
    const uint2* fl = list_of(min(gi, groups - 1)) + goff[q];
#pragma unroll
    for (int k = 0; k < kHold; ++k) held[q][k] = c ? fl[min((uint32_t)k, c - 1u)] : make_uint2(0u, 0u);
#pragma unroll
    for (int k = 0; k < kHold; ++k)
      if ((uint32_t)k < c) {
        const uint32_t pos = held[q][k].x & 0x3FFFu;
        atomicOr(&bits[pos >> 5], 1u << (pos & 31u));
      }
    for (uint32_t k0 = kHold; k0 < c; k0 += 8) {
      uint32_t x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = fl[min(k0 + k, c - 1u)].x;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k0 + k < c) {
          const uint32_t pos = x[k] & 0x3FFFu;
          atomicOr(&bits[pos >> 5], 1u << (pos & 31u));
        }
    }
  }
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) before += __shfl_xor(before, d, kWave);
  if (lane_id() == 0) s_before[wave_id()] = before;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int k = 0; k < kVwAssignThreads / kWave; ++k) base += s_before[k];
  // prefix over the bitmap words (words <= 2 * threads: two per thread)
  int tile_new;
  {
    const int w0 = threadIdx.x * 2;  // (words <= 512 <= 2 * threads)
    const int c0 = w0 < words ? __popc(bits[w0]) : 0, c1 = w0 + 1 < words ? __popc(bits[w0 + 1]) : 0;
    const int ex = block_exclusive_scan<kVwAssignThreads>(c0 + c1, scan_tmp, tile_new);
    if (w0 < words) wpre[w0] = (uint32_t)ex;
    if (w0 + 1 < words) wpre[w0 + 1] = (uint32_t)(ex + c0);
  }
  __syncthreads();
  if (tile == tiles - 1 && threadIdx.x == 0) totals[frame] = base + tile_new;
  const float inv_gx = 1.0f / (float)g.gx, inv_gy = 1.0f / (float)g.gy;
  auto write_row = [&](int vid, uint32_t word, uint32_t key) {
    const int64_t row = (int64_t)frame * max_voxels + vid;
    const uint32_t kept = word >> 24;
    vinfo[row] = make_uint2(word & 0xFFFFFFu, kept);
    const uint32_t t = gregion ? key / (uint32_t)g.gx : vt_div(key, (uint32_t)g.gx, inv_gx);
    const int cx = (int)(key - t * (uint32_t)g.gx);
    const uint32_t cz = gregion ? t / (uint32_t)g.gy : vt_div(t, (uint32_t)g.gy, inv_gy);
    const int cy = (int)(t - cz * (uint32_t)g.gy);
    const VtInt3 c3{(int)cz, cy, cx};
    __builtin_memcpy(coords + row * 3, &c3, sizeof(c3));
    num_pts[row] = (int)kept;
    // frame0: batch index of this launch's first frame (a half batch of the two-stream form starts at batch / 2)
    if (coors4) *reinterpret_cast<int4*>(coors4 + row * 4) = make_int4(frame0 + frame, (int)cz, cy, cx);
  };
  // pass B: records to their rank
  auto place = [&](int gi, uint2 e) {
    const uint32_t pos = e.x & 0x3FFFu;
    const uint32_t rank = wpre[pos >> 5] + (uint32_t)__popc(bits[pos >> 5] & ((1u << (pos & 31u)) - 1u));
    const uint32_t key = vt_group_to_key((uint32_t)gi, e.x >> 14, gbits);
    if (rank < (uint32_t)kVwAssignCap) recv[rank] = make_uint2(e.y, key);
    else if (base + (int)rank < max_voxels) write_row(base + (int)rank, e.y, key);
  };
#pragma unroll
  for (int q = 0; q < kGpt; ++q) {
    const int gi = group_of(q);
    if (gi >= groups) break;
    const uint32_t c = gcnt[q];
#pragma unroll
    for (int k = 0; k < kHold; ++k)
      if ((uint32_t)k < c) place(gi, held[q][k]);
    const uint2* fl = list_of(gi) + goff[q];
    for (uint32_t k0 = kHold; k0 < c; k0 += 8) {
      uint2 e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = fl[min(k0 + k, c - 1u)];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k0 + k < c) place(gi, e[k]);
    }
  }
  __syncthreads();
  const int staged = min(tile_new, kVwAssignCap);
  for (int j = threadIdx.x; j < staged; j += kVwAssignThreads) {
    if (base + j >= max_voxels) break;
    write_row(base + j, recv[j].x, recv[j].y);
  }
}

// ------------------------------------------------------------------------------------------------ E
// Row writer of the wave form.  vt_rows_gather_kernel gives a lane four consecutive floats of the flat output and
// spends ~170 instructions per float4 on finding out whose they are (the row writer was instruction bound: 1370
// issued instructions per eight float4s, wave64 instructions issue over several cycles); here a lane owns one
// (voxel, point slot): one index load, the point's D floats as one 16-byte + one 4-byte load (4-byte aligned, which
// is all global_load_dwordx4 asks for), the same two stores into the row -- lanes of consecutive slots write
// consecutive pieces, a wave writes one contiguous 64 * D * 4 bytes.  D = 4 / 5 (else the float4 form runs).
// 51 -> 47 us: what is left is the gather itself, 2.15 M random 20-byte reads at the ~57 G/s this machine serves
// them at (tools/hwcheck/memrates).  Tried on top and dropped: clearing the tensor with a fill on a second stream
// while the ranking kernels run and storing only the slots that hold a point (the row writer alone 47 -> 39 us, but
// the fill slows the kernels it runs beside by more than that: 125 -> 136 us).
constexpr int kVwRowsThreads = 256;
constexpr int kVwRowsIlp = 4;  // (voxel, slot) pairs per lane: four independent vinfo -> index -> point -> store chains

// A lane's chain is three dependent global round trips (vinfo, index list, point) in front of its stores; with one
// pair per lane the kernel's time was (waves / waves in flight) x that latency -- 150 k waves, 8 k in flight, ~3 us:
// 47-51 us for a 192 MB tensor that a bare fill writes in 27.  ILP pairs per lane (a workgroup covers ILP * 256
// consecutive slots, element u of lane i is slot i + 256 u: a wave's stores stay contiguous 64 * D * 4-byte runs).
template <int D, int ILP = kVwRowsIlp>
__global__ __launch_bounds__(kVwRowsThreads) void vw_rows_kernel(
    const float* __restrict__ points, int64_t n, const uint32_t* __restrict__ clist, int64_t cap,
    const uint2* __restrict__ vinfo, const int* __restrict__ totals, int batch, int units, int max_voxels, int max_pts,
    float* __restrict__ voxels, int32_t* __restrict__ coords, int32_t* __restrict__ num_pts,
    int32_t* __restrict__ num_voxels, int32_t* __restrict__ coors4) {
  int frame, unit;
  vt_unit(blockIdx.x, (uint32_t)units, (uint32_t)batch, frame, unit);
  const uint32_t total_q = (uint32_t)max_voxels * (uint32_t)max_pts;
  const uint32_t q0 = (uint32_t)unit * (kVwRowsThreads * ILP) + threadIdx.x;
  const float inv_p = 1.0f / (float)max_pts;
  uint32_t q[ILP], slot[ILP];
  uint2 info[ILP];
  bool in[ILP], live[ILP];
#pragma unroll
  for (int u = 0; u < ILP; ++u) {
    q[u] = q0 + (uint32_t)u * kVwRowsThreads;
    in[u] = q[u] < total_q;
    const uint32_t qq = in[u] ? q[u] : 0u;
    const uint32_t v = vt_div(qq, (uint32_t)max_pts, inv_p);
    slot[u] = qq - v * (uint32_t)max_pts;
    info[u] = vinfo[(int64_t)frame * max_voxels + v];
    if (in[u] && slot[u] == 0u) {
      const int nv = min(totals[frame], max_voxels);
      if (v == 0u) num_voxels[frame] = nv;
      if ((int)v >= nv) {  // padding rows of coords / count / coors4 (batch = -1)
        const int64_t row = (int64_t)frame * max_voxels + v;
        const VtInt3 z3{0, 0, 0};
        __builtin_memcpy(coords + row * 3, &z3, sizeof(z3));
        num_pts[row] = 0;
        if (coors4) *reinterpret_cast<int4*>(coors4 + row * 4) = make_int4(-1, 0, 0, 0);
      }
    }
  }
  uint32_t idx[ILP];
#pragma unroll
  for (int u = 0; u < ILP; ++u) {
    live[u] = in[u] && slot[u] < info[u].y;
    idx[u] = live[u] ? clist[(int64_t)frame * cap + info[u].x + slot[u]] : 0u;
  }
  vt_f32x4u a[ILP];
  float e[ILP];
#pragma unroll
  for (int u = 0; u < ILP; ++u) {
    const float* src = points + ((int64_t)frame * n + idx[u]) * D;
    a[u] = *reinterpret_cast<const vt_f32x4u*>(src);
    e[u] = D == 5 ? src[4] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < ILP; ++u) {
    if (!in[u]) continue;
    if (!live[u]) {
      a[u] = vt_f32x4u{0.f, 0.f, 0.f, 0.f};
      e[u] = 0.f;
    }
    float* dst = voxels + ((int64_t)frame * total_q + q[u]) * D;
    __builtin_nontemporal_store(a[u], reinterpret_cast<vt_f32x4u*>(dst));
    if (D == 5) __builtin_nontemporal_store(e[u], dst + 4);
  }
}

}  // namespace pd3
