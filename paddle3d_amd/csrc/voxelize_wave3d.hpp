// hard_voxelize, wave form for 3-D grids (path 14; the library's choice for grids of 2^20 .. 2^28 cells such as config
// 4's 1440 x 1440 x 40 = 82.9 M cells of 0.075 m, configs/centerpoint/centerpoint_voxels_0075voxel_nuscenes_10sweep.yml:
// 111-173, which used to fall to the radix-sort path at 0.085 of the HBM roofline).  Same order-independent restatement
// and the same four launches as voxelize_wave.hpp -- route, group, assign, rows -- with the one piece that cannot scale
// with the grid replaced:
//
//   * a group is still one of 1024 residue classes of the cell key (vt_key_to_group) and still belongs to ONE WAVE, but
//     its cells are 2^(bits - 10) <= 2^18 sparse possibilities of which a few hundred occur, so the per-cell tables
//     (records so far, place in the index list) are an LDS HASH TABLE of 1024 slots per wave keyed by the cell
//     (ds_cmpst_rtn claims a slot, linear probing) instead of arrays indexed by the cell.  What makes the result the
//     reference's: the slot of a point = the value a returning LDS add on its cell's counter hands back, issued for the
//     64 records of a step in ONE instruction (lanes of one cell are served in ascending lane order = stream order,
//     steps in program order: pd3_selfcheck_lds_atomic_order) -- which hash slot a cell ends up in never matters.
//   * the routed record is (index inside the route tile, 14 bits) << 18 | (cell inside the group, 18 bits): the tile is
//     what the directory search of the group kernel finds anyway, so 32 bits still do.
//   * a group's first-point list lies in the group's region of a [frame][N] array (sized by its records, like the
//     index list) instead of at [group][cells per group], which would be 1 GB per frame here; vw_assign_kernel takes
//     the region starts (gregion).
//   * a group with more records than the table takes at 3/4 load (768; config 4 has 266 +- 40) is processed in 2^k
//     PASSES over its record stream, pass p taking the cells whose low k bits (of an odd multiple) are p; a pass whose
//     cells still do not fit doubles the pass count and starts over (k = 18 is one cell per pass: it terminates for
//     any input).  Slots and places come out of the passes into a per-record word; a last sweep in stream order turns
//     them into the first-point list (whose order the assign kernel relies on).  Never taken by a LiDAR frame;
//     tests/test_voxelize_gpu.py::test_wave3d_heavy_group builds the input that takes it.
// Preconditions (else the sort path runs): 2^20 < cells <= 2^28, N < 2^22 - 3, P <= 254, tiles <= 1024.
#pragma once
#include "voxelize_wave.hpp"

namespace pd3 {

constexpr int kV3Gbits = 10;        // 1024 groups per frame
constexpr int kV3Low = 18;          // cell-in-group bits of a record; 14 bits of in-tile index above them
constexpr int kV3Slots = 1024;      // hash slots per wave: keys, places, counters = 12 KB
constexpr int kV3Load = 768;        // distinct cells a pass may hold
constexpr int kV3RegSteps = 12;     // groups of up to 768 records keep (tile, record, hash slot) in registers
constexpr uint32_t kV3Empty = 0xFFFFFFFFu;
constexpr uint32_t kV3CellMask = (1u << kV3Low) - 1u;

static inline VwPlan v3_plan(uint32_t ncells, int64_t n, int max_pts, int batch, int shape) {
  VwPlan p{};
  int bits = 0;
  while (((int64_t)1 << bits) < (int64_t)ncells) ++bits;
  p.low = kV3Low;
  p.gbits = kV3Gbits;
  p.groups = 1 << kV3Gbits;
  p.cpg = 0;  // not a table size here
  struct Shape { int threads, rounds; };
  static const Shape shapes[] = {{512, 8}, {1024, 8}, {1024, 10}};
  int pick = shape;
  if (pick < 0 || pick >= 3) {
    pick = 0;
    for (int k = 1; k < 3; ++k)
      // (measured on 8 frames of config 4: 10240-point tiles 141.8 us, 4096-point tiles 148.5 us -- the group and assign
      //  kernels walk a directory column per group, so fewer, longer tiles win as long as the route kernel covers the chip)
      if (ceil_div(n, (int64_t)shapes[k].threads * shapes[k].rounds) * batch >= 192) pick = k;
  }
  p.threads = shapes[pick].threads;
  p.rounds = shapes[pick].rounds;
  p.tile = p.threads * p.rounds;  // <= 10240 < 2^14
  p.tiles = (int)ceil_div(n, p.tile);
  p.ok = bits > 20 && bits <= kV3Gbits + kV3Low && p.tiles <= kVtMaxTiles && n < (int64_t)kVtCpMask - 1 &&
         max_pts <= kVtMaxPts;
  return p;
}

static inline size_t v3_group_lds(int tiles) { return ((size_t)3 * kV3Slots + (size_t)3 * vw_pow2_above(tiles)) * 4; }

__global__ __launch_bounds__(kWave) void v3_group_kernel(
    const uint32_t* __restrict__ recs, const uint32_t* __restrict__ dir, int gbits, int tiles, int tile_len, int tp,
    int batch, int max_pts, uint32_t* __restrict__ clist, int64_t cap, uint2* __restrict__ flist,
    uint2* __restrict__ fcnt, uint32_t* __restrict__ gregion, uint32_t* __restrict__ aux) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
  const int groups = 1 << gbits;
  uint32_t* K = reinterpret_cast<uint32_t*>(vt_smem);  // [slots] cell of the slot, kV3Empty: free
  uint32_t* A = K + kV3Slots;                           // [slots] (kept << 24) | place of the cell in the region
  uint32_t* B = A + kV3Slots;                           // [slots] records of the cell so far
  uint32_t* pre = B + kV3Slots;                         // [tp]    records of the group before tile t
  uint32_t* tsrc = pre + tp;                            // [tp]    routed position of the group's run in tile t
  uint32_t* cntT = tsrc + tp;                           // [tp]    first points of the group per tile
  int frame, grp;
  vt_unit(blockIdx.x, (uint32_t)groups, (uint32_t)batch, frame, grp);
  const int lane = threadIdx.x;
  // this group's column of [tp][groups]: (first points of the group in tile t, those in the tiles before it)
  uint2* fcol = fcnt + (int64_t)frame * tp * groups + grp;

  // directory column -> prefix of the run lengths (as vw_group_kernel): the group's region of the index list / of the
  // first-point list is sized by its record count and starts at the sum of its offsets inside the tiles' slices
  const uint32_t* dcol = dir + (int64_t)frame * tiles * groups + grp;
  for (int t = lane; t < tp; t += kWave) cntT[t] = 0u;
  uint32_t total = 0, region = 0;
  for (int t0 = 0; t0 < tp; t0 += kWave) {
    const int t = t0 + lane;
    const uint32_t d = t < tiles ? dcol[(int64_t)t * groups] : 0u;
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
  if (lane == 0) gregion[(int64_t)frame * groups + grp] = region;
  vt_wave_sync();
  if (total == 0u) {
    for (int t = lane; t < tiles; t += kWave) fcol[(int64_t)t * groups] = make_uint2(0u, 0u);
    return;
  }
  const uint32_t* rf = recs + (int64_t)frame * tiles * tile_len;
  uint32_t* cl = clist + (int64_t)frame * cap + region;
  uint2* fl = flist + (int64_t)frame * cap + region;  // at most one first point per record
  uint32_t* ax = aux + (int64_t)frame * cap + region;
  const int nsteps = (int)((total + 63u) >> 6);
  const uint32_t P = (uint32_t)max_pts;
  uint32_t nfirst = 0;  // first points written so far (wave-uniform)

  // record 64 s + lane of the group's stream (input order): its tile and its word; lanes past the end re-read record 0
  auto load = [&](int s, uint32_t& t, uint32_t& w) -> bool {
    const uint32_t r = (uint32_t)s * kWave + (uint32_t)lane;
    const bool valid = r < total;
    const uint32_t rr = valid ? r : 0u;
    t = vw_tile_of(pre, tp, rr);
    w = rf[tsrc[t] + (rr - pre[t])];
    return valid;
  };
  auto slot_of = [](uint32_t cell) -> uint32_t { return (cell * 0x9E3779B1u) >> 22; };  // 10 bits
  auto pass_of = [](uint32_t cell, int kbits) -> uint32_t { return (cell * 40503u) & ((1u << kbits) - 1u); };
  // a first point's record, appended in stream order; x = index inside its tile | cell in group << 14
  auto announce = [&](bool first, uint32_t t, uint32_t w, uint32_t word) {
    const unsigned long long m = __ballot(first);
    if (first) {
      atomicAdd(&cntT[t], 1u);
      fl[nfirst + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] =
          make_uint2((w >> kV3Low) | ((w & kV3CellMask) << 14), word);
    }
    nfirst += (uint32_t)__popcll(m);
  };

  // claim / find the slot of `cell` (linear probing); a claim that takes a free slot counts in `ndist`
  auto insert = [&](uint32_t cell, bool act, uint32_t& ndist) -> uint32_t {
    uint32_t h = slot_of(cell);
    bool pend = act;
    while (__ballot(pend)) {
      uint32_t old = cell;
      if (pend) old = atomicCAS(&K[h], kV3Empty, cell);
      ndist += (uint32_t)__popcll(__ballot(pend && old == kV3Empty));
      if (pend && (old == kV3Empty || old == cell)) pend = false;
      if (pend) h = (h + 1u) & (uint32_t)(kV3Slots - 1);
    }
    return h;
  };
  // slots -> places in the group's region, behind the earlier passes' (any order will do: the list is scratch)
  auto place_cells = [&](uint32_t& kept_base) {
    uint32_t cnt[kV3Slots / kWave];
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < kV3Slots / kWave; ++j) {
      cnt[j] = min(B[lane + j * kWave], P);
      sum += cnt[j];
    }
    const uint32_t inc = (uint32_t)wave_inclusive_scan((int)sum);
    uint32_t at = kept_base + inc - sum;
#pragma unroll
    for (int j = 0; j < kV3Slots / kWave; ++j) {
      A[lane + j * kWave] = (cnt[j] << 24) | at;
      B[lane + j * kWave] = 0u;
      at += cnt[j];
    }
    kept_base += (uint32_t)__shfl((int)inc, kWave - 1, kWave);
  };
  // a record whose cell sits in slot h: its slot in the cell (returning add: ONE instruction per step, lane order =
  // stream order), its entry of the index list; returns the word its first-point record carries, kV3Empty if it is
  // not a first point
  auto emit = [&](uint32_t t, uint32_t w, uint32_t h, bool act) -> uint32_t {
    const uint32_t slot = atomicAdd(&B[act ? h : (uint32_t)lane], act ? 1u : 0u);  // idle lanes: + 0 on a word of their own
    const uint32_t info = A[h];
    const uint32_t place = info & 0xFFFFFFu;
    if (act && slot < P) cl[place + slot] = t * (uint32_t)tile_len + (w >> kV3Low);
    return act && slot == 0u ? (region + place) | (info & 0xFF000000u) : kV3Empty;
  };

  int kbits = 0;
  while ((total >> kbits) > (uint32_t)kV3Load) ++kbits;
  if (kbits == 0 && nsteps <= kV3RegSteps) {
    // the usual case (config 4: 266 +- 40 records per group): one pass, tiles / records / hash slots of all steps stay
    // in registers between the sweeps; all directory searches advance level by level together, then all loads
    for (int c = lane; c < kV3Slots; c += kWave) {
      K[c] = kV3Empty;
      B[c] = 0u;
    }
    uint32_t tt[kV3RegSteps], ww[kV3RegSteps], hh[kV3RegSteps];
#pragma unroll
    for (int k = 0; k < kV3RegSteps; ++k) tt[k] = 0u;
    for (int st = tp >> 1; st > 0; st >>= 1) {
#pragma unroll
      for (int k = 0; k < kV3RegSteps; ++k)
        if (k < nsteps) {
          const uint32_t r = (uint32_t)k * kWave + (uint32_t)lane;
          if (pre[tt[k] + (uint32_t)st] <= (r < total ? r : 0u)) tt[k] += (uint32_t)st;
        }
    }
#pragma unroll
    for (int k = 0; k < kV3RegSteps; ++k) {
      ww[k] = 0u;
      if (k < nsteps) {
        const uint32_t r = (uint32_t)k * kWave + (uint32_t)lane, rr = r < total ? r : 0u;
        ww[k] = rf[tsrc[tt[k]] + (rr - pre[tt[k]])];
      }
    }
    vt_wave_sync();
    uint32_t ndist = 0;
#pragma unroll
    for (int k = 0; k < kV3RegSteps; ++k) {
      hh[k] = 0u;
      if (k < nsteps) {
        const bool act = (uint32_t)k * kWave + (uint32_t)lane < total;
        hh[k] = insert(ww[k] & kV3CellMask, act, ndist);
        if (act) atomicAdd(&B[hh[k]], 1u);
      }
    }
    vt_wave_sync();
    uint32_t kept_base = 0;
    place_cells(kept_base);
    vt_wave_sync();
#pragma unroll
    for (int k = 0; k < kV3RegSteps; ++k)
      if (k < nsteps) {
        const bool act = (uint32_t)k * kWave + (uint32_t)lane < total;
        const uint32_t word = emit(tt[k], ww[k], hh[k], act);
        announce(word != kV3Empty, tt[k], ww[k], word);
      }
  } else {
    for (;;) {
      bool overflow = false;
      uint32_t kept_base = 0;
      const int npass = 1 << kbits;
      for (int p = 0; p < npass && !overflow; ++p) {
        for (int c = lane; c < kV3Slots; c += kWave) {
          K[c] = kV3Empty;
          B[c] = 0u;
        }
        vt_wave_sync();
        // sweep 1: the pass's cells claim slots; records per cell
        uint32_t ndist = 0;  // wave-uniform
        for (int s = 0; s < nsteps && !overflow; ++s) {
          uint32_t t, w;
          const bool valid = load(s, t, w);
          const uint32_t cell = w & kV3CellMask;
          const bool act = valid && pass_of(cell, kbits) == (uint32_t)p;
          const uint32_t h = insert(cell, act, ndist);
          if (act) atomicAdd(&B[h], 1u);
          overflow = ndist > (uint32_t)kV3Load;  // (<= 768 + 64 slots are taken at this point: probing always ends)
        }
        if (overflow) break;
        vt_wave_sync();
        place_cells(kept_base);
        vt_wave_sync();
        // sweep 2: slots, the index list, and the first points -- straight into the list when this is the only pass,
        // else parked per record
        for (int s = 0; s < nsteps; ++s) {
          uint32_t t, w;
          const bool valid = load(s, t, w);
          const uint32_t cell = w & kV3CellMask;
          const bool act = valid && pass_of(cell, kbits) == (uint32_t)p;
          uint32_t nd = 0;
          const uint32_t h = insert(cell, act, nd);  // finds the slot claimed in sweep 1 (nothing is free on its way)
          const uint32_t word = emit(t, w, h, act);
          if (kbits == 0) announce(word != kV3Empty, t, w, word);
          else if (act) ax[(uint32_t)s * kWave + (uint32_t)lane] = word;
        }
        vt_wave_sync();
      }
      if (!overflow) break;
      ++kbits;  // some pass held more than 768 cells: twice as many passes, from the start (every write is repeated)
    }
  }
  if (kbits > 0) {  // the parked first points, in stream order
    for (int s = 0; s < nsteps; ++s) {
      uint32_t t, w;
      const bool valid = load(s, t, w);
      const uint32_t word = valid ? ax[(uint32_t)s * kWave + (uint32_t)lane] : kV3Empty;
      announce(valid && word != kV3Empty, t, w, word);
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

}  // namespace pd3
