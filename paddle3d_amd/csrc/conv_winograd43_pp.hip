// 3x3 / stride 1 / pad 1 convolution + bias + ReLU by Winograd F(4x4, 3x3) on the fp32 matrix cores: the ping-pong form
// whose multiply waves issue nothing but MFMAs and LDS reads (round 4).  Same arithmetic, tile shape and results (bit for
// bit) as conv_winograd43.hip (reference layers: second_backbone.py:72-120, center_head.py:43-220, cuDNN there).
#include "../../include/paddle3d_amd.h"
#include "common.hpp"
#include "conv_winograd43.hpp"

#include <type_traits>

namespace pd3 {

// ---------------------------------------------------------------------------------------------------------------------
// What the hardware does (gfx950, measured: tools/hwcheck/mfma_valu_overlap.hip, pingpong_skeleton.hip):
//   * a wave's own VALU instructions are not hidden behind its v_mfma_f32_16x16x4_f32: 36 MFMAs with K fmas behind each
//     run at 32.4 / 36.9 / 44.5 / 51.2 cycles per MFMA for K = 0 / 1 / 3 / 6 (a filler costs its issue time);
//   * a wave that streams these MFMAs back to back starves its partner on the SIMD: beyond the first four, the partner's
//     LDS reads return when the stream ends, its LDS stores likewise, its VALU instructions crawl (a transform-like phase
//     of 1700 cycles beside a 2370-cycle stream takes 3700); its buffer_load ... lds pieces do go out.  Two waves with the
//     same mixed stream do not interleave either: the older one runs first.
// So a SIMD's time is close to the SUM of its MFMAs (32 cycles each) and of everything else its waves issue, however
// the work is arranged.  Three arrangements of this convolution were built and measured in round 4 (same bytes out):
// this one; one with U computed in registers behind the MFMAs; one with the transform cut into micro-steps behind the
// MFMAs of the same wave, all eight waves in one role (git history: conv_winograd43_pl.hip).  Batch 16, 128 -> 128 @
// 128 x 128: packed form (conv_winograd43.hip) 0.254 ms, the others 0.254-0.259, this one 0.239 -- and 0.214 once what
// the partner's stream blocks HARD (LDS returns and LDS stores, not VALU issue, which only slows down) was moved out of
// its way: a wave reads the rows of its next transform slot at the END of its own multiply slot, in front of the barrier,
// sends its fetches first thing in the transform slot, and group 1 reads half of its U in front of the arithmetic.
//
// The arrangement: group g (waves 4g .. 4g+3, one per SIMD) owns tile row g for all 64 channels of the workgroup and
// alternates
//   T(s): its 256 threads turn the 8 channels x 16 tiles of its tile row into V (one patch per thread pair), from rows
//         that are already in registers; the slot opens with the fetches (U of slot s by group 0, the wave's rows of
//         slot s + 1), which the memory pipe takes while the other group's MFMAs hold the SIMD;
//   M(s): 72 MFMAs (two trips of 4 input channels) fed by ds_read_b128 alone: B from the group's V, A from a per-lane
//         packed copy of U in LDS.  No VALU, no fetches (one buffer_load ... lds costs an MFMA stream 60-185 cycles of
//         issue, a transform slot 25-60); behind its sixteenth group of MFMAs the wave reads the rows of T(s + 1) from
//         its planes of Raw into registers (fetched in T(s): they have landed).
// The other group runs the opposite role, one time slot off; ONE workgroup barrier per slot (w4_lds_barrier: LDS
// traffic only, fetches travel across it).
// Nothing goes through registers on its way into LDS: the raw rows of the next transform slot AND the next slot's U
// (pre-transformed on the host into lane order, 73.7 KB per slot and workgroup) travel by buffer_load_dwordx4 ... lds.
// U has ONE buffer (double buffering does not fit 160 KB): both groups read slot s's values in the same time slot (group
// 0 during its multiply slot, group 1 into registers during its transform slot), and group 0 refills the buffer
// at the start of its next transform slot, a whole slot ahead of the barrier that publishes it.  The raw rows are private
// to a wave (wave cb transforms channels 2 cb, 2 cb + 1 of the slot): it reads, refetches and waits for them itself -- no
// barrier is involved.
// Tried in round 6 and dropped: the walk over several work items per workgroup that pays in conv_winograd43_ppv.hip (next
// item's first fetches under this item's output transform).  Here the item loop costs 34 spilled registers at the 256 this
// kernel already fills: 64 -> 64 at 256^2 317 -> 310 us with two items per workgroup, the other layers unchanged, four
// items slower (too few workgroups).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPpKT = 2;                                   // trips per slot
constexpr int kPpCi = kPpKT * kW4Ci;                       // 8 input channels per slot
constexpr int kPpRawR = 6;                                 // staged input rows of one tile row
constexpr int kPpRawPl = kPpRawR * kW4RawW;                // 432 floats per channel
constexpr int kPpRawSz = kPpCi * kPpRawPl;                 // 3456 floats per group
constexpr int kPpVsz = kPpCi * kW4TC * kW4Cs;              // 4608 floats per group
constexpr int kPpXPT = (2 * kPpRawPl / 4 + 63) / 64;       // 4 fetch pieces per wave: its two planes are 216 float4
constexpr int kPpUHalf = 9 * 64 * 4;                       // 2304 floats: U of one trip for one wave (9 float4 per lane)
constexpr int kPpUsz = kPpKT * 4 * kPpUHalf;               // 18432 floats per slot: [trip][cb][q][lane][4]

__global__ __launch_bounds__(512, 1) void conv3x3_winograd43_pp_kernel(const float* __restrict__ x,
                                                                       const float* __restrict__ ulane,
                                                                       const float* __restrict__ bias,
                                                                       float* __restrict__ out, int cin, int cout, int h,
                                                                       int w, int wv, int relu, int ptiles,
                                                                       long long* __restrict__ dbg) {
  constexpr int CO = 64;
  long long t_tr = 0, t_mu = 0, t_ba = 0, t_all = dbg ? clock64() : 0;  // phase cycles of this wave (measurement)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(wave_id());  // (uniform: scalar slot control)
  const int grp = wave >> 2, cb = wave & 3;  // tile row / 16-channel block of this wave; waves w and w + 4 share a SIMD
  const int gt = cb * 64 + lane;             // thread inside its group
  float* Us = smem;                                          // [2 trips][4 cb][9][64 lanes][4]: U of the current slot
  float* Raw = smem + kPpUsz + grp * (kPpRawSz + kPpVsz);    // [8 ci][6 rows][72 cols] of this group's tile row
  float* Vs = Raw + kPpRawSz;                      // [8 ci][16 tiles][36]
  const int tiles_x = (w + 4 * kW4TC - 1) / (4 * kW4TC), tiles_y = (h + 4 * kW4TR - 1) / (4 * kW4TR);
  const int nct = cout / CO;
  const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
  const int ct = slot_id % nct, pt = (slot_id / nct) * 8 + xcd;
  if (pt >= ptiles) return;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int y0 = ty * 4 * kW4TR, x0 = tx * 4 * kW4TC;
  const int slots = cin / kPpCi;
  const int64_t plane = (int64_t)h * w;
  const float* xin = x + (int64_t)n * cin * plane;

  // staging pattern of this WAVE's raw rows (identical for every slot).  Wave cb transforms channels 2 cb and 2 cb + 1 of
  // the slot and nothing else, so those two planes of Raw ([2][6][72] = 216 float4) are private to it: it fetches them
  // itself and needs no barrier between its reads and the next fetch.  float4 e = lane + 64 i comes from byte offset
  // gofs[i] of the slot's 8 channel planes, or from beyond the buffer's range (-> zeros) for the padding.
  constexpr int kWvN4 = 2 * kPpRawPl / 4;  // 216
  unsigned gofs[kPpXPT];
#pragma unroll
  for (int i = 0; i < kPpXPT; ++i) {
    const int e = min(lane + i * 64, kWvN4 - 1);
    const int cl = e / (kPpRawR * (kW4RawW / 4)), rem = e - cl * (kPpRawR * (kW4RawW / 4));
    const int r = rem / (kW4RawW / 4), c4 = rem - r * (kW4RawW / 4);
    const int gy = y0 + 4 * grp - 1 + r, gx = x0 - 4 + c4 * 4;
    const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
    gofs[i] = ok ? (unsigned)(4 * ((2 * cb + cl) * plane + (int64_t)gy * w + gx)) : 0x7ffffff0u;
  }
  // transform assignment: thread pair (2p, 2p+1) of the group owns patch p = (ci 0..7, tile column 0..15)
  const int pidx = gt >> 1, hf = gt & 1;
  const int pci = pidx >> 4, ptile = pidx & 15;
  const int rsrc = pci * kPpRawPl + 4 * ptile + 3 + 3 * hf;
  const int vdst = (pci * kW4TC + ptile) * kW4Cs + 18 * hf;
  // MFMA operands: B = V[(trip * 4 + k) ci][tile][component]; A = U of (co, ci) = (lane & 15, lane >> 4), float4 q of the
  // lane's 36 components at Us[((trip * 4 + cb) * 9 + q) * 256 + 4 lane]
  const int bbase = ((lane >> 4) * kW4TC + (lane & 15)) * kW4Cs;
  const float* uct = ulane + (int64_t)ct * slots * kPpUsz;  // this workgroup's 64 output channels, all slots

  w4_f32x4 acc[36];
#pragma unroll
  for (int c = 0; c < 36; ++c) acc[c] = (w4_f32x4){0.f, 0.f, 0.f, 0.f};

  // the wave's raw rows of slot s go from global memory straight into its planes of Raw (buffer_load_dwordx4 ... lds: no
  // staging registers, no store pass)
  auto fetch_x = [&](int s) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(xin + (int64_t)s * kPpCi * plane), 0, (int)(kPpCi * plane * 4), 0x00020000);
#pragma unroll
    for (int i = 0; i < kPpXPT; ++i) {
      if (lane + i * 64 < kWvN4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rs, (__attribute__((address_space(3))) void*)(Raw + 2 * cb * kPpRawPl + i * 256), 16, gofs[i], 0, 0, 0);
    }
  };
  // half hh of slot s of this wave's U block: 9 KB, contiguous in global memory and in LDS alike
  const __amdgpu_buffer_rsrc_t urs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uct), 0, (int)((int64_t)slots * kPpUsz * 4), 0x00020000);
  auto fetch_u = [&](int s, int hh) {
    const int blk = (hh * 4 + cb) * kPpUHalf;
#pragma unroll
    for (int q = 0; q < 9; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(urs, (__attribute__((address_space(3))) void*)(Us + blk + q * 256), 16,
                                               lane * 16, (s * kPpUsz + blk + q * 256) * 4, 0, 0);
  };
  // The wave's raw rows of the NEXT transform slot are read into registers at the end of the wave's multiply slot, in front
  // of the barrier: behind it the other group's MFMAs hold the SIMD, and LDS reads issued then return when that stream
  // ends (pingpong_skeleton: more than four reads wait for it) -- 340-450 cycles of every time slot.
  float rv[3][6];
  auto read_rows = [&]() {
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float* d = Raw + rsrc + b;
#pragma unroll
      for (int r = 0; r < 6; ++r) rv[b][r] = d[r * kW4RawW];
    }
  };
  // V = B^T d B of this thread pair's patch (as W4_TRANSFORM above), from the rows in rv[].  The fetch of the NEXT slot's
  // rows (slot sn) goes out first -- the memory pipe takes it while the SIMD is held -- and has the rest of this slot and
  // the wave's multiply slot to land.
  auto transform = [&](int sn, auto&& after_fetch) {
    fetch_x(sn);
    after_fetch();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_sched_barrier(0);
    float lo[3][3], hi[3][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float t[6];
      w4_in(rv[b][0], rv[b][1], rv[b][2], rv[b][3], rv[b][4], rv[b][5], t);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a][b] = t[a];
        hi[a][b] = t[3 + a];
      }
    }
    float* v = Vs + vdst;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float f[3], l[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float ph = w4_swap_pair(hi[a][b]), pl = w4_swap_pair(lo[a][b]);
        f[b] = hf ? ph : lo[a][b];
        l[b] = hf ? hi[a][b] : pl;
      }
      float o[6];
      w4_in(f[0], f[1], f[2], l[0], l[1], l[2], o);
      *reinterpret_cast<w4_f32x2*>(v + a * 6 + 0) = (w4_f32x2){o[0], o[1]};
      *reinterpret_cast<w4_f32x2*>(v + a * 6 + 2) = (w4_f32x2){o[2], o[3]};
      *reinterpret_cast<w4_f32x2*>(v + a * 6 + 4) = (w4_f32x2){o[4], o[5]};
    }
  };
  // a multiply slot: 72 MFMAs, one stream over both trips, fed by ds_read_b128 alone; B through a ring of three reads, two
  // groups ahead.  No buffer_load ... lds here: one piece costs an MFMA stream 60-185 cycles of issue (measured: a group
  // that issued 18 of them per slot took 4200 cycles for its 2304 cycles of MFMAs), a transform slot 25-60.
  //   group 0: A through a second ring, straight from Us (whose refill landed before the barrier in front of this slot);
  //   group 1: A from the registers ua[], read from Us at the end of its transform slot -- the SAME time slot in which
  //            group 0 reads, so that Us is free for the refill one slot later.
  auto vptr = [&](int g) { return Vs + (g / 9) * (kW4Ci * kW4TC * kW4Cs) + bbase + (g % 9) * 4; };
  auto uptr = [&](int g) { return Us + (((g / 9) * 4 + cb) * 9 + (g % 9)) * 256 + lane * 4; };
  w4_f32x4 ua[18];
  auto multiply_ring = [&]() {
    w4_f32x4 a[3], b[3];
    a[0] = *reinterpret_cast<const w4_f32x4*>(uptr(0));
    b[0] = *reinterpret_cast<const w4_f32x4*>(vptr(0));
    a[1] = *reinterpret_cast<const w4_f32x4*>(uptr(1));
    b[1] = *reinterpret_cast<const w4_f32x4*>(vptr(1));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int g9 = 0; g9 < 9; ++g9) {
        const int g = t * 9 + g9;
        if (g + 2 < 18) {
          a[(g + 2) % 3] = *reinterpret_cast<const w4_f32x4*>(uptr(g + 2));
          b[(g + 2) % 3] = *reinterpret_cast<const w4_f32x4*>(vptr(g + 2));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[g9 * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g % 3][j], b[g % 3][j], acc[g9 * 4 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g == 15) {  // the rows of the wave's next transform slot (fetched a slot ago), behind the last ring reads
          __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
          read_rows();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  auto multiply_regs = [&]() {
    w4_f32x4 b[3];
    b[0] = *reinterpret_cast<const w4_f32x4*>(vptr(0));
    b[1] = *reinterpret_cast<const w4_f32x4*>(vptr(1));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int g9 = 0; g9 < 9; ++g9) {
        const int g = t * 9 + g9;
        if (g + 2 < 18) b[(g + 2) % 3] = *reinterpret_cast<const w4_f32x4*>(vptr(g + 2));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[g9 * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[g][j], b[g % 3][j], acc[g9 * 4 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g == 16) {  // as above, one group later: by then all but four of the 72 registers of ua[] are free
          __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
          read_rows();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };

  // prologue: the raw rows of slot 0 (each wave its own) and slot 0's U (group 0's waves, as in the loop)
  fetch_x(0);
  if (grp == 0) {
    fetch_u(0, 0);
    fetch_u(0, 1);
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  read_rows();
  w4_lds_barrier();
  // (every wave passes 2 * slots + 1 barriers: group 1 waits out the first time slot, group 0 the last)
  auto sync = [&]() {
    const long long c1 = dbg ? clock64() : 0;
    w4_lds_barrier();
    if (dbg) t_ba += clock64() - c1;
  };
  // one straight-line loop per group (a single loop with the group's role chosen inside keeps two copies of the 144
  // accumulators alive across the join and spills)
  auto run = [&](auto is_g0) {
    constexpr bool G0 = decltype(is_g0)::value;
    if (!G0) sync();
    for (int s = 0; s < slots; ++s) {
      {  // transform slot s
        const long long c0 = dbg ? clock64() : 0;
        const int sn = min(s + 1, slots - 1);  // (the last slot's refetch is never read)
        if (G0) {
          // Us is free: group 0 read slot s - 1 through its ring in the time slot before this one, group 1 into ua[].
          // The refill goes out at once -- the other group's MFMAs hold this SIMD for most of the slot (a dense fp32 MFMA
          // stream starves its partner of VALU issue and LDS returns: tools/hwcheck/pingpong_skeleton.hip), the memory
          // pipe takes the eighteen pieces meanwhile -- and lands before the barrier that ends the slot, in front of
          // group 0's multiply slot and group 1's read.
          if (s > 0) {
            fetch_u(s, 0);
            fetch_u(s, 1);
          }
          transform(sn, [&]() {});
          __builtin_amdgcn_s_waitcnt(0x0f70 | 4);  // vmcnt(4): everything but the four raw-row pieces issued last
        } else {
          // U of this slot into registers: trip 0's half goes out in front of the arithmetic (the reads return when the
          // other group's stream ends, the arithmetic does not wait for them), trip 1's half behind it (registers)
          transform(sn, [&]() {
#pragma unroll
            for (int g = 0; g < 9; ++g) ua[g] = *reinterpret_cast<const w4_f32x4*>(uptr(g));
          });
#pragma unroll
          for (int g = 9; g < 18; ++g) ua[g] = *reinterpret_cast<const w4_f32x4*>(uptr(g));
        }
        if (dbg) t_tr += clock64() - c0;
      }
      sync();
      {  // multiply slot s
        const long long c0 = dbg ? clock64() : 0;
        if (G0) multiply_ring();
        else multiply_regs();
        if (dbg) t_mu += clock64() - c0;
      }
      sync();
    }
    if (G0) sync();
  };
  if (grp == 0) run(std::true_type{});
  else run(std::false_type{});
  if (dbg && blockIdx.x == 8 && lane == 0) {
    dbg[wave * 4 + 0] = t_tr;
    dbg[wave * 4 + 1] = t_mu;
    dbg[wave * 4 + 2] = t_ba;
    dbg[wave * 4 + 3] = (clock64() - t_all) | ((long long)__builtin_amdgcn_s_getreg(2308) << 56);  // + SIMD id (HW_ID[5:4])
  }

  // epilogue (as above): Y = A^T M A; lane: tile column lane & 15, channels 4 (lane >> 4) + r of the co block
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = 0.f;
  const int co0 = ct * CO + cb * 16 + 4 * (lane >> 4);
  if (bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[co0 + r];
  }
  const int oy = y0 + 4 * grp, ox = x0 + 4 * (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float sm[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float c4[4];
      w4_out(acc[0 * 6 + j][r], acc[1 * 6 + j][r], acc[2 * 6 + j][r], acc[3 * 6 + j][r], acc[4 * 6 + j][r],
             acc[5 * 6 + j][r], c4);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) sm[kk][j] = c4[kk];
    }
    float* o = out + ((int64_t)n * cout + co0 + r) * plane + (int64_t)oy * w + ox;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float y4[4];
      w4_out(sm[kk][0], sm[kk][1], sm[kk][2], sm[kk][3], sm[kk][4], sm[kk][5], y4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y4[j] += bv[r];
        if (relu) y4[j] = fmaxf(y4[j], 0.f);
        if (ox + j >= wv) y4[j] = 0.f;
      }
      if (oy + kk < h && ox < w)
        __builtin_nontemporal_store((w4_f32x4){y4[0], y4[1], y4[2], y4[3]},
                                    reinterpret_cast<w4_f32x4*>(o + (int64_t)kk * w));
    }
  }
}

}  // namespace pd3

using namespace pd3;

static int launch_wino43_pp(const float* x, const float* u_lane, const float* bias, int batch, int cin, int cout, int h,
                            int w, int w_valid, int relu, float* out, hipStream_t s, long long* dbg = nullptr) {
  constexpr size_t lds = ((size_t)kPpUsz + 2 * (kPpRawSz + kPpVsz)) * sizeof(float);  // 138,240 B
  const void* fn = reinterpret_cast<const void*>(conv3x3_winograd43_pp_kernel);
  hipError_t e = pd3_max_dynamic_lds(fn, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int64_t ptiles = (int64_t)batch * ceil_div(h, 4 * kW4TR) * ceil_div(w, 4 * kW4TC);
  const int64_t nwg = (ptiles + 7) / 8 * 8 * (cout / 64);
  if (nwg >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  conv3x3_winograd43_pp_kernel<<<(unsigned)nwg, 512, lds, s>>>(x, u_lane, bias, out, cin, cout, h, w, w_valid, relu,
                                                               (int)ptiles, dbg);
  return launch_status();
}

static int check_wino43_pp(const float* x, const float* u_lane, const float* out, int batch, int cin, int cout, int h,
                           int w, int w_valid) {
  if (!x || !u_lane || !out || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || w_valid <= 0 || w_valid > w)
    return PD3_EINVAL;
  if (cin % kPpCi != 0 || cout % 64 != 0 || w % 4 != 0) return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(u_lane) % 16 != 0)
    return PD3_EINVAL;
  if ((int64_t)kPpCi * h * w >= (int64_t)1 << 29) return PD3_EUNSUPPORTED;       // 32-bit byte offsets inside a slot
  if ((int64_t)(cin / kPpCi) * kPpUsz >= (int64_t)1 << 29) return PD3_EUNSUPPORTED;  // and inside a channel tile's U
  return PD3_OK;
}

extern "C" int pd3_conv3x3_winograd43_pp_bias_relu(const float* x, const float* u_lane, const float* bias, int batch,
                                                   int cin, int cout, int h, int w, int w_valid, int relu, float* out,
                                                   void* stream) {
  const int st = check_wino43_pp(x, u_lane, out, batch, cin, cout, h, w, w_valid);
  if (st != PD3_OK) return st;
  return launch_wino43_pp(x, u_lane, bias, batch, cin, cout, h, w, w_valid, relu, out, static_cast<hipStream_t>(stream));
}

// measurement hook: + cycle counters of one workgroup (see include/paddle3d_amd.h for the layout of dbg)
extern "C" int pd3_conv3x3_winograd43_pp_trace(const float* x, const float* u_lane, const float* bias, int batch, int cin,
                                               int cout, int h, int w, int relu, float* out, long long* dbg,
                                               void* stream) {
  const int st = check_wino43_pp(x, u_lane, out, batch, cin, cout, h, w, w);
  if (st != PD3_OK || !dbg) return st != PD3_OK ? st : PD3_EINVAL;
  return launch_wino43_pp(x, u_lane, bias, batch, cin, cout, h, w, w, relu, out, static_cast<hipStream_t>(stream), dbg);
}
