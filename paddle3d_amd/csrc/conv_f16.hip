// Mixed-precision (AMP) form of the stride-1 3x3 convolutions of the dense BEV graph: fp16 activations and weights on
// the fp16 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate), bias + ReLU fused.  The reference ships an AMP
// configuration of the headline model (configs/centerpoint/centerpoint_pillars_02voxel_nuscenes_10sweep_ampO2_ultra.yml:
// 5-9, amp_cfg level O2: fp16 activations and weights in the convolutions) and publishes an FP16 figure next to the FP32
// one (docs/models/centerpoint/README.md:35); this is that path, reported as its own bench workload
// (`--workload centerpoint_pillars_amp`) with its error against the fp32 graph stated -- never as the fp32 headline.
//
// Direct implicit GEMM (no Winograd: F(4x4,3x3) amplifies fp16 rounding by its 4 .. 24x transform constants), laid out
// for what bounds an fp16 MFMA kernel on this machine -- LDS read bandwidth and the L2 -> CU ingest, not the matrix pipe:
//   * activations travel between the stride-1 layers as fp16 NHWC, so a pixel's 16 input channels of a K chunk are 32
//     contiguous bytes: the staged patch is a plain copy (no transpose, no conversion) and a lane's B operand of
//     v_mfma_f32_32x32x16_f16 (8 consecutive k of one column) is ONE aligned ds_read_b128; weights are packed on the host
//     as [cout tile][cin / 16][tap][co][16 ci] fp16, so the A operand is one ds_read_b128 too and a chunk is one linear
//     copy (round 6: in LDS the two 8-channel halves of the pixels live in two planes and the weights are packed
//     [tap][half][co][8], so that the 32 lanes of an operand read 32 consecutive 16-byte slots -- no bank conflict).
//     pd3_f32_nchw_to_f16_nhwc converts at the fp32 boundaries (after a stride-2 convolution, in front of the
//     head); the last layer of a chain writes fp32 NCHW for the fp32 kernels behind it.
//   * workgroup = 8 waves = (64 MB) output channels x 16 rows x 32 columns: a wave owns 64 channels x J = 2 MB rows = 2 x J
//     MFMA blocks; MB = 2 (128 channels) where cout % 128 == 0, MB = 1 for the 64-channel layers.  LDS: two patch buffers
//     of a 32-channel block (4 planes x 640 pixels x 16 B = 40 KB each) + two weight buffers of a 16-channel sub-chunk
//     (36 KB / 18 KB each): 152 KB / 116 KB, one workgroup per CU; everything arrives by LDS-DMA one sub-chunk (weights)
//     or one block (patch) ahead of its use, one barrier per sub-chunk.  See the kernel's header for what round 6 measured.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"

#include <algorithm>

namespace pd3 {

typedef _Float16 cf_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cf_h4 __attribute__((ext_vector_type(4)));
typedef float cf_f32x16 __attribute__((ext_vector_type(16)));
typedef float cf_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kCfCols = 32;               // output columns per workgroup (= N of one MFMA block)
constexpr int kCfPW = kCfCols + 2;        // staged patch width
constexpr int kCfKc = 16;                 // input channels per chunk (= K of the MFMA)
constexpr int kCfThreads = 512;

template <int MB>
struct CfShape {
  static constexpr int M = 64 * MB;                  // output channels per workgroup
  static constexpr int J = 2 * MB;                   // output rows per wave (each wave: 64 channels x J rows x 32 columns)
  static constexpr int R = 16;                       // output rows per workgroup: MB = 2: 2 channel blocks x 4 slabs of 4
                                                     // rows, MB = 1: 8 slabs of 2 rows
  static constexpr int NPIX = (R + 2) * kCfPW;       // staged pixels (612)
  static constexpr int NPP = (NPIX + 63) / 64 * 64;  // ... per plane, padded to whole LDS-DMA pieces (640)
  static constexpr int PG = NPP / 64;                // 64-pixel groups of the patch (10)
  static constexpr int PGW = (PG + 7) / 8;           // ... per wave
  // LDS: the patch travels in BLOCKS of 32 input channels = four planes [8-channel slot s][pixel][8 halfs] (a lane's B
  // operand = one 16-byte entry, the 32 lanes of a half wave read 32 consecutive entries: conflict-free ds_read_b128);
  // the weights in SUB-CHUNKS of 16 channels, [tap][kh][co][8] exactly as the host packs them.  Two buffers of each.
  static constexpr int PATCH = 4 * NPP * 8;          // halfs per patch buffer (40 KB)
  static constexpr int WTS = 9 * M * kCfKc;          // halfs per weight buffer
  static constexpr int WI = WTS / 512;               // LDS-DMA pieces (1 KB) of a sub-chunk's weights
  static constexpr int WPW = (WI + 7) / 8;           // ... per wave
  static constexpr int ES = 68;                      // halfs per pixel line of the epilogue's staging tile (136 B)
  // the epilogue's staging tiles (8 waves x 32 pixels x 68 halfs = 34 KB): MB = 2 in the weight buffer that is idle
  // between two items, MB = 1 (18 KB weight buffers) behind the operand buffers
  static constexpr int ETILES = 8 * 32 * ES;
  static constexpr size_t LDS = (size_t)(2 * (PATCH + WTS) + (MB == 1 ? ETILES : 0)) * sizeof(_Float16);
  static_assert(MB == 1 || ETILES <= WTS, "the staging tiles fit a weight buffer");
  static_assert(LDS <= 160 * 1024, "one workgroup per CU");
};

// One 1 KB piece global memory -> LDS by buffer_load_dwordx4 ... lds: lane l's 16 bytes from byte offset `voff` (+ the
// uniform `soff`) of the buffer [base, base + bytes) -- an offset beyond the buffer reads as zeros -- to lds + 16 l.
// (A plain function of plain types on purpose: inside a kernel TEMPLATE the address-space cast / the buffer-resource type
// make the host pass of hipcc 7.2 drop the kernel's stub without a diagnostic -- an undefined symbol at load time.)
__device__ __forceinline__ void cf_dma_piece(const _Float16* base, int bytes, _Float16* lds, unsigned voff, int soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base), 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// out_mode 0: fp16 NHWC (the next fp16 layer's input); 1: fp32 NCHW (what every fp32 kernel of the graph reads); 2: both
// (a block's last layer: fp16 NHWC for the next block's stride-2 convolution, fp32 NCHW for the FPN level -- out2)
//
// Round 6 (profiles/r06_conv_f16.txt): the round-5 form ran at 0.22-0.35 of the fp16 pipe with the matrix pipe 36 % busy.
// Switching its parts off showed a chunk taking 9000 cycles for 4608 of MFMAs, the prologue (one chunk's 56 KB) 6 us and
// the fp16 epilogue a quarter to a third of a launch; the counters, a two-way LDS bank conflict on every operand read.
// What it all came from:
//   * INGEST.  A 16-channel chunk is 32 of the 128 .. 768 bytes of a pixel in NHWC, and a lane fetched 16 of them: every
//     request pulled a whole cache line out of L2 for an eighth of its bytes, eight times over per layer -- the L2 -> CU path
//     moved 4.4 useful bytes per clock and CU (the fp32 Winograd kernel's whole-line fetches: 14).  Now the patch travels
//     in blocks of 32 channels, the four 16-byte pieces of a pixel's 64 bytes requested back to back by one wave (one L2
//     request per line and block), by buffer_load ... lds straight into LDS: no staging registers, no store pass, the
//     zero padding = out-of-range offsets.
//   * OPERANDS.  The two channel halves of a pixel side by side made a ds_read_b128's 16-lane groups touch every other
//     16-byte entry (two-way conflict); with 32 VGPRs holding the next chunk the A operand of a tap was read right in front
//     of its MFMAs, an exposed LDS round trip per tap.  Now: planes per 8-channel slot, and the operands of the next tap /
//     column offset are read while the current one multiplies (for one column offset the wave's rows and the three
//     row offsets share J + 2 row fragments).
//   * EPILOGUE.  32 eight-byte stores per lane, 256 .. 2304 bytes apart.  Now through a per-wave LDS tile: a lane stores
//     16 contiguous bytes, a store instruction eight whole 128-byte channel lines.
//   * PHASES.  With one workgroup per CU (152 KB of LDS) nothing ran beside a workgroup's prologue (a block of patch and a
//     sub-chunk of weights: 5-6 k cycles) -- a quarter of the time of a 64-input-channel layer's workgroup.  A workgroup
//     now walks `ipw` consecutive work items (channel tile fastest, so the head's nine / eighteen channel tiles of one
//     pixel tile follow each other): the next item's first pieces are sent during the current item's last sub-chunk, and
//     the patch of a layer with <= 64 input channels stays resident while the pixel tile does not change.
template <int MB, int OUT_MODE>
__global__ __launch_bounds__(kCfThreads, 1) void conv3x3_f16_kernel(const _Float16* __restrict__ x,
                                                                    const _Float16* __restrict__ wp,
                                                                    const float* __restrict__ bias,
                                                                    void* __restrict__ out, int cin, int cout, int h,
                                                                    int w, int relu, int ptiles, int ipw,
                                                                    float* __restrict__ out2 = nullptr) {
  using S = CfShape<MB>;
  constexpr int J = S::J;
  extern __shared__ __attribute__((aligned(16))) _Float16 cf_smem[];
  const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int tiles_x = (w + kCfCols - 1) / kCfCols, tiles_y = (h + S::R - 1) / S::R;  // partial border tiles are masked
  const int nct = cout / S::M;
  const int subs = cin / kCfKc, blocks = (subs + 1) >> 1;
  const int wbytes = (int)((int64_t)subs * S::WTS * 2);
  _Float16* Pbuf = cf_smem;
  _Float16* Wbuf = cf_smem + 2 * S::PATCH;
  // the epilogue's staging tiles: MB = 2 in the weight buffer that is idle between two items, MB = 1 behind the buffers
  _Float16* Ebase = MB == 2 ? nullptr : cf_smem + 2 * (S::PATCH + S::WTS);

  // XCD-aware order (as the fp32 kernels): work item = (pixel tile pt, channel tile ct); virtual block vb lives on XCD
  // vb % 8, its slot vb / 8 = (pixel tile group, ct) with ct fastest.  This workgroup walks slots s0 .. s0 + ipw - 1.
  const int xcd = blockIdx.x & 7;
  const int s0 = (blockIdx.x >> 3) * ipw;
  const int nslots = ((ptiles + 7) >> 3) * nct;
  const int l31 = lane & 31, kh = lane >> 5;
  const int mw = wave % MB, nw = wave / MB;  // the wave's 64-channel block and its slab of J rows

  struct Item {
    int ct, pt, n, y0, x0;
    bool valid;
  };
  auto item_of = [&](int sl) {
    Item it;
    it.ct = sl % nct;
    it.pt = (sl / nct) * 8 + xcd;
    it.valid = sl < nslots && sl < s0 + ipw && it.pt < ptiles;
    const int tx = it.pt % tiles_x, ty = (it.pt / tiles_x) % tiles_y;
    it.n = it.pt / (tiles_x * tiles_y);
    it.y0 = ty * S::R;
    it.x0 = tx * kCfCols;
    return it;
  };
  // the wave's 64-pixel groups of the patch: group g = wave, wave + 8; lane = staged pixel 64 g + lane, whose byte offset
  // in the frame (or an out-of-range offset: zeros) is pvo[]
  auto offsets = [&](const Item& it, unsigned (&pvo)[S::PGW]) {
#pragma unroll
    for (int i = 0; i < S::PGW; ++i) {
      const int pix = (wave + 8 * i) * 64 + lane;
      const int pr = pix / kCfPW, pc = pix - pr * kCfPW;
      const int gy = it.y0 - 1 + pr, gx = it.x0 - 1 + pc;
      const bool ok = pix < S::NPIX && gy >= 0 && gy < h && gx >= 0 && gx < w;
      pvo[i] = ok ? (unsigned)((gy * w + gx) * cin * 2) : 0x7ffffff0u;
    }
  };
  const int xbytes = (int)((int64_t)h * w * cin * 2);
  // block blk of an item's patch (input channels 32 blk .. 32 blk + 31): per pixel group the four 16-byte slots back to
  // back -- the same 64 cache lines four times in a row, so that L2 hands every line out once
  auto dma_patch = [&](const Item& it, const unsigned (&pvo)[S::PGW], int blk, int buf) {
    const _Float16* xin = x + (int64_t)it.n * h * w * cin;
    _Float16* P = Pbuf + buf * S::PATCH;
#pragma unroll
    for (int i = 0; i < S::PGW; ++i) {
      const int g = wave + 8 * i;
      if (S::PG % 8 == 0 || g < S::PG) {
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) cf_dma_piece(xin, xbytes, P + (sl * S::NPP + g * 64) * 8, pvo[i], blk * 64 + sl * 16);
      }
    }
  };
  // sub-chunk sc of channel tile ct's weights (input channels 16 sc .. 16 sc + 15): linear, packed in LDS order
  auto dma_w = [&](int ct, int sc, int buf) {
    const _Float16* wct = wp + (int64_t)ct * subs * S::WTS;
    _Float16* W = Wbuf + buf * S::WTS;
#pragma unroll
    for (int i = 0; i < S::WPW; ++i)
      if (S::WI % 8 == 0 || wave + 8 * i < S::WI)
        cf_dma_piece(wct, wbytes, W + (wave + 8 * i) * 512, lane * 16, (sc * S::WTS + (wave + 8 * i) * 512) * 2);
  };

  Item cur = item_of(s0);
  if (!cur.valid) return;
  unsigned pvo[S::PGW], pvn[S::PGW];
  offsets(cur, pvo);
  int wb = 0;         // weight buffer of the current sub-chunk
  int p0 = 0;         // patch buffer of the current item's block 0 (block b: p0 ^ (b & 1))
  bool fresh = true;  // only block 0 of the current item's patch has been sent so far
  dma_patch(cur, pvo, 0, 0);
  dma_w(cur.ct, 0, 0);
  __syncthreads();  // (drains vmcnt: the pieces have landed)

  // Waves w and w + 4 share a SIMD and its matrix pipe; the older one (w < 4) wins the arbitration, so the pair runs its
  // two MFMA streams one after the other (cycle stamps: 3200 cycles for waves 0-3, which then wait 2700 at the barrier,
  // 5500 for waves 4-7).  Sending the next pieces costs a wave 100-400 cycles of issue per piece (64 lanes, 64 cache lines)
  // -- with both waves sending first, the pipe idled that long in every sub-chunk.  So the halves send at different times:
  // waves 4-7 FIRST (while waves 0-3 multiply), waves 0-3 AFTER their MFMAs (while waves 4-7 multiply).
  const bool send_first = wave >= 4;
  for (int it_i = 0; it_i < ipw; ++it_i) {
    const Item nxt = item_of(s0 + it_i + 1);
    // the next item needs its patch sent unless it is the same pixel tile and the whole patch (<= 2 blocks) is resident
    const bool nxt_patch = nxt.valid && (nxt.pt != cur.pt || blocks > 2);
    if (nxt_patch) offsets(nxt, pvn);
    const int p0n = nxt_patch ? (p0 ^ ((blocks - 1) & 1) ^ 1) : p0;
    // the accumulators start at the bias (D's row = (reg & 3) + 8 (reg >> 2) + 4 kh of channel block i): its loads hide
    // behind the first sub-chunk and the epilogue has no add left
    const int co0 = cur.ct * S::M + mw * 64;
    cf_f32x16 acc[2][J];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float bvr = bias ? bias[co0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh] : 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) acc[i][j][r] = bvr;
      }
    // (block 0 of this item is in buffer p0: sent by the prologue / during the item before; blocks b >= 1 are sent during
    // sub-chunk 2 b - 2 unless the whole patch is resident from the item before: `fresh`)
    const bool stream_blocks = fresh;
    for (int sc = 0; sc < subs; ++sc) {
      const int blk = sc >> 1, k = sc & 1;
      auto send = [&]() {
        if (k == 0 && blk + 1 < blocks && stream_blocks) dma_patch(cur, pvo, blk + 1, p0 ^ ((blk + 1) & 1));
        if (sc + 1 < subs) {
          dma_w(cur.ct, sc + 1, wb ^ 1);
        } else if (nxt.valid) {  // the last sub-chunk: the next item's first pieces
          if (nxt_patch) dma_patch(nxt, pvn, 0, p0n);
          dma_w(nxt.ct, 0, wb ^ 1);
        }
      };
      if (send_first) send();
      const _Float16* P = Pbuf + (p0 ^ (blk & 1)) * S::PATCH;
      const _Float16* W = Wbuf + wb * S::WTS;
      // A: lane (m = l31, k = 8 kh ..) of channel block i, tap t;  B: lane (n = l31, k = 8 kh ..) of staged pixel row r at
      // column offset dx -- for one dx the wave's J output rows and the three dy share J + 2 row fragments
      const _Float16* wa = W + (kh * S::M + mw * 64 + l31) * 8;
      const _Float16* pb = P + ((2 * k + kh) * S::NPP + (nw * J) * kCfPW + l31) * 8;
      auto ld_a = [&](int t, cf_h8 (&a)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const cf_h8*>(wa + (t * 2 * S::M + i * 32) * 8);
      };
      auto ld_b = [&](int dx, cf_h8 (&b)[J + 2]) {
#pragma unroll
        for (int r = 0; r < J + 2; ++r) b[r] = *reinterpret_cast<const cf_h8*>(pb + (r * kCfPW + dx) * 8);
      };
      cf_h8 a[2][2], b[2][J + 2];
      ld_b(0, b[0]);
      ld_a(0, a[0]);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int step = dx * 3 + dy;
          // the operands of the NEXT step go out before this step's MFMAs (the fences keep the compiler from sinking them)
          if (dy < 2) {
            ld_a((dy + 1) * 3 + dx, a[(step + 1) & 1]);
          } else if (dx < 2) {
            ld_a(dx + 1, a[(step + 1) & 1]);
          }
          if (dy == 0 && dx < 2) ld_b(dx + 1, b[(dx + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[step & 1][i], b[dx & 1][j + dy], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!send_first) send();
      __syncthreads();  // what was sent during this sub-chunk has landed (vmcnt(0)), every wave is done with this one
      wb ^= 1;
    }

    // epilogue: D[row = (reg & 3) + 8 (reg >> 2) + 4 kh][col = l31] of block (i, j): channel co0 + 32 i + row, pixel
    // (y0 + J nw + j, x0 + l31).  (wb now names the buffer with the NEXT item's first weights; the other one is idle.)
    const int xg = cur.x0 + l31;
    // this wave's tile [32 pixels][64 channels], lines of 68 halfs
    _Float16* E = (MB == 2 ? Wbuf + (wb ^ 1) * S::WTS : Ebase) + wave * (32 * S::ES);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int yg = cur.y0 + nw * J + j;
      if (OUT_MODE == 0 || OUT_MODE == 2 || OUT_MODE == 3) {
        // registers -> the wave's LDS tile (8-byte stores, lines 136 bytes apart: conflict-free) -> 16 bytes per lane,
        // eight lanes per pixel: every global store instruction writes eight whole 128-byte channel lines
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // ReLU on the packed halfs: max(round(x), 0) = round(max(x, 0))
            cf_h4 pk = {(_Float16)acc[i][j][4 * q], (_Float16)acc[i][j][4 * q + 1], (_Float16)acc[i][j][4 * q + 2],
                        (_Float16)acc[i][j][4 * q + 3]};
            if (relu) pk = __builtin_elementwise_max(pk, (cf_h4){0, 0, 0, 0});
            *reinterpret_cast<cf_h4*>(E + l31 * S::ES + 32 * i + 8 * q + 4 * kh) = pk;
          }
        cf_h8 piece[4];  // (LDS operations of one wave execute in order: the reads see the stores above)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          piece[k] = *reinterpret_cast<const cf_h8*>(E + (8 * k + (lane >> 3)) * S::ES + (lane & 7) * 8);
        if (yg < h) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int px = cur.x0 + 8 * k + (lane >> 3);
            if (px < w) {
              // OUT_MODE 3: group-major [n][cout / 64][h][w][64] -- the 64 channels of a branch of the head lie pixel
              // after pixel, so the grouped final convolution fetches whole contiguous patch rows
              _Float16* o = OUT_MODE == 3
                                ? reinterpret_cast<_Float16*>(out) +
                                      ((((int64_t)cur.n * (cout >> 6) + (co0 >> 6)) * h + yg) * w + px) * 64 + (lane & 7) * 8
                                : reinterpret_cast<_Float16*>(out) + (((int64_t)cur.n * h + yg) * w + px) * cout + co0 +
                                      (lane & 7) * 8;
              *reinterpret_cast<cf_h8*>(o) = piece[k];
            }
          }
        }
      }
      if ((OUT_MODE == 1 || OUT_MODE == 2) && yg < h && xg < w) {
        const int64_t plane = (int64_t)h * w;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float* o = (OUT_MODE == 1 ? reinterpret_cast<float*>(out) : out2) +
                     (((int64_t)cur.n * cout + co0 + 32 * i + 4 * kh) * h + yg) * w + xg;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = relu ? fmaxf(acc[i][j][r], 0.f) : acc[i][j][r];
            __builtin_nontemporal_store(v, o + ((r & 3) + 8 * (r >> 2)) * plane);
          }
        }
      }
    }
    if (!nxt.valid) break;
    if (MB == 2) __syncthreads();  // the staging tiles lie in a weight buffer: the next item's sub-chunk 1 is sent into it
    if (nxt_patch) {
#pragma unroll
      for (int i = 0; i < S::PGW; ++i) pvo[i] = pvn[i];
    }
    p0 = p0n;
    fresh = nxt_patch;
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The stride-2 3x3 / pad 1 convolutions that open SecondBackbone's blocks 1 and 2 (second_backbone.py:84-113) under AMP:
// the same direct implicit GEMM on fp16 NHWC, output pixel (y, x) reading input pixels (2 y - 1 + dy, 2 x - 1 + dx).
// Workgroup = 8 waves = 128 output channels x 8 output rows x 32 output columns; wave (mw, nw) owns 64 channels x 2 rows =
// 2 x 2 MFMA blocks.  The staged patch is 17 input rows x 65 input columns x 16 channels with the COLUMNS DE-INTERLEAVED
// (even columns, then odd columns of a row): output column l reads input column 2 l + dx = entry l + (dx >> 1) of plane
// dx & 1, so the 32 lanes of a B operand read 32 consecutive 32-byte pixels exactly as in the stride-1 kernel (with the
// columns interleaved their stride would be 64 bytes: a four-way bank conflict on every ds_read_b128).
// fp16 NHWC out (the block's stride-1 layers follow).  cin % 16 == 0, cout % 128 == 0.
// Round 5, later: MW = 64-channel blocks per workgroup (2: 128 channels, 8 waves; 1: 64 channels, 4 waves -- the 64 -> 64
// layer that opens block 0) and GATHER = PointPillarsScatter fused in: the "image" is the pillar features [M, cin] fp16
// behind the inverse map (cell -> pillar row or -1, pd3_pointpillars_inverse_map), so the first layer of the backbone
// runs on the fp16 matrix cores as well and the canvas is never written (as in the fp32 pd3_scatter_conv3x3_bias_relu).
constexpr int kCs2R = 8;                         // output rows per workgroup
constexpr int kCs2PR = 2 * kCs2R + 1;            // staged input rows
constexpr int kCs2Half = kCfCols + 1;            // entries per parity plane of a staged row (33)
constexpr int kCs2Patch = kCs2PR * 2 * kCs2Half * kCfKc;  // halfs
constexpr int kCs2PPieces = kCs2Patch / 8;
template <int MW>
struct Cs2Shape {
  static constexpr int THREADS = 256 * MW;
  static constexpr int M = 64 * MW;
  static constexpr int WTS = 9 * M * kCfKc;
  static constexpr int WPIECES = WTS / 8;
  static constexpr int PPT = (kCs2PPieces + THREADS - 1) / THREADS;
  static constexpr int WPT = (WPIECES + THREADS - 1) / THREADS;
  static constexpr size_t LDS = (size_t)2 * (kCs2Patch + WTS) * sizeof(_Float16);
};

template <int MW, bool GATHER>
__global__ __launch_bounds__(256 * MW, 1) void conv3x3_s2_f16_kernel(const _Float16* __restrict__ x,
                                                                      const int32_t* __restrict__ inv,
                                                                      const _Float16* __restrict__ wp,
                                                                      const float* __restrict__ bias,
                                                                      _Float16* __restrict__ out, int cin, int cout,
                                                                      int h, int w, int ho, int wo, int relu,
                                                                      int ptiles) {
  using S = Cs2Shape<MW>;
  extern __shared__ __attribute__((aligned(16))) _Float16 cf_smem[];
  const int lane = lane_id(), wave = wave_id();
  const int tiles_x = (wo + kCfCols - 1) / kCfCols, tiles_y = (ho + kCs2R - 1) / kCs2R;
  const int nct = cout / S::M;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ct = slot % nct, pt = (slot / nct) * 8 + xcd;
  if (pt >= ptiles) return;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int y0 = ty * kCs2R, x0 = tx * kCfCols;  // output coordinates of the tile
  const int chunks = cin / kCfKc;
  const _Float16* xin = GATHER ? x : x + (int64_t)n * h * w * cin;
  const cf_h8* wsrc = reinterpret_cast<const cf_h8*>(wp) + (int64_t)ct * chunks * S::WPIECES;

  // staging pattern: piece e = (staged pixel, half of its 16 channels); staged pixel = (row pr, parity, entry)
  int64_t pofs[S::PPT];
  unsigned plive = 0;
#pragma unroll
  for (int i = 0; i < S::PPT; ++i) {
    const int e = min((int)threadIdx.x + i * S::THREADS, kCs2PPieces - 1);
    const int pix = e >> 1, hf = e & 1;
    const int pr = pix / (2 * kCs2Half), rem = pix - pr * (2 * kCs2Half);
    const int par = rem / kCs2Half, ent = rem - par * kCs2Half;
    const int pc = 2 * ent + par;  // input column of the patch, 0 .. 65 (65 = the odd plane's unused last entry)
    const int gy = 2 * y0 - 1 + pr, gx = 2 * x0 - 1 + pc;
    bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w && pc <= 2 * kCfCols;
    int64_t pixel = (int64_t)gy * w + gx;
    if (GATHER) {  // the cell's pillar row (or none)
      const int row = ok ? inv[(int64_t)n * h * w + pixel] : -1;
      ok = row >= 0;
      pixel = row;
    }
    pofs[i] = ok ? pixel * cin + 8 * hf : 0;
    plive |= ok ? (1u << i) : 0u;
  }
  cf_h8 preg[S::PPT], wreg[S::WPT];
  auto fetch = [&](int c) {
    const _Float16* xc = xin + c * kCfKc;
#pragma unroll
    for (int i = 0; i < S::PPT; ++i) preg[i] = *reinterpret_cast<const cf_h8*>(xc + pofs[i]);
    const cf_h8* wc = wsrc + (int64_t)c * S::WPIECES;
#pragma unroll
    for (int i = 0; i < S::WPT; ++i) wreg[i] = wc[min((int)threadIdx.x + i * S::THREADS, S::WPIECES - 1)];
  };
  auto stash = [&](int buf) {
    _Float16* P = cf_smem + buf * (kCs2Patch + S::WTS);
    _Float16* W = P + kCs2Patch;
#pragma unroll
    for (int i = 0; i < S::PPT; ++i) {
      const int e = (int)threadIdx.x + i * S::THREADS;
      const cf_h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      if (kCs2PPieces % S::THREADS == 0 || e < kCs2PPieces)  // [channel half][staged pixel][8], as the stride-1 kernel
        *reinterpret_cast<cf_h8*>(P + (e & 1) * (kCs2Patch / 2) + (e >> 1) * 8) = ((plive >> i) & 1u) ? preg[i] : z;
    }
#pragma unroll
    for (int i = 0; i < S::WPT; ++i) {
      const int e = (int)threadIdx.x + i * S::THREADS;
      if (S::WPIECES % S::THREADS == 0 || e < S::WPIECES) *reinterpret_cast<cf_h8*>(W + e * 8) = wreg[i];
    }
  };

  const int mw = wave % MW, nw = wave / MW;  // the wave's 64-channel block and its pair of output rows
  const int l31 = lane & 31, kh = lane >> 5;
  cf_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  fetch(0);
  stash(0);
  __syncthreads();
  for (int c = 0; c < chunks; ++c) {
    const bool more = c + 1 < chunks;
    if (more) fetch(c + 1);
    const _Float16* P = cf_smem + (c & 1) * (kCs2Patch + S::WTS);
    const _Float16* W = P + kCs2Patch;
    const _Float16* wa = W + (kh * S::M + mw * 64 + l31) * 8;
    const _Float16* pp = P + kh * (kCs2Patch / 2);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3, dx = t - 3 * dy;
      cf_h8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const cf_h8*>(wa + (t * 2 * S::M + i * 32) * 8);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int pr = 2 * (nw * 2 + j) + dy;
        b[j] = *reinterpret_cast<const cf_h8*>(pp + ((pr * 2 + (dx & 1)) * kCs2Half + l31 + (dx >> 1)) * 8);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) stash((c + 1) & 1);
    __syncthreads();
  }
  const int co0 = ct * S::M + mw * 64;
  const int xg = x0 + l31;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = bias ? bias[co0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh] : 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int yg = y0 + nw * 2 + j;
      if (yg >= ho || xg >= wo) continue;
      _Float16* o = out + (((int64_t)n * ho + yg) * wo + xg) * cout + co0 + 32 * i + 4 * kh;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][j][4 * q + e] + bv[4 * q + e];
          if (relu) v[e] = fmaxf(v[e], 0.f);
        }
        const cf_h4 pk = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *reinterpret_cast<cf_h4*>(o + 8 * q) = pk;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The final SeparateHead convolutions under AMP (center_head.py:99-118: per head a 3x3 convolution from the 64 channels
// of its first stage to 1 .. 3 maps): a grouped 3x3 convolution reading the first stage's fp16 NHWC output -- group g
// = channels 64 g .. 64 g + 63 of every pixel, 128 contiguous bytes -- and writing the fp32 NCHW maps the post-processing
// reads.  The fp32 form of this pair wrote the 2304-channel first-stage map as fp32 NCHW (2.4 GB per 16 frames, 1.02 ms)
// and fetched it back (0.53 ms); in fp16 NHWC it is half of that in each direction and every fetched line is used whole.
// Workgroup = (8 x 32-pixel tile, group, frame): the tile's (10 x 34) x 64 halfs are staged once (pixel lines padded to
// 68 halfs: conflict-free ds_read_b128); wave = two tile rows, v_mfma_f32_32x32x16_f16 with the weights as the A operand.
template <int CO>
__global__ __launch_bounds__(256) void grouped_conv3x3_small_f16_kernel(
    const _Float16* __restrict__ x, const _Float16* __restrict__ wg, const float* __restrict__ bias, int groups, int h,
    int w, float* __restrict__ out, int out_groups, int out_group0) {
  constexpr int TR = 8, TC = 32, PW = TC + 2, PS = 68;  // tile rows / columns, patch width, halfs per staged pixel
  __shared__ __attribute__((aligned(16))) _Float16 patch[(TR + 2) * PW * PS];
  const int tiles_x = (w + TC - 1) / TC;
  const int tx0 = (blockIdx.x % tiles_x) * TC, ty0 = (blockIdx.x / tiles_x) * TR;
  const int g = blockIdx.y, n = blockIdx.z;
  const int c = groups * 64;
  const _Float16* xin = x + (int64_t)n * h * w * c + g * 64;
  const int l31 = threadIdx.x & 31, kh = (threadIdx.x >> 5) & 1;
  cf_h8 aw[36];  // A fragments: W[tap][m = l31][16 s + 8 kh ..] for m < CO, zero rows above
  {
    const _Float16* wgrp = wg + (int64_t)g * 9 * CO * 64 + (l31 < CO ? l31 : 0) * 64 + 8 * kh;
    const cf_h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int sk = 0; sk < 4; ++sk) {
        const cf_h8 v = *reinterpret_cast<const cf_h8*>(wgrp + t * CO * 64 + 16 * sk);
        aw[t * 4 + sk] = l31 < CO ? v : z;
      }
  }
  {
    // all of a thread's pieces are requested before the first one is parked (one load per loop trip made the staging a
    // chain of eleven memory round trips: 0.59 ms per call, most of it here)
    constexpr int PIECES = (TR + 2) * PW * 8, PPT = (PIECES + 255) / 256;
    cf_h8 reg[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int e = min((int)threadIdx.x + i * 256, PIECES - 1);
      const int pix = e >> 3, q = e & 7;
      const int pr = pix / PW, pc = pix - pr * PW;
      const int gy = ty0 - 1 + pr, gx = tx0 - 1 + pc;
      const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
      const cf_h8 v = *reinterpret_cast<const cf_h8*>(xin + (ok ? ((int64_t)gy * w + gx) * c : 0) + q * 8);
      const cf_h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      reg[i] = ok ? v : z;
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int e = (int)threadIdx.x + i * 256;
      if (e < PIECES) *reinterpret_cast<cf_h8*>(patch + (e >> 3) * PS + (e & 7) * 8) = reg[i];
    }
  }
  __syncthreads();
  // The products run on the matrix cores (v_mfma_f32_32x32x16_f16): A = the group's weights, output channel m = lane &
  // 31 (rows CO .. 31 of the block are zero), B = the 32 pixels of a tile row, K = 16 channels per step -- 36 steps per
  // tap-and-chunk, two tile rows per wave.  (v_dot2c_f32_f16 with one pixel per thread issued at ~22 cycles per
  // instruction here: 864 of them per thread, 0.58 ms per call; the 32-wide M block wastes 29 of its 32 rows and is
  // still four times faster.)  The 36 A fragments were requested before the staging loads, so they have landed.
  const int ty = 2 * wave_id();
  cf_f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int dy = t / 3, dx = t - 3 * dy;
#pragma unroll
    for (int sk = 0; sk < 4; ++sk) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const cf_h8 bv = *reinterpret_cast<const cf_h8*>(patch + ((ty + j + dy) * PW + l31 + dx) * PS + 16 * sk + 8 * kh);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw[t * 4 + sk], bv, acc[j], 0, 0, 0);
      }
    }
  }
  // D[m = (reg & 3) + 8 (reg >> 2) + 4 kh][n = l31]: channels 0 .. CO - 1 are registers 0 .. CO - 1 of the kh = 0 lanes
  const int xg = tx0 + l31;
  if (kh != 0 || xg >= w) return;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int y = ty0 + ty + j;
    if (y >= h) continue;
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      const int ch = (out_group0 + g) * CO + o;
      out[(((int64_t)n * out_groups * CO + ch) * h + y) * w + xg] = acc[j][o] + (bias ? bias[g * CO + o] : 0.f);
    }
  }
}

// The same convolution, persistent and fed by LDS-DMA (round 6).  The form above stages a tile through registers behind a
// barrier and holds 144 registers of weights: two workgroups per CU, staging and MFMAs of a workgroup one after the other
// -- 0.48 ms per 16 frames for 1.2 GB of input (2.5 TB/s) with the matrix pipe idle three quarters of the time.  Here a
// workgroup walks a run of consecutive (frame, group, tile) items -- every CU the same number, at least kGpTiles -- with
// the group's weights in registers (reloaded where the run crosses into the next group); a FIFTH wave does nothing but
// fetch: the next-but-one tile's patch travels by buffer_load_dwordx4 ... lds into one of three buffers while the four
// multiplying waves run the MFMAs of this one (see the fetch wave below for why a wave of its own).  A pixel's 128-byte line
// is fetched whole by 8 lanes, but lane l takes 16-byte piece (l & 7) ^ (l >> 3) of it: the LDS image is XOR-swizzled by
// the fetch itself (piece p of patch pixel P sits in slot p ^ (P & 7)), so the B operand's ds_read_b128 over 32 consecutive
// pixels is conflict-free without padded lines -- which a fetch into LDS could not write.  Patch rows are 40 pixels apart
// (34 used), so P & 7 is the column's: the twelve swizzled read offsets of a lane are constants of the kernel.  Padding =
// fetches at an out-of-range offset (zeros), no branch.
constexpr int kGpTiles = 16;
constexpr unsigned kGpOob = 0x7ffffff0u;
template <int CO, bool GM>  // GM: x is group-major [n][groups][h][w][64] (out_mode 3 of the stride-1 kernel), else NHWC
__global__ __launch_bounds__(320, 1) void grouped_conv3x3_small_f16_pp_kernel(
    const _Float16* __restrict__ x, const _Float16* __restrict__ wg, const float* __restrict__ bias, int groups, int h,
    int w, float* __restrict__ out, int out_groups, int out_group0, int xbytes, int total, int per_wg) {
  constexpr int TR = 8, TC = 32, PWP = 40, PBYTES = (TR + 2) * PWP * 128;  // patch row pitch in pixels; bytes of a patch
  extern __shared__ __attribute__((aligned(16))) unsigned char gp_smem[];   // [3][PBYTES]
  const int tiles_x = (w + TC - 1) / TC, tiles = tiles_x * ((h + TR - 1) / TR);
  // work items = (frame, group, tile) in that order; this workgroup walks items [T0, T0 + nt): every CU gets the same
  // number of tiles whatever the number of groups (a grid of (tile strip, group, frame) left the last round of workgroups
  // half empty: 4.5 rounds for the head's 18-branch slice)
  const int T0 = blockIdx.x * per_wg, nt = min(per_wg, total - T0);
  const int c = groups * 64;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave == 4) {
    // THE FETCH WAVE.  A fetch that finds the memory pipe's queue full holds its wave, and a CU ingests a patch (43.5 KB) in
    // about the time its MFMAs take: issued by the multiplying waves themselves, the two ran one after the other (cycle
    // stamps: 3100 cycles of sending, THEN 3300 of MFMAs per tile).  This wave does nothing else: the next tile's 50
    // fetches, vmcnt(0), the barrier that publishes them.
    // fetch of patch row r, pixels 8 k .. 8 k + 7 (instruction (r, k)): lane l = pixel 8 k + (l >> 3), slot l & 7 <- piece
    // (l & 7) ^ (l >> 3)
    const unsigned line = GM ? 128u : 2u * (unsigned)c;  // bytes between neighbouring pixels
    const unsigned voff_n = (unsigned)(lane >> 3) * line + (unsigned)(((lane & 7) ^ (lane >> 3)) * 16);
    auto send = [&](int T, unsigned char* dst) {
      const int ng = T / tiles, t = T - ng * tiles;
      const int n = ng / groups, g = ng - n * groups;
      const int ty0 = (t / tiles_x) * TR, tx0 = (t - (t / tiles_x) * tiles_x) * TC;
      for (int r = 0; r < TR + 2; ++r) {
        const int gy = ty0 - 1 + r;
        const int gyc = min(max(gy, 0), h - 1);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const int gx = tx0 - 1 + 8 * k + (lane >> 3);  // this lane's pixel
          const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w && 8 * k + (lane >> 3) < TC + 2;
          // uniform part: the first pixel of the instruction, or pixel 0 of the row where that lies left of the image
          const int gx0 = tx0 - 1 + 8 * k, gxc = max(gx0, 0);
          const unsigned so = __builtin_amdgcn_readfirstlane(
              GM ? 128u * (unsigned)(((n * groups + g) * h + gyc) * w + gxc)
                 : 2u * ((unsigned)((n * h + gyc) * w + gxc) * (unsigned)c + (unsigned)(g * 64)));
          const unsigned vo = gx0 < 0 ? voff_n - line : voff_n;
          cf_dma_piece(x, xbytes, reinterpret_cast<_Float16*>(dst + (r * PWP + 8 * k) * 128), ok ? vo : kGpOob, (int)so);
        }
      }
    };
    // three patches: the fetches of tile it + 2 are in flight while the wave waits for tile it + 1's (50 younger: completion
    // is in order) -- with two the memory pipe ran empty at every barrier (3.0 TB/s)
    send(T0, gp_smem);
    if (nt > 1) {
      send(T0 + 1, gp_smem + PBYTES);
      __builtin_amdgcn_s_waitcnt(0x0f70 | (50 & 15) | ((50 >> 4) << 14));  // vmcnt(50)
    } else {
      __builtin_amdgcn_s_waitcnt(0x0f70);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int it = 0; it < nt; ++it) {
      if (it + 2 < nt) {
        send(T0 + it + 2, gp_smem + ((it + 2) % 3) * PBYTES);
        __builtin_amdgcn_s_waitcnt(0x0f70 | (50 & 15) | ((50 >> 4) << 14));
      } else {
        __builtin_amdgcn_s_waitcnt(0x0f70);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    return;
  }
  const int l31 = lane & 31, kh = lane >> 5;
  cf_h8 aw[36];  // A fragments: W[tap][m = l31][16 s + 8 kh ..] for m < CO, zero rows above
  float bv[CO];
  auto load_weights = [&](int g) {
    const _Float16* wgrp = wg + (int64_t)g * 9 * CO * 64 + (l31 < CO ? l31 : 0) * 64 + 8 * kh;
    const cf_h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int sk = 0; sk < 4; ++sk) {
        const cf_h8 v = *reinterpret_cast<const cf_h8*>(wgrp + t * CO * 64 + 16 * sk);
        aw[t * 4 + sk] = l31 < CO ? v : z;
      }
#pragma unroll
    for (int o = 0; o < CO; ++o) bv[o] = bias ? bias[g * CO + o] : 0.f;
  };
  int g_cur = (T0 / tiles) % groups;
  load_weights(g_cur);
  // B operand: lane (pixel l31 of tile row ty + j, kh), tap (dy, dx), K-step sk: piece 2 sk + kh of patch pixel
  // P = (ty + j + dy) * 40 + l31 + dx, in slot (2 sk + kh) ^ (P & 7) = (2 sk + kh) ^ ((l31 + dx) & 7)
  int boff[3][4];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int sk = 0; sk < 4; ++sk) boff[dx][sk] = (l31 + dx) * 128 + (((2 * sk + kh) ^ ((l31 + dx) & 7)) << 4);
  const int ty = 2 * wave;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the first patch has landed
  for (int it = 0; it < nt; ++it) {
    const int T = T0 + it;
    const int ng = T / tiles, t = T - ng * tiles;
    const int n = ng / groups, g = ng - n * groups;
    if (g != g_cur) {  // (uniform) the walk crossed into the next group: its weights
      g_cur = g;
      load_weights(g);
    }
    const unsigned char* P = gp_smem + (it % 3) * PBYTES;
    cf_f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // One wave per SIMD multiplies: nothing hides an LDS read but the wave's own other work, so the B fragments travel a
    // whole kernel row ahead: the 24 reads of row dy + 1 are in flight while the 24 MFMAs of row dy run.
    cf_h8 bq[2][24];
    auto load_row = [&](int dy, cf_h8 (&dst)[24]) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int sk = 0; sk < 4; ++sk)
            dst[(j * 3 + dx) * 4 + sk] = *reinterpret_cast<const cf_h8*>(P + (ty + j + dy) * (PWP * 128) + boff[dx][sk]);
    };
    load_row(0, bq[0]);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      if (dy < 2) load_row(dy + 1, bq[(dy + 1) & 1]);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int sk = 0; sk < 4; ++sk)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw[(dy * 3 + dx) * 4 + sk], bq[dy & 1][(j * 3 + dx) * 4 + sk],
                                                            acc[j], 0, 0, 0);
    }
    // D[m = (reg & 3) + 8 (reg >> 2) + 4 kh][n = l31]: channels 0 .. CO - 1 are registers 0 .. CO - 1 of the kh = 0 lanes
    const int ty0 = (t / tiles_x) * TR, tx0 = (t - (t / tiles_x) * tiles_x) * TC;
    const int xg = tx0 + l31;
    if (kh == 0 && xg < w) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int y = ty0 + ty + j;
        if (y < h) {
#pragma unroll
          for (int o = 0; o < CO; ++o) {
            const int ch = (out_group0 + g) * CO + o;
            out[(((int64_t)n * out_groups * CO + ch) * h + y) * w + xg] = acc[j][o] + bv[o];
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the fetch wave waited for the next patch)
  }
}

template <int CO, bool GM>
static int launch_grouped_pp(const _Float16* x, const _Float16* wg, const float* bias, int batch, int groups, int h, int w,
                             float* out, int out_groups, int out_group0, hipStream_t s) {
  constexpr int LDS = 3 * 10 * 40 * 128;
  const hipError_t e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(grouped_conv3x3_small_f16_pp_kernel<CO, GM>), LDS);
  if (e != hipSuccess) return (int)e;
  const int64_t tiles = ceil_div(w, 32) * ceil_div(h, 8), total = tiles * groups * batch;
  if (total >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  // one workgroup per CU where the work allows it, each with the same number of tiles (at least kGpTiles: the weights and
  // the first patch of a workgroup are not hidden)
  const int64_t per_wg = std::max<int64_t>(kGpTiles, ceil_div(total, 256));
  const int64_t xbytes = (int64_t)batch * h * w * groups * 64 * 2;
  grouped_conv3x3_small_f16_pp_kernel<CO, GM><<<(unsigned)ceil_div(total, per_wg), 320, LDS, s>>>(
      x, wg, bias, groups, h, w, out, out_groups, out_group0, (int)xbytes, (int)total, (int)per_wg);
  return launch_status();
}

// fp32 NCHW -> fp16 NHWC (the boundary in front of a chain of fp16 layers): one workgroup per (n, y, 64 columns),
// channels in chunks of 64 through an LDS tile (reads coalesced along x, writes along c)
__global__ __launch_bounds__(256) void f32_nchw_to_f16_nhwc_kernel(const float* __restrict__ x, int c, int h, int w,
                                                                   _Float16* __restrict__ out) {
  __shared__ float tile[64][65];
  const int tiles_x = (w + 63) / 64;
  const int bx = blockIdx.x % tiles_x, y = (blockIdx.x / tiles_x) % h, n = blockIdx.x / (tiles_x * h);
  const int x0 = bx * 64;
  const int tx = threadIdx.x & 63, tq = threadIdx.x >> 6;
  for (int c0 = 0; c0 < c; c0 += 64) {
    for (int r = tq; r < 64; r += 4) {
      const int ch = c0 + r, xx = x0 + tx;
      tile[r][tx] = (ch < c && xx < w) ? x[(((int64_t)n * c + ch) * h + y) * w + xx] : 0.f;
    }
    __syncthreads();
    for (int r = tq; r < 64; r += 4) {  // r = pixel, tx = channel
      const int xx = x0 + r, ch = c0 + tx;
      if (xx < w && ch < c) out[(((int64_t)n * h + y) * w + xx) * c + ch] = (_Float16)tile[tx][r];
    }
    __syncthreads();
  }
}

}  // namespace pd3

using namespace pd3;

template <int MB, int OUT_MODE>
static int launch_conv_f16(const void* x, const void* wp, const float* bias, int batch, int cin, int cout, int h, int w,
                           int relu, void* out, hipStream_t s, float* out2 = nullptr) {
  using S = CfShape<MB>;
  hipError_t e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(conv3x3_f16_kernel<MB, OUT_MODE>), (int)S::LDS);
  if (e != hipSuccess) return (int)e;
  const int64_t ptiles = (int64_t)batch * ceil_div(h, S::R) * ceil_div(w, kCfCols);
  const int nct = cout / S::M;
  const int64_t nslots = (ptiles + 7) / 8 * nct;  // work items per XCD lane (pixel tile group x channel tile)
  if (nslots * 8 >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  // items per workgroup: as many as leave >= 2 workgroups per CU, whole pixel tiles' worth of channel tiles where possible
  // (a patch of <= 64 input channels then stays resident over its channel tiles)
  int ipw = (int)std::min<int64_t>(16, std::max<int64_t>(1, nslots * 8 / 512));
  if (nct > 1 && ipw >= nct) ipw = ipw / nct * nct;
  else if (nct > 1 && nct % ipw != 0) {
    while (nct % ipw != 0) --ipw;
  }
  const int64_t nwg = 8 * ceil_div(nslots, ipw);
  conv3x3_f16_kernel<MB, OUT_MODE><<<(unsigned)nwg, kCfThreads, S::LDS, s>>>(
      static_cast<const _Float16*>(x), static_cast<const _Float16*>(wp), bias, out, cin, cout, h, w, relu, (int)ptiles,
      ipw, out2);
  return launch_status();
}

extern "C" int pd3_conv3x3_f16_bias_relu(const void* x_f16_nhwc, const void* w_packed_f16, const float* bias, int batch,
                                         int cin, int cout, int h, int w, int relu, void* out, int out_mode,
                                         int channels_per_tile, void* stream) {
  if (!x_f16_nhwc || !w_packed_f16 || !out || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return PD3_EINVAL;
  if ((out_mode != 0 && out_mode != 1 && out_mode != 3) || (channels_per_tile != 64 && channels_per_tile != 128))
    return PD3_EINVAL;
  if (reinterpret_cast<uintptr_t>(x_f16_nhwc) % 16 != 0 || reinterpret_cast<uintptr_t>(w_packed_f16) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out) % 16 != 0)
    return PD3_EINVAL;
  if (cin % kCfKc != 0 || cout % channels_per_tile != 0) return PD3_EUNSUPPORTED;
  if ((int64_t)h * w * cin >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;  // 32-bit staging offsets
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (channels_per_tile == 128)
    return out_mode == 0   ? launch_conv_f16<2, 0>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out, s)
           : out_mode == 1 ? launch_conv_f16<2, 1>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out, s)
                           : launch_conv_f16<2, 3>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out, s);
  return out_mode == 0   ? launch_conv_f16<1, 0>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out, s)
         : out_mode == 1 ? launch_conv_f16<1, 1>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out, s)
                         : launch_conv_f16<1, 3>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out, s);
}

extern "C" int pd3_conv3x3_f16_bias_relu_dual(const void* x_f16_nhwc, const void* w_packed_f16, const float* bias,
                                              int batch, int cin, int cout, int h, int w, int relu, void* out_f16_nhwc,
                                              float* out_f32_nchw, int channels_per_tile, void* stream) {
  if (!x_f16_nhwc || !w_packed_f16 || !out_f16_nhwc || !out_f32_nchw || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 ||
      w <= 0)
    return PD3_EINVAL;
  if (channels_per_tile != 64 && channels_per_tile != 128) return PD3_EINVAL;
  if (reinterpret_cast<uintptr_t>(x_f16_nhwc) % 16 != 0 || reinterpret_cast<uintptr_t>(w_packed_f16) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out_f16_nhwc) % 16 != 0 || reinterpret_cast<uintptr_t>(out_f32_nchw) % 16 != 0)
    return PD3_EINVAL;
  if (cin % kCfKc != 0 || cout % channels_per_tile != 0) return PD3_EUNSUPPORTED;
  if ((int64_t)h * w * cin >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (channels_per_tile == 128)
    return launch_conv_f16<2, 2>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out_f16_nhwc, s, out_f32_nchw);
  return launch_conv_f16<1, 2>(x_f16_nhwc, w_packed_f16, bias, batch, cin, cout, h, w, relu, out_f16_nhwc, s, out_f32_nchw);
}

template <int MW, bool GATHER>
static int launch_conv_s2_f16(const void* x, const int32_t* inv, const void* wp, const float* bias, int batch, int cin,
                              int cout, int h, int w, int relu, void* out, hipStream_t s) {
  using S = Cs2Shape<MW>;
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  hipError_t e = pd3_max_dynamic_lds(reinterpret_cast<const void*>(conv3x3_s2_f16_kernel<MW, GATHER>), (int)S::LDS);
  if (e != hipSuccess) return (int)e;
  const int64_t ptiles = (int64_t)batch * ceil_div(ho, kCs2R) * ceil_div(wo, kCfCols);
  const int64_t nwg = (ptiles + 7) / 8 * 8 * (cout / S::M);
  if (nwg >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  conv3x3_s2_f16_kernel<MW, GATHER><<<(unsigned)nwg, S::THREADS, S::LDS, s>>>(
      static_cast<const _Float16*>(x), inv, static_cast<const _Float16*>(wp), bias, static_cast<_Float16*>(out), cin, cout,
      h, w, ho, wo, relu, (int)ptiles);
  return launch_status();
}

extern "C" int pd3_conv3x3_s2_f16_bias_relu(const void* x_f16_nhwc, const void* w_packed_f16, const float* bias, int batch,
                                            int cin, int cout, int h, int w, int relu, void* out_f16_nhwc, void* stream) {
  if (!x_f16_nhwc || !w_packed_f16 || !out_f16_nhwc || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0)
    return PD3_EINVAL;
  if (reinterpret_cast<uintptr_t>(x_f16_nhwc) % 16 != 0 || reinterpret_cast<uintptr_t>(w_packed_f16) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out_f16_nhwc) % 8 != 0)
    return PD3_EINVAL;
  if (cin % kCfKc != 0 || cout % 128 != 0) return PD3_EUNSUPPORTED;
  if ((int64_t)h * w * cin >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  return launch_conv_s2_f16<2, false>(x_f16_nhwc, nullptr, w_packed_f16, bias, batch, cin, cout, h, w, relu,
                                      out_f16_nhwc, static_cast<hipStream_t>(stream));
}

extern "C" int pd3_scatter_conv3x3_s2_f16_bias_relu(const void* features_f16, const int32_t* inverse_map,
                                                    const void* w_packed_f16, const float* bias, int batch, int cin,
                                                    int cout, int ny, int nx, int relu, void* out_f16_nhwc,
                                                    int channels_per_tile, void* stream) {
  if (!features_f16 || !inverse_map || !w_packed_f16 || !out_f16_nhwc || batch <= 0 || cin <= 0 || cout <= 0 || ny <= 0 ||
      nx <= 0)
    return PD3_EINVAL;
  if (reinterpret_cast<uintptr_t>(features_f16) % 16 != 0 || reinterpret_cast<uintptr_t>(w_packed_f16) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(out_f16_nhwc) % 8 != 0)
    return PD3_EINVAL;
  if ((channels_per_tile != 64 && channels_per_tile != 128) || cin % kCfKc != 0 || cout % channels_per_tile != 0)
    return PD3_EUNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (channels_per_tile == 128)
    return launch_conv_s2_f16<2, true>(features_f16, inverse_map, w_packed_f16, bias, batch, cin, cout, ny, nx, relu,
                                       out_f16_nhwc, s);
  return launch_conv_s2_f16<1, true>(features_f16, inverse_map, w_packed_f16, bias, batch, cin, cout, ny, nx, relu,
                                     out_f16_nhwc, s);
}

static int grouped_small_f16(const void* x_f16_nhwc, const void* w_f16, const float* bias, int batch, int groups,
                             int channels_per_group, int out_per_group, int h, int w, float* out, int out_groups,
                             int out_group0, void* stream, bool group_major) {
  if (!x_f16_nhwc || !w_f16 || !out || batch <= 0 || groups <= 0 || h <= 0 || w <= 0 || out_groups < groups ||
      out_group0 < 0 || out_group0 + groups > out_groups)
    return PD3_EINVAL;
  if (channels_per_group != 64 || out_per_group < 1 || out_per_group > 4 || groups > 65535 || batch > 65535)
    return PD3_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(x_f16_nhwc) % 16 != 0 || reinterpret_cast<uintptr_t>(w_f16) % 16 != 0) return PD3_EINVAL;
  const dim3 grid((unsigned)(ceil_div(w, 32) * ceil_div(h, 8)), (unsigned)groups, (unsigned)batch);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const _Float16* x = static_cast<const _Float16*>(x_f16_nhwc);
  const _Float16* wg = static_cast<const _Float16*>(w_f16);
  if (group_major) {
    if ((int64_t)batch * h * w * groups * 64 * 2 >= (int64_t)kGpOob) return PD3_EUNSUPPORTED;  // 32-bit buffer offsets
    switch (out_per_group) {
      case 1: return launch_grouped_pp<1, true>(x, wg, bias, batch, groups, h, w, out, out_groups, out_group0, s);
      case 2: return launch_grouped_pp<2, true>(x, wg, bias, batch, groups, h, w, out, out_groups, out_group0, s);
      case 3: return launch_grouped_pp<3, true>(x, wg, bias, batch, groups, h, w, out, out_groups, out_group0, s);
      default: return launch_grouped_pp<4, true>(x, wg, bias, batch, groups, h, w, out, out_groups, out_group0, s);
    }
  }
  switch (out_per_group) {
    case 1: grouped_conv3x3_small_f16_kernel<1><<<grid, 256, 0, s>>>(x, wg, bias, groups, h, w, out, out_groups, out_group0); break;
    case 2: grouped_conv3x3_small_f16_kernel<2><<<grid, 256, 0, s>>>(x, wg, bias, groups, h, w, out, out_groups, out_group0); break;
    case 3: grouped_conv3x3_small_f16_kernel<3><<<grid, 256, 0, s>>>(x, wg, bias, groups, h, w, out, out_groups, out_group0); break;
    default: grouped_conv3x3_small_f16_kernel<4><<<grid, 256, 0, s>>>(x, wg, bias, groups, h, w, out, out_groups, out_group0); break;
  }
  return launch_status();
}

extern "C" int pd3_grouped_conv3x3_small_f16(const void* x_f16_nhwc, const void* w_f16, const float* bias, int batch,
                                             int groups, int channels_per_group, int out_per_group, int h, int w,
                                             float* out, int out_groups, int out_group0, void* stream) {
  return grouped_small_f16(x_f16_nhwc, w_f16, bias, batch, groups, channels_per_group, out_per_group, h, w, out, out_groups,
                           out_group0, stream, false);
}

extern "C" int pd3_grouped_conv3x3_small_f16_gm(const void* x_f16_group_major, const void* w_f16, const float* bias,
                                                int batch, int groups, int channels_per_group, int out_per_group, int h,
                                                int w, float* out, int out_groups, int out_group0, void* stream) {
  return grouped_small_f16(x_f16_group_major, w_f16, bias, batch, groups, channels_per_group, out_per_group, h, w, out,
                           out_groups, out_group0, stream, true);
}

extern "C" int pd3_f32_nchw_to_f16_nhwc(const float* x, int batch, int channels, int h, int w, void* out, void* stream) {
  if (!x || !out || batch <= 0 || channels <= 0 || h <= 0 || w <= 0) return PD3_EINVAL;
  const int64_t blocks = (int64_t)batch * h * ((w + 63) / 64);
  if (blocks >= (int64_t)1 << 31) return PD3_EUNSUPPORTED;
  f32_nchw_to_f16_nhwc_kernel<<<(unsigned)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(
      x, channels, h, w, static_cast<_Float16*>(out));
  return launch_status();
}
