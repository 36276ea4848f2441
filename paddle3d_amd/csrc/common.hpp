// Shared helpers for the gfx950 kernels.  Everything here is device/host glue with no policy.
//
// Build flags that matter for parity (see paddle3d_amd/build.py): -ffp-contract=off (the reference CPU
// path is plain IEEE fp32 without FMA contraction) and hipcc's default correctly-rounded fp32 divide.
#pragma once
#include <mutex>
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pd3 {

constexpr int kWave = 64;  // CDNA wavefront

#define PD3_OK 0
#define PD3_EINVAL (-1)    // bad argument
#define PD3_EWORKSPACE (-2)  // workspace too small
#define PD3_EUNSUPPORTED (-3)

// Returns the sticky launch error of the last kernel launch as a positive hipError_t (0 if none).
static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// hipFuncAttributeMaxDynamicSharedMemorySize for kernel `fn`, raised when `bytes` exceeds what this process has already
// asked for on the current device -- NOT set again at every launch.  Round 6: with a captured HIP graph alive, calling
// hipFuncSetAttribute for one of its kernels again (the eager form of the same operator, a few dozen times) made the
// graph's later replays compute with wrong launch state (detections lost, then memory faults; tools/prof/
// graph_replay_check.py reproduces it with the per-launch calls) -- and a call per launch is host time for nothing.
static inline hipError_t pd3_max_dynamic_lds(const void* fn, int bytes) {
  struct Entry {
    const void* fn;
    int dev, bytes;
  };
  static std::mutex mu;
  static std::vector<Entry> seen;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  for (Entry& x : seen)
    if (x.fn == fn && x.dev == dev) {
      if (bytes <= x.bytes) return hipSuccess;
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e == hipSuccess) x.bytes = bytes;
      return e;
    }
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) seen.push_back({fn, dev, bytes});
  return e;
}

__host__ __device__ static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller-provided workspace; all carve-outs are 256-byte aligned.
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + off);
    off = align_up(off + count * sizeof(T), 256);
    return p;
  }
};

// Tile of a sparse gather-GEMM block.  Blocks go to the 8 XCDs round-robin (block b runs on XCD b % 8, each XCD with its own
// 4 MB L2): the `per_win` tiles of one 8192-row window (the unit of the tile order: a few neighbouring lines of the sorted
// rows, whose tiles gather from the same three z-slabs) all run on ONE XCD, consecutive windows on consecutive XCDs -- the
// window's input rows are fetched into one L2 instead of into eight, and the eight XCDs still walk neighbouring windows at
// the same time (shared through the Infinity Cache).  Measured (profiles/r05_sparse_tilemap.txt): feature kernels of the
// CenterPoint-Voxel encoder 22.8 -> 22.1 ms (fp32 / bf16x3), 9.73 -> 9.06 ms (fp16); groups of 16, 64 or 128 tiles, and
// one contiguous eighth of all tiles per XCD, are slower than block order.
// Launch with sp_window_grid(tiles, per_win) blocks; -1 = nothing to do for this block.
__device__ __forceinline__ int sp_window_tile(int block, int tiles, int per_win) {
  const int xcd = block & 7, slot = block >> 3;
  const int t = ((slot / per_win) * 8 + xcd) * per_win + slot % per_win;
  return t < tiles ? t : -1;
}
static inline unsigned sp_window_grid(long long tiles, int per_win) {
  const long long group = 8LL * per_win;
  return (unsigned)((tiles + group - 1) / group * group);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// Inclusive scan of one int across the 64 lanes of a wave.
__device__ __forceinline__ int wave_inclusive_scan(int v) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    int n = __shfl_up(v, d, kWave);
    if (lane_id() >= d) v += n;
  }
  return v;
}

// Exclusive scan of one int per thread across a block of THREADS (multiple of 64, <= 1024).
// `total` receives the block sum.  `smem` must hold THREADS/64 + 1 ints.
template <int THREADS>
__device__ __forceinline__ int block_exclusive_scan(int v, int* smem, int& total) {
  constexpr int W = THREADS / kWave;
  const int inc = wave_inclusive_scan(v);
  if (lane_id() == kWave - 1) smem[wave_id()] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      int t = smem[w];
      smem[w] = run;
      run += t;
    }
    smem[W] = run;
  }
  __syncthreads();
  const int out = smem[wave_id()] + inc - v;
  total = smem[W];
  __syncthreads();  // smem may be reused by the caller right away
  return out;
}

}  // namespace pd3
