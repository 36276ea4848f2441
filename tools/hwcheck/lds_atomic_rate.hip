// Throughput of wave-wide ds_add_rtn_u32 per CU on gfx950, as a function of the address pattern
// (distinct banks / random cells of a table / pairs and runs of equal addresses), 16 waves per CU.
// The voxelizer's group kernel ranks ~4.3 M records per batch with one such instruction per 64 records.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kIters = 64, kRep = 8;

__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ addr, unsigned* __restrict__ out, int table,
                                         int returning) {
  extern __shared__ unsigned cnt[];
  for (int d = threadIdx.x; d < table; d += 256) cnt[d] = 0;
  __syncthreads();
  unsigned a[kRep];
  for (int r = 0; r < kRep; ++r) a[r] = addr[(r * 256 + threadIdx.x)];
  unsigned acc = 0;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int r = 0; r < kRep; ++r) {
      if (returning) acc += atomicAdd(&cnt[a[r]], 1u);
      else atomicAdd(&cnt[a[r]], 1u);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc + cnt[threadIdx.x % table];
}

int main() {
  const int blocks = 1024, table = 2048;
  std::vector<unsigned> h(kRep * 256);
  unsigned *da, *dout;
  hipMalloc(&da, h.size() * 4);
  hipMalloc(&dout, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[] = {"lane-distinct banks", "random cells (2048)", "pairs of equal cells", "runs of 4 equal cells",
                         "runs of 16 equal", "all lanes one cell"};
  for (int mode = 0; mode < 6; ++mode) {
    srand(3);
    for (size_t i = 0; i < h.size(); ++i) {
      const int lane = i % 64;
      unsigned v;
      switch (mode) {
        case 0: v = (lane % 32) + 32 * (rand() % (table / 32)); break;
        case 1: v = rand() % table; break;
        case 2: v = 0; break;
        default: v = 0;
      }
      h[i] = v;
    }
    if (mode >= 2) {
      const int run = mode == 2 ? 2 : mode == 3 ? 4 : mode == 4 ? 16 : 64;
      for (size_t i = 0; i < h.size(); i += run) {
        const unsigned v = rand() % table;
        for (int j = 0; j < run; ++j) h[i + j] = v;
      }
    }
    hipMemcpy(da, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int returning = 1; returning >= 0; --returning) {
      k<<<blocks, 256, table * 4>>>(da, dout, table, returning);
      hipEventRecord(e0);
      k<<<blocks, 256, table * 4>>>(da, dout, table, returning);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double instr = (double)blocks * 4 * kIters * kRep;
      printf("%-24s %s: %.1f us, %.2f G wave-instr/s, %.1f G lane-ops/s\n", names[mode],
             returning ? "rtn  " : "nortn", ms * 1e3, instr / ms / 1e6, instr * 64 / ms / 1e6);
    }
  }
  return 0;
}
