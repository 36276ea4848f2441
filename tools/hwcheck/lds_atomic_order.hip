// Hardware property check (gfx950): for ONE wave-wide ds_add_rtn_u32, lanes that hit the same LDS address
// receive their "old" values in ascending lane order; successive instructions of a wave are ordered.
// voxelize_tiled.hpp's stable ranking relies on this (the parity tests would also catch a violation).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k(const unsigned* __restrict__ addr, unsigned* __restrict__ old, int rounds, int table) {
  extern __shared__ unsigned cnt[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned* c = cnt + wave * table;
  for (int d = lane; d < table; d += 64) c[d] = 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const size_t base = ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * rounds * 64;
  for (int r = 0; r < rounds; ++r) {
    const unsigned a = addr[base + r * 64 + lane];
    unsigned o = 0xFFFFFFFFu;
    if (a != 0xFFFFFFFFu) o = atomicAdd(&c[a], 1u);
    old[base + r * 64 + lane] = o;
  }
}

int main() {
  const int blocks = 512, waves = 8, rounds = 16, table = 1024;
  const size_t n = (size_t)blocks * waves * rounds * 64;
  std::vector<unsigned> h(n), o(n);
  srand(1);
  for (size_t i = 0; i < n; ++i) {
    const int mode = (i / (64 * rounds)) % 4;
    unsigned a = mode == 0 ? rand() % table : mode == 1 ? rand() % 4 : mode == 2 ? 7 : (rand() % 37) * 27 % table;
    if (rand() % 11 == 0) a = 0xFFFFFFFFu;
    h[i] = a;
  }
  unsigned *da, *dout;
  hipMalloc(&da, n * 4);
  hipMalloc(&dout, n * 4);
  hipMemcpy(da, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<blocks, waves * 64, waves * table * 4>>>(da, dout, rounds, table);
  hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
  size_t bad = 0;
  std::vector<unsigned> c(table);
  for (size_t w = 0; w < (size_t)blocks * waves; ++w) {
    std::fill(c.begin(), c.end(), 0u);
    for (int i = 0; i < rounds * 64; ++i) {
      const unsigned a = h[w * rounds * 64 + i];
      const unsigned want = a == 0xFFFFFFFFu ? 0xFFFFFFFFu : c[a]++;
      if (o[w * rounds * 64 + i] != want) ++bad;
    }
  }
  printf("lds_atomic_order: %zu mismatches of %zu\n", bad, n);
  return bad != 0;
}
