// How should a wave gather ROWS (a sparse convolution's neighbours, an NHWC patch's pixels)?  Same rows, same bytes, three
// lane -> address maps (MI355X).  Rows of ROWB bytes, a wave-instruction loads 16 B per lane (1 KB):
//   A  lane = row: the 64 lanes of an instruction touch 64 different rows (16 B each); the row's other pieces come from the
//      wave's following instructions (what sparse_conv_x3 / the round-5 fp16 convolution do)
//   B  8 lanes = one row's 128 B: an instruction touches 8 rows, whole 128-byte lines
//   C  as B through buffer_load ... lds (no VGPR destination): the LDS-DMA form
// Indices: a window-local random walk like a rulebook's (rows of a tile's neighbours lie within a few thousand rows).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// every wave gathers `per_wave` rows x 128 B, 32 rows per step (a 32-row MFMA block)
template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ src, const int* __restrict__ idx, int rowf,
                                                int per_wave, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wave;
  const int* my = idx + (size_t)gw * per_wave;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  float* slot = lds + wave * (32 * 32 * 2);  // 2 x 4 KB per wave
  for (int r0 = 0; r0 < per_wave; r0 += 32) {
    if (MODE == 0) {  // lane = (row l & 31, half l >> 5): 64 B of the row's 128 as four 16-byte loads
      const int row = my[r0 + (lane & 31)];
      const f4* p = reinterpret_cast<const f4*>(src + (size_t)row * rowf + (lane >> 5) * 16);
      f4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
      acc += v0 + v1 + v2 + v3;
    } else if (MODE == 1) {  // lane = (row l >> 3, piece l & 7): four instructions of 8 rows each
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = my[r0 + q * 8 + (lane >> 3)];
        acc += *reinterpret_cast<const f4*>(src + (size_t)row * rowf + (lane & 7) * 4);
      }
    } else {  // the same through LDS-DMA, then each lane reads 64 B back
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
      float* dst = slot + ((r0 >> 5) & 1) * (32 * 32);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = my[r0 + q * 8 + (lane >> 3)];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + q * 256), 16,
                                                 (unsigned)(((size_t)row * rowf + (lane & 7) * 4) * 4), 0, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
      const f4* p = reinterpret_cast<const f4*>(dst + (lane & 31) * 32 + (lane >> 5) * 16);
      acc += p[0] + p[1] + p[2] + p[3];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <typename F> float timeit(F f, int iters = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters * 1e3f;  // us
}

int main() {
  for (int rowf : {32, 64, 128}) {          // floats per row: 128 / 256 / 512 B (the chunk fetched is always the first 128 B)
    const int nrows = 1400000;               // rows of a stage-3 feature matrix
    const int waves = 256 * 8 * 4, per_wave = 2048;  // 16.8 M row fetches = 2.1 GB gathered
    float *src, *sink; int* didx;
    CK(hipMalloc(&src, (size_t)nrows * rowf * 4)); CK(hipMalloc(&sink, 256)); CK(hipMalloc(&didx, (size_t)waves * per_wave * 4));
    CK(hipMemset(src, 0, (size_t)nrows * rowf * 4));
    std::vector<int> h((size_t)waves * per_wave);
    srand(5);
    for (int w = 0; w < waves; ++w) {
      const long base = (long)((double)w / waves * (nrows - 9000));
      for (int i = 0; i < per_wave; ++i) h[(size_t)w * per_wave + i] = (int)(base + rand() % 8192);
    }
    CK(hipMemcpy(didx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const double gb = (double)waves * per_wave * 128 / 1e9;
    float t0 = timeit([&] { k_gather<0><<<waves / 4, 256, 0>>>(src, didx, rowf, per_wave, sink); });
    float t1 = timeit([&] { k_gather<1><<<waves / 4, 256, 0>>>(src, didx, rowf, per_wave, sink); });
    float t2 = timeit([&] { k_gather<2><<<waves / 4, 256, 4 * 2 * 4096>>>(src, didx, rowf, per_wave, sink); });
    printf("rows of %3d B, %.2f GB gathered (128 B per row fetch): A lane-per-row %.0f us = %.2f TB/s | B 8 lanes per row %.0f us = %.2f TB/s | "
           "C LDS-DMA 8 lanes per row %.0f us = %.2f TB/s\n", rowf * 4, gb, t0, gb / t0 * 1e3, t1, gb / t1 * 1e3, t2, gb / t2 * 1e3);
    CK(hipFree(src)); CK(hipFree(didx)); CK(hipFree(sink));
  }
  return 0;
}
