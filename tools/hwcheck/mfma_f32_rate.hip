// Hardware check (gfx950): issue rate of v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x2_f32 from registers, one and two
// waves per SIMD, 36 / 8 independent accumulators, optionally with one ds_read_b128 per four MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>  // 0: 16x16x4 regs only; 1: 16x16x4 + LDS b128 read per 4; 2: 32x32x2 regs only
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  __shared__ float lds[512 * 36];
  for (int i = threadIdx.x; i < 512 * 36; i += blockDim.x) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float a = 1.0f + lane * 1e-3f, b = 0.5f - lane * 1e-3f;
  long long t0 = 0, t1 = 0;
  if (MODE < 2) {
    f4 acc[36];
#pragma unroll
    for (int c = 0; c < 36; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
      const float* vb = lds + ((lane >> 4) * 16 + (lane & 15)) * 36;
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        f4 bv = {b, b, b, b};
        if (MODE == 1) bv = *reinterpret_cast<const f4*>(vb + g * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[g * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[j], acc[g * 4 + j], 0, 0, 0);
      }
    }
    t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 36; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    f16v acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += acc[c][0] + acc[c][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
  if (blockIdx.x == 0 && lane == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 64);
  const int iters = 200;
  for (int mode = 0; mode < 3; ++mode)
    for (int threads : {256, 512}) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) k<0><<<256, threads>>>(out, cyc, iters);
        if (mode == 1) k<1><<<256, threads>>>(out, cyc, iters);
        if (mode == 2) k<2><<<256, threads>>>(out, cyc, iters);
        hipDeviceSynchronize();
      }
      long long h[8];
      hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
      const int per = mode < 2 ? 36 : 8;
      printf("mode %d (%s), %d waves/SIMD: %.1f cycles per MFMA per wave (wave 0), flops/clk/SIMD %.1f\n", mode,
             mode == 0 ? "16x16x4 regs" : mode == 1 ? "16x16x4 + ds_read_b128 per 4" : "32x32x2 regs", threads / 256,
             (double)h[0] / (iters * per), (double)(mode < 2 ? 2048 : 4096) * (threads / 256) * iters * per / (double)h[0]);
    }
  return 0;
}
