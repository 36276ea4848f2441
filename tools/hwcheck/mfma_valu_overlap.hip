// Hardware check (gfx950): how much independent VALU work fits behind one v_mfma_f32_16x16x4_f32 (32 matrix-pipe cycles)
// of the same wave -- K fmas on private registers after every MFMA, fenced in place -- alone on the SIMD and beside a
// second wave that runs VALU work only (the transform role of the ping-pong Winograd kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int K, int PARTNER, int SWAP = 0, int MLDS = 0>  // PARTNER: 0 none, 1 the other four waves run a VALU loop, 2 a VALU + LDS loop,
                                                     // 3 nine ds_read2_b32 + wait + 8 fmas; SWAP: MFMAs on waves 4-7; MLDS: + ds_read_b128 per 4 MFMAs
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  __shared__ float lds[64 * 36 * 8];
  for (int i = threadIdx.x; i < 64 * 36 * 8; i += blockDim.x) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float a = 1.0f + lane * 1e-3f, b = 0.5f - lane * 1e-3f;
  long long t0 = 0, t1 = 0;
  if ((wave < 4) != (SWAP != 0)) {
    f4 acc[36];
    float f[8];
#pragma unroll
    for (int c = 0; c < 36; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = a * (c + 1);
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < 36; ++c) {
        if (MLDS && c % 4 == 0) {
          const f4 v = *reinterpret_cast<const f4*>(lds + ((wave & 3) * 64 + lane) * 36 + c);
          b = v[0] + v[3];
        }
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < K; ++q) f[(c * K + q) % 8] = __builtin_fmaf(f[(c * K + q) % 8], 1.0001f, b);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 36; ++c) s += acc[c][0] + acc[c][3];
#pragma unroll
    for (int c = 0; c < 8; ++c) s += f[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else if (PARTNER) {
    float f[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = a * (c + 1);
    t0 = clock64();
    for (int it = 0; it < iters * 36; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] = __builtin_fmaf(f[q], 1.0001f, b);
      if (PARTNER == 3) {
        float r = 0.f;
#pragma unroll
        for (int q = 0; q < 18; ++q) r += lds[(wave & 3) * 2304 + lane * 2 + q * 72 + (it & 3)];
        f[2] += r;
      }
      if (PARTNER == 2) {
        f[0] += lds[(wave * 64 + lane) * 36 + (it & 31)];
        lds[(wave * 64 + lane) * 36 + ((it + 7) & 31)] = f[1];
      }
    }
    t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += f[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
  if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int K, int P, int SWAP = 0, int MLDS = 0>
void run(float* out, long long* cyc) {
  const int iters = 100;
  for (int rep = 0; rep < 2; ++rep) {
    k<K, P, SWAP, MLDS><<<256, 512>>>(out, cyc, iters);
    (void)hipDeviceSynchronize();
  }
  long long h[8];
  (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("K %d fmas behind every v_mfma_f32_16x16x4_f32 (36 accumulators, one wave per SIMD): %.1f cycles per MFMA\n", K,
         (double)h[SWAP ? 4 : 0] / (iters * 36));
}

int main() {
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&cyc, 64);
  // (the partner modes are for use under a profiler: the partner loop outlives the MFMA wave, so its mean says nothing
  // about the overlap -- tools/hwcheck/pingpong_skeleton.hip measures that with barrier-bounded phases)
  run<0, 0>(out, cyc);
  run<1, 0>(out, cyc);
  run<2, 0>(out, cyc);
  run<3, 0>(out, cyc);
  run<4, 0>(out, cyc);
  run<6, 0>(out, cyc);
  run<8, 0>(out, cyc);
  return 0;
}
