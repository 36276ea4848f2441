// Hardware check (gfx950): the skeleton of the ping-pong Winograd kernel.  Waves 0-3 and 4-7 (one of each per SIMD)
// alternate between a multiply phase (72 v_mfma_f32_16x16x4_f32 fed by ds_read_b128 through a ring) and a transform-like
// phase (18 LDS reads, wait, NV VALU operations in three chains, 9 ds_write_b64), one workgroup barrier per time slot.
// Prints, per role, the cycles of each phase when the partner is idle (SOLO) and when it runs the opposite phase.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NV, bool SOLO, bool RING, int MODE = 0, int NR = 18, int NW = 9>  // NR / NW: LDS reads / b64 writes of T; MODE bits: 1 no LDS reads in T, 2 no LDS writes in T, 4 no clocks, 8 T at priority 3,
                                                       // 16 s_nop after every MFMA, 32 T reads via one ds_read_b128-free path (b64)
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int slots) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 30000; i += blockDim.x) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, cb = wave & 3;
  f4 acc[36];
#pragma unroll
  for (int c = 0; c < 36; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + lane * 1e-3f;
  float keep = 0.f;
  long long tT = 0, tM = 0;
  const float* vb = lds + 8000 + grp * 4608 + ((lane >> 4) * 16 + (lane & 15)) * 36;
  float* raw = lds + grp * 3456 + cb * 864 + (lane >> 1) * 4 + (lane & 1) * 3;
  float* vout = lds + 8000 + grp * 4608 + (cb * 64 + lane) * 18;
  auto T = [&]() {
    const long long c0 = (MODE & 4) ? 0 : clock64();
    if (MODE & 8) __builtin_amdgcn_s_setprio(3);
    float r[18];
#pragma unroll
    for (int q = 0; q < 18; ++q) r[q] = q < NR ? raw[(q / 3) * 72 + (q % 3)] : a * q;
    float x0 = r[0] + r[3] + r[6] + r[9] + r[12] + r[15], x1 = r[1] + r[4] + r[7] + r[10] + r[13] + r[16],
          x2 = r[2] + r[5] + r[8] + r[11] + r[14] + r[17];
#pragma unroll
    for (int q = 0; q < NV / 3; ++q) {
      x0 = __builtin_fmaf(x0, 1.0001f, x1);
      x1 = __builtin_fmaf(x1, 0.9999f, x2);
      x2 = __builtin_fmaf(x2, 1.0002f, x0);
    }
    {
#pragma unroll
      for (int q = 0; q < NW; ++q) *reinterpret_cast<f2*>(vout + q * 2) = (f2){x0 + q, x1 - q};
    }
    keep += x2 + x0 + x1;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (MODE & 8) __builtin_amdgcn_s_setprio(0);
    if (!(MODE & 4)) tT += clock64() - c0;
  };
  auto M = [&]() {
    const long long c0 = (MODE & 4) ? 0 : clock64();
    f4 b[3];
    if (MODE & 64) {  // a VALU stream of about the same length instead of MFMAs
      float f[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) f[c] = a * (c + 1);
      for (int it = 0; it < 72; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = __builtin_fmaf(f[q], 1.0001f, a);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) keep += f[c];
      if (!(MODE & 4)) tM += clock64() - c0;
      return;
    }
    if (MODE & 32) {
      b[0] = b[1] = b[2] = (f4){a, a * 2, a * 3, a * 4};
    } else {
      b[0] = *reinterpret_cast<const f4*>(vb);
      b[1] = *reinterpret_cast<const f4*>(vb + 4);
    }
#pragma unroll
    for (int g = 0; g < 18; ++g) {
      if (RING && !(MODE & 32) && g + 2 < 18) b[(g + 2) % 3] = *reinterpret_cast<const f4*>(vb + (g / 9) * 2304 + ((g + 2) % 9) * 4);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        {
        acc[(g % 9) * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[g % 3][j], acc[(g % 9) * 4 + j], 0, 0, 0);
        if (MODE & 16) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0); }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(MODE & 4)) tM += clock64() - c0;
  };
  if (grp == 0) {
    for (int s = 0; s < slots; ++s) {
      T();
      __syncthreads();
      M();
      __syncthreads();
    }
    __syncthreads();
  } else {
    __syncthreads();
    for (int s = 0; s < slots; ++s) {
      if (!SOLO) T();
      __syncthreads();
      if (!SOLO) M();
      __syncthreads();
    }
  }
  float sres = keep;
#pragma unroll
  for (int c = 0; c < 36; ++c) sres += acc[c][0] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sres;
  if (blockIdx.x == 0 && lane == 0) cyc[wave * 2] = tT, cyc[wave * 2 + 1] = tM;
}

template <int NV, bool SOLO, bool RING, int MODE = 0, int NR = 18, int NW = 9>
void run(float* out, long long* cyc) {
  const int slots = 50;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<NV, SOLO, RING, MODE, NR, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            140000);
  long long t = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<NV, SOLO, RING, MODE, NR, NW><<<256, 512, 140000>>>(out, cyc, slots);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    t = (long long)(ms * 1e6);
  }
  long long h[16];
  (void)hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
  printf("NV %3d NR %2d NW %d %s %s mode %2d: group 0 T %5.0f M %5.0f | group 1 T %5.0f M %5.0f cycles per slot; kernel %lld ns = %.0f ns per time slot\n",
         NV, NR, NW, SOLO ? "solo   " : "partner", RING ? "ring" : "regs", MODE, (double)h[0] / slots, (double)h[1] / slots,
         (double)h[8] / slots, (double)h[9] / slots, t, (double)t / (2 * slots + 1));
}

int main() {
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&cyc, 128);
  // the transform-like phase alone and beside the other group's MFMA stream
  run<150, true, true>(out, cyc);
  run<150, false, true>(out, cyc);
  run<30, false, true>(out, cyc);
  run<300, false, true>(out, cyc);
  // which part of it is held up: LDS reads only (18 / 8 / 4 / 2), LDS writes only, VALU only
  run<0, true, true, 0, 18, 0>(out, cyc);
  run<0, false, true, 0, 18, 0>(out, cyc);
  run<0, false, true, 0, 8, 0>(out, cyc);
  run<0, false, true, 0, 4, 0>(out, cyc);
  run<0, false, true, 0, 2, 0>(out, cyc);
  run<0, true, true, 0, 2, 9>(out, cyc);
  run<0, false, true, 0, 2, 9>(out, cyc);
  run<150, true, true, 0, 2, 1>(out, cyc);
  run<150, false, true, 0, 2, 1>(out, cyc);
  // the multiply phase without any LDS read of its own; a VALU stream of the same length in its place
  run<0, false, true, 32, 18, 0>(out, cyc);
  run<0, false, true, 64, 18, 0>(out, cyc);
  run<150, false, true, 64>(out, cyc);
  // the transform-like phase at wave priority 3
  run<150, false, true, 8>(out, cyc);
  return 0;
}
