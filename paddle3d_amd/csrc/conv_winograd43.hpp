// Shared pieces of the F(4x4, 3x3) kernels (conv_winograd43.hip: the packed form; conv_winograd43_pp.hip: the ping-pong
// form of round 4): tile constants and the transform steps.
#pragma once
#include "common.hpp"

namespace pd3 {

typedef float w4_f32x4 __attribute__((ext_vector_type(4)));
typedef float w4_f32x2 __attribute__((ext_vector_type(2)));

constexpr int kW4Ci = 4;                  // input channels per trip (= MFMA K)
constexpr int kW4TR = 2, kW4TC = 16;      // tile rows / columns per workgroup (4x4 outputs each)
constexpr int kW4Cs = 36;                 // components per element
constexpr int kW4RawR = 4 * kW4TR + 2;    // 10 staged input rows
constexpr int kW4RawW = 4 * kW4TC + 8;    // 72 staged input columns: x0-4 .. x0+67
constexpr int kW4RawPl = kW4RawR * kW4RawW;                  // 720
constexpr int kW4Vsz = kW4TR * kW4Ci * kW4TC * kW4Cs;        // 4608 floats
constexpr int kW4RawSz = kW4Ci * kW4RawPl;                   // 2880 floats
constexpr int kW4XN4 = kW4RawSz / 4;                         // 720 float4

// B^T applied to six values (one column or one row of the patch)
__device__ __forceinline__ void w4_in(const float d0, const float d1, const float d2, const float d3, const float d4,
                                      const float d5, float (&t)[6]) {
  const float a = __builtin_fmaf(-4.f, d2, d4), b = __builtin_fmaf(-4.f, d1, d3);
  const float c = d4 - d2, e = d3 - d1;
  t[0] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
  t[1] = a + b;
  t[2] = a - b;
  t[3] = __builtin_fmaf(2.f, e, c);
  t[4] = __builtin_fmaf(-2.f, e, c);
  t[5] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
}

// A^T applied to six values -> four
__device__ __forceinline__ void w4_out(const float m0, const float m1, const float m2, const float m3, const float m4,
                                       const float m5, float (&s)[4]) {
  const float p = m1 + m2, q = m1 - m2, r = m3 + m4, u = m3 - m4;
  s[0] = m0 + p + r;
  s[1] = __builtin_fmaf(2.f, u, q);
  s[2] = __builtin_fmaf(4.f, r, p);
  s[3] = __builtin_fmaf(8.f, u, q) + m5;
}

__device__ __forceinline__ float w4_swap_pair(float v) {  // value of the neighbouring lane (lane ^ 1)
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() makes the compiler drain vmcnt as well -- a pending
// buffer_load ... lds counts as a store to LDS -- which puts the full latency of every fetch in flight in front of the
// barrier.  The kernels that use this wait for exactly the fetches a barrier has to publish (explicit s_waitcnt vmcnt(N))
// and let the others travel across it (LDS-DMA requests stay in flight across s_barrier).
__device__ __forceinline__ void w4_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace pd3
