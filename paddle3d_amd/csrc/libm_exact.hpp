// sinf / cosf / expf / atanf / atan2f with the BITS glibc returns on x86-64, glibc 2.28 .. 2.40 (checked against
// 2.35's libm.so.6, see tools/libm_exact_check.cpp and tests/test_libm_exact.py).  SUPPORTED RANGE: 2.28 introduced
// the Arm Optimized Routines sinf / cosf / expf restated here; 2.41 replaces atanf / atan2f (and later more) by the
// CORE-MATH correctly rounded routines, which fdlibm's float code below does NOT match in ~1e-3 of the arguments --
// on such a host the reference's own CPU code returns different keep lists than on an older one, and the CPU test
// skips itself by gnu_get_libc_version() (the device results stay those of the 2.28 .. 2.40 libm).
//
// Why: the reference's CPU code evaluates cos / sin / atan2 / exp on floats through <math.h>
// (iou3d_cpu.cpp:77-79,128-129,164-165; decode_kernel postprocess.cu:151-160 when it is compiled for the host),
// i.e. through glibc's float routines, and bit-exact NMS keep lists need the same bits, not the correctly
// rounded value (which those routines miss in ~1e-3 of the arguments).  The routines are small and their
// arithmetic is reproducible on the device:
//   * sinf / cosf / expf are the Arm Optimized Routines algorithms (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c,
//     sincosf.h, s_sincosf_data.c, e_expf.c, math/e_exp2f_data.c): fp64 polynomials on a reduced argument, rounded
//     to fp32 once.  On every x86-64 CPU with FMA glibc runs its `-mfma -mavx2` build of these files
//     (sysdeps/x86_64/fpu/multiarch/s_sinf-fma.c ...), in which the compiler has contracted a*b+c into fused
//     operations; which ones is read off the disassembly of libm.so.6 and written out below as explicit fma()
//     calls (the surrounding code is compiled with -ffp-contract=off).  A host without FMA runs the unfused build,
//     whose results differ in a few arguments per billion; this file follows the FMA build.
//   * atanf / atan2f are fdlibm's float routines (s_atanf.c, e_atan2f.c), plain fp32 arithmetic in source order,
//     built without FMA on x86-64 (no multiarch variant).
// Everything here is exact IEEE arithmetic (fp64 / fp32 add, mul, div, fma, conversions), which v_fma_f64 & co.
// implement identically, so host-side equality with glibc carries over to the device.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PD3_HD __host__ __device__ __forceinline__
#else
#define PD3_HD static inline
#endif

namespace pd3 {
namespace lm {

PD3_HD uint32_t f2u(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  return u;
}
PD3_HD float u2f(uint32_t u) {
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
PD3_HD uint64_t d2u(double d) {
  uint64_t u;
  __builtin_memcpy(&u, &d, 8);
  return u;
}
PD3_HD double u2d(uint64_t u) {
  double d;
  __builtin_memcpy(&d, &u, 8);
  return d;
}

// ---- sinf / cosf ------------------------------------------------------------------------------------------
// s_sincosf_data.c: table[1] is table[0] with the cosine coefficients negated
struct SinCos {
  double c0, c1, c2, c3, c4, s1, s2, s3;
};
PD3_HD SinCos sincos_table(bool negated) {
  const double sg = negated ? -1.0 : 1.0;
  return SinCos{sg * 0x1p0,
                sg * -0x1.ffffffd0c621cp-2,
                sg * 0x1.55553e1068f19p-5,
                sg * -0x1.6c087e89a359dp-10,
                sg * 0x1.99343027bf8c3p-16,
                -0x1.555545995a603p-3,
                0x1.1107605230bc4p-7,
                -0x1.994eb3774cf24p-13};
}
PD3_HD double sincos_sign(int q) { return (q == 1 || q == 2) ? -1.0 : 1.0; }  // sign[4] = {1, -1, -1, 1}

// sincosf.h sinf_poly, with the fused operations of the FMA build
PD3_HD float sinf_poly(double x, double x2, const SinCos& p, int n) {
  if ((n & 1) == 0) {
    const double x3 = x2 * x;
    const double s1 = __builtin_fma(p.s3, x2, p.s2);
    const double x7 = x2 * x3;
    const double s = __builtin_fma(x3, p.s1, x);
    return (float)__builtin_fma(s1, x7, s);
  }
  const double x4 = x2 * x2;
  const double c1 = __builtin_fma(x2, p.c1, p.c0);
  const double c2 = __builtin_fma(x2, p.c4, p.c3);
  const double x6 = x2 * x4;
  const double c = __builtin_fma(x4, p.c2, c1);
  return (float)__builtin_fma(c2, x6, c);
}

PD3_HD uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ffu; }

// sincosf.h reduce_fast (|x| < 120): n = round(x * 2/pi), x - n * pi/2
PD3_HD double reduce_fast(double x, int& n) {
  const double r = x * 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
  n = ((int32_t)r + 0x800000) >> 24;
  return __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
}

// sincosf.h reduce_large: 192 bits of 4/pi around the bits of x that matter
PD3_HD uint32_t inv_pio4(int i) {
  const uint32_t t[24] = {0xa2u,       0xa2f9u,     0xa2f983u,   0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u,
                          0x6e4e4415u, 0x4e441529u, 0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u,
                          0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u,
                          0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
  return t[i];
}
PD3_HD double reduce_large(uint32_t xi, int& np) {
  const int a = (int)((xi >> 26) & 15u);
  const int shift = (int)((xi >> 23) & 7u);
  xi = (xi & 0xffffffu) | 0x800000u;
  xi <<= shift;
  uint64_t res0 = (uint64_t)(uint32_t)(xi * inv_pio4(a));
  const uint64_t res1 = (uint64_t)xi * inv_pio4(a + 4);
  const uint64_t res2 = (uint64_t)xi * inv_pio4(a + 8);
  res0 = (res2 >> 32) | (res0 << 32);
  res0 += res1;
  const uint64_t n = (res0 + (1ull << 61)) >> 62;
  res0 -= n << 62;
  np = (int)n;
  return (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
}

PD3_HD float sinf(float y) {
  double x = (double)y;
  int n;
  const uint32_t top = abstop12(y);
  if (top < 0x3f4u) {  // |y| < pi/4
    if (top < 0x398u) return y;  // |y| < 2^-12 (the subnormal branch only raises a flag)
    return sinf_poly(x, x * x, sincos_table(false), 0);
  }
  if (top < 0x42fu) {  // |y| < 120
    x = reduce_fast(x, n);
    const double s = sincos_sign(n & 3);
    return sinf_poly(x * s, x * x, sincos_table((n & 2) != 0), n);
  }
  if (top < 0x7f8u) {
    const uint32_t xi = f2u(y);
    const int sign = (int)(xi >> 31);
    x = reduce_large(xi, n);
    const double s = sincos_sign((n + sign) & 3);
    return sinf_poly(x * s, x * x, sincos_table(((n + sign) & 2) != 0), n);
  }
  return y - y;  // inf -> NaN, NaN -> NaN (__math_invalidf)
}

PD3_HD float cosf(float y) {
  double x = (double)y;
  int n;
  const uint32_t top = abstop12(y);
  if (top < 0x3f4u) {
    if (top < 0x398u) return 1.0f;
    return sinf_poly(x, x * x, sincos_table(false), 1);
  }
  if (top < 0x42fu) {
    x = reduce_fast(x, n);
    const double s = sincos_sign(n & 3);
    return sinf_poly(x * s, x * x, sincos_table((n & 2) != 0), n ^ 1);
  }
  if (top < 0x7f8u) {
    const uint32_t xi = f2u(y);
    const int sign = (int)(xi >> 31);
    x = reduce_large(xi, n);
    const double s = sincos_sign((n + sign) & 3);
    return sinf_poly(x * s, x * x, sincos_table(((n + sign) & 2) != 0), n ^ 1);
  }
  return y - y;
}

// ---- expf -------------------------------------------------------------------------------------------------
// math/e_exp2f_data.c: T[i] = bits(2^(i/32)) - (i << 47)
PD3_HD uint64_t exp2f_tab(int i) {
  const uint64_t t[32] = {
      0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
      0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
      0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
      0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
      0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
      0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
      0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
      0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
  return t[i];
}

// |x| >= 88 or NaN: the arguments expf treats before its polynomial
PD3_HD bool expf_is_special(float x) { return ((f2u(x) >> 20) & 0x7ffu) >= 0x42bu; }

// The polynomial part (every argument that is not special), with the table read through `tab` -- straight-line code,
// so a caller with several arguments in registers can have the table reads of all of them in flight at once (and
// keep the table where it likes: postprocess.hip holds it in LDS).  Harmless on a special argument (the index is
// masked), but the value is then not expf's.
template <class Tab>
PD3_HD float expf_main(float x, Tab tab) {
  const double xd = (double)x;
  const double kInvLn2N = 0x1.71547652b82fep+5, kShift = 0x1.8p+52;
  // e_expf.c with TOINT_INTRINSICS == 0; the FMA build fuses z = InvLn2N * xd into both of its uses
  double kd = __builtin_fma(kInvLn2N, xd, kShift);
  const uint64_t ki = d2u(kd);
  kd -= kShift;
  const double r = __builtin_fma(kInvLn2N, xd, -kd);
  const uint64_t t = tab((int)(ki & 31u)) + (ki << 47);
  const double s = u2d(t);
  const double z = __builtin_fma(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
  const double r2 = r * r;
  double y = __builtin_fma(r, 0x1.62e42ff0c52d6p-6, 1.0);
  y = __builtin_fma(z, r2, y);
  y = y * s;
  return (float)y;
}

PD3_HD float expf(float x) {
  if (expf_is_special(x)) {
    const uint32_t abstop = (f2u(x) >> 20) & 0x7ffu;
    if (f2u(x) == 0xff800000u) return 0.0f;
    if (abstop >= 0x7f8u) return x + x;
    if (x > 0x1.62e42ep6f) return u2f(0x7f800000u);  // overflow
    if (x < -0x1.9fe368p6f) return 0.0f;             // underflow
    if (x < -0x1.9d1d9ep6f) return u2f(1u);          // __math_may_uflowf: 0x1.4p-75f * 0x1.4p-75f
  }
  return expf_main(x, [](int i) { return exp2f_tab(i); });
}

// ---- atanf / atan2f (fdlibm) ------------------------------------------------------------------------------
PD3_HD float atanf(float x) {
  const float atanhi[4] = {u2f(0x3eed6338u), u2f(0x3f490fdau), u2f(0x3f7b985eu), u2f(0x3fc90fdau)};
  const float atanlo[4] = {u2f(0x31ac3769u), u2f(0x33222168u), u2f(0x33140fb4u), u2f(0x33a22168u)};
  const float aT[11] = {u2f(0x3eaaaaabu), u2f(0xbe4ccccdu), u2f(0x3e124925u), u2f(0xbde38e38u),
                        u2f(0x3dba2e6eu), u2f(0xbd9d8795u), u2f(0x3d886b35u), u2f(0xbd6ef16bu),
                        u2f(0x3d4bda59u), u2f(0xbd15a221u), u2f(0x3c8569d7u)};
  const int32_t hx = (int32_t)f2u(x);
  const int32_t ix = hx & 0x7fffffff;
  int id;
  if (ix >= 0x4c000000) {  // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;
    return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {  // |x| < 0.4375
    if (ix < 0x31000000) return x;  // |x| < 2^-29 (huge + x > one: always)
    id = -1;
  } else {
    x = u2f((uint32_t)ix);  // fabsf
    if (ix < 0x3f980000) {
      if (ix < 0x3f300000) {
        id = 0;
        x = (2.0f * x - 1.0f) / (2.0f + x);
      } else {
        id = 1;
        x = (x - 1.0f) / (x + 1.0f);
      }
    } else {
      if (ix < 0x401c0000) {
        id = 2;
        x = (x - 1.5f) / (1.0f + 1.5f * x);
      } else {
        id = 3;
        x = -1.0f / x;
      }
    }
  }
  const float z = x * x;
  const float w = z * z;
  const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return hx < 0 ? -r : r;
}

PD3_HD float atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = u2f(0x3f490fdbu), pi_o_2 = u2f(0x3fc90fdbu), pi = u2f(0x40490fdbu),
              pi_lo = u2f(0xb3bbbd2eu);
  const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return atanf(y);
  const int m = (int)(((uint32_t)hy >> 31) & 1u) | (int)(((uint32_t)hx >> 30) & 2u);
  if (iy == 0) {
    switch (m) {
      case 0:
      case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0f * pi_o_4 + tiny;
        default: return -3.0f * pi_o_4 - tiny;
      }
    }
    switch (m) {
      case 0: return 0.0f;
      case 1: return -0.0f;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int32_t k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = atanf(u2f(f2u(y / x) & 0x7fffffffu));
  switch (m) {
    case 0: return z;
    case 1: return u2f(f2u(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

}  // namespace lm
}  // namespace pd3
