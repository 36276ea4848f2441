// Host-side check of paddle3d_amd/csrc/libm_exact.hpp against the libm of this machine (test infrastructure).
//   g++ -O2 -mfma -ffp-contract=off -pthread tools/libm_exact_check.cpp -o /tmp/libm_exact_check
//   /tmp/libm_exact_check [stride]      stride 1 = all 2^32 floats per unary function (minutes on 8 cores)
// Prints one line per function: arguments checked, mismatches (bit-for-bit; NaN matches NaN), first mismatch.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../paddle3d_amd/csrc/libm_exact.hpp"

using namespace pd3::lm;

static bool same(float a, float b) { return (isnan(a) && isnan(b)) || f2u(a) == f2u(b); }

template <typename F, typename G>
static void unary(const char* name, F ours, G ref, uint64_t stride, int threads) {
  std::atomic<uint64_t> bad{0}, first{UINT64_MAX};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      uint64_t mybad = 0;
      for (uint64_t u = (uint64_t)t * stride; u < (1ull << 32); u += stride * threads) {
        volatile float x = u2f((uint32_t)u);
        if (!same(ours(x), ref(x))) {
          ++mybad;
          uint64_t f = first.load();
          while (u < f && !first.compare_exchange_weak(f, u)) {}
        }
      }
      bad += mybad;
    });
  for (auto& th : pool) th.join();
  const uint64_t n = ((1ull << 32) + stride - 1) / stride;
  printf("%s: %llu arguments, %llu mismatches", name, (unsigned long long)n, (unsigned long long)bad.load());
  if (bad.load()) {
    const float x = u2f((uint32_t)first.load());
    printf(" (first: x = %a, ours %a, libm %a)", x, ours(x), ref(x));
  }
  printf("\n");
}

int main(int argc, char** argv) {
  const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
  const int threads = (int)std::thread::hardware_concurrency();
  unary("sinf", [](float x) { return pd3::lm::sinf(x); }, [](float x) { return ::sinf(x); }, stride, threads);
  unary("cosf", [](float x) { return pd3::lm::cosf(x); }, [](float x) { return ::cosf(x); }, stride, threads);
  unary("expf", [](float x) { return pd3::lm::expf(x); }, [](float x) { return ::expf(x); }, stride, threads);
  unary("atanf", [](float x) { return pd3::lm::atanf(x); }, [](float x) { return ::atanf(x); }, stride, threads);
  // atan2f: every pair of a set of special / boundary values, then pseudo-random pairs of all magnitudes and
  // pairs shaped like the reference's use (differences of BEV coordinates)
  std::vector<float> sp;
  const uint32_t bits[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x80000001u, 0x007fffffu, 0x00800000u, 0x3f800000u,
                           0xbf800000u, 0x3f7fffffu, 0x3f800001u, 0x7f7fffffu, 0xff7fffffu, 0x7f800000u, 0xff800000u,
                           0x7fc00000u, 0x40490fdbu, 0x3fc90fdbu, 0x3f000000u, 0x3ee00000u, 0x3f300000u, 0x3f980000u,
                           0x401c0000u, 0x4c000000u, 0x31000000u, 0x5e800000u, 0x1e800000u};
  for (uint32_t b : bits) sp.push_back(u2f(b));
  uint64_t n2 = 0, bad2 = 0;
  float fy = 0, fx = 0;
  auto chk = [&](float y, float x) {
    volatile float vy = y, vx = x;
    ++n2;
    if (!same(pd3::lm::atan2f(vy, vx), ::atan2f(vy, vx))) {
      if (!bad2) fy = y, fx = x;
      ++bad2;
    }
  };
  for (float y : sp)
    for (float x : sp) chk(y, x);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
  };
  const uint64_t pairs = (1ull << 30) / stride + 1000000;
  for (uint64_t i = 0; i < pairs; ++i) {
    const uint64_t r = rnd();
    chk(u2f((uint32_t)r), u2f((uint32_t)(r >> 32)));
    const uint64_t q = rnd();
    chk((float)((int64_t)(q & 0xFFFFF) - 0x80000) * 1.52587890625e-05f,
        (float)((int64_t)((q >> 32) & 0xFFFFF) - 0x80000) * 1.52587890625e-05f);
  }
  printf("atan2f: %llu pairs, %llu mismatches", (unsigned long long)n2, (unsigned long long)bad2);
  if (bad2) printf(" (first: y = %a, x = %a, ours %a, libm %a)", fy, fx, pd3::lm::atan2f(fy, fx), ::atan2f(fy, fx));
  printf("\n");
  return 0;
}
