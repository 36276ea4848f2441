// Shared pieces of the dense bf16x3 kernels (conv_patch_x3.hip, conv_s2_x3.hip): fp32 arithmetic on the bf16 matrix cores
// with every operand as three bf16 pieces (sparse_conv_x3.hip explains the scheme), operands fetched into LDS by
// buffer_load ... lds, results stored through buffer addressing.
#pragma once
#include "common.hpp"

namespace pd3 {

typedef __bf16 px_b8 __attribute__((ext_vector_type(8)));
typedef float px_f32x16 __attribute__((ext_vector_type(16)));
typedef float px_f32x4 __attribute__((ext_vector_type(4)));
typedef float px_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int px_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void px_split(const px_f32x4 lo4, const px_f32x4 hi4, px_b8& h, px_b8& m, px_b8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = e < 4 ? lo4[e & 3] : hi4[e & 3];
    const __bf16 a = (__bf16)x;
    const float r1 = x - (float)a;
    const __bf16 b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    h[e] = a;
    m[e] = b;
    l[e] = (__bf16)r2;
  }
}

// Buffer addressing: address = base + (lane's 32-bit offset) + (uniform 32-bit offset in an SGPR).  A lane offset of kPxOob
// is out of the buffer's range: its fetch delivers zeros, its store is dropped.
// (Plain functions, not lambdas of the kernel template: hipcc 7.2 drops the host stub of a kernel template whose body feeds
// a buffer resource from dependent expressions -- see conv_f16.hip.)
constexpr unsigned kPxOob = 0x7ffffff0u;
// 64 lanes x 16 bytes from base + voff + soff to lds .. lds + 1023 (lane l at lds + 16 l)
__device__ __forceinline__ void px_dma(const void* base, unsigned bytes, void* lds, unsigned voff, unsigned soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, (int)soff, 0, 0);
}
// 64 lanes x 4 bytes to lds .. lds + 255 (lane l at lds + 4 l)
__device__ __forceinline__ void px_dma4(const void* base, unsigned bytes, void* lds, unsigned voff, unsigned soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 4, (int)voff, (int)soff, 0, 0);
}
__device__ __forceinline__ void px_st1(float* base, unsigned bytes, unsigned voff, unsigned soff, float v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void px_st2(float* base, unsigned bytes, unsigned voff, unsigned soff, float v0, float v1) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
  const px_u32x2 v = {__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1)};
  __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)voff, (int)soff, 0);
}
// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt and lgkmcnt left at their maxima)
// (+ a compiler barrier: the LDS reads behind it must stay behind it)
#define PX_VMCNT(N)                                                              \
  do {                                                                           \
    __builtin_amdgcn_s_waitcnt(0x0f70 | ((N) & 15) | (((N) >> 4) << 14));        \
    asm volatile("" ::: "memory");                                               \
  } while (0)
// LDS traffic only; fetches in flight travel across it (a __syncthreads would drain every one of them: a fetch into LDS
// counts as an LDS store)
__device__ __forceinline__ void px_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Items per workgroup of the persistent kernels: one workgroup per CU at a time walks `ipw` consecutive slots of its XCD
// lane as one stream, the grid is 8 * ceil(nslots / ipw) workgroups.  Chosen to minimise (rounds of 256 workgroups) x (ipw
// + the half item a workgroup's unhidden first fetches cost): "as many as fill the CUs once" left 136 workgroups of 16
// items for CenterPoint-Voxel's 264 pixel tiles (16 item times where 8.25 are the work).
static inline int px_items_per_workgroup(int64_t nslots) {
  int best = 1;
  double best_cost = 1e30;
  for (int ipw = 1; ipw <= 32; ++ipw) {
    const int64_t nwg = 8 * ((nslots + ipw - 1) / ipw);
    const double cost = (double)((nwg + 255) / 256) * (ipw + 0.5);
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = ipw;
    }
  }
  return best;
}

}  // namespace pd3
