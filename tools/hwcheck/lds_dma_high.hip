// Hardware check (gfx950): buffer_load_dwordx4 ... lds into LDS offsets beyond 64 KB (M0 carries the wave's base), the
// 16-byte lane stride of the dwordx4 form, and zeros for out-of-range lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(const float* x, float* out, int n, int base_floats) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  for (int i = threadIdx.x; i < 40000; i += 256) sm[i] = -7.f;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n * 4, 0x00020000);
  unsigned off = (wave * 64 + lane) * 16;
  if (lane == 5) off = 0x7ffffff0u;  // out of range -> zeros
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(sm + base_floats + wave * 256),
                                           16, off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) out[i] = sm[base_floats + i];
}
int main() {
  float *x, *out, h[1024], hx[1024];
  (void)hipMalloc(&x, 4096);
  (void)hipMalloc(&out, 4096);
  for (int i = 0; i < 1024; ++i) hx[i] = (float)i;
  (void)hipMemcpy(x, hx, 4096, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
  for (int base : {0, 12000, 20000, 30000, 38000}) {
    k<<<1, 256, 160000>>>(x, out, 1024, base);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, out, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) {
      const int ln = (i / 4) % 64;
      const float want = ln == 5 ? 0.f : (float)i;
      bad += h[i] != want;
    }
    printf("LDS base %6d B: %d mismatches (h[0..3] %g %g %g %g, lane 5: %g)\n", base * 4, bad, h[0], h[1], h[2], h[3], h[20]);
  }
  return 0;
}
