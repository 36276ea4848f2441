// Hardware property the default hard_voxelize path rests on (voxelize_wave.hpp:125,296,345; voxelize_tiled.hpp:17-18):
// for ONE wave-wide returning LDS add (ds_add_rtn_u32), lanes that hit the same LDS word receive their "old" values
// in ASCENDING LANE ORDER, and successive LDS instructions of a wave execute in program order -- that is what makes
// the slot a point gets equal to the number of earlier points of its cell (the reference's sequential scan,
// voxelize_op.cc:47-78) without a sort.  The ISA documents neither; tools/hwcheck/lds_atomic_order.hip measured it
// (0 mismatches in 4.2 M on gfx950).  This entry point runs the same probe through the library, compiled with the
// library's flags, so that tests/test_voxelize_gpu.py::test_lds_atomic_lane_order can assert it on every GPU test run
// and a caller can assert it at start-up on a new part or driver.
#include "../../include/paddle3d_amd.h"
#include "common.hpp"

namespace pd3 {

__global__ void lds_atomic_order_kernel(const uint32_t* __restrict__ addr, uint32_t* __restrict__ old, int rounds,
                                        int table) {
  extern __shared__ uint32_t sc_cnt[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t* c = sc_cnt + wave * table;
  for (int d = lane; d < table; d += 64) c[d] = 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const size_t base = ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * rounds * 64;
  for (int r = 0; r < rounds; ++r) {
    const uint32_t a = addr[base + r * 64 + lane];
    uint32_t o = 0xFFFFFFFFu;
    if (a != 0xFFFFFFFFu) o = atomicAdd(&c[a], 1u);  // the form the voxelizer uses: divergent lanes skip the add
    old[base + r * 64 + lane] = o;
  }
}

}  // namespace pd3

extern "C" int pd3_selfcheck_lds_atomic_order(const uint32_t* addr, uint32_t* old, int blocks, int waves_per_block,
                                              int rounds, int table, void* stream) {
  if (!addr || !old || blocks <= 0 || waves_per_block <= 0 || waves_per_block > 16 || rounds <= 0 || table <= 0 ||
      (size_t)waves_per_block * table * 4 > 64 * 1024)
    return PD3_EINVAL;
  pd3::lds_atomic_order_kernel<<<(unsigned)blocks, waves_per_block * 64, (size_t)waves_per_block * table * 4,
                                 static_cast<hipStream_t>(stream)>>>(addr, old, rounds, table);
  return pd3::launch_status();
}
