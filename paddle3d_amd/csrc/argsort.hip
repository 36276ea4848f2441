// Stable argsort on the library's radix sort (radix_sort.hpp), for the two places of the host glue that still called
// the tensor library's: rotate_nms_pcdet's descending score order (reference paddle3d/models/layers/layer_libs.py:
// 230-236: paddle.argsort(descending=True) + the pre_max_size cut) and the re-sort by ranks_feat in front of
// bev_pool_v2_bkwd (bevdet_transformer.py:60-68).  Ties keep their input order (the reference's argsort leaves them
// open; a stable order is the deterministic choice and what the oracle restatements use).
#include "../../include/paddle3d_amd.h"
#include "common.hpp"
#include "radix_sort.hpp"

namespace pd3 {

// mode 0: int32 keys >= 0, ascending.  mode 1: fp32 keys, descending (NaN first like a descending sort of torch /
// paddle puts it; -0 and +0 tie): u = sign ? ~bits : bits | 0x80000000 is ascending, ~u descending.
__global__ __launch_bounds__(256) void argsort_key_kernel(const uint32_t* __restrict__ in, int64_t n, int mode,
                                                          uint32_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t b = in[i];
  if (mode == 1) {
    if ((b & 0x7FFFFFFFu) == 0u) b = 0u;                   // -0 == +0: a tie, kept in input order
    if ((b & 0x7FFFFFFFu) > 0x7F800000u) b = 0x7FC00000u;   // every NaN sorts as the largest value
    const uint32_t asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    b = ~asc;
  }
  keys[i] = b;
}

__global__ __launch_bounds__(256) void argsort_out_kernel(const uint32_t* __restrict__ vals, int64_t n,
                                                          int32_t* __restrict__ order) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) order[i] = (int32_t)vals[i];
}

struct ArgsortWs {
  uint32_t *ka, *va, *kb, *vb;
  int *hist, *partial;
  size_t bytes;
};

static ArgsortWs argsort_carve(void* base, int64_t n, const RadixPlan& p) {
  Carver c(base);
  ArgsortWs w;
  w.ka = c.take<uint32_t>((size_t)n);
  w.va = c.take<uint32_t>((size_t)n);
  w.kb = c.take<uint32_t>((size_t)n);
  w.vb = c.take<uint32_t>((size_t)n);
  w.hist = c.take<int>(radix_hist_ints(p));
  w.partial = c.take<int>((size_t)scan_num_tiles((int64_t)radix_hist_ints(p)));
  w.bytes = c.off;
  return w;
}

}  // namespace pd3

using namespace pd3;

extern "C" size_t pd3_stable_argsort_workspace(int64_t n, uint32_t max_key) {
  if (n <= 0) return 256;
  return argsort_carve(nullptr, n, radix_plan(max_key, n)).bytes;
}

extern "C" int pd3_stable_argsort(const void* keys, int64_t n, int mode, uint32_t max_key, int32_t* order,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  if (n < 0 || n >= ((int64_t)1 << 31) || (mode != 0 && mode != 1)) return PD3_EINVAL;
  if (n == 0) return 0;
  if (!keys || !order || !workspace) return PD3_EINVAL;
  if (mode == 1) max_key = 0xFFFFFFFFu;
  const RadixPlan plan = radix_plan(max_key, n);
  ArgsortWs w = argsort_carve(workspace, n, plan);
  if (workspace_bytes < w.bytes) return PD3_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned blocks = (unsigned)ceil_div(n, 256);
  argsort_key_kernel<<<blocks, 256, 0, s>>>(static_cast<const uint32_t*>(keys), n, mode, w.ka);
  const int where = enqueue_radix_sort(w.ka, w.va, w.kb, w.vb, n, n, 1, plan, /*identity_vals=*/true, w.hist, w.partial, s);
  argsort_out_kernel<<<blocks, 256, 0, s>>>(where ? w.vb : w.va, n, order);
  return launch_status();
}
