"""Phase cycle counters of one workgroup of the ping-pong Winograd kernel (measurement, not a test)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddle3d_amd._lib import lib  # noqa: E402
from paddle3d_amd.ops import conv  # noqa: E402
from paddle3d_amd.ops._common import check, ptr, stream_ptr  # noqa: E402

B, cin, cout, hw = 16, 128, 128, 128
x = torch.randn(B, cin, hw, hw, device="cuda")
w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
out = torch.empty(B, cout, hw, hw, device="cuda")
ul = conv.pack_winograd43_lane_weight(w)
for _once in (0,):
    dbg = torch.zeros(8, 4, dtype=torch.int64, device="cuda")
    for _ in range(2):
        check(lib().pd3_conv3x3_winograd43_pp_trace(ptr(x), ptr(ul), None, B, cin, cout, hw, hw, 1, ptr(out),
                                                     ptr(dbg), stream_ptr(x.device)), "trace")
    torch.cuda.synchronize()
    d = dbg.cpu().tolist()
    print(f"slots {cin // 8} per group; per wave [transform, multiply, barrier wait, kernel] cycles:")
    ns = cin // 8
    for wv, r in enumerate(d[:8]):
        print(f"  wave {wv} (group {wv >> 2}): T {r[0]:7d} ({r[0] // (cin // 8):5d}/slot)  M {r[1]:7d} ({r[1] // (cin // 8):5d}/slot)  "
              f"barrier {r[2]:7d}  total {r[3] & ((1 << 56) - 1):7d}  SIMD {r[3] >> 56}")
