"""Phase cycle counters of one workgroup of the ping-pong Winograd kernel (measurement, not a test)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddle3d_amd._lib import lib  # noqa: E402
from paddle3d_amd.ops._common import check, ptr, stream_ptr  # noqa: E402

B, cin, cout, hw = 16, 128, 128, 128
x = torch.randn(B, cin, hw, hw, device="cuda")
w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
out = torch.empty(B, cout, hw, hw, device="cuda")
for variant in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "4,0")]:
    dbg = torch.zeros(16, 4, dtype=torch.int64, device="cuda")
    for _ in range(2):
        check(lib().pd3_conv3x3_winograd43_raw_trace(ptr(x), ptr(w), None, B, cin, cout, hw, hw, 1, ptr(out), variant,
                                                     ptr(dbg), stream_ptr(x.device)), "trace")
    torch.cuda.synchronize()
    d = dbg.cpu().tolist()
    print(f"variant {variant}: slots {cin // 8} per group; per wave [transform, multiply, barrier wait, kernel] cycles:")
    ns = cin // 8
    for wv in range(8):
        e = d[8 + wv]
        print(f"  wave {wv} M split per slot: U(0)+fetch {e[0] // ns:5d}  trip0+roll {e[1] // ns:5d}  trip1 {e[2] // ns:5d}  stash {e[3] // ns:5d}")
    for wv, r in enumerate(d[:8]):
        print(f"  wave {wv} (group {wv >> 2}): T {r[0]:7d} ({r[0] // (cin // 8):5d}/slot)  M {r[1]:7d} ({r[1] // (cin // 8):5d}/slot)  "
              f"barrier {r[2]:7d}  total {r[3] & ((1 << 56) - 1):7d}  SIMD {r[3] >> 56}")
