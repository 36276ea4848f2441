"""Cycle stamps of one workgroup of the pipelined Winograd kernel (measurement, not a test)."""
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
dbg = torch.zeros(8, 8, 8, dtype=torch.int64, device="cuda")
for _ in range(2):
    check(lib().pd3_conv3x3_winograd43_pl_trace(ptr(x), ptr(ul), None, B, cin, cout, hw, hw, 1, ptr(out), ptr(dbg),
                                                stream_ptr(x.device)), "trace")
torch.cuda.synchronize()
d = dbg.cpu().tolist()
names = ["slot start", "at mid barrier", "past it", "MFMAs done", "past end barrier", "past V barrier"]
for wv in (0, 4, 3, 7):
    for sl in range(1, 4):
        print(f"wave {wv} slot {sl}: " + "  ".join(f"{n} {d[wv][sl][i]}" for i, n in enumerate(names)))
print("kernel end (wave 0):", d[0][7][6])
