"""Timing driver (not a test): the F(4x4,3x3) kernel on the stride-1 layer shapes of CenterPoint-Pillars, batch 16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddle3d_amd.ops import conv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for cin, cout, hw in [(64, 64, 256), (128, 128, 128), (256, 256, 64), (384, 64, 128), (64, 2304, 128)]:
    x = torch.randn(B, cin, hw, hw, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    b = torch.randn(cout, device="cuda")
    out = torch.empty(B, cout, hw, hw, device="cuda")
    up = conv.pack_winograd43_weight(w, 64)
    for _ in range(2):
        conv.conv3x3_winograd43_bias_relu(x, up, b, cout, True, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        conv.conv3x3_winograd43_bias_relu(x, up, b, cout, True, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * B * hw * hw * cin * cout * 9
    print(f"{cin:4d}->{cout:4d} @{hw:3d}: {ms:7.3f} ms  {fl / ms / 1e9 / 4:6.1f} TFLOP/s executed  {fl / ms / 1e9:6.1f} direct-form")
