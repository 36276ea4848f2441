"""Timing driver (not a test): the F(4x4,3x3) kernels on the stride-1 layer shapes of CenterPoint-Pillars, batch 16:
the packed form (waves 0-3 transform and multiply) and the ping-pong form of round 4 (multiply waves fed by LDS alone)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddle3d_amd.ops import conv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = [(64, 64, 256), (128, 128, 128), (256, 256, 64), (384, 64, 128), (64, 1152, 128)]
if len(sys.argv) > 3:
    shapes = shapes[1:2]
for cin, cout, hw in shapes:
    x = torch.randn(B, cin, hw, hw, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    b = torch.randn(cout, device="cuda")
    out = torch.empty(B, cout, hw, hw, device="cuda")
    out2 = torch.empty(B, cout, hw, hw, device="cuda")
    up = conv.pack_winograd43_weight(w, 64)
    forms = {"packed": lambda: conv.conv3x3_winograd43_bias_relu(x, up, b, cout, True, out=out)}
    ul = conv.pack_winograd43_lane_weight(w)
    forms["pingpong"] = lambda: conv.conv3x3_winograd43_pp_bias_relu(x, ul, b, cout, True, out=out2)
    res = {}
    for name, fn in list(forms.items()) * 4:
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) / 20)
    fl = 2.0 * B * hw * hw * cin * cout * 9
    diff = (out - out2).abs().max().item()
    line = f"{cin:4d}->{cout:4d} @{hw:3d}:"
    for name, ms in res.items():
        m = min(ms)
        line += f"  {name} {m:6.3f}"
    print(line + f"  max|packed - pingpong| {diff:.1e}")
