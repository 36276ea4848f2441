"""Profiling driver (not a test): the AMP graph's fp16 stride-1 3x3 layers alone (pd3_conv3x3_f16_bias_relu), HIP-event
time and TFLOP/s per layer shape of CenterPoint-Pillars at `batch` frames.  usage: prof_conv_f16.py [batch] [iters] [shape]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from paddle3d_amd.ops import conv as C  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
only = sys.argv[3] if len(sys.argv) > 3 else ""
shapes = [("64to64@256", 64, 64, 256), ("128to128@128", 128, 128, 128), ("256to256@64", 256, 256, 64),
          ("384to64@128", 384, 64, 128), ("64to1152@128", 64, 1152, 128)]
torch.manual_seed(0)
for name, cin, cout, hw in shapes:
    if only and only != name:
        continue
    x = torch.randn(batch, hw, hw, cin, device="cuda").half()
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    b = torch.randn(cout, device="cuda")
    wp = C.pack_conv3x3_f16_weight(w)
    out = C.conv3x3_f16_bias_relu(x, wp, b, cout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        C.conv3x3_f16_bias_relu(x, wp, b, cout, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2 * cin * cout * 9 * hw * hw * batch
    print(f"{name:14s} batch {batch}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  = {fl / ms / 1e9 / 2500:.3f} of the fp16 pipe")
