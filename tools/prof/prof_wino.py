"""Profiling driver (not a test): one Winograd convolution layer shape in a loop; use under rocprofv3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddle3d_amd.ops import conv  # noqa: E402

cin, cout, hw, B = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (128, 128, 128, 8)))
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
x = torch.randn(B, cin, hw, hw, device="cuda")
w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
b = torch.randn(cout, device="cuda")
algo = os.environ.get("PROF_ALGO", "43")
out = torch.empty(B, cout, hw, hw, device="cuda")
if algo == "43":
    up = conv.pack_winograd43_weight(w)
    fn = conv.conv3x3_winograd43_bias_relu
else:
    up = conv.pack_winograd_weight(w)
    fn = conv.conv3x3_winograd_bias_relu
for _ in range(iters):
    fn(x, up, b, cout, True, out=out)
torch.cuda.synchronize()
print("done")
