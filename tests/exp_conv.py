"""Experiment driver (not a test): conv3x3 MFMA kernel vs MIOpen on the dense-graph layer shapes."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddle3d_amd.ops import conv  # noqa: E402

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


for cin, cout, hw in [(64, 64, 256), (128, 128, 128), (256, 256, 64), (384, 64, 128), (64, 2304, 128)]:
    x = torch.randn(B, cin, hw, hw, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    b = torch.randn(cout, device="cuda")
    wp = conv.pack_conv3x3_weight(w)
    out = torch.empty(B, cout, hw, hw, device="cuda")
    up = conv.pack_winograd_weight(w)
    out2 = torch.empty(B, cout, hw, hw, device="cuda")
    t_mi = timeit(lambda: F.relu_(F.conv2d(x, w, b, padding=1))) if os.environ.get("EXP_MIOPEN", "1") == "1" else float("nan")
    t_me = timeit(lambda: conv.conv3x3_bias_relu(x, wp, b, cout, True, out=out))
    t_wg = timeit(lambda: conv.conv3x3_winograd_bias_relu(x, up, b, cout, True, out=out2))
    u43 = conv.pack_winograd43_weight(w)
    out3 = torch.empty(B, cout, hw, hw, device="cuda")
    t_43 = timeit(lambda: conv.conv3x3_winograd43_bias_relu(x, u43, b, cout, True, out=out3))
    e43 = (out3 - out).abs().max().item()
    fl = 2 * cin * cout * 9 * hw * hw * B / 1e12
    print(f"   F(4x4,3x3) {t_43:7.3f} ms ({fl / t_43 * 1e3:6.1f} TF eff)  max|diff vs direct| {e43:.2e}")
    err = (out2 - out).abs().max().item()
    print(f"cin {cin:4d} cout {cout:4d} hw {hw:3d}: miopen {t_mi:7.3f} ms ({fl / t_mi * 1e3:6.1f} TF)  "
          f"direct {t_me:7.3f} ms ({fl / t_me * 1e3:6.1f} TF)  winograd {t_wg:7.3f} ms ({fl / t_wg * 1e3:6.1f} TF eff)  "
          f"max|wino-direct| {err:.2e}")
