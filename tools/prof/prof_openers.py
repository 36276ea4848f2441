"""The two stride-2 block openers of SECOND at CenterPoint-Pillars size, 16 frames: bf16x3 kernel against the fp32 implicit GEMM."""
import sys, torch
sys.path.insert(0, '/root/repo')
from paddle3d_amd.ops import conv

torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for cin, cout, hw in ((64, 128, 256), (128, 256, 128), (128, 256, 180)):
    x = torch.randn(n, cin, hw, hw, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (cin * 9) ** 0.5
    b = torch.randn(cout, device='cuda')
    w32, wx3 = conv.pack_conv3x3_weight(w), conv.pack_conv3x3_s2_x3_weight(w)
    y32 = conv.conv3x3_bias_relu(x, w32, b, cout, stride=2)
    yx3 = conv.conv3x3_s2_x3_bias_relu(x, wx3, b, cout)
    t32 = timed(lambda: conv.conv3x3_bias_relu(x, w32, b, cout, stride=2))
    tx3 = timed(lambda: conv.conv3x3_s2_x3_bias_relu(x, wx3, b, cout))
    gf = 2 * n * (hw // 2) ** 2 * cin * cout * 9 / 1e9
    print("%d->%d @%d: fp32 %.1f us (%.0f TF), bf16x3 %.1f us (%.0f TF); max |diff| %.2e of %.1f" % (
        cin, cout, hw, t32, gf / t32 * 1e3, tx3, gf / tx3 * 1e3, (y32 - yx3).abs().max().item(), y32.abs().max().item()))
