"""CenterHead's first stage (64 -> 2304 as two slices of 1152) at CenterPoint-Pillars size, 16 frames: the ping-pong Winograd
kernel against the form with the input transform computed once (conv_winograd43_ppv.hip)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from paddle3d_amd.ops import conv
torch.manual_seed(0)
def timed(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for cin, cout, hw in ((64, 1152, 128), (256, 256, 64), (128, 128, 128)):
    x = torch.randn(16, cin, hw, hw, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (cin * 9) ** 0.5
    b = torch.randn(cout, device='cuda')
    ul = conv.pack_winograd43_lane_weight(w)
    y0 = conv.conv3x3_winograd43_pp_bias_relu(x, ul, b, cout, True)
    v = conv.winograd43_input_transform(x)
    y1 = conv.conv3x3_winograd43_ppv_bias_relu(v, x.shape, ul, b, cout, True)
    t0 = timed(lambda: conv.conv3x3_winograd43_pp_bias_relu(x, ul, b, cout, True))
    tv = timed(lambda: conv.winograd43_input_transform(x, out=v))
    t1 = timed(lambda: conv.conv3x3_winograd43_ppv_bias_relu(v, x.shape, ul, b, cout, True))
    print("%d->%d @%d: pp %.1f us; input transform %.1f us (%.0f MB) + ppv %.1f us; identical bytes: %s" % (
        cin, cout, hw, t0, tv, v.numel() * 4 / 1e6, t1, torch.equal(y0, y1)))
