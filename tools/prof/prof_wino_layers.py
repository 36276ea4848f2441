"""The backbone's stride-1 layers (and the head's shared convolution) on the ping-pong Winograd kernel, 16 frames."""
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
for cin, cout, hw in ((64, 64, 256), (128, 128, 128), (256, 256, 64), (384, 64, 128)):
    x = torch.randn(16, cin, hw, hw, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (cin * 9) ** 0.5
    b = torch.randn(cout, device='cuda')
    ul = conv.pack_winograd43_lane_weight(w)
    t = timed(lambda: conv.conv3x3_winograd43_pp_bias_relu(x, ul, b, cout, True))
    print("%d->%d @%d: %.1f us" % (cin, cout, hw, t))
