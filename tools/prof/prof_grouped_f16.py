"""The AMP form of the 36 final SeparateHead convolutions at CenterPoint-Pillars size (16 frames, 128 x 128, two slices of
18 groups): the persistent LDS-DMA kernel on the group-major form of the input against the register-staged one on NHWC."""
import os, sys, torch
sys.path.insert(0, '/root/repo')
from paddle3d_amd.ops import conv
torch.manual_seed(0)
n, groups, co = 16, 18, 3
x = torch.randn(n, 128, 128, groups * 64, device='cuda').half()
wt = torch.randn(groups * co, 64, 3, 3, device='cuda') / 24.0
b = torch.randn(groups * co, device='cuda')
wp = conv.pack_grouped_weight_f16(wt, groups)
out = torch.empty(n, groups * co, 128, 128, device='cuda')
def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = {}
xg = x.view(n, 128, 128, groups, 64).permute(0, 3, 1, 2, 4).contiguous()
for gm in (False, True):
    xx = xg if gm else x
    conv.grouped_conv3x3_small_f16(xx, wp, b, groups, out=out, out_groups=groups, out_group0=0, group_major=gm)
    res[gm] = out.clone()
    t = timed(lambda: conv.grouped_conv3x3_small_f16(xx, wp, b, groups, out=out, out_groups=groups, out_group0=0, group_major=gm))
    print(("group-major input, persistent LDS-DMA kernel" if gm else "NHWC input, register-staged kernel          "),
          "%.1f us per slice of 18 groups (%.2f TB/s of input)" % (t, x.numel() * 2 / t / 1e6))
print("identical bytes:", torch.equal(res[True], res[False]))
