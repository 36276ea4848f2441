"""SecondFPN (three kernel = stride levels) at CenterPoint-Pillars size, 16 frames: bf16x3 kernel against the fp32-MFMA kernel."""
import sys, torch
sys.path.insert(0, '/root/repo')
from paddle3d_amd import centerpoint as cpm
from paddle3d_amd.ops import conv

torch.manual_seed(0)
model = cpm.centerpoint_pillars_nuscenes(max_num_voxels=(30000, 30000)).cuda().eval()
fpn = model.neck
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
xs = [torch.randn(n, 64, 256, 256, device='cuda'), torch.randn(n, 128, 128, 128, device='cuda'),
      torch.randn(n, 256, 64, 64, device='cuda')]


def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


outs = {}
with torch.no_grad():
    for flag in (False, True):
        conv.PATCH_BF16X3 = flag
        outs[flag] = fpn(xs)
        print("bf16x3" if flag else "fp32  ", "whole FPN: %.1f us" % timed(lambda: fpn(xs)))
    plan, ctot = fpn._plan()
    out = torch.empty_like(outs[True])
    for p, x in zip(plan, xs):
        t32 = timed(lambda: conv.patch_conv_bias_relu(x, p["w"], p["b"], p["mode"], p["cout"], out, p["off"]))
        tx3 = timed(lambda: conv.patch_conv_x3_bias_relu(x, p["wx3"], p["b"], p["mode"], p["cout"], out, p["off"]))
        print("mode %d %d->%d: fp32 %.1f us, bf16x3 %.1f us" % (p["mode"], p["cin"], p["cout"], t32, tx3))
d = (outs[True] - outs[False]).abs().max().item()
print("max |bf16x3 - fp32 kernel| %.3e, max |value| %.2f" % (d, outs[False].abs().max().item()))
