"""Profiling driver (not a test): the dense BEV graph (SECOND + FPN + CenterHead) of CenterPoint-Pillars on a
batch of random pseudo-images, timed per section with HIP events.  PD3_DENSE_BACKEND selects miopen / hip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddle3d_amd import centerpoint as cpm  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
model = cpm.centerpoint_pillars_nuscenes().cuda().eval()
x = torch.randn(batch, 64, 512, 512, device="cuda")


def timed(fn, *a):
    for _ in range(2):
        out = fn(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn(*a)
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / iters


with torch.no_grad():
    _, total = timed(lambda t: model.bbox_head(model.dense_forward(t)), x)
    print(f"dense backend={model.dense_backend} batch={batch}: {total:.3f} ms  ({127.2e9 * batch / total / 1e9:.1f} TFLOP/s)")
    outs, cur = [], x
    for i, blk in enumerate(model._dense[0]):
        for j, layer in enumerate(blk):
            nxt, ms = timed(lambda t: model._run([layer], t), cur)
            fl = 2 * layer[1].numel() * nxt.shape[2] * nxt.shape[3] * batch
            print(f"  block{i} conv{j} {tuple(cur.shape[1:])}->{tuple(nxt.shape[1:])} stride {layer[3]}: {ms:.3f} ms {fl / ms / 1e9:.1f} TF")
            cur = nxt
        outs.append(cur)
    ups = []
    for i, (d, o) in enumerate(zip(model._dense[1], outs)):
        u, ms = timed(lambda t: model._run(d, t), o)
        print(f"  neck{i} {tuple(o.shape[1:])}->{tuple(u.shape[1:])}: {ms:.3f} ms")
        ups.append(u)
    cat, ms = timed(lambda: torch.cat(ups, dim=1))
    print(f"  concat: {ms:.3f} ms")
    fz, ms = timed(lambda: model._neck_fused(outs))
    if fz is not None:
        print(f"  fused FPN (3 patch GEMMs into the concat buffer): {ms:.3f} ms, max|diff| vs unfused {float((fz - cat).abs().max()):.2e}")
    head = model.bbox_head
    f = head._fused
    from paddle3d_amd.ops import conv as _conv
    import torch.nn.functional as F
    if head.dense_backend == "hip":
        s, ms0 = timed(lambda t: cpm._hip_conv3x3(t, f["w0"], f["b0"], 1, f["packed"]), cat)
        y, ms1 = timed(lambda t: cpm._hip_conv3x3(t, f["w1"], f["b1"], 1, f["packed"]), s)
        z, ms2 = timed(lambda t: _conv.grouped_conv3x3_small(t, f["pf"], f["bf"], f["groups"]), y)
    else:
        s, ms0 = timed(lambda t: F.relu(F.conv2d(t, f["w0"], f["b0"], padding=1)), cat)
        y, ms1 = timed(lambda t: F.relu(F.conv2d(t, f["w1"], f["b1"], padding=1)), s)
        z, ms2 = timed(lambda t: F.conv2d(t, f["wf"], f["bf"], padding=1, groups=f["groups"]), y)
    print(f"  head shared 384->64: {ms0:.3f} ms; first stage 64->{f['w1'].shape[0]}: {ms1:.3f} ms; final grouped: {ms2:.3f} ms")
    _, ms3 = timed(lambda t: head(t), cat)
    print(f"  head total (incl. slicing): {ms3:.3f} ms")
    # end-to-end deviation of the hand-written dense graph from the PyTorch-ROCm (MIOpen) one
    import copy
    ref_model = cpm.centerpoint_pillars_nuscenes().cuda().eval()
    ref_model.load_state_dict(model.state_dict())
    ref_model.dense_backend = "miopen"
    ref_model.bbox_head.dense_backend = "miopen"
    xs = x[:2]
    a = model.dense_forward(xs)
    b = ref_model.dense_forward(xs)
    pa, _ = model.bbox_head(a)
    pb, _ = ref_model.bbox_head(b)
    dmax = max(float((pa[t][k] - pb[t][k]).abs().max()) for t in range(len(pa)) for k in pa[t])
    print(f"  neck output: max|hip - miopen| {float((a - b).abs().max()):.2e} (max|ref| {float(b.abs().max()):.2f}); "
          f"head outputs: max|diff| {dmax:.2e}")
