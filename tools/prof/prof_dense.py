"""Profiling driver (not a test): the dense BEV graph (SECOND + FPN + CenterHead) of CenterPoint-Pillars on a
batch of random pseudo-images, timed per layer with HIP events, plus its deviation from the torch statement
of the same layers (oracle/pyoracle.py, MIOpen on the GPU).  usage: prof_dense.py [batch] [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
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
    print(f"dense graph batch={batch}: {total:.3f} ms  ({127.2e9 * batch / total / 1e9:.1f} TFLOP/s direct-form)")
    outs, cur = [], x
    for i, layers in enumerate(model.backbone._plan()):
        for j, conv in enumerate(layers):
            nxt, ms = timed(lambda t: conv(t)[0], cur)
            fl = 2 * conv.w.numel() * nxt.shape[2] * nxt.shape[3] * batch
            print(f"  block{i} conv{j} {tuple(cur.shape[1:])}->{tuple(nxt.shape[1:])} stride {conv.stride}: {ms:.3f} ms "
                  f"{fl / ms / 1e9:.1f} TF")
            cur = nxt
        outs.append(cur)
    cat, ms = timed(model.neck, outs)
    print(f"  FPN (patch GEMMs into the concatenated map): {ms:.3f} ms")
    head = model.bbox_head
    f = head._plan()
    from paddle3d_amd.ops import conv as _conv
    s, ms0 = timed(lambda t: f["shared"](t)[0], cat)
    y, ms1 = timed(lambda t: f["first"](t)[0], s)
    z, ms2 = timed(lambda t: _conv.grouped_conv3x3_small(t, f["pf"], f["bf"], f["groups"]), y)
    print(f"  head shared 384->64: {ms0:.3f} ms; first stage 64->{f['first'].cout}: {ms1:.3f} ms; final grouped: {ms2:.3f} ms")
    xs = x[:2]
    a = model.dense_forward(xs)
    b = O.dense_forward_torch(model, xs)
    pa, _ = model.bbox_head(a)
    pb, _ = O.center_head_torch(model.bbox_head, b)
    dmax = max(float((pa[t][k] - pb[t][k]).abs().max()) for t in range(len(pa)) for k in pa[t])
    print(f"  neck output: max|hip - torch| {float((a - b).abs().max()):.2e} (max|ref| {float(b.abs().max()):.2f}); "
          f"head outputs: max|diff| {dmax:.2e}")
