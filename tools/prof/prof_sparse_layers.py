"""Per-layer timing of the sparse encoder of CenterPoint-Voxel (not a test): rows, kernel volume, channels,
existing (row, offset) pairs, (16-row block, offset) pairs the matrix-core kernel executes, time, TFLOP/s."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddle3d_amd import centerpoint as cpm  # noqa: E402
from paddle3d_amd import synth  # noqa: E402
from paddle3d_amd.ops import sparse_conv3d as _sp  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
# second argument "raster": rows in raster order (no tile order), the round-4 behaviour
_sp.TILE_ORDER = not (len(sys.argv) > 2 and sys.argv[2] == "raster")
AMP = len(sys.argv) > 2 and sys.argv[2] == "amp"  # third form: the fp16 kernel where it applies
# fourth form "fp32": the fp32 matrix-core kernel in every layer (round 4 / early round 5); the default runs the bf16x3 form
_sp.SPLIT_BF16 = not (len(sys.argv) > 2 and sys.argv[2] in ("fp32", "raster"))
model = cpm.centerpoint_voxels_nuscenes(max_num_voxels=(120000, 160000)).cuda().eval()
pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(100 + i) for i in range(batch)])).cuda()
rows = []
_feat = _sp.features


def traced(in_feats, idx, weight, *a, **kw):
    for _ in range(2):
        _feat(in_feats, idx, weight, *a, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = _feat(in_feats, idx, weight, *a, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    nbr = idx.nbr[: idx.n_out]
    present = nbr >= 0
    pairs = int(present.sum())
    if idx.order is not None and _sp.TILE_ORDER:  # the blocks the kernel really forms: 16 consecutive slots of the order
        slots = idx.order.long()
        live = slots >= 0
        present = torch.where(live.unsqueeze(1), present[slots.clamp(min=0)], torch.zeros_like(present[:1]))
    n16 = (present.shape[0] + 15) // 16 * 16
    pad = torch.zeros(n16 - present.shape[0], nbr.shape[1], dtype=torch.bool, device=nbr.device)
    blocks = int(torch.cat([present, pad]).view(-1, 16, nbr.shape[1]).any(1).sum()) * 16
    cin, cout = int(weight.shape[-2]), int(weight.shape[-1])
    rows.append((idx.n_out, nbr.shape[1], cin, cout, pairs, blocks, ms))
    return out


_sp.features = traced
_feat16 = _sp.features_f16


def traced16(in_feats, idx, packed, cin, cout, *a, **kw):
    for _ in range(2):
        _feat16(in_feats, idx, packed, cin, cout, *a, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = _feat16(in_feats, idx, packed, cin, cout, *a, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    present = idx.nbr[: idx.n_out] >= 0
    pairs = int(present.sum())
    if idx.order is not None and _sp.TILE_ORDER:
        slots = idx.order.long()
        present = torch.where((slots >= 0).unsqueeze(1), present[slots.clamp(min=0)], torch.zeros_like(present[:1]))
    n32 = (present.shape[0] + 31) // 32 * 32
    pad = torch.zeros(n32 - present.shape[0], present.shape[1], dtype=torch.bool, device=present.device)
    blocks = int(torch.cat([present, pad]).view(-1, 32, present.shape[1]).any(1).sum()) * 32  # 32-row blocks here
    rows.append((idx.n_out, present.shape[1], cin, cout, pairs, blocks, ms))
    return out


_sp.features_f16 = traced16
_featx3 = _sp.features_bf16x3


def tracedx3(in_feats, idx, packed, cin, cout, *a, **kw):  # (same 32-row blocks as the fp16 form)
    global _feat16
    keep, _feat16 = _feat16, _featx3
    try:
        return traced16(in_feats, idx, packed, cin, cout, *a, **kw)
    finally:
        _feat16 = keep


_sp.features_bf16x3 = tracedx3
with torch.no_grad():
    voxels, coors, npv, nv = model.voxelizer(pts)
    b, v, p, d = voxels.shape
    keep = coors.view(b * v, 4)[:, 0] >= 0
    cs = coors.view(b * v, 4)[keep].contiguous()
    feats = model.voxel_encoder(voxels.view(b * v, p, d)[keep], npv.view(b * v)[keep], cs)
    model.middle_encoder.remember_capacities = False
    model.middle_encoder.amp = AMP
    model.middle_encoder(feats, cs, b)
tot = 0.0
print("fp32 layers:", "bf16x3 form where it applies (three bf16 pieces per fp32 operand, six products)" if _sp.SPLIT_BF16
      else "fp32 matrix-core kernel", "| amp" if AMP else "")
print("rows in", "tile order (windows of 8192 rows sorted by neighbour mask)" if _sp.TILE_ORDER else "raster order")
print(f"{'rows':>8} {'K':>3} {'cin':>4} {'cout':>4} {'pairs/row':>9} {'exec/useful':>11} {'ms':>8} {'useful TF':>9} {'exec TF':>8}")
for n, k, ci, co, pairs, blocks, ms in rows:
    tot += ms
    print(f"{n:8d} {k:3d} {ci:4d} {co:4d} {pairs / n:9.2f} {blocks / max(pairs, 1):11.2f} {ms:8.3f} "
          f"{2 * ci * co * pairs / ms / 1e9:9.1f} {2 * ci * co * blocks / ms / 1e9:8.1f}")
print("sum of feature kernels (ms):", tot)
