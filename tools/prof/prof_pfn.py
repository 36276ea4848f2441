"""Time the PillarFeatureNet kernel forms (per-pillar wave form vs packed form) on the C3 batch: 16 distinct synthetic
nuScenes sweeps voxelized on the device, the two-layer net of CenterPoint-Pillars (10 -> 32 | 64 -> 64).

    python tools/prof/prof_pfn.py [batch]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from paddle3d_amd import synth  # noqa: E402
from paddle3d_amd.ops import voxel_encoder as ve, voxelize  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(1000 + i) for i in range(batch)])).to(dev)
vox, co, npv, nv = voxelize.hard_voxelize_batch(pts, list(synth.NUSC_PILLAR), list(synth.NUSC_RANGE), 20, 30000)
vox = vox.reshape(-1, 20, 5)
npv = npv.reshape(-1)
c4 = co.reshape(-1, 4) if co.shape[-1] == 4 else torch.cat(
    [torch.arange(batch, device=dev, dtype=torch.int32).repeat_interleave(30000).unsqueeze(1), co.reshape(-1, 3)], 1)
g = torch.Generator(device="cuda").manual_seed(1)
w1 = torch.randn(10, 32, device=dev, generator=g) / 3
w2 = torch.randn(64, 64, device=dev, generator=g) / 8
s1, b1 = torch.rand(32, device=dev, generator=g) + 0.5, torch.randn(32, device=dev, generator=g) * 0.1
s2, b2 = torch.rand(64, device=dev, generator=g) + 0.5, torch.randn(64, device=dev, generator=g) * 0.1
args = (vox, npv, c4, 0.2, 0.2, -51.1, -51.1, w1, s1, b1, w2, s2, b2)
rows1 = (npv.clamp(max=20) + (npv < 20).int()) * (npv > 0).int()  # per-pillar form: + one padded row
rows2 = npv.clamp(min=0, max=20)                                    # packed form: stored points only
print(f"pillars {vox.shape[0]}, live {(npv > 0).sum().item()}, stored points {rows2.sum().item()}, "
      f"per-pillar blocks {((rows1 + 15) // 16).sum().item()}, "
      f"packed blocks {((rows2.reshape(-1, 8).sum(1) + 15) // 16).sum().item()} (+ one base block per chunk)")
outs = {}
for path in (1, 2, 0):
    for _ in range(3):
        outs[path] = ve.pillar_feature_net(*args, path=path)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ve.pillar_feature_net(*args, path=path)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    print(f"path {path}: {min(ts) * 1e3:.1f} us (median {sorted(ts)[2] * 1e3:.1f})")
d = (outs[1] - outs[2]).abs().max().item()
print("max |per-pillar - packed| =", d)
