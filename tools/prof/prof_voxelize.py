"""Profiling driver (not a test): hard_voxelize on C3-sized batches of DISTINCT frames, every path timed with HIP
events and checked for identical bytes against path 1; use under rocprofv3 for the per-kernel split.
usage: prof_voxelize.py [batch] [max_voxels] [iters] [paths, e.g. 2,3]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddle3d_amd import synth  # noqa: E402
from paddle3d_amd.ops import voxelize  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
v = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
paths = [int(p) for p in sys.argv[4].split(",")] if len(sys.argv) > 4 else [2, 3]
shuffle = len(sys.argv) > 5 and sys.argv[5] == "shuffle"
c4 = len(sys.argv) > 5 and sys.argv[5] == "c4"   # config 4: 0.075 m voxels on 1440 x 1440 x 40, P = 10 (pass max_voxels 160000)
pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(100 + i, shuffle=shuffle) for i in range(batch)])).cuda()
P = 10 if c4 else 20
args = ((list(synth.NUSC_VOXEL), list(synth.NUSC_VOXEL_RANGE), P, v) if c4 else
        (list(synth.NUSC_PILLAR), list(synth.NUSC_RANGE), P, v))
ref = voxelize.hard_voxelize_batch(pts, *args, with_batch_coors=True, path=1)
alg = (4 * 300000 * 5 + 4 * v * P * 5 + 16 * v + 4) * batch
for path in paths:
    for _ in range(3):
        out = voxelize.hard_voxelize_batch(pts, *args, with_batch_coors=True, path=path)
    same = all(torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a,
                           b.view(torch.int32) if b.dtype == torch.float32 else b) for a, b in zip(out, ref))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = voxelize.hard_voxelize_batch(pts, *args, with_batch_coors=True, path=path)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"path {path}: {ms * 1e3:.1f} us per {batch} frames = {alg / ms / 1e6:.0f} GB/s = "
          f"{alg / ms / 1e6 / 8000:.3f} of 8 TB/s; identical to the sort path: {same}; nv {out[3].tolist()[:4]}")
