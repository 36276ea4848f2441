"""hard_voxelize on 32 frames at once vs as two calls of 16, each after a 1 GiB flush (the state the op finds inside
the step): does a batch walked in slices keep its points in the last-level cache for the row writer?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddle3d_amd import synth  # noqa: E402
from paddle3d_amd.ops import voxelize  # noqa: E402

pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(100 + i) for i in range(32)])).cuda()
args = (list(synth.NUSC_PILLAR), list(synth.NUSC_RANGE), 20, 30000)
flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda")


def run(slices):
    n = 32 // slices
    for k in range(slices):
        voxelize.hard_voxelize_batch(pts[k * n:(k + 1) * n], *args, with_batch_coors=True)


for slices in (1, 2, 4, 1, 2, 4):
    ts = []
    for it in range(8):
        flush.fill_(float(it))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(slices)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"{slices} slice(s) of {32 // slices} frames after a flush: median {np.median(ts[2:]):.1f} us, min {min(ts[2:]):.1f} us per 32 frames")
