"""Profiling driver (not a test): runs hard_voxelize on C3-sized batches; use under rocprofv3."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddle3d_amd import synth  # noqa: E402
from paddle3d_amd.ops import voxelize  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
v = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
frames = np.stack([synth.nuscenes_sweep(100 + i) for i in range(min(batch, 4))])
frames = np.concatenate([frames] * (batch // len(frames) or 1))[:batch]
pts = torch.from_numpy(frames).cuda()
for _ in range(iters):
    out = voxelize.hard_voxelize_batch(pts, list(synth.NUSC_PILLAR), list(synth.NUSC_RANGE), 20, v)
torch.cuda.synchronize()
print("nv", out[3].tolist())
