"""Timing driver (not a test): CenterPoint-Voxel front half + whole graph per frame."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddle3d_amd import centerpoint as cpm  # noqa: E402
from paddle3d_amd import synth  # noqa: E402

torch.backends.cudnn.benchmark = True
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = cpm.centerpoint_voxels_nuscenes().cuda().eval()
pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(100 + i) for i in range(batch)])).cuda()


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters


from paddle3d_amd.ops import sparse_conv3d as _sp  # noqa: E402
_orig = _sp.indices
def _traced(coords, batch, spatial_shape, kernel_size, stride=1, padding=0, subm=False):
    r = _orig(coords, batch, spatial_shape, kernel_size, stride, padding, subm)
    print(f"  indices: n_in={coords.shape[0]} shape={tuple(spatial_shape)} k={kernel_size} s={stride} subm={subm} -> n_out={r.n_out}")
    return r
_sp.indices = _traced
with torch.no_grad():
    model.extract_pillars(pts)
_sp.indices = _orig
with torch.no_grad():
    t_front = timeit(lambda: model.extract_pillars(pts))
    t_all = timeit(lambda: model.test_forward(pts, device_only=True))
print(f"batch {batch}: front half {t_front * 1e3 / batch:.2f} ms/frame, whole graph {t_all * 1e3 / batch:.2f} ms/frame, "
      f"{batch / t_all:.1f} scenes/s")
