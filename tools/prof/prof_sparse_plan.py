"""Timing of the plan path's pieces on CenterPoint-Voxel coordinates (not a test)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddle3d_amd import centerpoint as cpm  # noqa: E402
from paddle3d_amd import synth  # noqa: E402
from paddle3d_amd import _lib  # noqa: E402
from paddle3d_amd.ops import sparse_conv3d as sp  # noqa: E402
from paddle3d_amd.sparse import _SparseConv  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = cpm.centerpoint_voxels_nuscenes(max_num_voxels=(120000, 160000)).cuda().eval()
pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(100 + i) for i in range(batch)])).cuda()
with torch.no_grad():
    voxels, coors, npv, nv = model.voxelizer(pts)
coors = coors.view(-1, 4)
enc = model.middle_encoder
convs = [m for m in enc.modules() if isinstance(m, _SparseConv)]
specs = [m.spec() for m in convs]

L = _lib.lib()
real = {}
for name in ("pd3_sparse_sort_coords", "pd3_sparse_conv_outputs", "pd3_sparse_rulebook"):
    fn = getattr(L, name)
    real[name] = fn

    def wrap(*a, _fn=fn, _name=name):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = _fn(*a)
        e1.record()
        torch.cuda.synchronize()
        print(f"{_name:28s} {e0.elapsed_time(e1) * 1e3:9.1f} us  n_in_cap={a[2] if _name != 'pd3_sparse_sort_coords' else a[1]}"
              + (f" n_out={a[5]}" if _name == "pd3_sparse_rulebook" else ""))
        return r

    setattr(L, name, wrap)
for _ in range(2):
    print("---")
    pl = sp.plan(coors, batch, enc.sparse_shape, specs)
print([i.n_out for i in pl.indices])
