"""A/B of two builds of the library on one box: hard_voxelize alone (path 5), HIP-event time per 16 frames.
usage: prof_vox_ab.py <lib A> <lib B> [iters]   (each library is timed in its own process, A B A B)"""
import os
import subprocess
import sys

if len(sys.argv) >= 2 and sys.argv[1] == "--one":
    sys.path.insert(0, ".")
    import paddle3d_amd._lib as L
    L.LIB_PATH = os.path.abspath(sys.argv[2])
    import numpy as np
    import torch
    from paddle3d_amd import synth
    from paddle3d_amd.ops import voxelize
    it = int(sys.argv[3])
    pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(1000 + i) for i in range(16)])).cuda()
    f = lambda: voxelize.hard_voxelize_batch(pts, list(synth.NUSC_PILLAR), list(synth.NUSC_RANGE), 20, 30000, path=5)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / it * 1e3)
    print(f"{os.path.basename(sys.argv[2])}: min {min(ts):.1f} us, median {sorted(ts)[2]:.1f} us")
else:
    a, b = sys.argv[1], sys.argv[2]
    it = sys.argv[3] if len(sys.argv) > 3 else "30"
    for lib in (a, b, a, b):
        subprocess.run([sys.executable, __file__, "--one", lib, it], check=False)
