"""First-light timing script for the GPU box (not a test): times each op on C3-sized inputs."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from paddle3d_amd import synth  # noqa: E402
from paddle3d_amd.ops import pointpillars_scatter as ps  # noqa: E402
from paddle3d_amd.ops import voxelize  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters


import os
res = {"device": torch.cuda.get_device_name(0)}
for path in ("tiled", "sort"):
  os.environ["PD3_VOXELIZE_PATH"] = path
  for batch in (1, 8, 32):
    frames = np.stack([synth.nuscenes_sweep(100 + i) for i in range(min(batch, 4))])
    frames = np.concatenate([frames] * (batch // len(frames) or 1))[:batch]
    pts = torch.from_numpy(frames).cuda()
    for v in (30000, 60000):
        dt = timeit(lambda: voxelize.hard_voxelize_batch(pts, list(synth.NUSC_PILLAR), list(synth.NUSC_RANGE), 20, v))
        alg = 4 * 300000 * 5 + 4 * v * 20 * 5 + 16 * v + 4
        res[f"voxelize_{path}_b{batch}_v{v}"] = dict(ms=round(dt * 1e3, 4), us_per_frame=round(dt * 1e6 / batch, 2),
                                              GBps=round(alg * batch / dt / 1e9, 1))
os.environ.pop("PD3_VOXELIZE_PATH")
feats = torch.randn(30000, 64, device="cuda")
co = torch.zeros(30000, 4, dtype=torch.int32, device="cuda")
cells = torch.randperm(512 * 512, device="cuda")[:30000]
co[:, 2] = (cells // 512).int()
co[:, 3] = (cells % 512).int()
for batch in (1, 8):
    f = feats.repeat(batch, 1)
    c = co.repeat(batch, 1)
    c[:, 0] = torch.arange(batch, device="cuda").repeat_interleave(30000).int()
    out = torch.empty(batch, 64, 512, 512, device="cuda")
    dt = timeit(lambda: ps.pointpillars_scatter(f, c, batch, 512, 512, out=out))
    res[f"scatter_b{batch}"] = dict(ms=dt * 1e3, GBps=75268864 * batch / dt / 1e9)
# write-only ceilings
z = torch.empty(100 * 1024 * 1024 // 4, device="cuda")
dt = timeit(lambda: z.zero_())
res["fill_100MB_GBps"] = z.numel() * 4 / dt / 1e9
z2 = torch.empty(12 * 1024 * 1024 // 4, device="cuda")
dt = timeit(lambda: z2.zero_())
res["fill_12MB_GBps"] = z2.numel() * 4 / dt / 1e9
# plain device copy ceiling
a = torch.empty(256 * 1024 * 1024 // 4, device="cuda")
b = torch.empty_like(a)
dt = timeit(lambda: b.copy_(a))
res["copy_256MB_GBps"] = 2 * a.numel() * 4 / dt / 1e9
print(json.dumps(res, indent=1))
