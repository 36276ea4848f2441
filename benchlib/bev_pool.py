"""bev_pool_v2 at BEVDet4D size."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403
from .common import _LAST_LOOP, _timed_loop, _timed_region  # noqa: F401

def bench_bev_pool(args, rank, world, dev):
    """bev_pool_v2 forward at BEVDet4D size: `--batch` frames x 6 cameras x 118 depth bins x 16 x 44, C = 80, 128 x 128
    BEV per frame (the op takes the whole batch in one launch: ranks_bev carries the frame, bev_pool.cc:30-54)."""
    from paddle3d_amd import synth
    from paddle3d_amd.bevdet import LSSViewTransformer
    from paddle3d_amd.ops import bev_pool_v2 as bp

    B = max(1, int(args.batch))
    # index sets from the real frustum geometry of bevdet4d_r50_depth_nuscenes.yml:174-186 (6 cameras of a synthetic
    # nuScenes-like rig per frame), built on the device by pd3_frustum_to_lidar + pd3_voxel_pooling_prepare
    vt = LSSViewTransformer()
    cams = synth.camera_rig(rank, batch=B)
    coor = vt.get_lidar_coor(*[torch.from_numpy(cams[k]).to(dev) for k in ("rots", "trans", "cam2imgs", "post_rots",
                                                                           "post_trans", "bda")])
    rb, rd, rf, st, ln = vt.voxel_pooling_prepare_v2(coor)
    rng = np.random.default_rng(0)
    t = dict(depth=torch.from_numpy(rng.random((B * 6, 118, 16, 44)).astype(np.float32)).to(dev),
             feat=torch.from_numpy(rng.normal(size=(B * 6, 16, 44, 80)).astype(np.float32)).to(dev),
             ranks_depth=rd, ranks_feat=rf, ranks_bev=rb, interval_lengths=ln, interval_starts=st)
    shape = (B, 128, 128, 80)
    names = ["start", "bev_pool_v2"]

    def run(events):
        if events is not None:
            events[0].record()
        out = bp.bev_pool_v2(t["depth"], t["feat"], t["ranks_depth"], t["ranks_feat"], t["ranks_bev"],
                             t["interval_lengths"], t["interval_starts"], shape)
        if events is not None:
            events[1].record()
        return out

    dt, per_op_ms, out, info = _timed_loop(run, args, world, dev, names)
    if rank != 0:
        return None
    n_pts, n_int, c = int(t["ranks_bev"].numel()), int(t["interval_lengths"].numel()), int(t["feat"].shape[-1])
    alg = 4 * (n_pts * (1 + c) + 3 * n_pts + 2 * n_int) + 4 * int(out.numel())
    return {
        "metric": "bev_pool_v2 forward frames/sec (BEVDet4D shapes)", "value": world * B * args.steps / dt,
        "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"bev_pool_v2 forward: {B} frame(s) per launch, {n_pts} frustum points in {n_int} "
                               f"intervals, C={c}, BEV {tuple(shape)}", "frames_per_gpu_per_step": B,
                   "parallelism": f"dp{world} (frames)"},
        "roofline": hbm_roofline(alg, per_op_ms["bev_pool_v2"], B,
                                 kernel="bev_pool_fwd_kernel (gathered operands counted once per use, SURVEY 8(d))"),
        "per_op_ms": per_op_ms,
    }
