"""--stub-ops: the launch / timing / collective path without a GPU."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403
from .common import _LAST_LOOP, _timed_loop, _timed_region  # noqa: F401

def bench_stub(args, rank, world, dev):
    """--stub-ops: the launch / timing / collective path of the default workload with the GPU ops replaced by a fixed
    synthetic detection set on the CPU, so that `bench.py --gpus N` (self-launch, gloo, barriers, MAX over ranks,
    all-gather inside the step) can be exercised without a GPU (tests/test_bench_dist_cpu.py).  The line it prints is
    marked `stub` and is not a measurement."""
    from paddle3d_amd import dist as pdist

    B, max_per_img = args.batch, 500
    g = torch.Generator().manual_seed(1234 + rank)
    bx = torch.randn(B, 498, 9, generator=g)
    sc = torch.rand(B, 498, generator=g)
    lb = torch.randint(0, 10, (B, 498), generator=g)
    cnt = torch.randint(1, 498, (B,), generator=g, dtype=torch.int32)
    names = ["start", "ops_stub", "gather"]
    pipe = pdist.GatherPipeline() if args.gather == "overlap" else None

    def hand_off(rec, c):
        if pipe is None:
            return pdist.gather_detections(rec, c)
        prev = pipe.submit(rec, c)
        return prev if prev is not None else (rec, c)

    def run_batch(frames, events=None):
        """`frames`: a [b, ...] tensor standing for a batch of scenes (only its length is used)."""
        b = frames.shape[0]
        if events is not None:
            events[0].record()
        time.sleep(0.002)  # stands for the device work of a step
        if events is not None:
            events[1].record()
        rec = pdist.pack_records(bx[:b], sc[:b], lb[:b], cnt[:b], max_per_img)
        out = hand_off(rec, cnt[:b])
        if events is not None:
            events[2].record()
        return out

    fake = torch.zeros(B, 4)
    finish = (lambda out: pipe.flush()) if pipe is not None else None
    dt, per_op_ms, out, _info = _timed_loop(lambda ev: run_batch(fake, ev), args, world, dev, names, finish=finish)
    assert out[0].shape[0] == world * B and out[1].shape[0] == world * B
    # what arrived is every rank's own record, in rank order (rank r's generator seed is 1234 + r)
    for r in range(world):
        gr = torch.Generator().manual_seed(1234 + r)
        want = pdist.pack_records(torch.randn(B, 498, 9, generator=gr), torch.rand(B, 498, generator=gr),
                                  torch.randint(0, 10, (B, 498), generator=gr),
                                  torch.randint(1, 498, (B,), generator=gr, dtype=torch.int32), max_per_img)
        assert torch.equal(out[0][r * B:(r + 1) * B], want), f"rank {r}'s records did not arrive intact"
    multi = {}
    if args.strong_frames > 0:
        loop = dict(_LAST_LOOP)
        flush = (lambda: pipe.flush()) if pipe is not None else None
        multi["strong_scaling"] = strong_scaling(lambda b: run_batch(b), flush, lambda ids: torch.zeros(len(ids), 4),
                                                 args.strong_frames, B, rank, world, dev, passes=2)
        multi["h2d_inclusive"] = h2d_inclusive(lambda b: run_batch(b), flush, torch.zeros(B, 4), torch.zeros(B, 4),
                                               args.steps, world, dev)
        multi["h2d_overlapped"] = h2d_overlapped(lambda b: run_batch(b), flush, [torch.zeros(B, 4), torch.ones(B, 4)],
                                                 args.steps, world, dev)
        _LAST_LOOP.clear()
        _LAST_LOOP.update(loop)
    if rank != 0:
        return None
    return {"metric": "scenes/sec CenterPoint-Pillars nuScenes 300k-pt sweeps", "stub": True,
            "value": world * B * args.steps / dt, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "STUB: no device ops; launch / collective path only", "frames_per_gpu_per_step": B,
                       "parallelism": f"dp{world} (frames)", "launch": "eager (stub: no device to capture on)",
                       "launch_reason": "stub"},
            # the line's shape as the real workloads print it: a roofline object built by the same helpers (the stub's
            # "kernel" is a 2 ms sleep that moves the records once)
            "roofline": hbm_roofline(float(out[0].numel() * 4), per_op_ms["ops_stub"], B, kernel="stub"),
            "rooflines": {"stub_mfma": mfma_roofline({"f32": 1e9, "f16": 4e9, "bf16x3": 2e9}, per_op_ms["ops_stub"], B)},
            "per_op_ms": per_op_ms, "frames_gathered": int(out[1].shape[0]), "extras": multi,
            "result_hand_off": "overlap" if pipe is not None else "sync"}
