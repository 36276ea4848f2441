#!/usr/bin/env python
"""bench.py -- scenes/s of the CenterPoint-Pillars nuScenes hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads (the default is BASELINE.json's headline; the others give the driver a clock on configs 4 / 5):
  centerpoint_pillars  hard_voxelize -> PFN -> pointpillars_scatter -> SECOND backbone + FPN -> CenterHead ->
                       centerpoint_postprocess (+ for N > 1 the RCCL all-gather of the per-frame box records) over
                       one batch of `--batch` synthetic nuScenes-shaped sweeps per GPU (300k points x 5, 0.2 m
                       pillars, 30k-voxel cap; BASELINE.json configs[2])
  centerpoint_voxel    the same graph with 0.075 m voxels, VoxelMean and the sparse-conv middle encoder (configs[3])
  bev_pool_v2          the camera->BEV pooling op at BEVDet4D size (configs[4]'s bev_pool)
A "step" is ONE pass of the whole path over one batch.  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.  Weak scaling: every rank processes its own batch, so value = N*B*K / time.

The line also carries
  roofline      hard_voxelize's launch sequence (the op the north star sets the >=50 % HBM target on):
                algorithmic bytes per launch / its HIP-event duration inside the timed region;
  rooflines     the same for the other ops: scatter / postprocess on HBM; PFN and the dense graph on the fp32
                MFMA peak, both as EXECUTED flops (what the matrix cores do) and as direct-form flops;
  extras        h2d_inclusive scenes/s (the batch copied from pinned host memory inside every step), batch-1
                latency, the copy / fill ceilings measured on this device;
  cpu_baseline  the oracle pipeline (reference CPU voxelizer compiled from /root/reference when present,
                otherwise the port; torch-CPU dense graph) on a bounded sample, rank 0 at N=1 only: all host
                threads, plus 1 thread and P processes x 1 thread.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.common import _dist_fields  # noqa: E402
from benchlib.registry import WORKLOADS  # noqa: E402
from benchlib.stub import bench_stub  # noqa: E402


def _self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script through torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1) and hand back its exit status."""
    import socket
    import subprocess

    if not args.stub_ops:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node; refusing to "
                             f"report a {args.gpus}-GPU number from fewer devices")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default 16; 8 for centerpoint_voxel; 32 gives the headline workload +2 %% scenes/s)")
    ap.add_argument("--max-voxels", type=int, default=30000)
    ap.add_argument("--workload", default="centerpoint_pillars",
                    choices=sorted(WORKLOADS))
    ap.add_argument("--vox-path", type=int, default=0, help="pd3_hard_voxelize_path selector (0 = library default, "
                    "1 generic sort, 2 tiled with a compact payload array, 3 tiled with gathered rows, 5 wave form)")
    ap.add_argument("--front", choices=["fused", "pair"], default="fused", help="front half of the pillar graphs: fused = "
                    "voxelizer -> PFN through an index of the points (the model path), pair = the two full operators")
    ap.add_argument("--graph", action="store_true", help="replay the step as five captured HIP graphs (one per op) "
                    "instead of launching every kernel from the host (centerpoint_pillars; same kernels and buffers)")
    ap.add_argument("--repeats", type=int, default=None, help="time the same K steps this many more times after the "
                    "contract block and report min / median / max (default 4 at N=1, 0 otherwise)")
    ap.add_argument("--gather", choices=["overlap", "sync"], default="overlap", help="result hand-off: the all-gather "
                    "of batch k overlapped with batch k + 1 (default) or inside the step")
    ap.add_argument("--strong-frames", type=int, default=128, help="frames of the fixed set of the strong-scaling extra "
                    "(sharded over the ranks by dist.shard_frames; 0 = off)")
    ap.add_argument("--no-affinity", action="store_true", help="do not pin the ranks of an N > 1 run to the cores next "
                    "to their GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--map-frames", type=int, default=16, help="frames of the mAP proxy in extras (device detections "
                    "scored against the oracle pipeline's; ~1 s of CPU per frame; 0 = off)")
    ap.add_argument("--no-extras", action="store_true", help="skip the h2d-inclusive / batch-1 / ceiling measurements "
                    "and the other workloads (profiling runs: only warm-up + timed steps are launched)")
    ap.add_argument("--stub-ops", action="store_true", help="test hook: no device ops, launch / collective path only "
                    "(CPU, gloo); the line is marked stub")
    args = ap.parse_args(argv)
    if args.batch is None:
        args.batch = WORKLOADS[args.workload][1]
    if args.repeats is None:
        args.repeats = 4 if args.gpus == 1 else 0

    # `python bench.py --gpus N` with no launcher around it: become the launcher
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_self_launch(args, argv))

    from paddle3d_amd import dist as pdist

    rank, world, local = pdist.init_from_env("gloo" if args.stub_ops else None)
    if world != args.gpus:
        if world > 1:
            torch.distributed.destroy_process_group()
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing "
                         "to print a line whose n_gpus would not be what was asked for")
    affinity = None
    if world > 1 and not args.no_affinity:
        affinity = pdist.set_cpu_affinity(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if args.stub_ops:
        dev = torch.device("cpu")
        line = bench_stub(args, rank, world, dev)
    else:
        from paddle3d_amd._lib import lib

        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP ops have no CPU path")
        lib()  # fail loudly if libpaddle3d_amd.so is missing
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        torch.manual_seed(0)
        line = WORKLOADS[args.workload][0](args, rank, world, dev)
    if rank == 0:
        if world > 1:
            line.setdefault("config", {})["cpu_affinity_rank0"] = (
                f"{len(affinity)} cpus {affinity[0]}-{affinity[-1]}" if affinity else "not pinned")
        print(json.dumps(_dist_fields(line, args, world)))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
