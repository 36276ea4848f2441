"""Workload table and the short runs of the other configurations inside the default invocation."""
from __future__ import annotations

import torch

from .bev_pool import bench_bev_pool
from .bevfusion import bench_bevfusion_lidar, bench_camera_pool, camera_pool_cpu_baseline
from .kitti import bench_pointpillars_kitti, c1_cpu_baseline
from .pillars import bench_pillars
from .stub import bench_stub
from .voxel import bench_voxel, c4_cpu_baseline

# --workload name -> (function, default frames per GPU per step)
WORKLOADS = {
    "centerpoint_pillars": (bench_pillars, 16),
    "centerpoint_pillars_amp": (bench_pillars, 16),
    "centerpoint_voxel": (bench_voxel, 8),
    "centerpoint_voxel_amp": (bench_voxel, 8),
    "pointpillars_kitti": (bench_pointpillars_kitti, 16),
    "bevfusion_lidar": (bench_bevfusion_lidar, 16),
    "bevfusion_camera_pool": (bench_camera_pool, 1),
    "bev_pool_v2": (bench_bev_pool, 1),
    "bev_pool_v2_b8": (bench_bev_pool, 8),   # the same op, eight frames per launch
}


def other_workloads(args, rank, world, dev):
    """Short runs of BASELINE.json's other single-GPU configurations inside the default invocation, so that one
    driver-run line carries every config that fits one GPU (value, ms per step, roofline fraction each)."""
    import copy

    out = {}
    todo = [(name, *WORKLOADS[name]) for name in WORKLOADS if name != "centerpoint_pillars"]
    for name, fn, batch in todo:
        a = copy.copy(args)
        a.batch, a.steps, a.warmup, a.repeats = batch, 5, 2, 0
        if fn is bench_voxel:  # (a 35-45 ms step whose first few runs still grow the allocator's pools)
            a.steps, a.warmup = 10, 4
        a.no_extras = a.no_cpu_baseline = True
        a.workload = name
        try:
            with torch.no_grad():
                line = fn(a, rank, world, dev)
            rf = line["roofline"]
            out[name] = dict(metric=line["metric"], value=line["value"], unit=line["unit"], steps=a.steps,
                             warmup=a.warmup, ms_per_step=line["ms_per_step"], workload=line["config"]["workload"],
                             roofline=dict(kernel=rf.get("kernel"), bound=rf["bound"], frac=rf["frac"],
                                           achieved=rf["achieved"], unit=rf["unit"]),
                             rooflines={k: dict(bound=v["bound"], frac=v.get("frac")) for k, v in
                                        line.get("rooflines", {}).items()},
                             per_op_ms=line["per_op_ms"])
            for extra in ("amp_error", "dtype", "forms", "sparse_plan"):
                if extra in line:
                    out[name][extra] = line[extra]
        except Exception as e:  # noqa: BLE001 -- reported extras, never required for the headline
            out[name] = dict(error=f"{type(e).__name__}: {e}")
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    for name, leg in (("centerpoint_voxel", c4_cpu_baseline), ("pointpillars_kitti", c1_cpu_baseline),
                      ("bevfusion_camera_pool", camera_pool_cpu_baseline)):
        if not args.no_cpu_baseline and name in out and "error" not in out[name]:
            try:
                out[name]["cpu_baseline"] = leg()
            except Exception as e:  # noqa: BLE001
                out[name]["cpu_baseline"] = dict(value=None, sample=f"failed: {type(e).__name__}: {e}")
    return out
