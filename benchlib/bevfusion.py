"""BEVFusion (BASELINE.json configs[4]): the LiDAR stream's front half and the camera -> BEV pooling."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403
from .common import _LAST_LOOP, _timed_loop, _timed_region  # noqa: F401

def bench_bevfusion_lidar(args, rank, world, dev):
    """BEVFusion LiDAR stream front half (config 5, configs/bevfusion/bevf_pp_2x8_1x_nusc.yaml:87-116): 0.25 m pillars
    on +-50 m (400 x 400), P = 64, V = 40 000, D = 4: hard_voxelize -> HardVFE (64, 64) -> PointPillarsScatter."""
    from paddle3d_amd import centerpoint as cpm
    from paddle3d_amd import synth

    B, V, PV, D4 = args.batch, 40000, 64, 4
    vs, pr = (0.25, 0.25, 8.0), (-50.0, -50.0, -5.0, 50.0, 50.0, 3.0)
    voxelizer = cpm.HardVoxelizer(vs, pr, PV, [30000, V]).eval()
    vfe = cpm.HardVFE(D4, (64, 64), False, True, True, vs, pr).to(dev).eval()
    scatter = cpm.PointPillarsScatter(64, vs, pr)
    pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(100 + B * rank + i, dims=D4) for i in range(B)])).to(dev)
    names = ["start", "hard_voxelize", "hard_vfe", "pointpillars_scatter"]

    def run(events):
        def mark(i):
            if events is not None:
                events[i].record()

        mark(0)
        voxels, coors, npv, nv = voxelizer(pts)
        mark(1)
        b, v, p, d = voxels.shape
        feats = vfe(voxels.view(b * v, p, d), npv.view(b * v), coors.view(b * v, 4))
        mark(2)
        canvas = scatter(feats, coors.view(b * v, 4), b)
        mark(3)
        return canvas, nv

    with torch.no_grad():
        dt, per_op_ms, out, info = _timed_loop(run, args, world, dev, names)
    if rank != 0:
        return None
    alg_v = 4 * N_POINTS * D4 + 4 * V * PV * D4 + 16 * V + 4
    a = alg_v * B / (per_op_ms["hard_voxelize"] * 1e-3) / 1e9
    alg_s = 4 * V * 64 + 16 * V + 4 * 64 * 400 * 400
    a_s = alg_s * B / (per_op_ms["pointpillars_scatter"] * 1e-3) / 1e9
    return {
        "metric": "frames/sec BEVFusion LiDAR stream front half (voxelize + HardVFE + scatter)",
        "value": world * B * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BEVFusion LiDAR stream: {N_POINTS} pts x {D4} per scene, 0.25 m pillars (400x400), "
                               f"P={PV}, max_voxels={V}, batch {B} distinct scenes/GPU/step, random-init weights, "
                               "hard_voxelize->HardVFE->PointPillarsScatter",
                   "frames_per_gpu_per_step": B, "max_voxels": V, "parallelism": f"dp{world} (frames)"},
        "roofline": dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBPS, unit="GB/s", frac=a / HBM_PEAK_GBPS, traffic=None,
                         ms_per_launch=per_op_ms["hard_voxelize"], units_per_launch=B,
                         algorithmic_bytes_per_unit=alg_v,
                         kernel="hard_voxelize launch sequence, wave form (vw_route + vw_group + vw_assign + vw_rows); "
                                "the fixed-shape [V, 64, 4] output dominates the bytes"),
        "rooflines": {"pointpillars_scatter": dict(bound="hbm", achieved=a_s, peak=HBM_PEAK_GBPS, unit="GB/s",
                                                   frac=a_s / HBM_PEAK_GBPS, traffic=None,
                                                   ms_per_launch=per_op_ms["pointpillars_scatter"],
                                                   units_per_launch=B, algorithmic_bytes_per_unit=alg_s)},
        "per_op_ms": per_op_ms, "voxels_first_frame": int(out[1][0]),
    }


def camera_pool_cpu_baseline():
    """The reference's own algorithm on the host cores, ONE scene of the same shape: frustum geometry (NumPy), lift
    x = depth (x) feat (cam_stream_lss.py:166) + voxel_pooling with the cumsum trick (:111-121, :318-373) as the NumPy
    restatement the oracle pins to the reference's executed Python (tests/test_python_golden.py)."""
    from oracle import pyoracle as O
    from paddle3d_amd import synth
    from paddle3d_amd.bevfusion import LiftSplatShoot

    lss = LiftSplatShoot()
    rig = synth.lss_camera_rig(0)
    depth, feat = synth.lss_camera_features(100, 6, lss.D, lss.fH, lss.fW, lss.camC)
    t0 = time.perf_counter()
    fr = lss.frustum.reshape(-1, 3)
    p = np.stack([fr[:, 0] * fr[:, 2], fr[:, 1] * fr[:, 2], fr[:, 2]], -1)
    geom = (np.einsum("nij,pj->npi", rig["rots"][0], p) + rig["trans"][0][:, None, :]).astype(np.float32)
    geom = geom.reshape(1, 6, lss.D, lss.fH, lss.fW, 3)
    x = (depth[:, :, :, :, None] * feat[:, None]).reshape(1, 6, lss.D, lss.fH, lss.fW, feat.shape[-1])
    t_lift = time.perf_counter() - t0
    O.lss_voxel_pooling_numpy(geom, x, lss.dx, lss.bx, lss.nx)
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit="frames/s", cores=1, kind="port",
                sample=f"1 scene of the same workload: NumPy geometry + lift ({t_lift:.1f} s) + quantise / filter / "
                       f"argsort / fp32 cumsum / differences / scatter ({dt - t_lift:.1f} s), single thread (the "
                       "restatement of cam_stream_lss.py:111-121,279-373; Paddle itself is not installed)")


def bench_camera_pool(args, rank, world, dev):
    """BEVFusion camera stream, camera -> BEV pooling at config 5's shape (configs/bevfusion/bevf_pp_2x8_1x_nusc.yaml:
    80-84; cam_stream_lss.py:279-373): 6 views x 41 depth bins x 112 x 200 feature pixels, C = 64, onto 200 x 200 x 16
    cells of 0.5 m.  A step = frustum geometry -> index build (quantise, filter, sort, run-length) -> pooling, per
    batch of `--batch` scenes; the headline form pools depth [B*6, 41, 112, 200] and feat [B*6, 112, 200, 64] directly
    (`lss_voxel_pooling_fused`), the second loop runs the reference's form on the lifted 1.41 GB / scene tensor."""
    from paddle3d_amd import synth
    from paddle3d_amd.bevfusion import LiftSplatShoot
    from paddle3d_amd.ops import bev_pool_v2 as bp

    B = args.batch
    lss = LiftSplatShoot()
    C = lss.camC
    rigs = [synth.lss_camera_rig(10 * rank + i) for i in range(B)]
    rots = torch.from_numpy(np.concatenate([r["rots"] for r in rigs])).to(dev)
    trans = torch.from_numpy(np.concatenate([r["trans"] for r in rigs])).to(dev)
    df = [synth.lss_camera_features(100 + 10 * rank + i, 6, lss.D, lss.fH, lss.fW, C) for i in range(B)]
    depth = torch.from_numpy(np.concatenate([d for d, _ in df])).to(dev)
    feat = torch.from_numpy(np.concatenate([f for _, f in df])).to(dev)
    names = ["start", "geometry", "index_build", "pool"]
    st = {}

    def run_fused(events):
        def mark(i):
            if events is not None:
                events[i].record()

        mark(0)
        geom = lss.get_geometry(rots, trans)
        mark(1)
        prep = bp.lss_pooling_prepare(geom, lss.dx, lss.bx, lss.nx)
        mark(2)
        out = bp.lss_voxel_pooling_fused(geom, depth, feat, lss.dx, lss.bx, lss.nx, prepared=prep)
        mark(3)
        st["prep"] = prep
        return out

    def run_lifted(events):
        def mark(i):
            if events is not None:
                events[i].record()

        mark(0)
        geom = lss.get_geometry(rots, trans)
        mark(1)
        x = (depth[:, :, :, :, None] * feat[:, None]).view(B, 6, lss.D, lss.fH, lss.fW, C)  # CamEncode's lift (:166)
        mark(2)
        out = lss.voxel_pooling(geom, x)  # index build + pooling of the lifted rows
        mark(3)
        return out

    with torch.no_grad():
        import copy as _copy

        a2 = _copy.copy(args)
        a2.repeats = 0
        dt_l, ms_l, out_l, _ = _timed_loop(run_lifted, a2, world, dev, ["start", "geometry", "lift", "index_build_and_pool"])
        del out_l
        torch.cuda.empty_cache()
        # the pooling kernel alone, fixed calibration (index sets built once: what BEVDet calls `accelerate`)
        geom0 = lss.get_geometry(rots, trans)
        prep0 = bp.lss_pooling_prepare(geom0, lss.dx, lss.bx, lss.nx)
        dt_a, ms_a, _o, _ = _timed_loop(
            lambda ev: (ev[0].record() if ev is not None else None,
                        bp.lss_voxel_pooling_fused(geom0, depth, feat, lss.dx, lss.bx, lss.nx, prepared=prep0),
                        ev[1].record() if ev is not None else None)[1], a2, world, dev, ["start", "pool"])
        dt, per_op_ms, out, info = _timed_loop(run_fused, args, world, dev, names)
    if rank != 0:
        return None
    cell, rd, rf, starts, lengths = st["prep"]
    n_pts, n_int = int(cell.numel()), int(starts.numel())
    out_elems = int(out.numel())
    alg = 4 * (n_pts * (1 + C) + 3 * n_pts + 2 * n_int) + 4 * out_elems   # SURVEY 8(d), gathered operands once per use
    compulsory = 4 * (int(depth.numel()) + int(feat.numel()) + 3 * n_pts + 2 * n_int + out_elems)
    lifted_bytes = 4 * B * 6 * lss.D * lss.fH * lss.fW * C
    # HBM-side traffic of the pooling launch from the committed PMC passes (tools/gpu_kernel_traffic.sh: the run launches
    # the kernel in its lifted form first, then on split operands: the later dispatches are the fused form), + the zero
    # fill of the output the entry point issues in front of it; one scene per launch
    traffic, traffic_src = None, None
    try:
        import glob

        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_camera_pool_kernel_traffic.json")))[-1]
        kt = [v for k, v in json.load(open(path))["kernels"].items() if "bev_pool_fwd" in k][0]
        if B == 1:
            traffic = float(min(kt["fetch_bytes_x2_each"]) + min(kt["write_bytes_each"]) + 4 * out_elems)
            traffic_src = "profiles/" + os.path.basename(path)
    except Exception:  # noqa: BLE001 -- no committed profile: null
        pass
    line = {
        "metric": "frames/sec BEVFusion camera->BEV pooling (LiftSplatShoot.voxel_pooling, config 5 shape)",
        "value": world * B * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BEVFusion camera->BEV pooling: {B} scene(s)/GPU/step x 6 views x {lss.D} depth bins x "
                               f"{lss.fH} x {lss.fW} pixels (5510400 frustum points/scene), C={C}, grid "
                               f"{lss.nx[0]}x{lss.nx[1]}x{lss.nx[2]} of 0.5 m: frustum geometry -> index build -> pooling "
                               "of depth and feat as separate operands (the lifted depth (x) feat tensor is never formed)",
                   "frames_per_gpu_per_step": B, "parallelism": f"dp{world} (frames)",
                   "host_syncs_per_step": 1,
                   "note": "one host read of (kept points, intervals) per step to size the index tensors, like the "
                           "reference's boolean-mask filter (:340-341)"},
        "roofline": hbm_roofline(alg, per_op_ms["pool"], B, traffic, traffic_src,
                                 kernel="bev_pool_fwd_kernel on split operands (pd3_bev_pool_v2, prepare mode 2)",
                                 compulsory_bytes_per_launch=compulsory,
                                 note="algorithmic bytes per SURVEY 8(d)'s bev_pool formula (every gathered row counted "
                                      "once per use); compulsory = each tensor once (depth + feat + index sets + output): "
                                      "the 34 MB / scene of feat rows are re-read from cache 13 x on average"),
        "per_op_ms": per_op_ms, "kept_points": n_pts, "intervals": n_int,
        "forms": {
            "fused_fixed_calibration": dict(value=world * B * a2.steps / dt_a, unit="frames/s", ms_per_step=dt_a / a2.steps * 1e3,
                                            per_op_ms=ms_a, note="index sets built once (fixed rig), the step is the "
                                                                 "pooling launch alone"),
            "lifted_reference_form": dict(value=world * B * a2.steps / dt_l, unit="frames/s", ms_per_step=dt_l / a2.steps * 1e3,
                                          per_op_ms=ms_l, lifted_tensor_bytes=lifted_bytes,
                                          note="CamEncode's lift x = depth (x) feat written to HBM (1.41 GB / scene), then "
                                               "voxel_pooling(geom, x) through the same kernels with unit weights: the "
                                               "reference's data flow; same result bit for bit "
                                               "(tests/test_lss_c5_gpu.py)")},
    }
    if world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = camera_pool_cpu_baseline()
        except Exception as e:  # noqa: BLE001 -- reported, never required
            line["cpu_baseline"] = dict(value=None, unit="frames/s", cores=0, kind="port", sample=f"failed: {e}")
    return line
