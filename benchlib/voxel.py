"""CenterPoint-Voxel nuScenes (BASELINE.json configs[3]): sparse-conv middle encoder."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403
from .common import _LAST_LOOP, _timed_loop, _timed_region  # noqa: F401

def c4_cpu_baseline():
    """CenterPoint-Voxel on the host cores, bounded: the dense conv3d statement of the sparse encoder (the only CPU
    statement there is: Paddle's sparse kernels are not vendored) fits a CPU only on a cropped grid, so this is ONE
    frame of a quarter-range copy of config 4 (41 x 256 x 256 cells, 120 k points): reference voxelizer (oracle/_ref
    when present) -> voxel mean -> dense conv3d stack -> torch dense graph -> C postprocess."""
    from oracle import pyoracle as O
    from paddle3d_amd import centerpoint as cpm
    from paddle3d_amd import synth

    pcr = [-9.6, -9.6, -5.0, 9.6, 9.6, 3.0]
    torch.manual_seed(8)
    cpu = cpm.centerpoint_voxels_nuscenes(max_num_voxels=(40000, 40000), point_cloud_range=pcr).eval()
    kind = "ref" if O.have_ref() else "port"
    cfg = cpu.test_cfg
    pts = synth.nuscenes_sweep(93, n_points=120_000)
    t0 = time.perf_counter()
    vox, co, npv, nv = O.hard_voxelize(pts, synth.NUSC_VOXEL, pcr, 10, 40000, kind)
    mean = O.voxel_mean(vox[:nv], npv[:nv])
    c4 = np.concatenate([np.zeros((nv, 1), np.int32), co[:nv]], 1)
    bev = O.sparse_encoder_dense_torch(cpu.middle_encoder, mean, c4, 1)
    with torch.no_grad():
        preds, _ = O.center_head_torch(cpu.bbox_head, O.dense_forward_torch(cpu, bev))
    tasks = [{k: v.numpy() for k, v in p.items()} for p in preds]
    O.centerpoint_postprocess(tasks, cfg["voxel_size"] + [8.0], cfg["point_cloud_range"] + [0.0] * 4,
                              cfg["post_center_limit_range"], [0, 1, 3, 5, 6, 8], cfg["down_ratio"],
                              cfg["score_threshold"], cfg["nms"]["nms_iou_threshold"], cfg["nms"]["nms_pre_max_size"],
                              cfg["nms"]["nms_post_max_size"], True)
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit="cropped scenes/s", cores=torch.get_num_threads(),
                kind="reference" if kind == "ref" else "port",
                sample="1 frame of a QUARTER-RANGE crop of config 4 (0.075 m voxels on +-9.6 m: 41 x 256 x 256 cells, "
                       f"120000 points, {int(nv)} voxels): reference voxelizer, sparse encoder as dense torch conv3d "
                       "(the full 41 x 1440 x 1440 grid has no dense CPU statement that finishes), torch dense graph, "
                       "C postprocess; 1/32 of the full grid's cells, so not comparable with `value` one to one")


def bench_voxel(args, rank, world, dev):
    """CenterPoint-Voxel (config 4): 0.075 m voxels, sort-path hard_voxelize, VoxelMean, SparseResNet3D, dense
    graph at 180 x 180, postprocess."""
    from paddle3d_amd import centerpoint as cpm
    from paddle3d_amd import dist as pdist

    B = args.batch
    V = 160000  # the reference's test-time cap (max_num_voxels: [120000, 160000])
    model = cpm.centerpoint_voxels_nuscenes(max_num_voxels=(120000, V)).to(dev).eval()
    amp = args.workload == "centerpoint_voxel_amp"
    model.set_amp(amp)  # the sparse encoder from 16 -> 32 on (the 180-wide dense maps are not the fp16 kernel's shape)
    # the unsynced sparse plan is opt-in (the default plans with a host sync and can never truncate): this loop reads
    # take_overflow() after the timed region and voids the line if a set outgrew its capacity
    model.middle_encoder.remember_capacities = True
    # DISTINCT batches in rotation (different scenes, different voxel counts): the capacities remembered from the first
    # one (x 1.25) have to hold for the others, as they would on a stream of frames
    n_batches = max(1, int(getattr(args, "voxel_batches", 4)))
    batches = [make_batch(B, 100 + B * rank + 1000 * j, dev) for j in range(n_batches)]
    pts = batches[0]
    cfg = model.test_cfg
    names = ["start", "hard_voxelize", "voxel_mean_sparse_encoder", "dense", "postprocess", "gather"]
    stats = {"step": 0}

    def run(events):
        def mark(i):
            if events is not None:
                events[i].record()

        pts = batches[stats["step"] % n_batches]
        stats["step"] += 1
        mark(0)
        voxels, coors, npv, nv = model.voxelizer(pts)
        mark(1)
        b, v, p, d = voxels.shape
        voxels, coors, npv = voxels.view(b * v, p, d), coors.view(b * v, 4), npv.view(b * v)
        feats = model.voxel_encoder(voxels, npv, coors)  # padding rows included: the encoder skips them
        x = model.middle_encoder(feats, coors, b)
        mark(2)
        x = model.dense_forward(x)
        preds, _ = model.bbox_head(x, want_shared=False)  # (as CenterPoint.test_forward calls it)
        mark(3)
        _bx, _sc, _lb, cnt, rec = model.bbox_head.predict_by_custom_op(preds, cfg, device_only=True,
                                                                      records=cfg["max_per_img"])
        mark(4)
        all_rec, all_cnt = pdist.gather_detections(rec, cnt)
        mark(5)
        return all_rec, all_cnt

    dt, per_op_ms, out, info = _timed_loop(run, args, world, dev, names)
    # the encoder planned every timed step from remembered capacities (no host round trip inside the step); did a set
    # outgrow its capacity?  (one read-back, after the timed region)
    overflow = bool(model.middle_encoder.take_overflow())
    if rank != 0:
        return None
    alg = 4 * N_POINTS * DIMS + 4 * V * 10 * DIMS + 16 * V + 4
    a = alg * B / (per_op_ms["hard_voxelize"] * 1e-3) / 1e9
    with torch.no_grad():  # untimed: how many multiply-adds the encoder's rulebooks hold for this batch
        from paddle3d_amd import sparse as _sparse

        stats["active_per_batch"] = [int((model.voxelizer(bt)[1].view(-1, 4)[:, 0] >= 0).sum().item()) for bt in batches]

        voxels, coors, npv, nv = model.voxelizer(pts)
        b, v, p, d = voxels.shape
        keep = coors.view(b * v, 4)[:, 0] >= 0
        cs = coors.view(b * v, 4)[keep].contiguous()
        stats["active_voxels"] = int(cs.shape[0])
        sp = _sparse.count_flops(model.middle_encoder, model.voxel_encoder(voxels.view(b * v, p, d)[keep],
                                                                          npv.view(b * v)[keep], cs), cs, b)
    line = {
        "metric": "scenes/sec CenterPoint-Voxel nuScenes 300k-pt sweeps" + (
            " (AMP O2: the sparse encoder's convolutions on the fp16 matrix cores)" if amp else ""),
        "value": world * B * args.steps / dt, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("f16 x f16 -> f32 (sparse convolutions from 16 -> 32 channels on), f32 elsewhere" if amp else
                  "f32 (sparse layers with >= 64 output channels: fp32 arithmetic as bf16x3 on the bf16 matrix cores)"),
        "data": "synthetic",
        "config": {"workload": "CenterPoint-Voxel nuScenes 10-sweep: 300000 pts x 5 per scene, 0.075 m voxels "
                               f"(1440x1440x40), P=10, max_voxels={V}, batch {B} distinct scenes/GPU/step, random-init "
                               "weights, voxelize->VoxelMean->SparseResNet3D->SECOND+FPN->CenterHead->postprocess",
                   "frames_per_gpu_per_step": B, "max_voxels": V, "parallelism": f"dp{world} (frames)"},
        "roofline": dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBPS, unit="GB/s", frac=a / HBM_PEAK_GBPS, traffic=None,
                         ms_per_launch=per_op_ms["hard_voxelize"], units_per_launch=B, algorithmic_bytes_per_unit=alg,
                         kernel="hard_voxelize launch sequence, 3-D wave form (voxelize_wave3d.hpp: route + group with an "
                                "LDS hash table per wave + assign + rows) on the 82.9 M-cell grid"),
        "per_op_ms": per_op_ms, "active_voxels_per_batch": stats.get("active_voxels"),
        "detections_first_frame": int(out[1][0].item()),
        "sparse_plan": dict(host_syncs_per_step=0, capacity_overflow=overflow, distinct_batches=n_batches,
                            active_voxels_per_batch=stats.get("active_per_batch"),
                            note="index sets planned from capacities remembered at the FIRST batch (x 1.25, one sync "
                                 f"there); the timed steps rotate over {n_batches} distinct batches of other scenes; an "
                                 "overflow would make the timed steps invalid (the line then carries `error`); "
                                 "opt-in: the library default plans with the sync"),
    }
    if overflow:
        line["error"] = "sparse plan: an index set outgrew its remembered capacity during the timed steps"
    if not amp:
        from paddle3d_amd.ops import sparse_conv3d as _sp3

        with torch.no_grad():  # (untimed) the encoder's map by the fp32 matrix-core kernel in every layer, for comparison
            bev_x3 = model.extract_pillars(pts)
            _sp3.SPLIT_BF16 = False
            try:
                bev_32 = model.extract_pillars(pts)
            finally:
                _sp3.SPLIT_BF16 = True
        line["sparse_arithmetic"] = dict(
            form="fp32; the layers with >= 64 output channels multiply on the bf16 matrix cores with every fp32 operand "
                 "cut into three bf16 pieces (hi + mid + lo = the value exactly) and six of the nine piece products "
                 "accumulated in fp32 (csrc/sparse_conv_x3.hip): the error against exact arithmetic is that of the fp32 "
                 "matrix-core kernel (tests/test_sparse_conv_gpu.py::test_features_bf16x3_is_fp32_arithmetic)",
            encoder_map_max_abs_diff_vs_fp32_kernel=float((bev_x3 - bev_32).abs().max()),
            encoder_map_max_abs=float(bev_32.abs().max()))
    if amp:
        from paddle3d_amd import nuscenes_bridge as nb

        import copy

        from paddle3d_amd import synth

        with torch.no_grad():  # what the mode costs in accuracy on this batch (untimed)
            bev16 = model.extract_pillars(pts)
            model.set_amp(False)
            bev32 = model.extract_pillars(pts)
            model.set_amp(True)
            # detections: a copy with BatchNorm statistics and heads like a trained net's (plain random-init weights put
            # every score of a class into one band 0.003 wide, where the comparison measures tie-breaking)
            m2 = copy.deepcopy(model)
            m2.middle_encoder.remember_capacities = None
            m2.set_amp(False)
            synth.trained_like_batchnorm(m2, 7)
            synth.trained_like_heads(m2, pts[:2])
            d32 = m2.test_forward(pts)
            m2.middle_encoder.amp = True      # the sparse encoder alone in fp16
            d16e = m2.test_forward(pts)
            m2.set_amp(True)                  # the whole graph (the benchmarked mode)
            d16 = m2.test_forward(pts)
            del m2

        def twins(d):
            fwd = nb.unmatched_detections(d, d32, score_tol=2e-2)
            back = nb.unmatched_detections(d32, d, score_tol=2e-2)
            return dict(fp32_boxes_without_amp_twin=fwd, amp_boxes_without_fp32_twin=back,
                        twin_fraction=1.0 - max(fwd["unmatched"] / max(1, fwd["total"]),
                                                back["unmatched"] / max(1, back["total"])))

        line["amp_error"] = dict(
            bev_map_max_abs=float((bev16 - bev32).abs().max()), bev_map_max_magnitude=float(bev32.abs().max()),
            sparse_encoder_fp16_only=twins(d16e), whole_graph_fp16=twins(d16),
            note="the encoder's [B, 256, 180, 180] map of the AMP graph against the fp32 graph's (the benchmarked "
                 "random-init weights); detections on a copy with BatchNorm statistics and heads like a trained net's "
                 "(synth.trained_like_batchnorm / trained_like_heads): boxes of one graph without a twin in the other "
                 "(same frame and class, centre within 0.5 m, score within 0.02).  The fp16 sparse encoder alone keeps "
                 ">= 99 % (asserted in tests/test_sparse_conv_gpu.py::test_sparse_encoder_amp_close_to_fp32_and_voxel_"
                 "model); with the dense graph in fp16 too, ~11 % of the boxes change cell on this model's calibrated "
                 "random heads although every head map stays within 5e-3 of its magnitude -- analysis in "
                 "profiles/r06_amp_voxel_twins.txt")
    if sp:
        ms = per_op_ms["voxel_mean_sparse_encoder"]
        line["rooflines"] = {"sparse_encoder": mfma_roofline(
            sp["pairs_by_pipe"], ms, B, flops_existing_pairs=sp["pairs"], flops_dense_equivalent=sp["dense"],
            note="flops of the (output row, kernel offset) pairs that exist, 2*Cin*Cout each (counted on the first "
                 "batch), over the whole encoder stage time (index building included), split by the matrix pipe each "
                 "layer's kernel issues on (`pipes`); peak = that mix with every layer at its own pipe's dense peak "
                 "(bf16x3 = fp32-equivalent flops at a sixth of the bf16 pipe); dense-equivalent counts all 27 offsets"
                 + ("; this mode is bound by the gather, not by the fp16 pipe" if amp else ""))}
    return line
