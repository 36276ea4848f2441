"""PointPillars-KITTI, the whole inference graph (BASELINE.json configs[0])."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403
from .common import _LAST_LOOP, _timed_loop, _timed_region  # noqa: F401

def c1_cpu_baseline(frames=2):
    """PointPillars-KITTI (BASELINE config 1, "on the Paddle CPU reference path") on the host cores: reference
    voxelizer (oracle/_ref when present), torch-CPU PFN / SECOND / FPN / head, NumPy anchor mask + decode + NMS (the
    oracle's statement of SSDHead.post_process), `frames` frames of the same synthetic KITTI clouds."""
    from oracle import pyoracle as O
    from paddle3d_amd import pointpillars as ppm
    from paddle3d_amd import synth

    torch.manual_seed(4)
    cpu = ppm.pointpillars_kitti_car().eval()
    with torch.no_grad():
        cpu.head.cls_head.bias.fill_(-2.0)
    kind = "ref" if O.have_ref() else "port"
    gen, h = cpu.anchor_generator, cpu.head
    an, bv = gen.anchors.numpy(), gen.anchors_bv.numpy().astype(np.int64)
    vs, pcr = cpu.voxelizer.voxel_size, cpu.voxelizer.point_cloud_range
    p_max, v_max = cpu.voxelizer.max_num_points_in_voxel, cpu.voxelizer.max_num_voxels[1]
    nx, ny = gen.grid_size
    apl, ncls = h.num_anchor_per_loc, h.num_classes
    c_cls, c_box = apl * ncls, apl * 7
    params = [dict(weight=l.linear.weight.t().detach().numpy(), gamma=l.norm.weight.detach().numpy(),
                   beta=l.norm.bias.detach().numpy(), mean=l.norm.running_mean.numpy(), var=l.norm.running_var.numpy())
              for l in cpu.pillar_encoder.pfn_layers]
    t0 = time.perf_counter()
    for i in range(frames):
        pts = synth.kitti_frame(100 + i, 16384)
        vox, co, npv, nv = O.hard_voxelize(pts, vs, pcr, p_max, v_max, kind)
        c4 = np.concatenate([np.zeros((nv, 1), np.int32), co[:nv]], 1)
        feats = O.pfn_forward_torch(vox[:nv], npv[:nv], c4, params, vs, pcr)
        bev = torch.from_numpy(O.pillar_scatter(feats, c4, 1, ny, nx))
        with torch.no_grad():
            x = O.second_fpn_torch(cpu.neck, O.second_backbone_torch(cpu.backbone, bev))
            m = torch.cat([h.cls_head(x), h.box_head(x), h.dir_head(x)], 1)[0].numpy()
        pr = m.reshape(m.shape[0], -1).T
        mask = O.ssd_anchor_mask_numpy(co[:nv], bv, gen.grid_size, 1.0)
        O.ssd_post_process_frame_numpy(pr[:, c_cls:c_cls + c_box].reshape(-1, 7), pr[:, :c_cls].reshape(-1, ncls),
                                       pr[:, c_cls + c_box:].reshape(-1, 2), an, mask, h.nms_score_threshold,
                                       h.pred_center_limit_range, h.nms_pre_max_size, h.nms_post_max_size,
                                       h.nms_iou_threshold)
    dt = time.perf_counter() - t0
    return dict(value=frames / dt, unit="frames/s", cores=torch.get_num_threads(),
                kind="reference" if kind == "ref" else "port",
                sample=f"{frames} frames of the same workload: hard_voxelize = "
                       f"{'reference voxelize_op.cc:19-82 compiled from /root/reference' if kind == 'ref' else 'C port'} "
                       "(1 thread), PFN / SECOND / FPN / head = torch CPU fp32, anchor mask / decode / NMS = NumPy + C port")


def bench_pointpillars_kitti(args, rank, world, dev):
    """PointPillars-KITTI, the whole inference graph (config 1, configs/pointpillars/pointpillars_xyres16_kitti_car.yml:
    86-146): 16 384 camera-FOV points x 4, 0.16 m pillars (432 x 496), P = 32, V = 40 000: hard_voxelize ->
    PillarFeatureNet (64) -> PointPillarsScatter -> SECOND backbone -> FPN (transposed convolutions 1 / 2 / 4) -> SSD
    head (one 1x1 GEMM) -> anchor masks + decode + rotated NMS (ssd_postprocess)."""
    from paddle3d_amd import pointpillars as ppm
    from paddle3d_amd import synth

    B, V, PV, D4, NK = args.batch, 40000, 32, 4, 16384
    model = ppm.pointpillars_kitti_car((16000, V)).to(dev).eval()
    with torch.no_grad():
        model.head.cls_head.bias.fill_(-2.0)  # random weights: a few hundred anchors per frame pass the 0.05 threshold
    pts = torch.from_numpy(np.stack([synth.kitti_frame(100 + B * rank + i, NK) for i in range(B)])).to(dev)
    names = ["start", "hard_voxelize", "pillar_feature_net", "pointpillars_scatter", "dense", "ssd_head_postprocess"]

    def run(events):
        def mark(i):
            if events is not None:
                events[i].record()

        mark(0)
        voxels, coors, npv, nv = model.voxelizer(pts)
        mark(1)
        b, v, p, d = voxels.shape
        c4 = coors.view(b * v, 4)
        feats = model.pillar_encoder(voxels.view(b * v, p, d), npv.view(b * v), c4)
        mark(2)
        canvas = model.scatter(feats, c4, b)
        mark(3)
        x = model.neck(model.backbone(canvas))
        mark(4)
        out = model.head.post_process(model.head.head_map(x), model.anchor_generator, c4, device_only=True)
        mark(5)
        return out, nv

    with torch.no_grad():
        dt, per_op_ms, out, info = _timed_loop(run, args, world, dev, names)
    if rank != 0:
        return None
    alg_v = 4 * NK * D4 + 4 * V * PV * D4 + 16 * V + 4
    a = alg_v * B / (per_op_ms["hard_voxelize"] * 1e-3) / 1e9
    alg_s = 4 * V * 64 + 16 * V + 4 * 64 * 432 * 496
    a_s = alg_s * B / (per_op_ms["pointpillars_scatter"] * 1e-3) / 1e9

    def conv(cin, cout, k, h, w):
        return 2 * cin * cout * k * k * h * w

    s1 = 3 * conv(64, 64, 3, 248, 216) + 5 * conv(128, 128, 3, 124, 108) + 5 * conv(256, 256, 3, 62, 54)
    s2 = conv(64, 64, 3, 248, 216) + conv(64, 128, 3, 124, 108) + conv(128, 256, 3, 62, 54)
    other = conv(64, 128, 1, 248, 216) + conv(128, 128, 2, 124, 108) + conv(256, 128, 4, 62, 54)
    direct, executed = s1 + s2 + other, s1 / 4 + s2 + other
    return {
        "metric": "frames/sec PointPillars-KITTI (whole inference graph)",
        "value": world * B * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"PointPillars-KITTI: {NK} pts x {D4} per frame, 0.16 m pillars (432x496), P={PV}, "
                               f"max_voxels={V}, batch {B} distinct frames/GPU/step, random-init weights, "
                               "hard_voxelize->PillarFeatureNet(64)->PointPillarsScatter->SECOND+FPN->SSDHead->"
                               "anchor mask + decode + rotated NMS",
                   "frames_per_gpu_per_step": B, "max_voxels": V, "parallelism": f"dp{world} (frames)"},
        "roofline": dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBPS, unit="GB/s", frac=a / HBM_PEAK_GBPS, traffic=None,
                         ms_per_launch=per_op_ms["hard_voxelize"], units_per_launch=B,
                         algorithmic_bytes_per_unit=alg_v,
                         kernel="hard_voxelize launch sequence, tiled path; the fixed-shape [V, 32, 4] output is "
                                "20.5 of the 20.8 MB per frame"),
        "rooflines": {"pointpillars_scatter": (dict(bound="hbm", fused_into="dense_backbone_fpn", achieved=None,
                                                    peak=HBM_PEAK_GBPS, unit="GB/s", frac=None, traffic=None,
                                                    ms_per_launch=per_op_ms["pointpillars_scatter"], units_per_launch=B,
                                                    note="fused into the first backbone convolution: inverse-map "
                                                         "kernels only, no canvas written")
                                               if getattr(model, "fuse_scatter", False) else
                                               dict(bound="hbm", achieved=a_s, peak=HBM_PEAK_GBPS, unit="GB/s",
                                                    frac=a_s / HBM_PEAK_GBPS, traffic=None,
                                                    ms_per_launch=per_op_ms["pointpillars_scatter"],
                                                    units_per_launch=B, algorithmic_bytes_per_unit=alg_s)),
                      "dense_backbone_fpn": mfma_roofline(
                          {"f32": executed * B}, per_op_ms["dense"], B, executed_flops_per_unit=executed,
                          direct_form_flops_per_unit=direct,
                          note="executed flops: stride-1 3x3 layers by Winograd F(4x4,3x3) (a quarter of the direct "
                               "multiplies), the rest direct GEMMs, all on the fp32 matrix pipe")},
        "per_op_ms": per_op_ms, "voxels_first_frame": int(out[1][0]),
        "detections_first_frame": int(out[0][3][0]),
    }
