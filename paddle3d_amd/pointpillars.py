"""PointPillars (KITTI) inference graph over the HIP ops -- the host-side mirror of the reference's
paddle3d/models/detection/pointpillars/ package: AnchorGenerator (anchors_generator.py:21-156), SSDHead
(pointpillars_head.py:31-196) and PointPillars.test_forward (pointpillars.py:107-127), with the reference's layer and
parameter names (`head.cls_head.weight` ...) so that a converted ``.pdparams`` state dict drops in.

Reference call stack (configs/pointpillars/pointpillars_xyres16_kitti_car.yml:86-126): HardVoxelize (a dataset
transform in the reference, transforms/reader + ops/voxelize's numba twin; the device op here) -> PillarFeatureNet ->
PointPillarsScatter -> SecondBackbone -> SecondFPN (transposed convolutions 1 / 2 / 4) -> SSDHead.forward (three 1x1
convolutions) -> AnchorGenerator(coords) per frame -> SSDHead.post_process (decode, mask, sigmoid, filter, rotated NMS).

What differs from the reference, by design: the three head convolutions run as ONE 1x1 GEMM whose NCHW output is
consumed in place (the reference transposes each map to [B, H*W*A, width]); anchor masks, decoding, filtering, NMS
and result assembly of the whole batch are one launch sequence without a host round trip (`ops/ssd_head.py`); a
frame without detections comes back with zero rows (the reference hands `_box_empty`'s marker row to
`_parse_result_to_sample`, which turns it into a Sample without boxes).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from ._lib import Paddle3DAmdError
from .centerpoint import (HardVoxelizer, PillarFeatureNet, PointPillarsScatter, SecondBackbone, SecondFPN,
                          _InferenceCache)
from .ops import conv as _conv
from .ops import ssd_head as _ssd

__all__ = ["AnchorGenerator", "SSDHead", "PointPillars", "pointpillars_kitti_car",
           "pointpillars_kitti_cyclist_pedestrian"]


def _limit_period(val, offset=0.5, period=math.pi):
    f32 = np.float32
    return val - np.floor(val / f32(period) + f32(offset)) * f32(period)  # anchors_generator.py:178-179


class AnchorGenerator(nn.Module):
    """anchors_generator.py:21-156.  `anchors` [A, 7] fp32 (x, y, z, w, l, h, r) in (y, x, config, rotation) order and
    `anchors_bv` [A, 4] int32 (pillar-index boxes xmin, ymin, xmax, ymax) are built once on the host in float32,
    step by step as the reference does; the per-frame mask (generate_anchors_mask) is part of the device op."""

    def __init__(self, output_stride_factor, point_cloud_range, voxel_size, anchor_configs, anchor_area_threshold=1):
        super().__init__()
        f32 = np.float32
        pr, vs = np.asarray(point_cloud_range, f32), np.asarray(voxel_size, f32)
        grid = np.round((pr[3:6] - pr[:3]) / vs).astype(np.int64)
        self.grid_size = (int(grid[0]), int(grid[1]))
        fw, fh = int(grid[0] // output_stride_factor), int(grid[1] // output_stride_factor)
        self.feature_map_size = (fh, fw)
        groups = []
        for cfg in anchor_configs:  # AnchorGeneratorStride.generate, :103-137 ([z, y, x, size, rotation, 7])
            sx, sy, sz = (f32(v) for v in cfg["anchor_strides"])
            ox, oy, oz = (f32(v) for v in cfg["anchor_offsets"])
            xc = np.arange(fw, dtype=f32) * sx + ox
            yc = np.arange(fh, dtype=f32) * sy + oy
            zc = np.arange(1, dtype=f32) * sz + oz
            sizes = np.asarray(cfg["sizes"], f32).reshape(-1, 3)
            rots = np.asarray(cfg["rotations"], f32)
            a = np.empty((fh, fw, sizes.shape[0], rots.shape[0], 7), f32)
            a[..., 0] = xc[None, :, None, None]
            a[..., 1] = yc[:, None, None, None]
            a[..., 2] = zc[0]
            a[..., 3:6] = sizes[None, None, :, None, :]
            a[..., 6] = rots[None, None, None, :]
            groups.append(a.reshape(fh, fw, -1, 7))
        anchors = np.concatenate(groups, axis=2)
        self.num_anchors_per_loc = int(anchors.shape[2])
        anchors = np.ascontiguousarray(anchors.reshape(-1, 7))
        # rbbox2d_to_circumscribed (:158-176), then pillar indices clipped on one side each (:62-78)
        r = np.abs(_limit_period(anchors[:, 6]))
        lying = r > f32(math.pi / 4)
        dx = np.where(lying, anchors[:, 4], anchors[:, 3])
        dy = np.where(lying, anchors[:, 3], anchors[:, 4])
        bv = np.empty((anchors.shape[0], 4), f32)
        bv[:, 0] = np.maximum(np.floor((anchors[:, 0] - dx / f32(2) - pr[0]) / vs[0]), 0)
        bv[:, 1] = np.maximum(np.floor((anchors[:, 1] - dy / f32(2) - pr[1]) / vs[1]), 0)
        bv[:, 2] = np.minimum(np.floor((anchors[:, 0] + dx / f32(2) - pr[0]) / vs[0]), f32(grid[0] - 1))
        bv[:, 3] = np.minimum(np.floor((anchors[:, 1] + dy / f32(2) - pr[1]) / vs[1]), f32(grid[1] - 1))
        bv = bv.astype(np.int32)
        if (bv < 0).any() or (bv[:, [0, 2]] >= grid[0]).any() or (bv[:, [1, 3]] >= grid[1]).any():
            raise Paddle3DAmdError("AnchorGenerator: an anchor's bird's-eye box leaves the pillar grid "
                                   "(unsupported configuration, status -3)")
        self.register_buffer("anchors", torch.from_numpy(anchors), persistent=False)
        self.register_buffer("anchors_bv", torch.from_numpy(bv), persistent=False)
        self.anchor_area_threshold = float(anchor_area_threshold)


class SSDHead(_InferenceCache, nn.Module):
    """pointpillars_head.py:31-196.  The modules hold the parameters (reference names cls_head / box_head / dir_head);
    forward runs the three 1x1 convolutions as one GEMM on the patch-GEMM kernel (bias fused, no activation)."""

    def __init__(self, num_classes, feature_channels=384, num_anchor_per_loc=2, encode_background_as_zeros=True,
                 use_direction_classifier=True, box_code_size=7, nms_score_threshold=0.05, nms_pre_max_size=1000,
                 nms_post_max_size=300, nms_iou_threshold=0.5, prediction_center_limit_range=None):
        super().__init__()
        if box_code_size != 7:
            raise NotImplementedError("SSDHead: box_code_size 7 is the one the decoder on the path defines")
        self.encode_background_as_zeros = bool(encode_background_as_zeros)
        self.use_direction_classifier = bool(use_direction_classifier)
        self.box_code_size = box_code_size
        self.nms_score_threshold = float(nms_score_threshold)
        self.nms_pre_max_size = int(nms_pre_max_size)
        self.nms_post_max_size = int(nms_post_max_size)
        self.nms_iou_threshold = float(nms_iou_threshold)
        self.pred_center_limit_range = (None if prediction_center_limit_range is None
                                        else [float(v) for v in prediction_center_limit_range])
        self.num_classes = int(num_classes)
        self._num_classes = self.num_classes if self.encode_background_as_zeros else self.num_classes + 1
        self.num_anchor_per_loc = int(num_anchor_per_loc)
        self.cls_head = nn.Conv2d(feature_channels, self.num_anchor_per_loc * self._num_classes, 1)
        self.box_head = nn.Conv2d(feature_channels, self.num_anchor_per_loc * box_code_size, 1)
        if self.use_direction_classifier:
            self.dir_head = nn.Conv2d(feature_channels, self.num_anchor_per_loc * 2, 1)

    def _plan(self):
        if not self._cache_valid():
            heads = [self.cls_head, self.box_head] + ([self.dir_head] if self.use_direction_classifier else [])
            w = torch.cat([h.weight.detach() for h in heads], 0).contiguous()
            b = torch.cat([h.bias.detach() for h in heads], 0).contiguous()
            cls_c, box_c = self.cls_head.out_channels, self.box_head.out_channels
            self._store_cache(dict(w=_conv.pack_patch_weight(w, 1, False), b=b, cout=int(w.shape[0]),
                                   cin=int(w.shape[1]), cls0=0, box0=cls_c,
                                   dir0=cls_c + box_c if self.use_direction_classifier else -1))
        return self._cache

    def head_map(self, features):
        """[B, C, H, W] -> the fused head map [B, cls | box | dir channels, H, W]."""
        self._require_eval()
        f = self._plan()
        h, w = int(features.shape[2]), int(features.shape[3])
        if features.shape[1] != f["cin"] or not _conv.patch_supported(1, f["cin"], f["cout"], h, w):
            raise Paddle3DAmdError(f"SSDHead: unsupported configuration ({features.shape[1]} channels into a "
                                   f"{f['cin']} -> {f['cout']} 1x1 head on a {h}x{w} map) (status -3)")
        out = torch.empty((features.shape[0], f["cout"], h, w), dtype=torch.float32, device=features.device)
        return _conv.patch_conv_bias_relu(features, f["w"], f["b"], 1, f["cout"], out, 0, relu=False)

    def forward(self, features):
        """pointpillars_head.py:79-96: dict(cls_preds [B, A, classes], box_preds [B, A, 7], dir_preds [B, A, 2]) as
        permuted VIEWS of the fused map (which rides along under "head_map" for post_process)."""
        m = self.head_map(features)
        f, b = self._plan(), int(features.shape[0])

        def view(c0, c1, width):
            return m[:, c0:c1].permute(0, 2, 3, 1).reshape(b, -1, width)

        ret = dict(cls_preds=view(f["cls0"], f["box0"], self._num_classes),
                   box_preds=view(f["box0"], f["box0"] + self.box_head.out_channels, self.box_code_size), head_map=m)
        if self.use_direction_classifier:
            ret["dir_preds"] = view(f["dir0"], f["cout"], 2)
        return ret

    @torch.no_grad()
    def post_process(self, preds, anchor_generator: AnchorGenerator, coors, device_only=False, full_sort=False):
        """pointpillars_head.py:86-196 for the whole batch.  `preds` = forward()'s dict (or the fused head map);
        coors [M, 4] int32 (batch, z, y, x) of the batch's pillars, padding rows with batch -1.
        -> per frame dict(box3d_lidar [K, 7], scores [K], label_preds [K] int64); K = 0 where the reference returns
        its empty marker.  device_only=True returns the padded device tensors and counts (no sync)."""
        m = preds["head_map"] if isinstance(preds, dict) else preds
        f = self._plan()
        if tuple(m.shape[2:]) != tuple(anchor_generator.feature_map_size) or \
                anchor_generator.num_anchors_per_loc != self.num_anchor_per_loc:
            raise Paddle3DAmdError("SSDHead.post_process: the anchors do not match the head map "
                                   f"({tuple(m.shape[2:])} x {self.num_anchor_per_loc} vs "
                                   f"{anchor_generator.feature_map_size} x {anchor_generator.num_anchors_per_loc})")
        b, s, l, n = _ssd.ssd_postprocess_device(
            m, f["cls0"], f["box0"], f["dir0"], self.num_anchor_per_loc, self.num_classes,
            self.encode_background_as_zeros, anchor_generator.anchors, anchor_generator.anchors_bv, coors,
            anchor_generator.grid_size, anchor_generator.anchor_area_threshold, self.nms_score_threshold,
            self.pred_center_limit_range, self.nms_iou_threshold, self.nms_pre_max_size, self.nms_post_max_size,
            full_sort=full_sort)
        if device_only:
            return b, s, l, n
        counts = n.cpu().tolist()
        return [dict(box3d_lidar=b[i, :k], scores=s[i, :k], label_preds=l[i, :k]) for i, k in enumerate(counts)]


class PointPillars(nn.Module):
    """pointpillars.py:35-127, inference path.  `test_forward(points)` takes [B, N, D] points (or a list of [N_i, D]
    tensors) and voxelizes on the device; `test_forward_voxels(voxels, coords, num_points_per_voxel, batch_size)`
    takes the reference's pre-voxelized samples (its HardVoxelize transform runs in the data loader)."""

    def __init__(self, voxelizer, pillar_encoder, middle_encoder, backbone, neck, head, anchor_configs,
                 anchor_area_threshold=1):
        super().__init__()
        self.voxelizer = voxelizer
        self.pillar_encoder = pillar_encoder
        self.middle_encoder = middle_encoder
        # the scatter is fused into the backbone's first convolution where it can be (the module itself keeps
        # returning the dense pseudo image to any other caller)
        self.fuse_scatter = isinstance(middle_encoder, PointPillarsScatter) and isinstance(backbone, SecondBackbone)
        self.backbone = backbone
        self.neck = neck
        self.head = head
        # pointpillars.py:65-71: output stride of the head map = first downsample stride // first upsample stride
        ds = [blk[0].stride[0] for blk in backbone.blocks]
        us = neck.deblocks[0][0].stride[0]
        self.anchor_generator = AnchorGenerator(output_stride_factor=ds[0] // us,
                                                point_cloud_range=voxelizer.point_cloud_range,
                                                voxel_size=voxelizer.voxel_size, anchor_configs=anchor_configs,
                                                anchor_area_threshold=anchor_area_threshold)

    def scatter(self, feats, coords, batch_size):
        """The middle encoder as this model's forward runs it (a SparseCanvas when the scatter is fused)."""
        if self.fuse_scatter:
            return self.middle_encoder.sparse(feats, coords, batch_size)
        return self.middle_encoder(feats, coords, batch_size)

    def _pack(self, points):
        if isinstance(points, torch.Tensor):
            return points, None
        n = max(p.shape[0] for p in points)
        out = torch.zeros((len(points), n, points[0].shape[1]), dtype=torch.float32, device=points[0].device)
        for i, p in enumerate(points):
            out[i, : p.shape[0]] = p
        lens = torch.tensor([p.shape[0] for p in points], dtype=torch.int32, device=points[0].device)
        return out, lens

    @torch.no_grad()
    def test_forward_voxels(self, voxels, coords, num_points_per_voxel, batch_size, device_only=False):
        """pointpillars.py:107-127 on voxelized input: voxels [M, P, D], coords [M, 4] int32 (batch, z, y, x; -1 on
        padding rows), num_points_per_voxel [M] int32."""
        feats = self.pillar_encoder(voxels, num_points_per_voxel, coords)
        x = self.scatter(feats, coords, batch_size)
        x = self.neck(self.backbone(x))
        return self.head.post_process(self.head.head_map(x), self.anchor_generator, coords, device_only=device_only)

    @torch.no_grad()
    def test_forward(self, points, device_only=False):
        pts, lens = self._pack(points)
        voxels, coors, npv, _ = self.voxelizer(pts, lens)
        b, v, p, d = voxels.shape
        return self.test_forward_voxels(voxels.view(b * v, p, d), coors.view(b * v, 4), npv.view(b * v), b,
                                        device_only=device_only)

    forward = test_forward


KITTI_CAR_ANCHORS = [dict(sizes=[1.6, 3.9, 1.56], anchor_strides=[0.32, 0.32, 0.0], anchor_offsets=[0.16, -39.52, -1.78],
                          rotations=[0, 1.57], matched_threshold=0.6, unmatched_threshold=0.45)]


def pointpillars_kitti_car(max_num_voxels=(16000, 40000)) -> PointPillars:
    """configs/pointpillars/pointpillars_xyres16_kitti_car.yml:86-146, random init."""
    pcr, vs = [0.0, -39.68, -3.0, 69.12, 39.68, 1.0], [0.16, 0.16, 4.0]
    return PointPillars(
        voxelizer=HardVoxelizer(vs, pcr, 32, list(max_num_voxels)),
        pillar_encoder=PillarFeatureNet(4, (64,), False, 32, vs, pcr, legacy=False),
        middle_encoder=PointPillarsScatter(64, vs, pcr),
        backbone=SecondBackbone(64, (64, 128, 256), (3, 5, 5), (2, 2, 2)),
        neck=SecondFPN((64, 128, 256), (128, 128, 128), (1, 2, 4), use_conv_for_no_stride=False),
        head=SSDHead(num_classes=1, feature_channels=384, num_anchor_per_loc=2, encode_background_as_zeros=True,
                     use_direction_classifier=True, box_code_size=7, nms_score_threshold=0.05, nms_pre_max_size=1000,
                     nms_post_max_size=300, nms_iou_threshold=0.5,
                     prediction_center_limit_range=[0.0, -39.68, -5.0, 69.12, 39.68, 5.0]),
        anchor_configs=KITTI_CAR_ANCHORS, anchor_area_threshold=1)


KITTI_CYCLIST_PEDESTRIAN_ANCHORS = [
    dict(sizes=[0.6, 1.76, 1.73], anchor_strides=[0.16, 0.16, 0.0], anchor_offsets=[0.08, -19.76, -1.465],
         rotations=[0, 1.57], matched_threshold=0.5, unmatched_threshold=0.35),
    dict(sizes=[0.6, 0.8, 1.73], anchor_strides=[0.16, 0.16, 0.0], anchor_offsets=[0.08, -19.76, -1.465],
         rotations=[0, 1.57], matched_threshold=0.5, unmatched_threshold=0.35)]


def pointpillars_kitti_cyclist_pedestrian(max_num_voxels=(12000, 12000)) -> PointPillars:
    """configs/pointpillars/pointpillars_xyres16_kitti_cyclist_pedestrian.yml:86-150, random init: 100 points per
    pillar, a stride-1 first backbone block (the head map has the pillar grid's resolution, 248 x 296), two classes,
    four anchors per location."""
    pcr, vs = [0.0, -19.84, -2.5, 47.36, 19.84, 0.5], [0.16, 0.16, 3.0]
    return PointPillars(
        voxelizer=HardVoxelizer(vs, pcr, 100, list(max_num_voxels)),
        pillar_encoder=PillarFeatureNet(4, (64,), False, 100, vs, pcr, legacy=False),
        middle_encoder=PointPillarsScatter(64, vs, pcr),
        backbone=SecondBackbone(64, (64, 128, 256), (3, 5, 5), (1, 2, 2)),
        neck=SecondFPN((64, 128, 256), (128, 128, 128), (1, 2, 4), use_conv_for_no_stride=False),
        head=SSDHead(num_classes=2, feature_channels=384, num_anchor_per_loc=4, encode_background_as_zeros=True,
                     use_direction_classifier=True, box_code_size=7, nms_score_threshold=0.05, nms_pre_max_size=1000,
                     nms_post_max_size=300, nms_iou_threshold=0.5,
                     prediction_center_limit_range=[0.0, -19.84, -2.5, 47.36, 19.84, 0.5]),
        anchor_configs=KITTI_CYCLIST_PEDESTRIAN_ANCHORS, anchor_area_threshold=1)
