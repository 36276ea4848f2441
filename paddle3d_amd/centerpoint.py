"""CenterPoint / PointPillars inference graph over the HIP ops -- the host-side mirror of the reference's
forward path, with the reference's layer names, constructor arguments and parameter names so that the
forward graph (and a converted ``.pdparams`` state dict) drops in unchanged.

Reference call stack: CenterPoint.test_forward (paddle3d/models/detection/centerpoint/
centerpoint.py:155-166) -> extract_feat (:126-138) -> HardVoxelizer (models/voxelizers/voxelize.py:27-82)
-> PillarFeatureNet (models/voxel_encoders/pillar_encoder.py:64-210) -> PointPillarsScatter
(models/middle_encoders/pillar_scatter.py:34-105) -> SecondBackbone (models/backbones/second_backbone.py:
72-120) -> SecondFPN (models/necks/second_fpn.py:99-157) -> CenterHead.forward (center_head.py:212-220)
-> CenterHead.predict_by_custom_op (:294-339).

What differs from the reference, by design:
  * voxelize / PFN / scatter / postprocess are single fused HIP ops taking the WHOLE batch (the reference
    loops over samples in Python and syncs on num_voxels per sample, voxelize.py:43);
  * padded rows beyond num_voxels are never sliced off on the host: fixed-shape [B, V, ...] tensors flow
    through PFN and scatter, which ignore rows >= num_voxels through the batch column -1;
  * the dense 2-D convolutions run on the library's own fp32-MFMA kernels in eval mode (Winograd F(2x2,3x3)
    for the stride-1 3x3 layers, an implicit-GEMM kernel for the stride-2 ones, a patch-GEMM kernel for the
    FPN levels writing into the concatenated map, a grouped kernel for the final head convolutions), with
    BatchNorm folded; the 36 first-stage head convolutions that read the same shared feature map are issued
    as ONE convolution.  PD3_DENSE_BACKEND=miopen routes them through PyTorch-ROCm instead (also the fallback
    for shapes the kernels do not take, and the training-mode path).
torch is plumbing here (device memory, streams); the work is in libpaddle3d_amd.so.
"""
from __future__ import annotations

import math
import os
from typing import Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .ops import centerpoint_postprocess as _cp
from .ops import conv as _conv
from .ops import pointpillars_scatter as _ps
from .ops import voxel_encoder as _ve
from .ops import voxelize as _vox

__all__ = ["HardVoxelizer", "PillarFeatureNet", "HardVFE", "VoxelMean", "PointPillarsScatter", "SecondBackbone",
           "SecondFPN", "CenterHead", "CenterPoint", "centerpoint_pillars_nuscenes",
           "centerpoint_voxels_nuscenes", "load_paddle_state_dict"]


def _grid(voxel_size, point_cloud_range):
    pr = np.array(point_cloud_range, dtype=np.float32)
    vs = np.array(voxel_size, dtype=np.float32)
    return np.round((pr[3:] - pr[:3]) / vs).astype(np.int64)  # pillar_scatter.py:46-51


class HardVoxelizer(nn.Module):
    """models/voxelizers/voxelize.py:27-82.  forward(points [B, N, D]) -> (voxels [B,V,P,D],
    coors [B,V,4] int32 (batch, z, y, x; batch = -1 on padding rows), num_points [B,V], num_voxels [B])."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points_in_voxel, max_num_voxels):
        super().__init__()
        self.voxel_size = list(map(float, voxel_size))
        self.point_cloud_range = list(map(float, point_cloud_range))
        self.max_num_points_in_voxel = int(max_num_points_in_voxel)
        if isinstance(max_num_voxels, (tuple, list)):
            self.max_num_voxels = list(max_num_voxels)
        else:
            self.max_num_voxels = [max_num_voxels, max_num_voxels]

    def forward(self, points: torch.Tensor, num_points: torch.Tensor | None = None):
        v = self.max_num_voxels[0] if self.training else self.max_num_voxels[1]
        # the batch column (voxelize.py:51-57 builds it through a float cast + F.pad) is written by the op
        # itself; -1 marks padding rows
        voxels, _, npv, nv, coors = _vox.hard_voxelize_batch(points, self.voxel_size, self.point_cloud_range,
                                                             self.max_num_points_in_voxel, v, num_points,
                                                             with_batch_coors=True)
        return voxels, coors, npv, nv


class PFNLayer(nn.Module):
    """pillar_encoder.py:64-105; holds parameters only -- the arithmetic runs inside the fused HIP op."""

    def __init__(self, in_channels, out_channels, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        self.units = out_channels if last_layer else out_channels // 2
        self.linear = nn.Linear(in_channels, self.units, bias=False)
        self.norm = nn.BatchNorm1d(self.units, eps=1e-3, momentum=0.01)
        bound = 1 / math.sqrt(in_channels)
        nn.init.uniform_(self.linear.weight, -bound, bound)


class PillarFeatureNet(nn.Module):
    """pillar_encoder.py:108-210 (legacy=False, with_distance=False)."""

    def __init__(self, in_channels=4, feat_channels=(64,), with_distance=False, max_num_points_in_voxel=20,
                 voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1), legacy=False):
        super().__init__()
        if with_distance or legacy:
            raise NotImplementedError("with_distance / legacy PFN variants are not on the hot path")
        if len(feat_channels) not in (1, 2):
            raise NotImplementedError("the fused PFN op covers one or two PFN layers")
        self.in_channels = in_channels
        chans = [in_channels + 5] + list(feat_channels)
        self.pfn_layers = nn.ModuleList(
            PFNLayer(chans[i], chans[i + 1], last_layer=(i == len(chans) - 2)) for i in range(len(chans) - 1))
        # layer i+1 consumes [y_i | max(y_i)]: 2 * units_i inputs
        for i in range(1, len(self.pfn_layers)):
            assert self.pfn_layers[i].linear.in_features == 2 * self.pfn_layers[i - 1].units
        self.vx, self.vy = float(voxel_size[0]), float(voxel_size[1])
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.max_num_points_in_voxel = max_num_points_in_voxel
        self._folded = None

    def _fold(self):
        out = []
        for l in self.pfn_layers:
            scale, shift = _ve.fold_batchnorm(l.norm.weight, l.norm.bias, l.norm.running_mean,
                                              l.norm.running_var, l.norm.eps)
            out += [l.linear.weight.t().contiguous(), scale, shift]  # torch [out,in] -> Paddle [in,out]
        return [t.detach() for t in out]

    def forward(self, features, num_points_per_voxel, coors):
        """features [M,P,D], num_points [M] int32, coors [M,4] int32 -> [M, C]."""
        if self.training or self._folded is None:
            self._folded = self._fold()
        return _ve.pillar_feature_net(features, num_points_per_voxel, coors, self.vx, self.vy, self.x_offset,
                                      self.y_offset, *self._folded)


class HardVFE(nn.Module):
    """voxel_encoder.py:142-283 with with_cluster_center = with_voxel_center = True, two VFE layers (the
    BEVFusion LiDAR stream, configs/bevfusion/bevf_pp_2x8_1x_nusc.yaml:93-101); parameters only, arithmetic
    in the fused HIP op."""

    def __init__(self, in_channels=4, feat_channels=(64, 64), with_distance=False, with_cluster_center=True,
                 with_voxel_center=True, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1)):
        super().__init__()
        if with_distance or not (with_cluster_center and with_voxel_center) or len(feat_channels) != 2:
            raise NotImplementedError("HardVFE: only the BEVFusion configuration is on the hot path")
        self.voxel_size, self.point_cloud_range = tuple(voxel_size), tuple(point_cloud_range)
        chans = [in_channels + 6] + list(feat_channels)

        class _VFE(nn.Module):
            def __init__(self, cin, cout):
                super().__init__()
                self.linear = nn.Linear(cin, cout, bias=False)
                self.norm = nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01)

        self.vfe_layers = nn.ModuleList([_VFE(chans[0], chans[1]), _VFE(2 * chans[1], chans[2])])

    def forward(self, features, num_points, coors):
        args = []
        for l in self.vfe_layers:
            scale, shift = _ve.fold_batchnorm(l.norm.weight, l.norm.bias, l.norm.running_mean, l.norm.running_var,
                                              l.norm.eps)
            args += [l.linear.weight.t().contiguous().detach(), scale.detach(), shift.detach()]
        return _ve.hard_vfe(features, num_points, coors, self.voxel_size, self.point_cloud_range, *args)


class VoxelMean(nn.Module):
    """voxel_encoder.py:44-57: mean of the points of a voxel."""

    def __init__(self, in_channels=4):
        super().__init__()
        self.in_channels = in_channels

    def forward(self, features, num_points_per_voxel, coors=None):
        assert self.in_channels == features.shape[-1]
        return _ve.voxel_mean(features, num_points_per_voxel)


class PointPillarsScatter(nn.Module):
    """pillar_scatter.py:34-105."""

    def __init__(self, in_channels, voxel_size, point_cloud_range):
        super().__init__()
        self.in_channels = in_channels
        g = _grid(voxel_size, point_cloud_range)
        self.nx, self.ny = int(g[0]), int(g[1])

    def forward(self, voxel_features, coords, batch_size):
        return _ps.pointpillars_scatter(voxel_features, coords, batch_size, self.ny, self.nx)


def _conv_bn_relu(cin, cout, k, stride=1, padding=0, transpose=False, eps=1e-3, momentum=0.01, bias=False):
    conv = (nn.ConvTranspose2d if transpose else nn.Conv2d)(cin, cout, k, stride=stride, padding=padding, bias=bias)
    return [conv, nn.BatchNorm2d(cout, eps=eps, momentum=momentum), nn.ReLU()]


class SecondBackbone(nn.Module):
    """second_backbone.py:72-120 (parameter names blocks.<i>.<j>.* as in the reference)."""

    def __init__(self, in_channels=128, out_channels=(128, 128, 256), layer_nums=(3, 5, 5),
                 downsample_strides=(2, 2, 2)):
        super().__init__()
        in_filters = [in_channels, *out_channels[:-1]]
        blocks = []
        for i, n in enumerate(layer_nums):
            block = _conv_bn_relu(in_filters[i], out_channels[i], 3, stride=downsample_strides[i], padding=1)
            for _ in range(n):
                block += _conv_bn_relu(out_channels[i], out_channels[i], 3, padding=1)
            blocks.append(nn.Sequential(*block))
        self.blocks = nn.ModuleList(blocks)

    def forward(self, x):
        outs = []
        for blk in self.blocks:
            x = blk(x)
            outs.append(x)
        return tuple(outs)


class SecondFPN(nn.Module):
    """second_fpn.py:99-157 (use_spatial_attn_before_concat unsupported: unused on the path)."""

    def __init__(self, in_channels=(128, 128, 256), out_channels=(256, 256, 256), upsample_strides=(1, 2, 4),
                 use_conv_for_no_stride=False):
        super().__init__()
        deblocks = []
        for i, oc in enumerate(out_channels):
            stride = upsample_strides[i]
            if stride > 1 or (stride == 1 and not use_conv_for_no_stride):
                layer = _conv_bn_relu(in_channels[i], oc, int(stride), stride=int(stride), transpose=True)
            else:
                s = round(1 / stride)
                layer = _conv_bn_relu(in_channels[i], oc, s, stride=s)
            deblocks.append(nn.Sequential(*layer))
        self.deblocks = nn.ModuleList(deblocks)

    def forward(self, xs):
        ups = [d(x) for d, x in zip(self.deblocks, xs)]
        return torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]


class ConvModule(nn.Module):
    """center_head.py:43-78: conv(bias) -> BN(eps 1e-5) -> ReLU."""

    def __init__(self, cin, cout, k, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding, bias=True)
        self.bn = nn.BatchNorm2d(cout, eps=1e-5, momentum=0.1)
        self.activate = nn.ReLU()

    def forward(self, x):
        return self.activate(self.bn(self.conv(x)))


class SeparateHead(nn.Module):
    """center_head.py:81-153: per head `num_conv-1` ConvModules then a biased conv to `classes` maps."""

    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, init_bias=-2.19):
        super().__init__()
        self.heads = heads
        for head, (classes, num_conv) in heads.items():
            layers, c_in = [], in_channels
            for _ in range(num_conv - 1):
                layers.append(ConvModule(c_in, head_conv, final_kernel, padding=final_kernel // 2))
                c_in = head_conv
            layers.append(nn.Conv2d(head_conv, classes, final_kernel, padding=final_kernel // 2, bias=True))
            setattr(self, head, nn.Sequential(*layers))
        with torch.no_grad():
            getattr(self, "hm")[-1].bias.fill_(init_bias)

    def forward(self, x):
        return {h: getattr(self, h)(x) for h in self.heads}


def _fold_conv_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    if isinstance(conv, nn.ConvTranspose2d):
        w = conv.weight * scale.reshape(1, -1, 1, 1)
    else:
        w = conv.weight * scale.reshape(-1, 1, 1, 1)
    b = (conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean))
    return w.detach().contiguous(), ((b - bn.running_mean) * scale + bn.bias).detach().contiguous()


def _hip_conv3x3(x, w, b, stride, cache):
    """3x3 / pad 1 convolution + bias + ReLU on the hand-written kernels: stride 1 by Winograd F(4x4,3x3)
    (PD3_CONV_ALGO=winograd selects F(2x2,3x3), =direct turns Winograd off), the direct implicit-GEMM kernel
    otherwise; None if no kernel takes the shape.  `cache` keeps the packed weights."""
    cout, cin = w.shape[:2]
    h, wd = x.shape[2], x.shape[3]
    algo = os.environ.get("PD3_CONV_ALGO", "winograd43")
    if stride == 1 and algo == "winograd43" and _conv.winograd43_supported(cin, cout, h, wd):
        key = ("wino43", w.data_ptr())
        if key not in cache:
            cache[key] = _conv.pack_winograd43_weight(w)
        return _conv.conv3x3_winograd43_bias_relu(x, cache[key], b, cout, relu=True)
    if stride == 1 and algo in ("winograd", "winograd43") and _conv.winograd_supported(cin, cout, h, wd):
        key = ("wino", w.data_ptr())
        if key not in cache:
            cache[key] = _conv.pack_winograd_weight(w)
        return _conv.conv3x3_winograd_bias_relu(x, cache[key], b, cout, relu=True)
    if _conv.supported(cin, cout, h, wd, stride):
        key = ("direct", w.data_ptr())
        if key not in cache:
            cache[key] = _conv.pack_conv3x3_weight(w)
        return _conv.conv3x3_bias_relu(x, cache[key], b, cout, relu=True, stride=stride)
    return None


class CenterHead(nn.Module):
    """center_head.py:156-220 forward + :294-339 predict_by_custom_op (inference only)."""

    def __init__(self, in_channels, tasks, common_heads, init_bias=-2.19, share_conv_channel=64, num_hm_conv=2,
                 **_unused):
        super().__init__()
        self.num_classes = [len(t["class_names"]) for t in tasks]
        self.class_names = [t["class_names"] for t in tasks]
        self.with_velocity = "vel" in common_heads
        self.box_n_dim = 9 if self.with_velocity else 7
        self.shared_conv = ConvModule(in_channels, share_conv_channel, 3, padding=1)
        self.tasks = nn.ModuleList()
        for ncls in self.num_classes:
            heads = dict(common_heads)
            heads.update(hm=(ncls, num_hm_conv))
            self.tasks.append(SeparateHead(share_conv_channel, heads, final_kernel=3, init_bias=init_bias))
        self._fused = None
        self.dense_backend = os.environ.get("PD3_DENSE_BACKEND", "hip")

    # -- reference-shaped forward: list of per-task dicts -------------------------------------------
    def forward(self, x):
        if not self.training:
            return self._forward_fused(x)
        return self.forward_layers(x)

    def forward_layers(self, x):
        """Layer-by-layer statement of center_head.py:212-220 (what the fused path must equal)."""
        x = self.shared_conv(x)
        return [task(x) for task in self.tasks], x

    # -- inference: BN folded; the 36 first-stage 3x3 convs on the shared map run as ONE convolution -----
    def _build_fused(self):
        w0, b0 = _fold_conv_bn(self.shared_conv.conv, self.shared_conv.bn)
        ws, bs, plan = [], [], []
        finals = []
        for t, task in enumerate(self.tasks):
            for head in task.heads:
                seq = getattr(task, head)
                if len(seq) != 2:
                    raise NotImplementedError("fused CenterHead expects num_conv == 2 for every head")
                w, b = _fold_conv_bn(seq[0].conv, seq[0].bn)
                ws.append(w)
                bs.append(b)
                plan.append((t, head))
                finals.append((seq[1].weight.detach(), seq[1].bias.detach()))
        cmax = max(f[0].shape[0] for f in finals)
        hc = ws[0].shape[0]
        # grouped second stage: group g maps its own hc channels to cmax (zero padded) outputs
        wf = torch.zeros(len(finals) * cmax, hc, 3, 3, device=w0.device, dtype=w0.dtype)
        bf = torch.zeros(len(finals) * cmax, device=w0.device, dtype=w0.dtype)
        for g, (w, b) in enumerate(finals):
            wf[g * cmax:g * cmax + w.shape[0]] = w
            bf[g * cmax:g * cmax + w.shape[0]] = b
        self._fused = dict(w0=w0, b0=b0, w1=torch.cat(ws, 0).contiguous(), b1=torch.cat(bs, 0).contiguous(),
                           wf=wf, bf=bf, plan=plan, cmax=cmax, groups=len(finals),
                           ncls=[f[0].shape[0] for f in finals])

    def _forward_fused(self, x):
        if self._fused is None:
            self._build_fused()
        f = self._fused
        y = None
        if self.dense_backend == "hip" and x.is_cuda:
            pk = f.setdefault("packed", {})
            x1 = _hip_conv3x3(x, f["w0"], f["b0"], 1, pk)
            if x1 is not None:
                y = _hip_conv3x3(x1, f["w1"], f["b1"], 1, pk)
                if y is not None:
                    x = x1
        if y is None:
            x = F.relu(F.conv2d(x, f["w0"], f["b0"], padding=1))
            y = F.relu(F.conv2d(x, f["w1"], f["b1"], padding=1))
        if (self.dense_backend == "hip" and y.is_cuda
                and _conv.grouped_small_supported(f["wf"].shape[1], f["cmax"], y.shape[2], y.shape[3])):
            if "pf" not in f:
                f["pf"] = _conv.pack_grouped_weight(f["wf"], f["groups"])
            z = _conv.grouped_conv3x3_small(y, f["pf"], f["bf"], f["groups"])
        else:
            z = F.conv2d(y, f["wf"], f["bf"], padding=1, groups=f["groups"])
        rets = [dict() for _ in self.tasks]
        for g, (t, head) in enumerate(f["plan"]):
            # channel slices of z stay views: the postprocess op takes them with their common batch stride
            rets[t][head] = z[:, g * f["cmax"]:g * f["cmax"] + f["ncls"][g]]
        return rets, x

    @torch.no_grad()
    def predict_by_custom_op(self, preds_dicts, test_cfg, device_only=False):
        """center_head.py:294-339.  Returns per frame dict(box3d_lidar, label_preds, scores)."""
        hm, reg, height, dim, vel, rot = [], [], [], [], [], []
        for preds in preds_dicts:
            hm.append(preds["hm"])
            reg.append(preds["reg"])
            height.append(preds["height"])
            dim.append(preds["dim"])
            vel.append(preds["vel"] if self.with_velocity else preds["reg"])
            rot.append(preds["rot"])
        # the reference builds a len(tasks)**2 list of running class offsets (:303-309); entry t is task t's
        offsets = np.concatenate([[0], np.cumsum(self.num_classes)[:-1]]).astype(int).tolist()
        num_classes = offsets * len(preds_dicts)
        b, s, l, n = _cp.centerpoint_postprocess_device(
            hm, reg, height, dim, vel, rot, test_cfg["voxel_size"], test_cfg["point_cloud_range"],
            test_cfg["post_center_limit_range"], num_classes, test_cfg["down_ratio"], test_cfg["score_threshold"],
            test_cfg["nms"]["nms_iou_threshold"], test_cfg["nms"]["nms_pre_max_size"],
            test_cfg["nms"]["nms_post_max_size"], self.with_velocity, allow_batch=True)
        if device_only:
            return b, s, l, n
        counts = n.cpu().tolist()
        return [dict(box3d_lidar=b[i, :k], label_preds=l[i, :k], scores=s[i, :k]) for i, k in enumerate(counts)]


class CenterPoint(nn.Module):
    """centerpoint.py:45-166, inference path.  `test_forward(points)` takes a [B, N, D] tensor (or a list of
    [N_i, D] tensors, padded internally) and returns the per-frame detections."""

    def __init__(self, voxelizer, voxel_encoder, middle_encoder, backbone, neck, bbox_head, test_cfg,
                 box_with_velocity=True):
        super().__init__()
        self.voxelizer = voxelizer
        self.voxel_encoder = voxel_encoder
        self.middle_encoder = middle_encoder
        self.backbone = backbone
        self.neck = neck
        self.bbox_head = bbox_head
        self.test_cfg = test_cfg
        self.box_with_velocity = box_with_velocity
        self._dense = None
        # "hip" (default): the hand-written fp32-MFMA kernel (ops/conv.py) for every 3x3 convolution of the
        # backbone and the FPN levels; "miopen": PyTorch-ROCm convolutions
        self.dense_backend = os.environ.get("PD3_DENSE_BACKEND", "hip")
        self._packed = {}

    def _pack(self, points):
        if isinstance(points, torch.Tensor):
            return points, None
        n = max(p.shape[0] for p in points)
        d = points[0].shape[1]
        out = torch.zeros((len(points), n, d), dtype=torch.float32, device=points[0].device)
        for i, p in enumerate(points):
            out[i, : p.shape[0]] = p
        lens = torch.tensor([p.shape[0] for p in points], dtype=torch.int32, device=points[0].device)
        return out, lens

    def extract_pillars(self, points, num_points=None):
        """voxelize -> voxel encoder -> middle encoder: the LiDAR front half (dense BEV features)."""
        voxels, coors, npv, nv = self.voxelizer(points, num_points)
        b, v, p, d = voxels.shape
        voxels, coors, npv = voxels.view(b * v, p, d), coors.view(b * v, 4), npv.view(b * v)
        if not isinstance(self.middle_encoder, PointPillarsScatter):
            # the sparse middle encoder works on the occupied voxels only (one host sync, like the
            # reference's voxels[0:num_voxels] slice, voxelize.py:43)
            keep = coors[:, 0] >= 0
            voxels, coors, npv = voxels[keep], coors[keep].contiguous(), npv[keep]
        feats = self.voxel_encoder(voxels, npv, coors)
        return self.middle_encoder(feats, coors, b)

    def _dense_fold(self):
        """BatchNorm folded into the backbone / neck convolutions for inference."""
        seqs = []
        for blk in list(self.backbone.blocks) + list(self.neck.deblocks):
            layers = []
            mods = list(blk)
            for i in range(0, len(mods), 3):
                conv, bn = mods[i], mods[i + 1]
                w, bias = _fold_conv_bn(conv, bn)
                layers.append((isinstance(conv, nn.ConvTranspose2d), w, bias, conv.stride, conv.padding))
            seqs.append(layers)
        nb = len(self.backbone.blocks)
        self._dense = (seqs[:nb], seqs[nb:])

    def _run(self, layers, x):
        for tr, w, b, stride, padding in layers:
            if (self.dense_backend == "hip" and x.is_cuda and not tr and tuple(w.shape[2:]) == (3, 3)
                    and tuple(stride) in ((1, 1), (2, 2)) and tuple(padding) == (1, 1)):
                y = _hip_conv3x3(x, w, b, stride[0], self._packed)
                if y is not None:
                    x = y
                    continue
            x = F.conv_transpose2d(x, w, b, stride=stride, padding=padding) if tr else \
                F.conv2d(x, w, b, stride=stride, padding=padding)
            x = F.relu_(x)
        return x

    def _neck_fused(self, outs):
        """The FPN levels on the patch-GEMM kernel, each writing its channel slice of the concatenated map."""
        if self.dense_backend != "hip" or not outs[0].is_cuda:
            return None
        plan, hw, ctot = [], None, 0
        for layers, o in zip(self._dense[1], outs):
            if len(layers) != 1:
                return None
            tr, w, b, stride, padding = layers[0]
            mode = _conv.patch_mode(w, stride[0], tr)
            cout = w.shape[1] if tr else w.shape[0]
            cin = w.shape[0] if tr else w.shape[1]
            if (mode is None or tuple(padding) != (0, 0) or stride[0] != stride[1]
                    or not _conv.patch_supported(mode, cin, cout, o.shape[2], o.shape[3])):
                return None
            scale = {0: 0.5, 1: 1, 2: 2}[mode]
            size = (int(o.shape[2] * scale), int(o.shape[3] * scale))
            if hw is not None and size != hw:
                return None
            hw = size
            plan.append((mode, w, b, tr, cout, ctot))
            ctot += cout
        out = torch.empty((outs[0].shape[0], ctot, hw[0], hw[1]), dtype=torch.float32, device=outs[0].device)
        for (mode, w, b, tr, cout, off), o in zip(plan, outs):
            key = ("patch", w.data_ptr())
            if key not in self._packed:
                self._packed[key] = _conv.pack_patch_weight(w, mode, tr)
            _conv.patch_conv_bias_relu(o, self._packed[key], b, mode, cout, out, off, relu=True)
        return out

    def dense_forward(self, x):
        if self.training:
            return self.neck(self.backbone(x))
        if self._dense is None:
            self._dense_fold()
        outs = []
        for blk in self._dense[0]:
            x = self._run(blk, x)
            outs.append(x)
        fused = self._neck_fused(outs)
        if fused is not None:
            return fused
        ups = [self._run(d, o) for d, o in zip(self._dense[1], outs)]
        return torch.cat(ups, dim=1)

    @torch.no_grad()
    def test_forward(self, points, device_only=False):
        pts, lens = self._pack(points)
        x = self.extract_pillars(pts, lens)
        x = self.dense_forward(x)
        preds, _ = self.bbox_head(x)
        return self.bbox_head.predict_by_custom_op(preds, self.test_cfg, device_only=device_only)

    forward = test_forward


NUSC_TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "construction_vehicle"]),
              dict(num_class=2, class_names=["bus", "trailer"]), dict(num_class=1, class_names=["barrier"]),
              dict(num_class=2, class_names=["motorcycle", "bicycle"]),
              dict(num_class=2, class_names=["pedestrian", "traffic_cone"])]


def centerpoint_pillars_nuscenes(max_num_voxels=(30000, 60000)) -> CenterPoint:
    """configs/centerpoint/centerpoint_pillars_02voxel_nuscenes_10sweep.yml:110-179, random init."""
    pcr, vs = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], [0.2, 0.2, 8]
    test_cfg = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_per_img=500,
                    nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.2),
                    score_threshold=0.1, point_cloud_range=[-51.2, -51.2], down_ratio=4, voxel_size=[0.2, 0.2])
    return CenterPoint(
        voxelizer=HardVoxelizer(vs, pcr, 20, list(max_num_voxels)),
        voxel_encoder=PillarFeatureNet(5, (64, 64), False, 20, vs, pcr, legacy=False),
        middle_encoder=PointPillarsScatter(64, vs, pcr),
        backbone=SecondBackbone(64, (64, 128, 256), (3, 5, 5), (2, 2, 2)),
        neck=SecondFPN((64, 128, 256), (128, 128, 128), (0.5, 1, 2), use_conv_for_no_stride=True),
        bbox_head=CenterHead(384, NUSC_TASKS, dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2))),
        test_cfg=test_cfg, box_with_velocity=True)


def centerpoint_voxels_nuscenes(max_num_voxels=(120000, 160000)) -> CenterPoint:
    """configs/centerpoint/centerpoint_voxels_0075voxel_nuscenes_10sweep.yml:111-173, random init."""
    from .sparse import SparseResNet3D

    pcr, vs = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], [0.075, 0.075, 0.2]
    test_cfg = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_per_img=500,
                    nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.2),
                    score_threshold=0.1, point_cloud_range=[-54.0, -54.0], down_ratio=8, voxel_size=[0.075, 0.075])
    return CenterPoint(
        voxelizer=HardVoxelizer(vs, pcr, 10, list(max_num_voxels)),
        voxel_encoder=VoxelMean(5),
        middle_encoder=SparseResNet3D(5, vs, pcr),
        backbone=SecondBackbone(256, (128, 256), (5, 5), (1, 2)),
        neck=SecondFPN((128, 256), (256, 256), (1, 2), use_conv_for_no_stride=True),
        bbox_head=CenterHead(512, NUSC_TASKS, dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2))),
        test_cfg=test_cfg, box_with_velocity=True)


def load_paddle_state_dict(model: nn.Module, state: dict) -> list:
    """Copy a Paddle state dict (name -> ndarray, what ``paddle.load`` of a ``.pdparams`` returns; reference
    apis/checkpoint.py:148-212) into the torch mirror.  Layout rules: Linear weights
    are [in, out] in Paddle (transposed here), BatchNorm statistics are ``_mean`` / ``_variance``.
    Returns the list of keys that could not be placed."""
    own = dict(model.state_dict())
    missing = []
    with torch.no_grad():
        for k, v in state.items():
            t = torch.as_tensor(np.asarray(v))
            name = k.replace("._mean", ".running_mean").replace("._variance", ".running_var")
            if name not in own:
                missing.append(k)
                continue
            if name.endswith("linear.weight") and t.dim() == 2:
                t = t.t()
            if own[name].shape != t.shape:
                missing.append(k)
                continue
            own[name].copy_(t)
    for m in model.modules():
        for attr in ("_folded", "_fused", "_dense"):
            if hasattr(m, attr):
                setattr(m, attr, None)
    return missing
