"""Sparse middle encoder of CenterPoint-Voxel over the sparse_conv3d ops: layer mirrors of what the reference
builds from Paddle-core sparse layers (paddle3d/models/middle_encoders/sparse_resnet.py:31-206), inference only,
BatchNorm folded into the convolution epilogue, rulebooks shared through `indice_key` like the reference's `key=`.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from .ops import sparse_conv3d as _sp

__all__ = ["SparseConvTensor", "SubmConv3D", "Conv3D", "SparseBasicBlock", "SparseResNet3D", "SparseNet3D"]


class SparseConvTensor:
    """features [N, C] fp32 + indices [N, 4] int32 (b, z, y, x) + dense spatial shape (D, H, W)."""

    def __init__(self, features, indices, spatial_shape, batch_size, cache=None, plan=None, n_dev=None, amp=False):
        self.amp = bool(amp)  # mixed precision: convolutions the fp16 kernel serves take / leave fp16 rows
        self.features = features
        self.indices = indices
        self.spatial_shape = tuple(int(s) for s in spatial_shape)
        self.batch_size = int(batch_size)
        self.cache = {} if cache is None else cache  # indice_key -> SparseIndices
        self.plan = plan  # id(conv module) -> SparseIndices, when the encoder planned its index sets up front
        # [1] int32 on the device: the real row count when the arrays are at a remembered capacity (a plan made
        # without a host sync); rows past it hold nothing
        self.n_dev = n_dev

    def replace(self, features):
        return SparseConvTensor(features, self.indices, self.spatial_shape, self.batch_size, self.cache, self.plan,
                                self.n_dev, self.amp)

    def dense(self):
        return _sp.to_dense(self.features.float(), self.indices, self.batch_size, self.spatial_shape, self.n_dev)


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(int(x) for x in v)


_STATS = None  # dict(pairs=..., dense=...) while count_flops() runs


def count_flops(encoder, voxel_features, coors, batch_size) -> dict:
    """Flops of one forward of a sparse encoder: `pairs` = 2 * Cin * Cout per existing (output row, kernel offset)
    pair, `dense` = the same with all kernel offsets counted, `pairs_by_pipe` = `pairs` split by the matrix pipe the
    layer's kernel runs on ("f32", "bf16x3" = fp32 arithmetic as six bf16 products, "f16").  Host syncs per
    convolution: not for timed code."""
    global _STATS
    _STATS = dict(pairs=0, dense=0, pairs_by_pipe={})
    keep = getattr(encoder, "remember_capacities", None)
    if keep is not None:
        encoder.remember_capacities = False  # exact-size rulebooks: every row of `nbr` is a real row
    try:
        encoder(voxel_features, coors, batch_size)
        return dict(_STATS)
    finally:
        _STATS = None
        if keep is not None:
            encoder.remember_capacities = keep


class _SparseConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, subm=False,
                 key=None):
        super().__init__()
        self.ks, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        self.subm, self.key = subm, key
        # Paddle layout [kd, kh, kw, Cin, Cout]
        self.weight = nn.Parameter(torch.empty(*self.ks, in_channels, out_channels))
        fan_in = in_channels * math.prod(self.ks)
        nn.init.uniform_(self.weight, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def spec(self):
        # the tile order of the rulebook (ops.sparse_conv3d) costs one sort per rulebook: it pays for the strided layers
        # and from 32 output channels on (16 -> 16 over 1 M rows: 0.18 ms with it, 0.18 ms without)
        worth = (not self.subm) or int(self.weight.shape[-1]) >= 32
        return _sp.ConvSpec(self.ks, self.stride, self.padding, self.subm, self.key, worth)

    def _indices(self, x: SparseConvTensor):
        if x.plan is not None and id(self) in x.plan:
            return x.plan[id(self)]
        if self.subm and self.key is not None and self.key in x.cache:
            return x.cache[self.key]
        pad = tuple(k // 2 for k in self.ks) if self.subm else self.padding
        idx = _sp.indices(x.indices, x.batch_size, x.spatial_shape, self.ks, self.stride, pad, self.subm)
        if self.subm and self.key is not None:
            x.cache[self.key] = idx
        return idx

    out_f32 = False  # mixed precision: this convolution writes fp32 rows (an encoder's last layer, in front of to_dense)

    def _packed_f16(self):
        """The weight in the fp16 kernel's operand order, repacked when the parameter changes."""
        tag = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        hit = getattr(self, "_pd3_packed", None)
        if hit is None or hit[0] != tag:
            hit = (tag, _sp.pack_weight_f16(self.weight.detach()))
            object.__setattr__(self, "_pd3_packed", hit)
        return hit[1]

    def _packed_bf16x3(self):
        """The weight's three bf16 pieces in the bf16x3 kernel's operand order, repacked when the parameter changes."""
        tag = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        hit = getattr(self, "_pd3_packed_x3", None)
        if hit is None or hit[0] != tag:
            hit = (tag, _sp.pack_weight_bf16x3(self.weight.detach()))
            object.__setattr__(self, "_pd3_packed_x3", hit)
        return hit[1]

    def forward(self, x: SparseConvTensor, scale=None, shift=None, residual=None, relu=False):
        idx = self._indices(x)
        cin, cout = int(self.weight.shape[-2]), int(self.weight.shape[-1])
        if _STATS is not None:  # measurement aid (bench.py): multiply-adds of the pairs that exist
            fl = 2 * cin * cout * int((idx.nbr >= 0).sum().item())
            pipe = ("f16" if x.amp and _sp.f16_supported(cin, cout, idx.kernel_volume) else
                    "bf16x3" if _sp.SPLIT_BF16 and _sp.bf16x3_pays(cin, cout, idx.kernel_volume) else "f32")
            _STATS["pairs"] += fl
            _STATS["pairs_by_pipe"][pipe] = _STATS["pairs_by_pipe"].get(pipe, 0) + fl
            _STATS["dense"] += 2 * cin * cout * idx.n_out * idx.kernel_volume
        if x.amp and _sp.f16_supported(cin, cout, idx.kernel_volume):
            # fp16 rows in (converted once where the chain leaves the narrow fp32 layers), fp16 rows out
            feats = x.features if x.features.dtype == torch.float16 else x.features.half()
            res = residual if residual is None or residual.dtype == torch.float16 else residual.half()
            out = _sp.features_f16(feats, idx, self._packed_f16(), cin, cout, self.bias, scale, shift, res, relu,
                                   out_f32=self.out_f32)
        else:
            feats = x.features if x.features.dtype == torch.float32 else x.features.float()
            res = residual if residual is None or residual.dtype == torch.float32 else residual.float()
            if _sp.SPLIT_BF16 and _sp.bf16x3_pays(cin, cout, idx.kernel_volume):
                # fp32 arithmetic on the bf16 matrix cores (three bf16 pieces per operand, six products)
                out = _sp.features_bf16x3(feats, idx, self._packed_bf16x3(), cin, cout, self.bias, scale, shift, res,
                                          relu)
            else:
                out = _sp.features(feats, idx, self.weight, self.bias, scale, shift, res, relu)
        if self.subm:
            return x.replace(out)
        return SparseConvTensor(out, idx.out_coords, idx.out_shape, x.batch_size, plan=x.plan, n_dev=idx.n_out_dev,
                                amp=x.amp)


class SubmConv3D(_SparseConv):
    """paddle.sparse.nn.SubmConv3D: output index set = input index set.  The reference passes padding=1 with
    kernel 3 (sparse_resnet.py:31-44); a submanifold conv is defined with "same" padding k // 2."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=None, bias=True, key=None):
        super().__init__(in_channels, out_channels, kernel_size, 1, 0, bias, subm=True, key=key)


class Conv3D(_SparseConv):
    """paddle.sparse.nn.Conv3D: regular sparse convolution (output set dilates / downsamples)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias, subm=False)


def _fold(bn: nn.BatchNorm1d):
    """(scale, shift) of an eval-mode BatchNorm, computed once and kept on the module.  The cache entry is tied to
    the identity AND the in-place version of the four tensors, so load_state_dict / .to() / an optimizer step all
    invalidate it."""
    src = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
    tag = tuple((t.data_ptr(), t._version, t.device) for t in src)
    hit = getattr(bn, "_pd3_folded", None)
    if hit is not None and hit[0] == tag:
        return hit[1], hit[2]
    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().contiguous()
    shift = (bn.bias - bn.running_mean * scale).detach().contiguous()
    object.__setattr__(bn, "_pd3_folded", (tag, scale, shift))
    return scale, shift


def _planned_input(encoder, voxel_features, coors, batch_size, caps=None):
    """The encoder's index sets are planned before any feature is touched (ops.sparse_conv3d.plan: sorted key
    sets, one host sync, exact-size rulebooks); rows travel in raster order, so tiles of consecutive rows are
    spatial neighbours (a tile's kernel offsets are mostly all-present or all-absent, its gathers share cache
    lines).  `coors` may carry the voxelizer's padding rows (batch = -1)."""
    convs = [m for m in encoder.modules() if isinstance(m, _SparseConv)]
    pl = _sp.plan(coors, batch_size, encoder.sparse_shape, [m.spec() for m in convs], caps=caps)
    feats = voxel_features.index_select(0, pl.order)
    return SparseConvTensor(feats, pl.coords, encoder.sparse_shape, batch_size,
                            plan={id(m): idx for m, idx in zip(convs, pl.indices)}, n_dev=pl.n_in_dev,
                            amp=getattr(encoder, "amp", False)), pl


def _bn(channels):
    """paddle.sparse.nn.BatchNorm(eps 1e-3) acts on the values only: a BatchNorm1d over the feature rows.  The
    parameter names (weight, bias, running_mean / running_var <- _mean / _variance) follow the reference's."""
    return nn.BatchNorm1d(channels, eps=1e-3, momentum=0.01)


def _run_sequential(seq: nn.Sequential, x: SparseConvTensor) -> SparseConvTensor:
    """Runs a reference-shaped Sequential(conv, BatchNorm, ReLU, blocks ...): every conv + BatchNorm (+ ReLU)
    triple is ONE fused kernel call (BatchNorm folded into the convolution epilogue)."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, _SparseConv):
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) else None
            relu = bn is not None and i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            if bn is None:
                x = m(x)
                i += 1
            else:
                s, b = _fold(bn)
                x = m(x, s, b, None, relu=relu)
                i += 3 if relu else 2
        elif isinstance(m, nn.Sequential):
            x = _run_sequential(m, x)
            i += 1
        elif isinstance(m, (nn.BatchNorm1d, nn.ReLU)):
            raise RuntimeError("sparse Sequential: BatchNorm / ReLU without a convolution in front")
        else:
            x = m(x)
            i += 1
    return x


class SparseBasicBlock(nn.Module):
    """sparse_resnet.py:63-111: conv-bn-relu, conv-bn, + identity, relu (all submanifold, shared rulebook)."""

    def __init__(self, in_channels, out_channels, indice_key=None):
        super().__init__()
        self.conv1 = SubmConv3D(in_channels, out_channels, 3, bias=True, key=indice_key)
        self.bn1 = _bn(out_channels)
        self.conv2 = SubmConv3D(out_channels, out_channels, 3, bias=True, key=indice_key)
        self.bn2 = _bn(out_channels)

    def forward(self, x):
        s1, b1 = _fold(self.bn1)
        out = self.conv1(x, s1, b1, None, relu=True)
        s2, b2 = _fold(self.bn2)
        return self.conv2(out, s2, b2, x.features, relu=True)  # relu(bn2(conv2) + identity), :104-109


def _sparse_shape(voxel_size, point_cloud_range):
    pcr = np.array(point_cloud_range, dtype=np.float32)
    vs = np.array(voxel_size, dtype=np.float32)
    grid = np.round((pcr[3:] - pcr[:3]) / vs).astype(np.int64)
    return tuple(int(v) for v in (np.array(grid[::-1]) + [1, 0, 0]))  # sparse_resnet.py:173 / sparsenet.py:121


def _densify(x: SparseConvTensor):
    return x.dense()  # to_dense + transpose([0, 4, 1, 2, 3]) + reshape [N, C * D, H, W], sparse_resnet.py:202-205


class SparseResNet3D(nn.Module):
    """sparse_resnet.py:115-206, with the reference's parameter names (Sequential indices: conv_input.0.weight,
    conv_input.1._mean, conv2.0.weight, conv2.3.conv1.weight ...) so that a converted checkpoint places every key.
    forward(voxel_features [M, C], coors [M, 4] (b,z,y,x), batch_size) -> dense [B, 128 * D', H', W'] BEV map."""

    def __init__(self, in_channels=128, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1)):
        super().__init__()
        self.conv_input = nn.Sequential(SubmConv3D(in_channels, 16, 3, bias=False, key="res0"), _bn(16), nn.ReLU())
        self.conv1 = nn.Sequential(SparseBasicBlock(16, 16, "res0"), SparseBasicBlock(16, 16, "res0"))
        self.conv2 = nn.Sequential(Conv3D(16, 32, 3, 2, padding=1, bias=False), _bn(32), nn.ReLU(),
                                   SparseBasicBlock(32, 32, "res1"), SparseBasicBlock(32, 32, "res1"))
        self.conv3 = nn.Sequential(Conv3D(32, 64, 3, 2, padding=1, bias=False), _bn(64), nn.ReLU(),
                                   SparseBasicBlock(64, 64, "res2"), SparseBasicBlock(64, 64, "res2"))
        self.conv4 = nn.Sequential(Conv3D(64, 128, 3, 2, padding=(0, 1, 1), bias=False), _bn(128), nn.ReLU(),
                                   SparseBasicBlock(128, 128, "res3"), SparseBasicBlock(128, 128, "res3"))
        self.extra_conv = nn.Sequential(Conv3D(128, 128, (3, 1, 1), (2, 1, 1), bias=False), _bn(128), nn.ReLU())
        self.extra_conv[0].out_f32 = True  # (mixed precision: the densified map is fp32 like the fp32 path's)
        self.sparse_shape = _sparse_shape(voxel_size, point_cloud_range)
        self.in_channels = in_channels

    accepts_padding_rows = True  # rows with batch index < 0 are ignored (no boolean-mask sync in the caller)
    # Mixed precision (CenterPoint.set_amp; the reference's amp_cfg level O2): the convolutions from 16 -> 32 on run on the
    # fp16 matrix cores with fp16 feature rows between them (fp32 accumulation, fp32 BatchNorm fold); the 5 -> 16 and
    # 16 -> 16 layers and the densified map stay fp32.  Never the default.
    amp = False

    # remember_capacities: the first forward of a (batch size, input rows) shape plans with the ONE host sync and
    # remembers every index set's row count (x 1.25); later forwards of that shape plan without any host round trip
    # (sparse_conv3d.plan(caps=...)).  A denser later scene can make a set reach its capacity: the map is then
    # TRUNCATED, and only `take_overflow()` says so.  The reference never truncates (its layers read nnz on the host
    # after every sparse op), so the unsynced plan is OPT-IN:
    #   None (default)  used only inside `with encoder.overflow_checked():`, i.e. by callers that read take_overflow()
    #                   where they synchronise anyway and recompute (CenterPoint.test_forward does);
    #   True            always -- the caller promises to read take_overflow() itself (bench.py, tools/prof);
    #   False           never.
    # Direct calls (`encoder(...)`, `CenterPoint.extract_pillars`, `test_forward(device_only=True)`) therefore plan
    # with the sync and return exact maps unless the flag was set.
    remember_capacities = None
    _overflow_checked = False

    def overflow_checked(self):
        """Context: forwards inside it may plan from remembered capacities; the caller MUST read `take_overflow()`
        after it has synchronised and recompute the frame if it returns True."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev = self._overflow_checked
            object.__setattr__(self, "_overflow_checked", True)
            try:
                yield self
            finally:
                object.__setattr__(self, "_overflow_checked", prev)

        return ctx()

    def _unsynced(self) -> bool:
        return self.remember_capacities is True or (self.remember_capacities is None and self._overflow_checked)

    def _plan_input(self, voxel_features, coors, batch_size):
        shape_key = (int(batch_size), int(coors.shape[0]))
        remember = self._unsynced()
        caps = getattr(self, "_caps", {}).get(shape_key) if remember else None
        x, pl = _planned_input(self, voxel_features, coors, batch_size, caps)
        if caps is None:
            if remember:
                if not hasattr(self, "_caps"):
                    object.__setattr__(self, "_caps", {})
                self._caps[shape_key] = _sp.plan_caps(pl)
        else:
            prev = getattr(self, "_overflow", None)
            object.__setattr__(self, "_overflow", pl.overflow if prev is None else (prev | pl.overflow))
        return x

    def take_overflow(self) -> bool:
        """True if an index set of a forward since the last call outgrew its remembered capacity (ONE host sync; the
        capacities are dropped so that the next forward plans with exact sizes).  False when nothing ran unsynced."""
        flag = getattr(self, "_overflow", None)
        object.__setattr__(self, "_overflow", None)
        if flag is None or not bool(flag.item()):
            return False
        object.__setattr__(self, "_caps", {})
        return True

    @torch.no_grad()
    def forward(self, voxel_features, coors, batch_size):
        x = self._plan_input(voxel_features, coors, batch_size)
        for stage in (self.conv_input, self.conv1, self.conv2, self.conv3, self.conv4, self.extra_conv):
            x = _run_sequential(stage, x)
        return _densify(x)


def _sparse_conv_bn_relu(in_channels, out_channels, kernel_size, stride=1, padding=0, conv_type="subm"):
    """sparsenet.py:31-64."""
    if conv_type == "subm":
        conv = SubmConv3D(in_channels, out_channels, kernel_size, bias=False)
    elif conv_type == "spconv":
        conv = Conv3D(in_channels, out_channels, kernel_size, stride, padding, bias=False)
    else:
        raise NotImplementedError(conv_type)
    return nn.Sequential(conv, _bn(out_channels), nn.ReLU())


class SparseNet3D(nn.Module):
    """sparsenet.py:68-182 (the plain conv-bn-relu sparse encoder of the reference's voxel R-CNN family), with the
    reference's parameter names.  forward -> the reference's batch_dict: spatial_features [B, 128 * D', H', W'],
    spatial_features_stride 8, multi_scale_3d_features x_conv1..4 (SparseConvTensor) and their strides."""

    def __init__(self, in_channels=128, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1)):
        super().__init__()
        self.conv_input = nn.Sequential(SubmConv3D(in_channels, 16, 3, bias=False), _bn(16), nn.ReLU())
        self.conv1 = nn.Sequential(_sparse_conv_bn_relu(16, 16, 3, padding=1))
        self.conv2 = nn.Sequential(_sparse_conv_bn_relu(16, 32, 3, stride=2, padding=1, conv_type="spconv"),
                                   _sparse_conv_bn_relu(32, 32, 3, padding=1), _sparse_conv_bn_relu(32, 32, 3, padding=1))
        self.conv3 = nn.Sequential(_sparse_conv_bn_relu(32, 64, 3, stride=2, padding=1, conv_type="spconv"),
                                   _sparse_conv_bn_relu(64, 64, 3, padding=1), _sparse_conv_bn_relu(64, 64, 3, padding=1))
        self.conv4 = nn.Sequential(_sparse_conv_bn_relu(64, 64, 3, stride=2, padding=(0, 1, 1), conv_type="spconv"),
                                   _sparse_conv_bn_relu(64, 64, 3, padding=1), _sparse_conv_bn_relu(64, 64, 3, padding=1))
        self.extra_conv = nn.Sequential(Conv3D(64, 128, (3, 1, 1), (2, 1, 1), padding=0, bias=False), _bn(128), nn.ReLU())
        self.sparse_shape = _sparse_shape(voxel_size, point_cloud_range)
        self.in_channels = in_channels
        self.num_point_features = 128
        self.backbone_channels = {"x_conv1": 16, "x_conv2": 32, "x_conv3": 64, "x_conv4": 64}

    accepts_padding_rows = True

    @torch.no_grad()
    def forward(self, voxel_features, coors, batch_size):
        # (always the synced plan: the intermediate sparse tensors are part of the result, at their exact sizes)
        x, _ = _planned_input(self, voxel_features, coors, batch_size)
        x = _run_sequential(self.conv_input, x)
        x1 = _run_sequential(self.conv1, x)
        x2 = _run_sequential(self.conv2, x1)
        x3 = _run_sequential(self.conv3, x2)
        x4 = _run_sequential(self.conv4, x3)
        out = _densify(_run_sequential(self.extra_conv, x4))
        return {"spatial_features": out, "spatial_features_stride": 8,
                "multi_scale_3d_features": {"x_conv1": x1, "x_conv2": x2, "x_conv3": x3, "x_conv4": x4},
                "multi_scale_3d_strides": {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8}}
