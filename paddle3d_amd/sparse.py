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

__all__ = ["SparseConvTensor", "SubmConv3D", "Conv3D", "SparseBasicBlock", "SparseResNet3D"]


class SparseConvTensor:
    """features [N, C] fp32 + indices [N, 4] int32 (b, z, y, x) + dense spatial shape (D, H, W)."""

    def __init__(self, features, indices, spatial_shape, batch_size, cache=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = tuple(int(s) for s in spatial_shape)
        self.batch_size = int(batch_size)
        self.cache = {} if cache is None else cache  # indice_key -> SparseIndices

    def replace(self, features):
        return SparseConvTensor(features, self.indices, self.spatial_shape, self.batch_size, self.cache)

    def dense(self):
        return _sp.to_dense(self.features, self.indices, self.batch_size, self.spatial_shape)


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(int(x) for x in v)


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

    def _indices(self, x: SparseConvTensor):
        if self.subm and self.key is not None and self.key in x.cache:
            return x.cache[self.key]
        pad = tuple(k // 2 for k in self.ks) if self.subm else self.padding
        idx = _sp.indices(x.indices, x.batch_size, x.spatial_shape, self.ks, self.stride, pad, self.subm)
        if self.subm and self.key is not None:
            x.cache[self.key] = idx
        return idx

    def forward(self, x: SparseConvTensor, scale=None, shift=None, residual=None, relu=False):
        idx = self._indices(x)
        out = _sp.features(x.features, idx, self.weight, self.bias, scale, shift, residual, relu)
        if self.subm:
            return x.replace(out)
        return SparseConvTensor(out, idx.out_coords, idx.out_shape, x.batch_size)


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
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return scale.detach().contiguous(), (bn.bias - bn.running_mean * scale).detach().contiguous()


class _ConvBnRelu(nn.Module):
    def __init__(self, conv):
        super().__init__()
        self.conv = conv
        self.bn = nn.BatchNorm1d(conv.weight.shape[-1], eps=1e-3, momentum=0.01)

    def forward(self, x, residual=None):
        s, b = _fold(self.bn)
        return self.conv(x, s, b, residual, relu=True)


class SparseBasicBlock(nn.Module):
    """sparse_resnet.py:63-111: conv-bn-relu, conv-bn, + identity, relu (all submanifold, shared rulebook)."""

    def __init__(self, in_channels, out_channels, indice_key=None):
        super().__init__()
        self.conv1 = SubmConv3D(in_channels, out_channels, 3, bias=True, key=indice_key)
        self.bn1 = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        self.conv2 = SubmConv3D(out_channels, out_channels, 3, bias=True, key=indice_key)
        self.bn2 = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)

    def forward(self, x):
        s1, b1 = _fold(self.bn1)
        out = self.conv1(x, s1, b1, None, relu=True)
        s2, b2 = _fold(self.bn2)
        return self.conv2(out, s2, b2, x.features, relu=True)  # relu(bn2(conv2) + identity), :104-109


class SparseResNet3D(nn.Module):
    """sparse_resnet.py:115-206.  forward(voxel_features [M, C], coors [M, 4] (b,z,y,x), batch_size) ->
    dense [B, 128 * D', H', W'] BEV map."""

    def __init__(self, in_channels=128, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1)):
        super().__init__()
        self.conv_input = _ConvBnRelu(SubmConv3D(in_channels, 16, 3, bias=False, key="res0"))
        self.conv1 = nn.Sequential(SparseBasicBlock(16, 16, "res0"), SparseBasicBlock(16, 16, "res0"))
        self.conv2_down = _ConvBnRelu(Conv3D(16, 32, 3, 2, padding=1, bias=False))
        self.conv2 = nn.Sequential(SparseBasicBlock(32, 32, "res1"), SparseBasicBlock(32, 32, "res1"))
        self.conv3_down = _ConvBnRelu(Conv3D(32, 64, 3, 2, padding=1, bias=False))
        self.conv3 = nn.Sequential(SparseBasicBlock(64, 64, "res2"), SparseBasicBlock(64, 64, "res2"))
        self.conv4_down = _ConvBnRelu(Conv3D(64, 128, 3, 2, padding=(0, 1, 1), bias=False))
        self.conv4 = nn.Sequential(SparseBasicBlock(128, 128, "res3"), SparseBasicBlock(128, 128, "res3"))
        self.extra_conv = _ConvBnRelu(Conv3D(128, 128, (3, 1, 1), (2, 1, 1), bias=False))
        pcr = np.array(point_cloud_range, dtype=np.float32)
        vs = np.array(voxel_size, dtype=np.float32)
        grid = np.round((pcr[3:] - pcr[:3]) / vs).astype(np.int64)
        self.sparse_shape = tuple(int(v) for v in (np.array(grid[::-1]) + [1, 0, 0]))  # :173
        self.in_channels = in_channels

    @torch.no_grad()
    def forward(self, voxel_features, coors, batch_size):
        # rows in raster order: tiles of consecutive rows are then spatial neighbours, so a tile's kernel
        # offsets are mostly all-present or all-absent (skipped) and its gathers share cache lines.  The
        # final dense map does not depend on the row order.
        d, h, w = self.sparse_shape
        c64 = coors.long()
        order = torch.argsort(((c64[:, 0] * d + c64[:, 1]) * h + c64[:, 2]) * w + c64[:, 3])
        x = SparseConvTensor(voxel_features[order].contiguous(), coors[order].contiguous(), self.sparse_shape,
                             batch_size)
        x = self.conv_input(x)
        x = self.conv1(x)
        x = self.conv2(self.conv2_down(x))
        x = self.conv3(self.conv3_down(x))
        x = self.conv4(self.conv4_down(x))
        x = self.extra_conv(x)
        return x.dense()
