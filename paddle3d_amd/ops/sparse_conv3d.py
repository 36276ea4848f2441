"""`sparse_conv3d`: the ops behind paddle.sparse.nn.SubmConv3D / Conv3D as the reference's CenterPoint-Voxel
middle encoder uses them (paddle3d/models/middle_encoders/sparse_resnet.py:31-59, :115-206).  The reference
has no `paddle3d.ops.sparse_conv3d` module -- the arithmetic is Paddle core -- so the signatures here are ours:

  indices(coords, batch, spatial_shape, kernel_size, stride, padding, subm) -> SparseIndices
  features(in_feats, idx, weight, bias=None, scale=None, shift=None, residual=None, relu=False) -> out_feats
  to_dense(feats, coords, batch, spatial_shape) -> [B, C*D, H, W]
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

from ._common import check, host_i32, lib, ptr, require_gpu, stream_ptr, workspace

__all__ = ["SparseIndices", "indices", "features", "to_dense", "out_spatial_shape"]


@dataclass
class SparseIndices:
    out_coords: torch.Tensor   # [n_out, 4] int32 (b, z, y, x)
    nbr: torch.Tensor          # [n_out, K] int32, input row or -1
    n_out: int
    out_shape: tuple           # (D, H, W)
    kernel_volume: int


def out_spatial_shape(spatial_shape, kernel_size, stride, padding):
    return tuple((s + 2 * p - k) // st + 1 for s, k, st, p in zip(spatial_shape, kernel_size, stride, padding))


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(int(x) for x in v)


def indices(coords: torch.Tensor, batch: int, spatial_shape, kernel_size, stride=1, padding=0,
            subm: bool = False) -> SparseIndices:
    c = require_gpu(coords, "sparse_conv3d", torch.int32)
    if c.dim() != 2 or c.shape[1] != 4:
        raise RuntimeError("sparse_conv3d: coords must be [N, 4] int32 (batch, z, y, x)")
    ks, st, pd = _triple(kernel_size), _triple(stride), _triple(padding)
    n_in = c.shape[0]
    kvol = ks[0] * ks[1] * ks[2]
    dev = c.device
    if n_in == 0:  # an empty voxel set gives an empty sparse tensor (the reference's layers accept nnz == 0)
        shape = tuple(spatial_shape) if subm else out_spatial_shape(spatial_shape, ks, st, pd)
        return SparseIndices(torch.empty((0, 4), dtype=torch.int32, device=dev),
                             torch.empty((0, kvol), dtype=torch.int32, device=dev), 0, shape, kvol)
    if subm:
        out_shape = tuple(spatial_shape)
        cap = n_in
    else:
        out_shape = out_spatial_shape(spatial_shape, ks, st, pd)
        per_in = math.prod(-(-k // s) for k, s in zip(ks, st))
        cap = int(min(n_in * per_in, batch * math.prod(out_shape)))
    out_coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    nbr = torch.empty((cap, kvol), dtype=torch.int32, device=dev)
    n_out = torch.empty((1,), dtype=torch.int32, device=dev)
    L = lib()
    hk, hs, hp, hsh = host_i32(ks), host_i32(st), host_i32(pd), host_i32(spatial_shape)
    ws_bytes = L.pd3_sparse_conv3d_workspace(n_in, ptr(hk), int(subm), cap)
    if ws_bytes == 0:
        raise RuntimeError("sparse_conv3d: invalid sizes")
    ws = workspace(ws_bytes, dev)
    check(L.pd3_sparse_conv3d_indices(ptr(c), n_in, batch, ptr(hsh), ptr(hk), ptr(hs), ptr(hp), int(subm),
                                      ptr(out_coords), ptr(nbr), ptr(n_out), cap, ptr(ws), ws.numel(),
                                      stream_ptr(dev)), "sparse_conv3d_indices")
    n = n_in if subm else int(n_out.item())  # one host sync per strided convolution (v1)
    return SparseIndices(out_coords[:n], nbr[:n], n, out_shape, kvol)


def features(in_feats: torch.Tensor, idx: SparseIndices, weight: torch.Tensor, bias=None, scale=None,
             shift=None, residual=None, relu: bool = False) -> torch.Tensor:
    """weight [kd, kh, kw, Cin, Cout] (Paddle layout)."""
    f = require_gpu(in_feats, "sparse_conv3d")
    w = require_gpu(weight, "sparse_conv3d")
    cin, cout = int(w.shape[-2]), int(w.shape[-1])
    if f.shape[1] != cin or w.numel() != idx.kernel_volume * cin * cout:
        raise RuntimeError("sparse_conv3d: weight / feature shapes do not match")
    out = torch.empty((idx.n_out, cout), dtype=torch.float32, device=f.device)
    if idx.n_out == 0:
        return out
    opt = [None if t is None else require_gpu(t, "sparse_conv3d") for t in (bias, scale, shift, residual)]
    check(lib().pd3_sparse_conv3d_features(ptr(f), ptr(idx.nbr), None, idx.n_out, idx.kernel_volume, cin, cout,
                                           ptr(w), ptr(opt[0]), ptr(opt[1]), ptr(opt[2]), ptr(opt[3]),
                                           int(bool(relu)), ptr(out), stream_ptr(f.device)),
          "sparse_conv3d_features")
    return out


def to_dense(feats: torch.Tensor, coords: torch.Tensor, batch: int, spatial_shape) -> torch.Tensor:
    f = require_gpu(feats, "sparse_to_dense")
    c = require_gpu(coords, "sparse_to_dense", torch.int32)
    d, h, w = (int(x) for x in spatial_shape)
    ch = f.shape[1]
    out = torch.empty((batch, ch * d, h, w), dtype=torch.float32, device=f.device)
    n = f.shape[0]
    if n == 0:
        return out.zero_()
    check(lib().pd3_sparse_to_dense(ptr(f), ptr(c), None, n, ch, batch, ptr(host_i32(spatial_shape)), ptr(out),
                                    stream_ptr(f.device)), "sparse_to_dense")
    return out
