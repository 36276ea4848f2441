"""PointPillarsScatter forward as one op: pointpillars_scatter(voxel_features, coords, batch_size, ny, nx).

Reference layer: paddle3d/models/middle_encoders/pillar_scatter.py:57-93 (zeros canvas, paddle.scatter
with overwrite, transpose, concat).  Returns the [B, C, ny, nx] fp32 pseudo image.
"""
from __future__ import annotations

import torch

from ._common import check, lib, ptr, require_gpu, stream_ptr, workspace

__all__ = ["pointpillars_scatter", "inverse_map", "SparseCanvas"]


def pointpillars_scatter(voxel_features: torch.Tensor, coords: torch.Tensor, batch_size: int, ny: int,
                         nx: int, out: torch.Tensor | None = None) -> torch.Tensor:
    f = require_gpu(voxel_features, "pointpillars_scatter")
    c = require_gpu(coords, "pointpillars_scatter", torch.int32)
    if f.dim() != 2 or c.dim() != 2 or c.shape[1] != 4 or c.shape[0] != f.shape[0]:
        raise RuntimeError("pointpillars_scatter: voxel_features [M, C], coords [M, 4] expected")
    dev = f.device
    m, ch = f.shape
    if out is None:
        out = torch.empty((batch_size, ch, ny, nx), dtype=torch.float32, device=dev)
    L = lib()
    ws = workspace(L.pd3_pointpillars_scatter_workspace(batch_size, ny, nx), dev)
    check(L.pd3_pointpillars_scatter(ptr(f), ptr(c), m, ch, batch_size, ny, nx, ptr(out), ptr(ws),
                                     ws.numel(), stream_ptr(dev)), "pointpillars_scatter")
    return out


def inverse_map(coords: torch.Tensor, batch_size: int, ny: int, nx: int) -> torch.Tensor:
    """[B, ny * nx] int32: the row of `coords` ([M, 4] = batch, z, y, x) that names the cell, -1 for an empty cell;
    on duplicates the highest row wins (paddle.scatter(overwrite=True), pillar_scatter.py:83-90)."""
    c = require_gpu(coords, "pointpillars_scatter", torch.int32)
    if c.dim() != 2 or c.shape[1] != 4:
        raise RuntimeError("pointpillars_scatter: coords [M, 4] expected")
    inv = torch.empty((batch_size, ny * nx), dtype=torch.int32, device=c.device)
    check(lib().pd3_pointpillars_inverse_map(ptr(c), c.shape[0], batch_size, ny, nx, ptr(inv), stream_ptr(c.device)),
          "pointpillars_scatter")
    return inv


class SparseCanvas:
    """A PointPillarsScatter result that has not been written out: the pillar features, the inverse map and the
    canvas shape.  The first backbone convolution consumes it directly (ops.conv.scatter_conv3x3_bias_relu);
    `dense()` materialises the [B, C, ny, nx] pseudo image for any other consumer (same bytes as the op)."""

    def __init__(self, voxel_features, coords, batch_size, ny, nx):
        self.features = require_gpu(voxel_features, "pointpillars_scatter")
        self.coords = require_gpu(coords, "pointpillars_scatter", torch.int32)
        if self.features.dim() != 2 or self.coords.shape[0] != self.features.shape[0]:
            raise RuntimeError("pointpillars_scatter: voxel_features [M, C], coords [M, 4] expected")
        self.batch, self.ny, self.nx = int(batch_size), int(ny), int(nx)
        self.inv = inverse_map(self.coords, self.batch, self.ny, self.nx)
        self.shape = (self.batch, int(self.features.shape[1]), self.ny, self.nx)

    def dense(self):
        return pointpillars_scatter(self.features, self.coords, self.batch, self.ny, self.nx)
