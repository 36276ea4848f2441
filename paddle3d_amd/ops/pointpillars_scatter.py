"""PointPillarsScatter forward as one op: pointpillars_scatter(voxel_features, coords, batch_size, ny, nx).

Reference layer: paddle3d/models/middle_encoders/pillar_scatter.py:57-93 (zeros canvas, paddle.scatter
with overwrite, transpose, concat).  Returns the [B, C, ny, nx] fp32 pseudo image.
"""
from __future__ import annotations

import torch

from ._common import check, lib, ptr, require_gpu, stream_ptr, workspace

__all__ = ["pointpillars_scatter"]


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
