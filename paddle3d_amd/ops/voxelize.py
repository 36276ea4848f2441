"""`paddle3d.ops.voxelize` mirror: hard_voxelize(points, voxel_size, point_cloud_range,
max_num_points_in_voxel, max_voxels) -> (voxels, coords, num_points_per_voxel, num_voxels).

Reference operator: paddle3d/ops/voxel/voxelize_op.cc:149-191 (PD_BUILD_OP(hard_voxelize)); caller
paddle3d/models/voxelizers/voxelize.py:39-58.  Same positional arguments, same four outputs with the
same fixed shapes / dtypes (fp32 [V,P,D], int32 [V,3] (z,y,x), int32 [V], int32 [1]).
"""
from __future__ import annotations


import torch

from ._common import check, host_f32, lib, ptr, require_gpu, stream_ptr, workspace

__all__ = ["dynamic_voxelize", "hard_voxelize", "hard_voxelize_batch"]


def hard_voxelize_batch(points: torch.Tensor, voxel_size, point_cloud_range, max_num_points_in_voxel: int,
                        max_voxels: int, num_points: torch.Tensor | None = None, with_batch_coors: bool = False,
                        path: int = 0):
    """points [B, N, D] fp32 on the GPU; num_points optional int32 [B] (valid rows per frame).

    Returns voxels [B,V,P,D], coords [B,V,3], num_points_per_voxel [B,V], num_voxels [B] -- the
    reference's per-sample op results stacked (HardVoxelizer's python loop, voxelize.py:60-82).
    with_batch_coors=True additionally returns coors [B,V,4] = (batch, z, y, x), batch -1 on padding rows.
    path: 0 automatic, 1 generic sort path, 2 tiled path (compact payload array), 3 tiled path (gathered rows),
    5 wave form of the tiled path (6 .. 10: route tile shape forced) (pd3_hard_voxelize_path; the tests run them).
    """
    pts = require_gpu(points, "hard_voxelize")
    if pts.dim() != 3:
        raise RuntimeError("hard_voxelize_batch expects points of shape [B, N, D]")
    b, n, d = pts.shape
    dev = pts.device
    vs, pr = host_f32(voxel_size, 3), host_f32(point_cloud_range, 6)
    p, v = int(max_num_points_in_voxel), int(max_voxels)
    voxels = torch.empty((b, v, p, d), dtype=torch.float32, device=dev)
    coords = torch.empty((b, v, 3), dtype=torch.int32, device=dev)
    npv = torch.empty((b, v), dtype=torch.int32, device=dev)
    nv = torch.empty((b,), dtype=torch.int32, device=dev)
    coors4 = torch.empty((b, v, 4), dtype=torch.int32, device=dev) if with_batch_coors else None
    if num_points is not None:
        num_points = require_gpu(num_points, "hard_voxelize", torch.int32)
    L = lib()
    ws_bytes = L.pd3_hard_voxelize_workspace(b, n, d, ptr(vs), ptr(pr), p, v)
    if ws_bytes == 0:
        raise RuntimeError("hard_voxelize: invalid voxel_size / point_cloud_range / sizes")
    ws = workspace(ws_bytes, dev)
    check(L.pd3_hard_voxelize_path(ptr(pts), ptr(num_points), b, n, d, ptr(vs), ptr(pr), p, v, ptr(voxels),
                                   ptr(coords), ptr(npv), ptr(nv), ptr(coors4), ptr(ws), ws.numel(),
                                   stream_ptr(dev), int(path)),
          "hard_voxelize")
    if with_batch_coors:
        return voxels, coords, npv, nv, coors4
    return voxels, coords, npv, nv


def hard_voxelize(points: torch.Tensor, voxel_size, point_cloud_range, max_num_points_in_voxel: int,
                  max_voxels: int, path: int = 0):
    pts = require_gpu(points, "hard_voxelize")
    if pts.dim() != 2:
        raise RuntimeError("hard_voxelize expects points of shape [N, D]")
    voxels, coords, npv, nv = hard_voxelize_batch(pts.unsqueeze(0), voxel_size, point_cloud_range,
                                                  max_num_points_in_voxel, max_voxels, path=path)
    return voxels[0], coords[0], npv[0], nv


def dynamic_voxelize(points: torch.Tensor, voxel_size, point_cloud_range) -> torch.Tensor:
    """Per-point voxel coordinates [N, 3] int32 = (z, y, x), -1 outside the range; hard_voxelize's cell rule."""
    pts = require_gpu(points, "dynamic_voxelize")
    if pts.dim() != 2:
        raise RuntimeError("dynamic_voxelize expects points of shape [N, D]")
    n, d = pts.shape
    vs, pr = host_f32(voxel_size, 3), host_f32(point_cloud_range, 6)
    coors = torch.empty((n, 3), dtype=torch.int32, device=pts.device)
    check(lib().pd3_dynamic_voxelize(ptr(pts), n, d, ptr(vs), ptr(pr), ptr(coors), stream_ptr(pts.device)),
          "dynamic_voxelize")
    return coors
