"""Fused voxel encoders: pillar_feature_net (PillarFeatureNet eval forward) and voxel_mean (VoxelMean).

Reference layers: paddle3d/models/voxel_encoders/pillar_encoder.py:156-210 / :81-105 and
voxel_encoder.py:44-57.  BatchNorm1D is folded on the host into (scale, shift).
"""
from __future__ import annotations

import ctypes as C

import torch

from ._common import check, lib, ptr, require_gpu, stream_ptr

__all__ = ["pillar_feature_net", "pillar_feature_net_indexed", "hard_vfe", "voxel_mean", "fold_batchnorm"]


def fold_batchnorm(gamma, beta, mean, var, eps):
    scale = gamma / torch.sqrt(var + eps)
    return scale.contiguous(), (beta - mean * scale).contiguous()


def pillar_feature_net(voxels, num_points, coors, vx, vy, x_offset, y_offset, w1, scale1, shift1,
                       w2=None, scale2=None, shift2=None, vz=0.0, z_offset=0.0, voxel_center_dims=2, path=0):
    """voxels [M,P,D], num_points [M] i32, coors [M,4] i32 (b,z,y,x); w* in Paddle Linear layout [in,out].
    voxel_center_dims=3 (+ vz, z_offset) is the HardVFE decoration (voxel_encoder.py:252-266).
    path: 0 = the library's choice, 1 = per-pillar kernels, 2 = packed kernel (tests run all of them)."""
    v = require_gpu(voxels, "pillar_feature_net")
    n = require_gpu(num_points, "pillar_feature_net", torch.int32)
    c = require_gpu(coors, "pillar_feature_net", torch.int32)
    m, p, d = v.shape
    w1 = require_gpu(w1, "pillar_feature_net")
    c1 = w1.shape[1]
    if w1.shape[0] != d + 3 + voxel_center_dims:
        raise RuntimeError(f"pillar_feature_net: w1 must be [{d + 3 + voxel_center_dims}, C1]")
    two = w2 is not None
    if two:
        w2 = require_gpu(w2, "pillar_feature_net")
        if w2.shape[0] != 2 * c1:
            raise RuntimeError("pillar_feature_net: w2 must be [2*C1, C2]")
        c2 = w2.shape[1]
    else:
        c2 = 0
    out = torch.empty((m, c2 if two else c1), dtype=torch.float32, device=v.device)
    check(lib().pd3_pillar_feature_net_path(ptr(v), ptr(n), ptr(c), m, p, d, int(voxel_center_dims), C.c_float(vx),
                                            C.c_float(vy), C.c_float(vz), C.c_float(x_offset),
                                            C.c_float(y_offset), C.c_float(z_offset), ptr(w1),
                                            ptr(scale1.contiguous()), ptr(shift1.contiguous()), c1,
                                            ptr(w2), ptr(scale2.contiguous() if two else None),
                                            ptr(shift2.contiguous() if two else None), c2, ptr(out), int(path),
                                            stream_ptr(v.device)), "pillar_feature_net")
    return out


def pillar_feature_net_indexed(points, vox_span, point_list, coors, max_points, vx, vy, x_offset, y_offset, w1, scale1,
                               shift1, w2, scale2, shift2, vz=0.0, z_offset=0.0, voxel_center_dims=2):
    """pillar_feature_net on the voxelizer's index instead of a padded [M, P, D] tensor (pd3_pillar_feature_net_indexed):
    points [B, N, D], vox_span [B, V, 2] / point_list from ops.voxelize.hard_voxelize_index_batch, coors [B * V, 4].
    Returns [B * V, C2] -- the same bytes as hard_voxelize_batch + pillar_feature_net -- or None for shapes the
    indexed kernel does not serve (the caller then runs the pair)."""
    pts = require_gpu(points, "pillar_feature_net_indexed")
    sp = require_gpu(vox_span, "pillar_feature_net_indexed", torch.int32)
    pl = require_gpu(point_list, "pillar_feature_net_indexed", torch.int32)
    c = require_gpu(coors, "pillar_feature_net_indexed", torch.int32)
    b, n, d = pts.shape
    v = sp.shape[1]
    w1 = require_gpu(w1, "pillar_feature_net_indexed")
    w2 = require_gpu(w2, "pillar_feature_net_indexed")
    c1, c2 = w1.shape[1], w2.shape[1]
    if w1.shape[0] != d + 3 + voxel_center_dims or w2.shape[0] != 2 * c1:
        raise RuntimeError("pillar_feature_net_indexed: weight shapes do not match the point width")
    out = torch.empty((b * v, c2), dtype=torch.float32, device=pts.device)
    rc = lib().pd3_pillar_feature_net_indexed(
        ptr(pts), n, ptr(sp), ptr(pl), n, ptr(c), b * v, v, int(max_points), d, int(voxel_center_dims), C.c_float(vx),
        C.c_float(vy), C.c_float(vz), C.c_float(x_offset), C.c_float(y_offset), C.c_float(z_offset), ptr(w1),
        ptr(scale1.contiguous()), ptr(shift1.contiguous()), c1, ptr(w2), ptr(scale2.contiguous()),
        ptr(shift2.contiguous()), c2, ptr(out), stream_ptr(pts.device))
    if rc == -3:
        return None
    check(rc, "pillar_feature_net_indexed")
    return out


def hard_vfe(voxels, num_points, coors, voxel_size, point_cloud_range, w1, scale1, shift1, w2, scale2, shift2, path=0):
    """HardVFE.forward (eval, with_cluster_center + with_voxel_center, two VFE layers).  path as in
    pillar_feature_net (0 = the library's choice: the packed form for the 64 / 64 net of the BEVFusion config)."""
    vx, vy, vz = (float(v) for v in voxel_size)
    return pillar_feature_net(voxels, num_points, coors, vx, vy, vx / 2 + point_cloud_range[0],
                              vy / 2 + point_cloud_range[1], w1, scale1, shift1, w2, scale2, shift2, vz=vz,
                              z_offset=vz / 2 + point_cloud_range[2], voxel_center_dims=3, path=path)


def voxel_mean(voxels, num_points):
    v = require_gpu(voxels, "voxel_mean")
    n = require_gpu(num_points, "voxel_mean", torch.int32)
    m, p, d = v.shape
    out = torch.empty((m, d), dtype=torch.float32, device=v.device)
    check(lib().pd3_voxel_mean(ptr(v), ptr(n), m, p, d, ptr(out), stream_ptr(v.device)), "voxel_mean")
    return out
