"""Geometry + pooling half of BEVFusion's camera stream over the HIP ops -- the host-side mirror of
`LiftSplatShoot` (reference paddle3d/models/detection/bevfusion/cam_stream_lss.py:170-420) without its depth network
and BEV encoder (the camera branch's convolutions are outside the LiDAR-detection hot path): frustum template
(`create_frustum`, :262-277), frustum points in the LiDAR frame (`get_geometry`, :279-304), the camera -> BEV pooling
(`voxel_pooling`, :318-373) and the channel fold in front of the BEV encoder (`s2c`, :386-390).

Config 5 (configs/bevfusion/bevf_pp_2x8_1x_nusc.yaml:80-84): 6 views of 900 x 1600, downsample 8 -> 112 x 200 feature
pixels, depth 4 .. 45 m at 1 m (41 bins), camC = 64, grid 0.5 m on [-50, 50] x [-50, 50] x [-5, 3] -> nx = (200, 200,
16): 5 510 400 frustum points per scene; the reference's lifted tensor x = depth (x) feat is 1.41 GB per scene."""
from __future__ import annotations

import numpy as np
import torch

from .ops import bev_pool_v2 as _bp
from .ops._common import check, lib, ptr, require_gpu, stream_ptr

__all__ = ["LiftSplatShoot", "gen_dx_bx"]


def gen_dx_bx(xbound, ybound, zbound):
    """cam_stream_lss.py:101-108: cell size, centre of the first cell, cells per axis."""
    rows = (xbound, ybound, zbound)
    dx = np.array([r[2] for r in rows], np.float32)
    bx = np.array([r[0] + r[2] / 2.0 for r in rows], np.float32)
    nx = [int((r[1] - r[0]) / r[2]) for r in rows]
    return dx, bx, nx


class LiftSplatShoot:
    def __init__(self, final_dim=(900, 1600), camera_depth_range=(4.0, 45.0, 1.0),
                 pc_range=(-50, -50, -5, 50, 50, 3), downsample=8, grid=0.5, camC=64):
        self.pc_range, self.final_dim, self.grid, self.downsample, self.camC = pc_range, final_dim, grid, downsample, camC
        self.grid_conf = dict(xbound=[pc_range[0], pc_range[3], grid], ybound=[pc_range[1], pc_range[4], grid],
                              zbound=[pc_range[2], pc_range[5], grid], dbound=list(camera_depth_range))
        self.dx, self.bx, self.nx = gen_dx_bx(self.grid_conf["xbound"], self.grid_conf["ybound"], self.grid_conf["zbound"])
        self.fH, self.fW = final_dim[0] // downsample, final_dim[1] // downsample
        self.frustum = self.create_frustum()
        self.D = int(self.frustum.shape[0])
        self._frustum_dev = None
        self._prepared = None

    def create_frustum(self):
        """:262-277: (u, v, depth) [D, fH, fW, 3] in pixels of the full-size image."""
        ogfH, ogfW = self.final_dim
        ds = np.arange(*self.grid_conf["dbound"], dtype=np.float32).reshape(-1, 1, 1)
        D = ds.shape[0]
        xs = np.linspace(0, ogfW - 1, self.fW, dtype=np.float32).reshape(1, 1, self.fW)
        ys = np.linspace(0, ogfH - 1, self.fH, dtype=np.float32).reshape(1, self.fH, 1)
        sh = (D, self.fH, self.fW)
        return np.stack([np.broadcast_to(xs, sh), np.broadcast_to(ys, sh), np.broadcast_to(ds, sh)], -1).astype(np.float32)

    def get_geometry(self, rots, trans):
        """:279-304.  rots [B, N, 3, 3] = inverse(lidar2img)[:3, :3], trans [B, N, 3] = inverse(lidar2img)[:3, 3] (GPU
        fp32) -> [B, N, D, fH, fW, 3]: rots @ (u * d, v * d, d) + trans, one HIP kernel (pd3_frustum_to_lidar with
        identity image / BEV augmentation)."""
        op = "lss_get_geometry"
        rots, trans = require_gpu(rots, op), require_gpu(trans, op)
        B, N = int(rots.shape[0]), int(rots.shape[1])
        dev = rots.device
        if self._frustum_dev is None or self._frustum_dev.device != dev:
            self._frustum_dev = torch.from_numpy(np.ascontiguousarray(self.frustum)).to(dev)
        eye = torch.eye(3, dtype=torch.float32, device=dev)
        ident = eye.expand(B * N, 3, 3).contiguous()
        zero = torch.zeros(B * N, 3, dtype=torch.float32, device=dev)
        out = torch.empty((B, N, self.D, self.fH, self.fW, 3), dtype=torch.float32, device=dev)
        check(lib().pd3_frustum_to_lidar(ptr(self._frustum_dev), self.D * self.fH * self.fW, B, N, ptr(ident), ptr(zero),
                                         ptr(rots.reshape(B * N, 3, 3)), ptr(trans.reshape(B * N, 3)),
                                         ptr(eye.expand(B, 3, 3).contiguous()), ptr(out), stream_ptr(dev)), op)
        return out

    def voxel_pooling(self, geom_feats, x):
        """:318-373 on the lifted tensor x [B, N, D, H, W, C] -> [B, C, Z, X, Y]."""
        return _bp.lss_voxel_pooling(geom_feats, x, self.dx, self.bx, self.nx)

    def init_acceleration(self, geom_feats):
        """Fixed calibration: build the index sets once (what BEVDet calls `accelerate`, bevdet_transformer.py:194-207)."""
        self._prepared = _bp.lss_pooling_prepare(geom_feats, self.dx, self.bx, self.nx)

    def voxel_pooling_fused(self, geom_feats, depth, feat):
        """The same map from the two factors depth [B*N, D, H, W] and feat [B*N, H, W, C]: x is never formed."""
        return _bp.lss_voxel_pooling_fused(geom_feats, depth, feat, self.dx, self.bx, self.nx, prepared=self._prepared)

    @staticmethod
    def s2c(x):
        """:386-390: [B, C, Z, X, Y] -> [B, C * Z, Y, X]."""
        B, C, H, W, L = x.shape
        return x.reshape(B, C * H, W, L).permute(0, 1, 3, 2)
