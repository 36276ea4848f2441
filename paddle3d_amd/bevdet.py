"""Geometry + pooling half of BEVDet's view transformer over the HIP ops -- the host-side mirror of
LSSViewTransformer (reference paddle3d/models/transformers/bevdet_transformer.py:90-274) without its depth
network (the camera branch's convolutions are outside the LiDAR-detection hot path): frustum template, frustum
points in the ego frame, the bev_pool_v2 index build and the pooling itself.

BEVDet4D-R50 (configs/bevdet/bevdet4d_r50_depth_nuscenes.yml:174-186): grid x, y in [-51.2, 51.2] at 0.8 m,
z in [-5, 3] at 8 m, depth 1 .. 60 m at 0.5 m (118 bins), 256 x 704 input, downsample 16 -> 6 x 118 x 16 x 44
= 498 432 frustum points per frame."""
from __future__ import annotations

import numpy as np
import torch

from .ops import bev_pool_v2 as _bp
from .ops._common import check, lib, ptr, require_gpu, stream_ptr

__all__ = ["LSSViewTransformer", "BEVDET4D_GRID"]

BEVDET4D_GRID = dict(x=[-51.2, 51.2, 0.8], y=[-51.2, 51.2, 0.8], z=[-5, 3, 8], depth=[1.0, 60.0, 0.5])


class LSSViewTransformer:
    def __init__(self, grid_config=None, input_size=(256, 704), downsample=16, accelerate=False):
        grid_config = dict(BEVDET4D_GRID if grid_config is None else grid_config)
        self.grid_config, self.downsample, self.accelerate = grid_config, downsample, accelerate
        self.create_grid_infos(**grid_config)
        self.create_frustum(grid_config["depth"], input_size, downsample)
        self._prepared = None

    def create_grid_infos(self, x, y, z, **_kw):
        """:120-124: lower bound, interval and size of the grid, as float32."""
        self.grid_lower_bound = np.array([c[0] for c in (x, y, z)], np.float32)
        self.grid_interval = np.array([c[2] for c in (x, y, z)], np.float32)
        self.grid_size = np.array([(c[1] - c[0]) / c[2] for c in (x, y, z)], np.float32)

    def create_frustum(self, depth_cfg, input_size, downsample):
        """:126-140: the (u, v, depth) template [D, H, W, 3] shared by all cameras."""
        h_in, w_in = input_size
        hf, wf = h_in // downsample, w_in // downsample
        d = np.arange(*depth_cfg, dtype=np.float32).reshape(-1, 1, 1)
        self.D = int(d.shape[0])
        x = np.linspace(0, w_in - 1, wf, dtype=np.float32).reshape(1, 1, wf)
        y = np.linspace(0, h_in - 1, hf, dtype=np.float32).reshape(1, hf, 1)
        shape = (self.D, hf, wf)
        self.frustum = np.stack([np.broadcast_to(x, shape), np.broadcast_to(y, shape), np.broadcast_to(d, shape)],
                                -1).astype(np.float32)
        self._frustum_dev = None

    def get_lidar_coor(self, rots, trans, cam2imgs, post_rots, post_trans, bda):
        """:142-192.  rots / cam2imgs / post_rots [B, N, 3, 3], trans / post_trans [B, N, 3], bda [B, 3, 3] (GPU fp32)
        -> frustum points in the ego frame [B, N, D, H, W, 3].  The per-camera 3x3 inverses and products are a few
        dozen floats (torch); the per-point transform is one HIP kernel."""
        op = "frustum_to_lidar"
        rots = require_gpu(rots, op)
        B, N = int(rots.shape[0]), int(rots.shape[1])
        dev = rots.device
        if self._frustum_dev is None or self._frustum_dev.device != dev:
            self._frustum_dev = torch.from_numpy(np.ascontiguousarray(self.frustum)).to(dev)
        D, H, W, _ = self.frustum.shape
        inv_post = torch.linalg.inv(require_gpu(post_rots, op)).reshape(B * N, 3, 3).contiguous()
        cam2ego = torch.matmul(rots, torch.linalg.inv(require_gpu(cam2imgs, op).float())).reshape(B * N, 3, 3).contiguous()
        out = torch.empty((B, N, D, H, W, 3), dtype=torch.float32, device=dev)
        check(lib().pd3_frustum_to_lidar(ptr(self._frustum_dev), D * H * W, B, N, ptr(inv_post),
                                         ptr(require_gpu(post_trans, op).reshape(B * N, 3)), ptr(cam2ego),
                                         ptr(require_gpu(trans, op).reshape(B * N, 3)), ptr(require_gpu(bda, op)),
                                         ptr(out), stream_ptr(dev)), op)
        return out

    def voxel_pooling_prepare_v2(self, coor):
        """:230-274 on the device."""
        return _bp.voxel_pooling_prepare_v2(coor, self.grid_lower_bound, self.grid_interval, self.grid_size)

    def init_acceleration_v2(self, coor):
        """:194-207: the index sets are computed once and reused (accelerate=True, fixed calibration)."""
        self._prepared = self.voxel_pooling_prepare_v2(coor)

    def voxel_pooling_v2(self, coor, depth, feat):
        """:209-228.  depth [B*N, D, H, W], feat [B*N, C, H, W] -> BEV features [B, C, Y, X]."""
        B = int(coor.shape[0])
        prep = self._prepared if (self.accelerate and self._prepared is not None) else self.voxel_pooling_prepare_v2(coor)
        gx, gy = int(self.grid_size[0]), int(self.grid_size[1])
        if prep[0] is None:
            return torch.zeros((B, feat.shape[1], gx, gy), dtype=feat.dtype, device=feat.device)
        ranks_bev, ranks_depth, ranks_feat, starts, lengths = prep
        f = feat.permute(0, 2, 3, 1).contiguous()
        out = _bp.bev_pool_v2(depth.contiguous(), f, ranks_depth, ranks_feat, ranks_bev, lengths, starts,
                              (B, gy, gx, f.shape[-1]))
        return out.permute(0, 3, 1, 2)
