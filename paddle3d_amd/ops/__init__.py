"""Op modules with the names and call signatures of the reference's `paddle3d.ops` plugin API
(reference registry: paddle3d/ops/__init__.py:27-104).  Everything here binds the C ABI of
libpaddle3d_amd.so (include/paddle3d_amd.h) through ctypes; nothing has a CPU / PyTorch fallback."""
from . import (bev_pool_v2, centerpoint_postprocess, conv, iou3d_nms, pointpillars_scatter, sparse_conv3d, ssd_head,
               sweeps, voxel_encoder, voxelize)

bev_pool_v2_backward = bev_pool_v2  # the reference exposes the backward op as its own module

__all__ = ["voxelize", "pointpillars_scatter", "voxel_encoder", "iou3d_nms", "centerpoint_postprocess",
           "bev_pool_v2", "bev_pool_v2_backward", "sparse_conv3d", "sweeps", "conv", "ssd_head"]
