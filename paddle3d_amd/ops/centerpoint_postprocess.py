"""`paddle3d.ops.centerpoint_postprocess` mirror.

centerpoint_postprocess(hm, reg, height, dim, vel, rot, voxel_size, point_cloud_range,
    post_center_range, num_classes, down_ratio, score_threshold, nms_iou_threshold, nms_pre_max_size,
    nms_post_max_size, with_velocity) -> (bboxes [K, 9|7] fp32, scores [K] fp32, labels [K] int64)

Reference operator: paddle3d/ops/centerpoint_postprocess/postprocess.cc:91-104, postprocess.cu:104-280;
caller CenterHead.predict_by_custom_op, paddle3d/models/detection/centerpoint/center_head.py:294-339.
hm..rot are lists (one entry per task) of [1, c, H, W] GPU tensors.  `num_classes` is the list the
reference caller builds (len(tasks)**2 long, center_head.py:306-309); only entry t is used for task t.
The single device->host read is the final row count K (the reference syncs twice per task).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._common import check, host_f32, lib, ptr, require_gpu, stream_ptr, workspace

__all__ = ["centerpoint_postprocess", "centerpoint_postprocess_device"]


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def _shared_batch_stride(tensors) -> int:
    """Batch stride (elements) if every tensor is a [B, c, H, W] view whose frames are dense [c, H, W] blocks with
    ONE common, non-trivial batch stride; 0 if they are ordinary contiguous tensors (or anything else)."""
    bs = 0
    for t in tensors:
        if not isinstance(t, torch.Tensor) or t.dim() != 4 or not t.is_cuda or t.dtype != torch.float32:
            return 0
        b, c, h, w = t.shape
        if t.stride(3) != 1 or t.stride(2) != w or t.stride(1) != h * w:
            return 0
        if b > 1 and t.stride(0) == c * h * w:
            return 0  # plain contiguous
        s0 = int(t.stride(0))
        if bs and s0 != bs:
            return 0
        bs = s0
    if tensors and int(tensors[0].shape[0]) == 1:
        return 0
    return bs


def _require_view(t: torch.Tensor, op: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"Unsupported device type for {op} operator.")
    return t


def centerpoint_postprocess_device(hm, reg, height, dim, vel, rot, voxel_size, point_cloud_range,
                                   post_center_range, num_classes, down_ratio, score_threshold,
                                   nms_iou_threshold, nms_pre_max_size, nms_post_max_size, with_velocity,
                                   allow_batch=False, full_sort=False, records=0):
    """No-sync variant: returns padded (bboxes [B,R,dims], scores [B,R], labels [B,R]; rows >= count are zero)
    plus the device int32 row counts [B].  allow_batch=True lifts the reference's batch-1 restriction (frames
    are processed independently in one launch sequence).  full_sort=True forces the reference's own selection
    (a full stable sort of all cells) instead of the in-LDS top-K selection (`selection` of the C ABI).
    records = max_per_img > 0 additionally returns the rows as the [B, max_per_img, 11] record of the multi-GPU
    hand-off (paddle3d_amd/dist.py), written by the operator itself (heads must be views of one fused map)."""
    op = "centerpoint postprocess"
    t_n = len(hm)
    groups = (hm, reg, height, dim, vel, rot)
    if any(len(g) != t_n for g in groups):
        raise RuntimeError("centerpoint_postprocess: every head list needs one tensor per task")
    # heads that are channel slices of ONE wider [B, C, H, W] map (a fused CenterHead) are passed as views with
    # their common batch stride instead of being copied out
    bstride = _shared_batch_stride([t for g in groups for t in g])
    lists = []
    for group in groups:
        lists.append([_require_view(t, op) if bstride else require_gpu(t, op) for t in group])
    hm0 = lists[0][0]
    batch = int(hm0.shape[0])
    if batch != 1 and not allow_batch:
        raise RuntimeError("hm[0] batch size must be 1.")  # CHECK_INPUT_BATCHSIZE, postprocess.cu:19-20
    if any(int(t.shape[0]) != batch for g in lists for t in g):
        raise RuntimeError("centerpoint_postprocess: inconsistent batch size")
    h, w = int(hm0.shape[2]), int(hm0.shape[3])
    dev = hm0.device
    dims = 9 if with_velocity else 7
    rows = t_n * max(int(nms_post_max_size), 1)
    # (rows behind the last one are zeroed by the operator)
    out_b = torch.empty((batch, rows, dims), dtype=torch.float32, device=dev)
    out_s = torch.empty((batch, rows), dtype=torch.float32, device=dev)
    out_l = torch.empty((batch, rows), dtype=torch.int64, device=dev)
    out_n = torch.empty((batch,), dtype=torch.int32, device=dev)
    ncls = np.ascontiguousarray([int(t.shape[1]) for t in lists[0]], dtype=np.int32)
    offs = np.ascontiguousarray([int(num_classes[t]) for t in range(t_n)], dtype=np.int32)
    vs, pr, pcr = host_f32(voxel_size)[:2], host_f32(point_cloud_range)[:2], host_f32(post_center_range, 6)
    vs = np.ascontiguousarray(np.concatenate([vs, [0.0]]).astype(np.float32))
    pr = np.ascontiguousarray(np.concatenate([pr, [0.0] * 4]).astype(np.float32))
    L = lib()
    ws = workspace(L.pd3_centerpoint_postprocess_workspace(batch, t_n, h, w, int(nms_pre_max_size),
                                                           int(nms_post_max_size)), dev)
    arrays = [_ptr_array(g) for g in lists]
    if full_sort and not bstride:
        if batch != 1:
            raise RuntimeError("centerpoint_postprocess: full_sort needs batch 1 or heads with one common batch stride")
        bstride = 1  # never applied with a single frame
    if records:
        if full_sort or (not bstride and batch != 1):
            raise RuntimeError("centerpoint_postprocess: records need batch 1 or heads with one common batch stride")
        bstride = bstride or 1  # never applied with a single frame
        out_r = torch.empty((batch, int(records), 11), dtype=torch.float32, device=dev)
        check(L.pd3_centerpoint_postprocess_records(
            *[C.cast(a, C.c_void_p) for a in arrays], C.c_int64(bstride), batch, t_n, ptr(ncls), h, w, ptr(vs),
            ptr(pr), ptr(pcr), ptr(offs), int(down_ratio), C.c_float(score_threshold),
            C.c_float(nms_iou_threshold), int(nms_pre_max_size), int(nms_post_max_size), int(bool(with_velocity)),
            ptr(out_b), ptr(out_s), ptr(out_l), ptr(out_n), ptr(out_r), int(records), ptr(ws), ws.numel(),
            stream_ptr(dev)), op)
        return out_b, out_s, out_l, out_n, out_r
    if bstride:
        check(L.pd3_centerpoint_postprocess_strided(
            *[C.cast(a, C.c_void_p) for a in arrays], C.c_int64(bstride), batch, t_n, ptr(ncls), h, w, ptr(vs),
            ptr(pr), ptr(pcr), ptr(offs), int(down_ratio), C.c_float(score_threshold),
            C.c_float(nms_iou_threshold), int(nms_pre_max_size), int(nms_post_max_size), int(bool(with_velocity)),
            ptr(out_b), ptr(out_s), ptr(out_l), ptr(out_n), ptr(ws), ws.numel(), stream_ptr(dev),
            int(bool(full_sort))), op)
        return out_b, out_s, out_l, out_n
    check(L.pd3_centerpoint_postprocess(*[C.cast(a, C.c_void_p) for a in arrays], batch, t_n, ptr(ncls), h, w,
                                        ptr(vs), ptr(pr), ptr(pcr), ptr(offs), int(down_ratio),
                                        C.c_float(score_threshold), C.c_float(nms_iou_threshold),
                                        int(nms_pre_max_size), int(nms_post_max_size),
                                        int(bool(with_velocity)), ptr(out_b), ptr(out_s), ptr(out_l),
                                        ptr(out_n), ptr(ws), ws.numel(), stream_ptr(dev)), op)
    return out_b, out_s, out_l, out_n


def centerpoint_postprocess(hm, reg, height, dim, vel, rot, voxel_size, point_cloud_range,
                            post_center_range, num_classes, down_ratio, score_threshold,
                            nms_iou_threshold, nms_pre_max_size, nms_post_max_size, with_velocity):
    b, s, l, n = centerpoint_postprocess_device(hm, reg, height, dim, vel, rot, voxel_size,
                                                point_cloud_range, post_center_range, num_classes,
                                                down_ratio, score_threshold, nms_iou_threshold,
                                                nms_pre_max_size, nms_post_max_size, with_velocity)
    k = int(n.item())
    return b[0, :k], s[0, :k], l[0, :k]
