"""SSDHead.post_process of PointPillars as one device op (`pd3_ssd_postprocess`).

Reference: paddle3d/models/detection/pointpillars/pointpillars_head.py:86-196 (post_process, _single_post_process,
_box_not_empty, _box_empty), pointpillars_coder.py:126-148 (decode), anchors_generator.py:103-121 + :191-210
(generate_anchors_mask, which PointPillars.test_forward runs per frame, pointpillars.py:120-127), layer_libs.py:210-249
(rotate_nms_pcdet).  The whole batch goes through one launch sequence; nothing is read back to the host.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._common import check, host_f32, lib, ptr, require_gpu, stream_ptr, workspace

__all__ = ["ssd_postprocess_device"]


def ssd_postprocess_device(head_map, cls_channel0, box_channel0, dir_channel0, anchors_per_loc, num_classes,
                           encode_background_as_zeros, anchors, anchors_bv, coors, grid_xy, anchor_area_threshold,
                           score_threshold, center_limit_range, nms_iou_threshold, nms_pre_max_size,
                           nms_post_max_size, full_sort=False):
    """head_map [B, C, H, W] fp32 (the head convolutions' NCHW output; a batch-strided view of a wider map is taken
    as it is), anchors [H*W*apl, 7] fp32, anchors_bv [H*W*apl, 4] int32, coors [M, 4] int32 (batch, z, y, x; rows
    with batch < 0 are padding) -> (boxes [B, R, 7], scores [B, R], labels [B, R] int64, counts [B] int32) on the
    device, R = max(nms_post_max_size, 1).  count 0 = the reference's empty result (row 0 then holds its
    `_box_empty` row: zeros, -1, -1).  full_sort=True forces the reference's own selection (a stable sort of all
    anchors) instead of the top-K selection kernel (`selection` of the C ABI; identical results)."""
    op = "ssd_postprocess"
    if not isinstance(head_map, torch.Tensor) or not head_map.is_cuda or head_map.dtype != torch.float32:
        raise RuntimeError(f"Unsupported device type for {op} operator.")
    b, c, h, w = head_map.shape
    if head_map.stride(3) != 1 or head_map.stride(2) != w or head_map.stride(1) != h * w:
        head_map = head_map.contiguous()
    anchors = require_gpu(anchors, op)
    anchors_bv = require_gpu(anchors_bv, op, torch.int32)
    coors = require_gpu(coors, op, torch.int32)
    a = h * w * int(anchors_per_loc)
    if tuple(anchors.shape) != (a, 7) or tuple(anchors_bv.shape) != (a, 4) or coors.dim() != 2 or coors.shape[1] != 4:
        raise RuntimeError(f"{op}: anchors must be [{a}, 7], anchors_bv [{a}, 4], coors [M, 4]")
    width = int(num_classes) + (0 if encode_background_as_zeros else 1)
    need = [cls_channel0 + anchors_per_loc * width, box_channel0 + anchors_per_loc * 7]
    if dir_channel0 >= 0:
        need.append(dir_channel0 + anchors_per_loc * 2)
    if max(need) > c:
        raise RuntimeError(f"{op}: the head map has {c} channels, the channel groups need {max(need)}")
    dev = head_map.device
    rows = max(int(nms_post_max_size), 1)
    out_b = torch.zeros((b, rows, 7), dtype=torch.float32, device=dev)
    out_s = torch.zeros((b, rows), dtype=torch.float32, device=dev)
    out_l = torch.zeros((b, rows), dtype=torch.int64, device=dev)
    out_n = torch.empty((b,), dtype=torch.int32, device=dev)
    lim = None if center_limit_range is None else host_f32(center_limit_range, 6)
    L = lib()
    gx, gy = int(grid_xy[0]), int(grid_xy[1])
    ws = workspace(L.pd3_ssd_postprocess_workspace(b, h, w, int(anchors_per_loc), gx, gy, int(nms_pre_max_size)), dev)
    check(L.pd3_ssd_postprocess(ptr(head_map), C.c_int64(head_map.stride(0)), int(cls_channel0), int(box_channel0),
                                int(dir_channel0), b, h, w, int(anchors_per_loc), int(num_classes),
                                int(bool(encode_background_as_zeros)), ptr(anchors), ptr(anchors_bv), ptr(coors),
                                C.c_int64(coors.shape[0]), gx, gy, C.c_float(anchor_area_threshold),
                                C.c_float(score_threshold), ptr(lim), C.c_float(nms_iou_threshold),
                                int(nms_pre_max_size), int(nms_post_max_size), ptr(out_b), ptr(out_s), ptr(out_l),
                                ptr(out_n), ptr(ws), ws.numel(), stream_ptr(dev), int(bool(full_sort))), op)
    return out_b, out_s, out_l, out_n
