"""`paddle3d.ops.iou3d_nms` mirror (reference: paddle3d/ops/iou3d_nms/iou3d_nms_api.cpp:73-108).

  nms_gpu(boxes, thresh)        -> (keep int32 CPU [N], num_to_keep int32 CPU [1])   iou3d_nms.cpp:86-141
  nms_normal_gpu(boxes, thresh) -> same                                              iou3d_nms.cpp:143-204
  boxes_iou_bev_gpu(a, b)       -> iou  [N, M] GPU                                   iou3d_nms.cpp:65-84
  boxes_overlap_bev_gpu(a, b)   -> area [N, M] GPU                                   iou3d_nms.cpp:44-63
`keep` / `num_to_keep` come back as CPU int32 tensors exactly like the reference, so the caller idiom
`order[keep[:num_out]]` (paddle3d/models/layers/layer_libs.py:244) works unchanged.  The *_device
variants keep everything on the GPU (no sync) for fused pipelines.
  boxes_iou_bev_cpu(a, b)       -> iou  [N, M] CPU (host tensors in and out)         iou3d_cpu.cpp:241-264
boxes_iou_bev_cpu keeps the reference's contract (CPU tensors in, CPU tensor out) but not its arithmetic unit:
the library has no CPU path, so the host tensors are staged through the same pd3_boxes_iou_bev kernel, which
evaluates the reference's fp32 expressions (iou3d_cpu.cpp:36-239 == iou3d_nms_kernel.cu:28-273).
"""
from __future__ import annotations

import ctypes as C

import torch

from ._common import check, lib, ptr, require_gpu, stream_ptr, workspace

__all__ = ["nms_gpu", "nms_normal_gpu", "nms_gpu_device", "nms_normal_gpu_device", "boxes_iou_bev_gpu",
           "boxes_overlap_bev_gpu", "boxes_iou_bev_cpu"]


def _check_boxes(b, op):
    b = require_gpu(b, op)
    if b.dim() != 2 or b.shape[1] != 7:
        raise RuntimeError(f"{op}: boxes must be [N, 7]")
    return b


def _nms_device(boxes, thresh, normal):
    op = "nms_normal_gpu" if normal else "nms_gpu"
    b = _check_boxes(boxes, op)
    n = b.shape[0]
    dev = b.device
    keep = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    num = torch.empty((1,), dtype=torch.int32, device=dev)
    L = lib()
    ws = workspace(L.pd3_nms_workspace(n), dev)
    fn = L.pd3_nms_normal if normal else L.pd3_nms_bev
    check(fn(ptr(b), n, C.c_float(thresh), ptr(keep), ptr(num), ptr(ws), ws.numel(), stream_ptr(dev)), op)
    return keep[:n], num


def nms_gpu_device(boxes, nms_overlap_thresh):
    return _nms_device(boxes, nms_overlap_thresh, False)


def nms_normal_gpu_device(boxes, nms_overlap_thresh):
    return _nms_device(boxes, nms_overlap_thresh, True)


def nms_gpu(boxes, nms_overlap_thresh):
    keep, num = _nms_device(boxes, nms_overlap_thresh, False)
    return keep.cpu(), num.cpu()


def nms_normal_gpu(boxes, nms_overlap_thresh):
    keep, num = _nms_device(boxes, nms_overlap_thresh, True)
    return keep.cpu(), num.cpu()


def _pairwise(a, b, iou):
    op = "boxes_iou_bev_gpu" if iou else "boxes_overlap_bev_gpu"
    a, b = _check_boxes(a, op), _check_boxes(b, op)
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    fn = lib().pd3_boxes_iou_bev if iou else lib().pd3_boxes_overlap_bev
    check(fn(ptr(a), a.shape[0], ptr(b), b.shape[0], ptr(out), stream_ptr(a.device)), op)
    return out


def boxes_iou_bev_gpu(boxes_a, boxes_b):
    return _pairwise(boxes_a, boxes_b, True)


def boxes_overlap_bev_gpu(boxes_a, boxes_b):
    return _pairwise(boxes_a, boxes_b, False)


def boxes_iou_bev_cpu(boxes_a, boxes_b, device=None):
    """iou3d_cpu.cpp:241-264: [N, 7] x [M, 7] CPU fp32 tensors -> [N, M] CPU fp32."""
    for t in (boxes_a, boxes_b):
        if not isinstance(t, torch.Tensor) or t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != 7:
            raise RuntimeError("boxes_iou_bev_cpu: boxes must be CPU float32 tensors of shape [N, 7]")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    return boxes_iou_bev_gpu(boxes_a.to(dev), boxes_b.to(dev)).cpu()


def libm_eval(op: str, x: torch.Tensor, y: torch.Tensor | None = None) -> torch.Tensor:
    """Diagnostic: the float math routines the geometry / decode kernels use (glibc's sinf / cosf / expf / atanf /
    atan2f bit for bit, csrc/libm_exact.hpp) over a float32 GPU tensor; `y` only for atan2f(x, y)."""
    code = {"sinf": 0, "cosf": 1, "expf": 2, "atanf": 3, "atan2f": 4}[op]
    x = require_gpu(x, "libm_eval")
    y = require_gpu(y, "libm_eval") if y is not None else None
    out = torch.empty_like(x)
    check(lib().pd3_libm_eval(code, ptr(x), ptr(y), ptr(out), x.numel(), stream_ptr(x.device)), "libm_eval")
    return out
