"""`paddle3d.models.layers.layer_libs` mirror, the part on the hot path: rotate_nms_pcdet
(reference paddle3d/models/layers/layer_libs.py:210-249), the caller of iou3d_nms.nms_gpu in
CenterHead.single_post_processing (center_head.py:480-489) and the RoI heads."""
from __future__ import annotations

import math

import torch

from .ops import iou3d_nms
from .ops.sort import stable_argsort

__all__ = ["rotate_nms_pcdet"]


def rotate_nms_pcdet(boxes: torch.Tensor, scores: torch.Tensor, thresh: float, pre_max_size=None, post_max_size=None):
    """boxes [N, 7+] (x, y, z, l, w, h, ..., theta) and scores [N] on the GPU -> indices (int64, GPU) of the kept
    boxes, best first.  Same steps as the reference: columns reordered to the NMS kernel's (x, y, z, w, l, h, theta),
    heading -> -theta - pi/2, descending sort, top `pre_max_size`, rotated NMS, `post_max_size` cap.
    Ties between equal scores are broken by index (a stable sort; the reference's argsort leaves them open)."""
    if boxes.dim() != 2 or boxes.shape[1] < 7 or scores.dim() != 1 or scores.shape[0] != boxes.shape[0]:
        raise RuntimeError("rotate_nms_pcdet: boxes must be [N, >=7] and scores [N]")
    if boxes.shape[0] == 0:
        return torch.zeros((0,), dtype=torch.int64, device=boxes.device)
    cols = torch.tensor([0, 1, 2, 4, 3, 5, boxes.shape[1] - 1], device=boxes.device)
    b = boxes.index_select(1, cols).contiguous()       # transform back to pcdet's coordinate (:222-226)
    b[:, -1] = -b[:, -1] - math.pi / 2                 # fp32 tensor op with a scalar, as in the reference (:229)
    order = stable_argsort(scores, descending=True)  # the library's radix sort (ties by index)
    if pre_max_size is not None:
        order = order[:pre_max_size]
    b = b[order].reshape(-1, 7).contiguous()
    keep, num_out = iou3d_nms.nms_gpu(b, thresh)        # CPU int32, like the reference op
    selected = order[keep[: int(num_out)].to(order.device).long()]
    if post_max_size is not None:
        selected = selected[:post_max_size]
    return selected
