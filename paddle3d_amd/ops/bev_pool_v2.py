"""`paddle3d.ops.bev_pool_v2` / `bev_pool_v2_backward` mirrors.

bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_lengths, interval_starts,
            bev_feat_shape) -> out [B, Y, X, C]               (note: lengths BEFORE starts, bev_pool.cc:30-35)
bev_pool_v2_bkwd(out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_lengths,
                 interval_starts) -> (depth_grad, feat_grad)   (bev_pool_bkwd.cc:24-57)
Reference caller: QuickCumsumCuda, paddle3d/models/transformers/bevdet_transformer.py:27-79.
"""
from __future__ import annotations

import torch

from ._common import check, lib, ptr, require_gpu, stream_ptr

__all__ = ["bev_pool_v2", "bev_pool_v2_bkwd", "BevPoolV2", "lss_voxel_pooling"]


def _i32(t, op):
    return require_gpu(t, op, torch.int32)


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_lengths, interval_starts,
                bev_feat_shape):
    op = "bev_pool_v2"
    d, f = require_gpu(depth, op), require_gpu(feat, op)
    rd, rf, rb = _i32(ranks_depth, op), _i32(ranks_feat, op), _i32(ranks_bev, op)
    il, is_ = _i32(interval_lengths, op), _i32(interval_starts, op)
    c = f.shape[-1]
    out = torch.empty(tuple(int(x) for x in bev_feat_shape), dtype=torch.float32, device=f.device)
    check(lib().pd3_bev_pool_v2(ptr(d), ptr(f), ptr(rd), ptr(rf), ptr(rb), ptr(il), ptr(is_),
                                il.numel(), c, out.numel(), ptr(out), stream_ptr(f.device)), op)
    return out


def bev_pool_v2_bkwd(out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_lengths,
                     interval_starts):
    op = "bev_pool_v2_bkwd"
    g, d, f = require_gpu(out_grad, op), require_gpu(depth, op), require_gpu(feat, op)
    rd, rf, rb = _i32(ranks_depth, op), _i32(ranks_feat, op), _i32(ranks_bev, op)
    il, is_ = _i32(interval_lengths, op), _i32(interval_starts, op)
    c = g.shape[-1]
    dg, fg = torch.empty_like(d), torch.empty_like(f)
    check(lib().pd3_bev_pool_v2_bkwd(ptr(g), ptr(d), ptr(f), ptr(rd), ptr(rf), ptr(rb), ptr(il), ptr(is_),
                                     il.numel(), rd.numel(), c, d.numel(), f.numel(), ptr(dg), ptr(fg),
                                     stream_ptr(f.device)), op)
    return dg, fg


class BevPoolV2(torch.autograd.Function):
    """Autograd wrapper with the structure of the reference's QuickCumsumCuda PyLayer
    (bevdet_transformer.py:27-79): backward re-sorts the index sets by ranks_feat and calls the
    separate backward op."""

    @staticmethod
    def forward(ctx, depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                interval_lengths):
        out = bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_lengths,
                          interval_starts, bev_feat_shape)
        ctx.save_for_backward(ranks_bev, depth, feat, ranks_feat, ranks_depth)
        return out

    @staticmethod
    def backward(ctx, out_grad):
        ranks_bev, depth, feat, ranks_feat, ranks_depth = ctx.saved_tensors
        order = torch.argsort(ranks_feat.long(), stable=True)
        rb, rd, rf = ranks_bev[order], ranks_depth[order], ranks_feat[order]
        kept = torch.ones(rb.shape[0], dtype=torch.bool, device=rb.device)
        kept[1:] = rf[1:] != rf[:-1]
        starts = torch.nonzero(kept).squeeze(1).to(torch.int32)
        lengths = torch.empty_like(starts)
        lengths[:-1] = starts[1:] - starts[:-1]
        lengths[-1] = rb.shape[0] - starts[-1]
        dg, fg = bev_pool_v2_bkwd(out_grad.contiguous(), depth, feat, rd.contiguous(), rf.contiguous(),
                                  rb.contiguous(), lengths, starts)
        return dg, fg, None, None, None, None, None, None


def lss_voxel_pooling(geom_feats: torch.Tensor, x: torch.Tensor, dx, bx, nx) -> torch.Tensor:
    """BEVFusion's camera->BEV pooling, `LiftSplatShoot.voxel_pooling`
    (paddle3d/models/detection/bevfusion/cam_stream_lss.py:318-373), expressed through the bev_pool kernel.

    geom_feats [B,N,D,H,W,3] metric frustum coordinates, x [B,N,D,H,W,C] lifted features ->
    [B, C, Z, X, Y] (the reference's layout).  The reference sorts every frustum point by voxel rank and
    takes differences of one global cumulative sum (the "cumsum trick", :111-121); here the sorted points
    are run-length encoded into intervals and each interval is summed on its own by bev_pool_v2 with unit
    depth weights -- the same sums without the cancellation error of subtracting large running totals.
    """
    op = "lss_voxel_pooling"
    xg = require_gpu(x, op)
    gg = require_gpu(geom_feats, op)
    B, C = int(xg.shape[0]), int(xg.shape[-1])
    nprime = xg.numel() // C
    dev = xg.device
    dxt = torch.as_tensor(dx, dtype=torch.float32, device=dev)
    bxt = torch.as_tensor(bx, dtype=torch.float32, device=dev)
    g = ((gg.reshape(nprime, 3) - (bxt - dxt / 2.0)) / dxt).to(torch.int64)  # trunc toward zero like astype
    batch_ix = torch.arange(B, device=dev).repeat_interleave(nprime // B)
    kept = (g[:, 0] >= 0) & (g[:, 0] < nx[0]) & (g[:, 1] >= 0) & (g[:, 1] < nx[1]) & (g[:, 2] >= 0) & (g[:, 2] < nx[2])
    idx = torch.nonzero(kept).squeeze(1)
    g, b = g[idx], batch_ix[idx]
    # output cell in the reference's final layout [B, Z, X, Y]
    cell = ((b * nx[2] + g[:, 2]) * nx[0] + g[:, 0]) * nx[1] + g[:, 1]
    order = torch.argsort(cell, stable=True)
    cell, src = cell[order], idx[order]
    head = torch.ones(cell.shape[0], dtype=torch.bool, device=dev)
    head[1:] = cell[1:] != cell[:-1]
    starts = torch.nonzero(head).squeeze(1).to(torch.int32)
    lengths = torch.empty_like(starts)
    if starts.numel() > 0:
        lengths[:-1] = starts[1:] - starts[:-1]
        lengths[-1] = cell.shape[0] - starts[-1]
    ones = torch.ones(1, dtype=torch.float32, device=dev)
    zeros = torch.zeros(cell.shape[0], dtype=torch.int32, device=dev)
    out = bev_pool_v2(ones, xg.reshape(nprime, C), zeros, src.to(torch.int32), cell.to(torch.int32), lengths,
                      starts, (B, nx[2], nx[0] * nx[1], C))
    return out.reshape(B, nx[2], nx[0], nx[1], C).permute(0, 4, 1, 2, 3)
