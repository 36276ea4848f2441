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

__all__ = ["bev_pool_v2", "bev_pool_v2_bkwd", "BevPoolV2"]


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
