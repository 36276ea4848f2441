"""`paddle3d.ops.bev_pool_v2` / `bev_pool_v2_backward` mirrors.

bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_lengths, interval_starts,
            bev_feat_shape) -> out [B, Y, X, C]               (note: lengths BEFORE starts, bev_pool.cc:30-35)
bev_pool_v2_bkwd(out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_lengths,
                 interval_starts) -> (depth_grad, feat_grad)   (bev_pool_bkwd.cc:24-57)
Reference caller: QuickCumsumCuda, paddle3d/models/transformers/bevdet_transformer.py:27-79.
"""
from __future__ import annotations

import numpy as np
import torch

from ._common import check, lib, ptr, require_gpu, stream_ptr

__all__ = ["bev_pool_v2", "bev_pool_v2_bkwd", "BevPoolV2", "lss_voxel_pooling", "lss_voxel_pooling_fused",
           "lss_pooling_prepare", "voxel_pooling_prepare_v2"]


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
        from .sort import stable_argsort

        order = stable_argsort(ranks_feat, descending=False)  # the library's radix sort
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


def _prepare(coor, batch, depth_bins, feat_hw, lower, interval, size, mode):
    """pd3_voxel_pooling_prepare; one host read of the two counts to hand back exact-size index tensors."""
    from ._common import host_f32, workspace

    op = "voxel_pooling_prepare"
    c = require_gpu(coor, op).reshape(-1, 3)
    n = int(c.shape[0])
    dev = c.device
    outs = [torch.empty((n,), dtype=torch.int32, device=dev) for _ in range(5)]
    counts = torch.empty((2,), dtype=torch.int32, device=dev)
    L = lib()
    ws = workspace(L.pd3_voxel_pooling_prepare_workspace(n), dev)
    hlo, hiv, hsz = host_f32(lower, 3), host_f32(interval, 3), host_f32(size, 3)  # must outlive the call
    check(L.pd3_voxel_pooling_prepare(ptr(c), n, int(batch), int(depth_bins), int(feat_hw), ptr(hlo),
                                      ptr(hiv), ptr(hsz), int(mode),
                                      *[ptr(o) for o in outs], ptr(counts), ptr(ws), ws.numel(), stream_ptr(dev)), op)
    n_kept, n_int = counts.cpu().tolist()
    rb, rd, rf, st, ln = outs
    return rb[:n_kept], rd[:n_kept], rf[:n_kept], st[:n_int], ln[:n_int]


def voxel_pooling_prepare_v2(coor: torch.Tensor, grid_lower_bound, grid_interval, grid_size):
    """LSSViewTransformer.voxel_pooling_prepare_v2 (paddle3d/models/transformers/bevdet_transformer.py:230-274) on
    the device.  coor [B, N, D, H, W, 3] fp32 -> (ranks_bev, ranks_depth, ranks_feat, interval_starts,
    interval_lengths) int32, or five Nones when no frustum point falls inside the grid."""
    B, N, D, H, W, _ = coor.shape
    out = _prepare(coor, B, D, H * W, grid_lower_bound, grid_interval, grid_size, 0)
    if out[3].numel() == 0:
        return None, None, None, None, None
    return out


def _lss_lower(dx, bx):
    dxn = np.asarray(dx, dtype=np.float32).reshape(3)
    bxn = np.asarray(bx, dtype=np.float32).reshape(3)
    return bxn - dxn / np.float32(2.0), dxn  # (bx - dx / 2.) in fp32, as the reference's tensors evaluate it (:328-329)


def lss_pooling_prepare(geom_feats: torch.Tensor, dx, bx, nx, split: bool = True):
    """The index build of `LiftSplatShoot.voxel_pooling` (cam_stream_lss.py:325-346: quantise, filter, sort by cell)
    on the device, for reuse across forwards with a fixed calibration.  geom_feats [B, N, D, H, W, 3] ->
    (cell, ranks_depth, ranks_feat, interval_starts, interval_lengths) int32; `split` = ranks_feat addresses
    feat [B*N, H, W, C] (the fused form), otherwise the lifted [B*N*D*H*W, C] tensor."""
    gg = require_gpu(geom_feats, "lss_pooling_prepare")
    B, N, D, H, W, _ = gg.shape
    lower, dxn = _lss_lower(dx, bx)
    return _prepare(gg, B, D if split else 1, H * W if split else 1, lower, dxn, np.asarray(nx, dtype=np.float32),
                    2 if split else 1)


def lss_voxel_pooling_fused(geom_feats: torch.Tensor, depth: torch.Tensor, feat: torch.Tensor, dx, bx, nx,
                            prepared=None) -> torch.Tensor:
    """BEVFusion's camera->BEV pooling WITHOUT the lifted tensor.  The reference forms x = depth (x) feat
    [B, N, D, H, W, C] in `CamEncode.get_depth_feat` (cam_stream_lss.py:166: 1.41 GB per scene at config 5) and
    pools it (`voxel_pooling`, :318-373); here the two factors go to the pooling kernel as they are:

        geom_feats [B, N, D, H, W, 3], depth [B*N, D, H, W] (softmax over D), feat [B*N, H, W, C]  ->  [B, C, Z, X, Y]

    through pd3_bev_pool_v2 with ranks_depth = frustum point, ranks_feat = camera pixel (prepare mode 2).  Every
    product depth * feat is the fp32 product the reference stores in x, summed per cell in point order: the result is
    bit-identical to `lss_voxel_pooling` on the materialised x (tests/test_lss_c5_gpu.py).  `prepared`: the index
    sets of `lss_pooling_prepare(geom_feats, dx, bx, nx)` when the calibration is fixed."""
    op = "lss_voxel_pooling_fused"
    d, f = require_gpu(depth, op), require_gpu(feat, op)
    B = int(geom_feats.shape[0])
    C = int(f.shape[-1])
    cell, rd, rf, starts, lengths = prepared if prepared is not None else lss_pooling_prepare(geom_feats, dx, bx, nx)
    out = bev_pool_v2(d, f.reshape(-1, C), rd, rf, cell, lengths, starts, (B, nx[2], nx[0] * nx[1], C))
    return out.reshape(B, nx[2], nx[0], nx[1], C).permute(0, 4, 1, 2, 3)


def lss_voxel_pooling(geom_feats: torch.Tensor, x: torch.Tensor, dx, bx, nx) -> torch.Tensor:
    """BEVFusion's camera->BEV pooling, `LiftSplatShoot.voxel_pooling`
    (paddle3d/models/detection/bevfusion/cam_stream_lss.py:318-373), expressed through the bev_pool kernel.

    geom_feats [B,N,D,H,W,3] metric frustum coordinates, x [B,N,D,H,W,C] lifted features ->
    [B, C, Z, X, Y] (the reference's layout).  The reference sorts every frustum point by voxel rank and
    takes differences of one global cumulative sum (the "cumsum trick", :111-121); here the sorted points
    are run-length encoded into intervals and each interval is summed on its own by bev_pool_v2 with unit
    depth weights -- the same sums without the cancellation error of subtracting large running totals.  The
    index build (quantise, filter, stable sort by cell, run-length) is pd3_voxel_pooling_prepare, mode 1.
    """
    op = "lss_voxel_pooling"
    xg = require_gpu(x, op)
    gg = require_gpu(geom_feats, op)
    B, C = int(xg.shape[0]), int(xg.shape[-1])
    nprime = xg.numel() // C
    dev = xg.device
    lower, dxn = _lss_lower(dx, bx)
    cell, _, src, starts, lengths = _prepare(gg, B, 1, 1, lower, dxn, np.asarray(nx, dtype=np.float32), 1)
    ones = torch.ones(1, dtype=torch.float32, device=dev)
    zeros = torch.zeros(cell.shape[0], dtype=torch.int32, device=dev)
    out = bev_pool_v2(ones, xg.reshape(nprime, C), zeros, src, cell, lengths, starts, (B, nx[2], nx[0] * nx[1], C))
    return out.reshape(B, nx[2], nx[0], nx[1], C).permute(0, 4, 1, 2, 3)
