"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY.

NumPy-facing access to the two CPU oracles:

* ``port``  -- oracle/_build/liboracle_port.so, our C restatement (oracle/port/oracle_port.c); travels to
  the GPU box.
* ``ref``   -- oracle/_ref/libref_oracle.so, the reference's OWN arithmetic compiled from the line
  ranges where they lie under /root/reference (oracle/ref_wrap.cpp); built in the dev container and
  shipped as a binary, never as source.

plus NumPy / torch-CPU restatements of the Python layers on the path (PointPillarsScatter, PFN,
VoxelMean, HardVoxelizer glue).  Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline``
leg may import this module; the product package ``paddle3d_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from functools import lru_cache

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(_HERE, "_build", "liboracle_port.so")
REF_SO = os.path.join(_HERE, "_ref", "libref_oracle.so")

_f = np.float32
_i = np.int32


def build(ref: bool = True) -> None:
    """(Re)build the oracle libraries with gcc; ``ref`` only when /root/reference exists."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "port"])
    if ref and os.path.isdir("/root/reference/paddle3d/ops"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


@lru_cache(maxsize=None)
def _lib(kind: str):
    path = PORT_SO if kind == "port" else REF_SO
    if not os.path.exists(path):
        if kind == "port":
            build(ref=False)
        else:
            raise FileNotFoundError(f"{path} missing: run `make -C oracle ref` where /root/reference exists")
    lib = C.CDLL(path)
    return lib


def have_ref() -> bool:
    return os.path.exists(REF_SO)


# ------------------------------------------------------------------------------------------------
# hard_voxelize
# ------------------------------------------------------------------------------------------------
def hard_voxelize(points, voxel_size, pc_range, max_pts, max_voxels, kind="port"):
    """Returns (voxels [V,P,D] f32, coords [V,3] i32 (z,y,x), num_points [V] i32, num_voxels int)."""
    f64 = np.asarray(points).dtype == np.float64  # the reference's kernel instantiated for double
    pts = _c(points, np.float64 if f64 else _f)
    n, d = pts.shape
    vs, pr = _c(voxel_size, _f), _c(pc_range, _f)
    voxels = np.empty((max_voxels, max_pts, d), pts.dtype)
    coords = np.empty((max_voxels, 3), _i)
    npts = np.empty((max_voxels,), _i)
    nv = np.zeros((1,), _i)
    name = ("port" if kind == "port" else "ref") + "_hard_voxelize" + ("_f64" if f64 else "")
    fn = getattr(_lib(kind), name)
    fn.restype = C.c_int
    fn(_p(pts), C.c_int64(n), C.c_int(d), _p(vs), _p(pr), C.c_int(max_pts), C.c_int(max_voxels),
       _p(voxels), _p(coords), _p(npts), _p(nv))
    return voxels, coords, npts, int(nv[0])


# ------------------------------------------------------------------------------------------------
# iou3d_nms
# ------------------------------------------------------------------------------------------------
def boxes_iou_bev(a, b, kind="port"):
    a, b = _c(a, _f), _c(b, _f)
    out = np.empty((len(a), len(b)), _f)
    fn = getattr(_lib(kind), ("port" if kind == "port" else "ref") + "_boxes_iou_bev")
    fn(_p(a), C.c_int(len(a)), _p(b), C.c_int(len(b)), _p(out))
    return out


def boxes_overlap_bev(a, b, kind="port"):
    a, b = _c(a, _f), _c(b, _f)
    out = np.empty((len(a), len(b)), _f)
    fn = getattr(_lib(kind), ("port" if kind == "port" else "ref") + "_boxes_overlap_bev")
    fn(_p(a), C.c_int(len(a)), _p(b), C.c_int(len(b)), _p(out))
    return out


def libm_eval(op, x, y=None):
    """The host libm's sinf / cosf / expf / atanf / atan2f (op 0..4) over float32 arrays."""
    x = _c(x, _f)
    y = _c(y if y is not None else x, _f)
    out = np.empty_like(x)
    _lib("port").port_libm_eval(C.c_int(op), _p(x), _p(y), _p(out), C.c_int64(x.size))
    return out


def nms(boxes, thresh, normal=False, kind="port"):
    """Greedy NMS over score-sorted boxes [N,7]; returns keep indices (int32 [num])."""
    bx = _c(boxes, _f)
    keep = np.empty((max(len(bx), 1),), _i)
    num = np.zeros((1,), _i)
    fn = getattr(_lib(kind), ("port" if kind == "port" else "ref") + "_nms")
    fn(_p(bx), C.c_int(len(bx)), C.c_float(thresh), C.c_int(int(normal)), _p(keep), _p(num))
    return keep[: int(num[0])].copy()


def iou_margin(boxes, thresh, normal=False):
    """min |IoU - thresh| over all pairs: how far a test vector is from a libm-sensitive flip."""
    iou = boxes_iou_bev(boxes, boxes) if not normal else None
    if iou is None:
        raise NotImplementedError
    tri = iou[np.triu_indices(len(boxes), 1)]
    return float(np.min(np.abs(tri - thresh))) if tri.size else float("inf")


# ------------------------------------------------------------------------------------------------
# PointPillarsScatter / voxel encoders (NumPy restatements of the Python layers)
# ------------------------------------------------------------------------------------------------
def pillar_scatter(feats, coords, batch, ny, nx):
    """pillar_scatter.py:57-93 -> canvas [B, C, ny, nx]."""
    f, c4 = _c(feats, _f), _c(coords, _i)
    out = np.empty((batch, f.shape[1], ny, nx), _f)
    _lib("port").port_pillar_scatter(_p(f), _p(c4), C.c_int64(len(f)), C.c_int(f.shape[1]),
                                     C.c_int(batch), C.c_int(ny), C.c_int(nx), _p(out))
    return out


def pillar_scatter_numpy(feats, coords, batch, ny, nx):
    """Independent NumPy statement of the same layer (zeros canvas, indexed assignment, transpose)."""
    c = feats.shape[1]
    out = []
    for b in range(batch):
        canvas = np.zeros((nx * ny, c), feats.dtype)
        m = coords[:, 0] == b
        idx = coords[m, 2] * nx + coords[m, 3]
        canvas[idx] = feats[m]
        out.append(canvas.T.reshape(1, c, ny, nx))
    return np.concatenate(out, 0)


def voxel_mean(voxels, num_points):
    """voxel_encoder.py:44-57: sum over P / count."""
    return voxels.sum(1, dtype=np.float32) / num_points.astype(np.float32).reshape(-1, 1)


def pfn_forward_torch(voxels, num_points, coors, params, voxel_size, pc_range):
    """PillarFeatureNet.forward in eval mode (pillar_encoder.py:156-210, PFNLayer :81-105), torch CPU fp32.

    params: list of dicts per PFN layer with 'weight' [in, units] (Paddle Linear layout), 'gamma',
    'beta', 'mean', 'var' (BatchNorm1D, eps 1e-3).
    """
    import torch

    f = torch.as_tensor(voxels, dtype=torch.float32)
    npv = torch.as_tensor(num_points).to(torch.float32).reshape(-1, 1, 1)
    co = torch.as_tensor(coors)
    P = f.shape[1]
    vx, vy = float(voxel_size[0]), float(voxel_size[1])
    x_off, y_off = vx / 2 + pc_range[0], vy / 2 + pc_range[1]
    mean = f[:, :, :3].sum(1, keepdim=True) / npv
    f_cluster = f[:, :, :3] - mean
    f_center = torch.zeros_like(f[:, :, :2])
    f_center[:, :, 0] = f[:, :, 0] - (co[:, 3].reshape(-1, 1).to(torch.float32) * vx + x_off)
    f_center[:, :, 1] = f[:, :, 1] - (co[:, 2].reshape(-1, 1).to(torch.float32) * vy + y_off)
    x = torch.cat([f, f_cluster, f_center], -1)
    mask = (torch.as_tensor(num_points).reshape(-1, 1) > torch.arange(P).reshape(1, -1)).to(torch.float32)
    x = x * mask.unsqueeze(-1)
    for li, p in enumerate(params):
        w, g, b, mu, var = (torch.as_tensor(np.asarray(p[k]), dtype=torch.float32)
                            for k in ("weight", "gamma", "beta", "mean", "var"))
        y = x @ w
        y = (y - mu) / torch.sqrt(var + 1e-3)
        y = y * g + b
        y = torch.relu(y)
        ymax = y.max(dim=1, keepdim=True).values
        if li == len(params) - 1:
            x = ymax
        else:
            x = torch.cat([y, ymax.expand(-1, P, -1)], dim=2)
    return x.squeeze(1).numpy()


def hard_vfe_forward_torch(voxels, num_points, coors, params, voxel_size, pc_range):
    """HardVFE.forward in eval mode with cluster + voxel centre (voxel_encoder.py:220-283, VFELayer :94-138)."""
    import torch

    f = torch.as_tensor(voxels, dtype=torch.float32)
    npv = torch.as_tensor(num_points).to(torch.float32).reshape(-1, 1, 1)
    co = torch.as_tensor(coors).to(torch.float32)
    P = f.shape[1]
    vx, vy, vz = (float(v) for v in voxel_size)
    offs = (vx / 2 + pc_range[0], vy / 2 + pc_range[1], vz / 2 + pc_range[2])
    f_cluster = f[:, :, :3] - f[:, :, :3].sum(1, keepdim=True) / npv
    f_center = torch.zeros(f.shape[0], P, 3)
    f_center[:, :, 0] = f[:, :, 0] - (co[:, 3].unsqueeze(1) * vx + offs[0])
    f_center[:, :, 1] = f[:, :, 1] - (co[:, 2].unsqueeze(1) * vy + offs[1])
    f_center[:, :, 2] = f[:, :, 2] - (co[:, 1].unsqueeze(1) * vz + offs[2])
    x = torch.cat([f, f_cluster, f_center], -1)
    mask = (torch.as_tensor(num_points).reshape(-1, 1) > torch.arange(P).reshape(1, -1)).to(torch.float32)
    x = x * mask.unsqueeze(-1)
    for li, p in enumerate(params):
        w, g, b, mu, var = (torch.as_tensor(np.asarray(p[k]), dtype=torch.float32)
                            for k in ("weight", "gamma", "beta", "mean", "var"))
        y = torch.relu(((x @ w) - mu) / torch.sqrt(var + 1e-3) * g + b)
        ymax = y.max(dim=1, keepdim=True).values
        x = ymax if li == len(params) - 1 else torch.cat([y, ymax.expand(-1, P, -1)], dim=2)
    return x.squeeze(1).numpy()


def merge_sweeps_numpy(key_frame, sweeps, ref_from_curr, time_lags, use_dim=None, use_time_lag=True, radius=1.0):
    """LoadPointCloud.__call__ multi-sweep merge (reader.py:118-164) for a GIVEN sweep order."""
    data = key_frame if use_dim is None else key_frame[:, :use_dim]
    if use_time_lag:
        data = np.hstack([data, np.zeros((data.shape[0], 1), dtype=data.dtype)])
    out = [data]
    for sw, m, lag in zip(sweeps, ref_from_curr, time_lags):
        sd = (sw if use_dim is None else sw[:, :use_dim]).T.copy()
        not_close = np.logical_not(np.logical_and(np.abs(sd[0, :]) < radius, np.abs(sd[1, :]) < radius))
        sd = sd[:, not_close]
        if m is not None:
            sd[:3, :] = np.asarray(m).dot(np.vstack((sd[:3, :], np.ones(sd.shape[1]))))[:3, :]
        sd = sd.T
        if use_time_lag:
            sd = np.hstack([sd, lag * np.ones((sd.shape[0], 1)).astype(sd.dtype)])
        out.append(sd)
    return np.concatenate(out, axis=0)


def lss_voxel_pooling_numpy(geom, x, dx, bx, nx):
    """LiftSplatShoot.voxel_pooling with the cumsum trick (cam_stream_lss.py:111-121, :318-373), NumPy.
    geom [B,N,D,H,W,3] metric coordinates, x [B,N,D,H,W,C] -> [B, C, Z, X, Y]."""
    B = x.shape[0]
    C = x.shape[-1]
    nprime = int(np.prod(x.shape[:-1]))
    xf = x.reshape(nprime, C).astype(np.float32)
    g = ((geom - (bx - dx / 2.0)) / dx).astype(np.int64).reshape(nprime, 3)
    batch_ix = np.repeat(np.arange(B), nprime // B).reshape(-1, 1)
    g = np.concatenate([g, batch_ix], 1)
    kept = (g[:, 0] >= 0) & (g[:, 0] < nx[0]) & (g[:, 1] >= 0) & (g[:, 1] < nx[1]) & (g[:, 2] >= 0) & (g[:, 2] < nx[2])
    xf, g = xf[kept], g[kept]
    ranks = g[:, 0] * (nx[1] * nx[2] * B) + g[:, 1] * (nx[2] * B) + g[:, 2] * B + g[:, 3]
    order = np.argsort(ranks, kind="stable")
    xf, g, ranks = xf[order], g[order], ranks[order]
    cs = np.cumsum(xf, 0, dtype=np.float32)
    keep = np.ones(len(xf), bool)
    keep[:-1] = ranks[1:] != ranks[:-1]
    cs, g = cs[keep], g[keep]
    cs = np.concatenate([cs[:1], cs[1:] - cs[:-1]])
    final = np.zeros((B * nx[2] * nx[0] * nx[1], C), np.float32)
    final[g[:, 3] * (nx[2] * nx[0] * nx[1]) + g[:, 2] * (nx[0] * nx[1]) + g[:, 0] * nx[1] + g[:, 1]] = cs
    return final.reshape(B, nx[2], nx[0], nx[1], C).transpose(0, 4, 1, 2, 3)


def lss_voxel_pooling_exact(geom, x, dx, bx, nx, chunk=16):
    """The map `LiftSplatShoot.voxel_pooling` DEFINES (cam_stream_lss.py:318-373: the sum of the lifted features of
    the frustum points of every cell), with every per-cell sum taken in float64 -- the yardstick both the reference's
    cumsum trick (fp32 running totals over millions of rows, then differences) and the device's per-cell fp32 sums are
    measured against.  Same quantisation and filter as `lss_voxel_pooling_numpy`; channels in chunks to bound memory.
    Returns float64 [B, C, Z, X, Y]."""
    B, C = x.shape[0], x.shape[-1]
    nprime = int(np.prod(x.shape[:-1]))
    xf = x.reshape(nprime, C)
    g = ((geom - (bx - dx / 2.0)) / dx).astype(np.int64).reshape(nprime, 3)
    kept = (g[:, 0] >= 0) & (g[:, 0] < nx[0]) & (g[:, 1] >= 0) & (g[:, 1] < nx[1]) & (g[:, 2] >= 0) & (g[:, 2] < nx[2])
    idx = np.nonzero(kept)[0]
    g = g[idx]
    cell = ((idx // (nprime // B)) * nx[2] + g[:, 2]) * (nx[0] * nx[1]) + g[:, 0] * nx[1] + g[:, 1]
    order = np.argsort(cell, kind="stable")
    idx, cell = idx[order], cell[order]
    head = np.ones(len(cell), bool)
    head[1:] = cell[1:] != cell[:-1]
    starts = np.nonzero(head)[0]
    final = np.zeros((B * nx[2] * nx[0] * nx[1], C), np.float64)
    for c0 in range(0, C, chunk):
        rows = xf[idx, c0:c0 + chunk].astype(np.float64)
        final[cell[starts], c0:c0 + chunk] = np.add.reduceat(rows, starts, axis=0) if len(starts) else 0.0
    return final.reshape(B, nx[2], nx[0], nx[1], C).transpose(0, 4, 1, 2, 3)


# ------------------------------------------------------------------------------------------------
# centerpoint_postprocess
# ------------------------------------------------------------------------------------------------
def centerpoint_postprocess(tasks, voxel_size, pc_range, post_center_range, num_classes, down_ratio,
                            score_threshold, nms_iou_threshold, nms_pre_max_size, nms_post_max_size,
                            with_velocity=True, return_margins=False):
    """tasks: list of dict(hm, reg, height, dim, vel, rot) float32 [1,c,H,W].  num_classes: label
    offset per task (postprocess.cu:268-270).  Returns (bboxes [K,9|7], scores [K], labels int64 [K])."""
    lib = _lib("port")
    fn = lib.port_centerpoint_postprocess_task
    fn.restype = C.c_int
    dims = 9 if with_velocity else 7
    vs, pr, pcr = _c(voxel_size, _f), _c(pc_range, _f), _c(post_center_range, _f)
    outs_b, outs_s, outs_l, margins = [], [], [], []
    h, w = tasks[0]["hm"].shape[2:]
    for t_id, t in enumerate(tasks):
        hm = _c(t["hm"], _f)
        cap = max(1, nms_post_max_size)
        ob = np.zeros((cap, dims), _f)
        os_ = np.zeros((cap,), _f)
        ol = np.zeros((cap,), np.int64)
        mm = np.zeros((2,), _f)
        arrs = [_c(t[k], _f) for k in ("reg", "height", "dim", "vel", "rot")]
        rows = fn(_p(hm), C.c_int(hm.shape[1]), _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]),
                  _p(arrs[4]), C.c_int(h), C.c_int(w), _p(vs), _p(pr), _p(pcr),
                  C.c_int(int(num_classes[t_id])), C.c_int(int(down_ratio)), C.c_float(score_threshold),
                  C.c_float(nms_iou_threshold), C.c_int(nms_pre_max_size), C.c_int(nms_post_max_size),
                  C.c_int(int(with_velocity)), _p(ob), _p(os_), _p(ol), _p(mm))
        outs_b.append(ob[:rows])
        outs_s.append(os_[:rows])
        outs_l.append(ol[:rows])
        margins.append(mm.copy())
    res = (np.concatenate(outs_b), np.concatenate(outs_s), np.concatenate(outs_l))
    if return_margins:
        return res + (np.stack(margins),)
    return res


def centerpoint_pillars_pipeline(model_cpu, points, max_pts, max_voxels, kind=None, dense_batch=8):
    """The CPU statement of the WHOLE CenterPoint-Pillars path for a list / array of frames [B, N, D]:
    hard_voxelize (the reference kernel when oracle/_ref is built) -> PFN -> scatter (C port) -> SECOND + FPN +
    CenterHead (torch CPU fp32, `dense_batch` frames at a time) -> centerpoint_postprocess (C port).
    `model_cpu`: a paddle3d_amd.centerpoint.CenterPoint on the CPU (parameter container only).  Returns a list of
    dict(box3d_lidar, scores, label_preds) NumPy arrays -- what `test_forward` returns on the device."""
    import torch

    kind = kind or ("ref" if have_ref() else "port")
    cfg = model_cpu.test_cfg
    vs3, rng6 = list(model_cpu.voxelizer.voxel_size), list(model_cpu.voxelizer.point_cloud_range)
    nx = int(round((rng6[3] - rng6[0]) / vs3[0]))
    ny = int(round((rng6[4] - rng6[1]) / vs3[1]))
    params = []
    for l in model_cpu.voxel_encoder.pfn_layers:
        params.append(dict(weight=l.linear.weight.t().detach().numpy(), gamma=l.norm.weight.detach().numpy(),
                           beta=l.norm.bias.detach().numpy(), mean=l.norm.running_mean.numpy(),
                           var=l.norm.running_var.numpy()))
    label_offsets = np.concatenate([[0], np.cumsum([t.hm[-1].out_channels for t in model_cpu.bbox_head.tasks])[:-1]])
    out = []
    for b0 in range(0, len(points), dense_batch):
        canvases = []
        for pts in points[b0:b0 + dense_batch]:
            vox, co, npv, nv = hard_voxelize(np.ascontiguousarray(pts), vs3, rng6, max_pts, max_voxels, kind)
            c4 = np.concatenate([np.zeros((nv, 1), np.int32), co[:nv]], 1)
            feats = pfn_forward_torch(vox[:nv], npv[:nv], c4, params, vs3, rng6)
            canvases.append(pillar_scatter(feats, c4, 1, ny, nx))
        with torch.no_grad():
            preds, _ = center_head_torch(model_cpu.bbox_head,
                                         dense_forward_torch(model_cpu, torch.from_numpy(np.concatenate(canvases))))
        for i in range(len(canvases)):
            tasks = [{k: v[i:i + 1].numpy() for k, v in p.items()} for p in preds]
            bb, sc, lab = centerpoint_postprocess(
                tasks, cfg["voxel_size"] + [vs3[2]], cfg["point_cloud_range"] + [0.0] * 4, cfg["post_center_limit_range"],
                [int(v) for v in label_offsets], cfg["down_ratio"], cfg["score_threshold"],
                cfg["nms"]["nms_iou_threshold"], cfg["nms"]["nms_pre_max_size"], cfg["nms"]["nms_post_max_size"], True)
            out.append(dict(box3d_lidar=bb, scores=sc, label_preds=lab))
    return out


def centerpoint_decode_ref(score, reg, height, expdim, vel, rot, score_threshold, feat_w, down_ratio,
                           voxel_size, pc_range, post_center_range, with_velocity=True):
    """The reference decode_kernel itself (oracle/_ref), for pinning the port's decode stage."""
    lib = _lib("ref")
    hw = score.size
    dims = 9 if with_velocity else 7
    boxes = np.zeros((hw, dims), _f)
    mask = np.zeros((hw,), np.uint8)
    sidx = np.zeros((hw,), _i)
    pcr = _c(post_center_range, _f)
    args = [_c(a, _f) for a in (score, reg, height, expdim, vel, rot)]
    lib.ref_centerpoint_decode(*[_p(a) for a in args], C.c_float(score_threshold), C.c_int(feat_w),
                               C.c_float(down_ratio), C.c_float(voxel_size[0]), C.c_float(voxel_size[1]),
                               C.c_float(pc_range[0]), C.c_float(pc_range[1]), _p(pcr), C.c_int(hw),
                               C.c_int(int(with_velocity)), _p(boxes), _p(mask), _p(sidx))
    return boxes, mask.astype(bool), sidx


# ------------------------------------------------------------------------------------------------
# bev_pool_v2
# ------------------------------------------------------------------------------------------------
def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_lengths, interval_starts,
                bev_feat_shape, kind="port"):
    d, f = _c(depth, _f), _c(feat, _f)
    rd, rf, rb = _c(ranks_depth, _i), _c(ranks_feat, _i), _c(ranks_bev, _i)
    il, is_ = _c(interval_lengths, _i), _c(interval_starts, _i)
    c = f.shape[-1]
    out = np.zeros(tuple(bev_feat_shape), _f)
    fn = getattr(_lib(kind), ("port" if kind == "port" else "ref") + "_bev_pool_v2")
    fn(C.c_int(c), C.c_int(len(il)), _p(d), _p(f), _p(rd), _p(rf), _p(rb), _p(is_), _p(il), _p(out))
    return out


def bev_pool_v2_bkwd(out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_lengths,
                     interval_starts, kind="port"):
    g, d, f = _c(out_grad, _f), _c(depth, _f), _c(feat, _f)
    rd, rf, rb = _c(ranks_depth, _i), _c(ranks_feat, _i), _c(ranks_bev, _i)
    il, is_ = _c(interval_lengths, _i), _c(interval_starts, _i)
    c = g.shape[-1]
    dg, fg = np.zeros_like(d), np.zeros_like(f)
    fn = getattr(_lib(kind), ("port" if kind == "port" else "ref") + "_bev_pool_v2_bkwd")
    fn(C.c_int(c), C.c_int(len(il)), _p(g), _p(d), _p(f), _p(rd), _p(rf), _p(rb), _p(is_), _p(il),
       _p(dg), _p(fg))
    return dg, fg


# ---------------------------------------------------------------------------------------------------------
# Dense BEV graph: the reference's layers stated with torch's own convolutions (MIOpen on a GPU tensor, MKL-DNN
# on a CPU tensor).  The product (paddle3d_amd/centerpoint.py) runs these layers on its hand-written kernels
# only; the tests and bench.py's cpu_baseline compare / time against the functions below.
# ---------------------------------------------------------------------------------------------------------
def second_backbone_torch(backbone, x):
    """second_backbone.py:114-120: the blocks are Sequential(conv, bn, relu, ...) -- torch runs them as written."""
    outs = []
    for blk in backbone.blocks:
        x = blk(x)
        outs.append(x)
    return tuple(outs)


def second_fpn_torch(neck, xs):
    """second_fpn.py:140-157: per level deblock, then channel concat."""
    import torch

    ups = [d(x) for d, x in zip(neck.deblocks, xs)]
    return torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]


def dense_forward_torch(model, x):
    """CenterPoint.extract_feat's dense half (centerpoint.py:133-137): backbone -> neck."""
    return second_fpn_torch(model.neck, second_backbone_torch(model.backbone, x))


def center_head_torch(head, x):
    """CenterHead.forward (center_head.py:212-220): shared ConvModule, then every SeparateHead branch
    (ConvModule -> conv); returns (list of per-task dicts, shared map) like the fused product path."""
    def conv_module(m, t):
        return m.activate(m.bn(m.conv(t)))

    x = conv_module(head.shared_conv, x)
    rets = []
    for task in head.tasks:
        d = {}
        for name in task.heads:
            seq = getattr(task, name)
            t = x
            for layer in seq:
                t = conv_module(layer, t) if hasattr(layer, "bn") else layer(t)
            d[name] = t
        rets.append(d)
    return rets, x


# ---------------------------------------------------------------------------------------------------------
# Sparse middle encoders by densification: the public definition of paddle.sparse.nn.SubmConv3D / Conv3D /
# BatchNorm (Paddle core, not vendored: parity unpinned against Paddle's own kernels) stated with dense torch
# conv3d on the CPU -- a submanifold convolution keeps the input's active set, a regular one activates every
# output some active input reaches, BatchNorm / ReLU / add act on active sites only.
# ---------------------------------------------------------------------------------------------------------
def sparse_encoder_dense_torch(net, feats, coords, batch):
    """Dense statement of SparseResNet3D.forward (sparse_resnet.py:184-206) / SparseNet3D.forward
    (sparsenet.py:140-182) for the module mirrors in paddle3d_amd/sparse.py.  Returns the [B, C*D, H, W] map."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    d, h, w = net.sparse_shape
    co = torch.as_tensor(np.asarray(coords)).long()
    x = torch.zeros(batch, feats.shape[1], d, h, w)
    x[co[:, 0], :, co[:, 1], co[:, 2], co[:, 3]] = torch.as_tensor(np.asarray(feats))
    mask = torch.zeros(batch, 1, d, h, w)
    mask[co[:, 0], 0, co[:, 1], co[:, 2], co[:, 3]] = 1

    def conv(m, t, mk):
        wt = m.weight.detach().cpu().permute(4, 3, 0, 1, 2).contiguous()  # [kd,kh,kw,ci,co] -> [co,ci,kd,kh,kw]
        b = None if m.bias is None else m.bias.detach().cpu()
        if m.subm:
            return F.conv3d(t, wt, b, padding=tuple(k // 2 for k in m.ks)) * mk, mk
        nm = (F.conv3d(mk, torch.ones(1, 1, *m.ks), stride=m.stride, padding=m.padding) > 0).float()
        return F.conv3d(t, wt, b, stride=m.stride, padding=m.padding) * nm, nm

    def bn(m, t, mk):
        sc = (m.weight / torch.sqrt(m.running_var + m.eps)).detach().cpu().view(1, -1, 1, 1, 1)
        sh = (m.bias.detach().cpu().view(1, -1, 1, 1, 1) - m.running_mean.detach().cpu().view(1, -1, 1, 1, 1) * sc)
        return (t * sc + sh) * mk

    def run(mod, t, mk):
        if isinstance(mod, nn.Sequential):
            for sub in mod:
                t, mk = run(sub, t, mk)
            return t, mk
        if hasattr(mod, "conv1") and hasattr(mod, "bn2"):  # SparseBasicBlock, sparse_resnet.py:92-111
            o, _ = conv(mod.conv1, t, mk)
            o = torch.relu(bn(mod.bn1, o, mk))
            o, _ = conv(mod.conv2, o, mk)
            return torch.relu(bn(mod.bn2, o, mk) + t), mk
        if isinstance(mod, nn.BatchNorm1d):
            return bn(mod, t, mk), mk
        if isinstance(mod, nn.ReLU):
            return torch.relu(t), mk
        return conv(mod, t, mk)

    with torch.no_grad():
        for stage in (net.conv_input, net.conv1, net.conv2, net.conv3, net.conv4, net.extra_conv):
            x, mask = run(stage, x, mask)
    n, c, dd, hh, ww = x.shape
    return x.reshape(n, c * dd, hh, ww)


# ---------------------------------------------------------------------------------------------------------
# The same sparse encoders WITHOUT a dense grid (round 5): index sets are sorted int64 key arrays
# ((b * D + z) * H + y) * W + x, a neighbour is found with np.searchsorted, the sum over kernel offsets runs in
# ascending offset order in fp32.  Needs no [B, C, D, H, W] tensor, so it runs at the full config-4 grid
# (41 x 1440 x 1440, ~130 k voxels per scene) where the dense statement above cannot.  Same public definition of
# paddle.sparse.nn.SubmConv3D / Conv3D as above (Paddle core arithmetic absent: parity unpinned against Paddle's
# own kernels); tests/test_oracle.py pins this statement to the dense one on small grids.
# ---------------------------------------------------------------------------------------------------------
def _sp_decode(keys, shape):
    d, h, w = (int(v) for v in shape)
    x = keys % w
    r = keys // w
    y = r % h
    r = r // h
    return r // d, r % d, y, x  # b, z, y, x


def sparse_keys_numpy(coords, shape):
    """coords [n, 4] (b, z, y, x) -> (sorted unique int64 keys, order) with keys[i] = key of coords[order[i]]."""
    c = np.asarray(coords).astype(np.int64)
    d, h, w = (int(v) for v in shape)
    keys = ((c[:, 0] * d + c[:, 1]) * h + c[:, 2]) * w + c[:, 3]
    order = np.argsort(keys, kind="stable")
    keys = keys[order]
    if keys.size > 1 and (np.diff(keys) <= 0).any():
        raise ValueError("duplicate voxel coordinates")
    return keys, order


def sparse_conv_numpy(feats, keys, batch, shape, weight, stride=(1, 1, 1), padding=(0, 0, 0), subm=False, bias=None):
    """One sparse convolution on a sorted key set.  feats [n, cin] fp32 (row i at keys[i]), weight [kd, kh, kw, cin,
    cout] (Paddle layout).  Submanifold: the output set is the input set ("same" padding k // 2, stride 1);
    regular: every output cell some input reaches, in raster order.  Returns (out_feats, out_keys, out_shape, pairs)
    -- pairs = the number of (output row, offset) pairs that exist."""
    feats = np.ascontiguousarray(feats, _f)
    weight = np.asarray(weight, _f)
    kd, kh, kw, cin, cout = weight.shape
    d, h, w = (int(v) for v in shape)
    if subm:
        stride, padding = (1, 1, 1), (kd // 2, kh // 2, kw // 2)
        oshape, okeys = (d, h, w), keys
    else:
        stride, padding = tuple(int(v) for v in stride), tuple(int(v) for v in padding)
        oshape = tuple((s + 2 * p - k) // st + 1 for s, k, st, p in zip((d, h, w), (kd, kh, kw), stride, padding))
        b, z, y, x = _sp_decode(keys, shape)
        cand = []
        for kz in range(kd):
            for ky in range(kh):
                for kx in range(kw):
                    nz, ny, nx = z + padding[0] - kz, y + padding[1] - ky, x + padding[2] - kx  # = q * stride
                    ok = (nz >= 0) & (ny >= 0) & (nx >= 0) & (nz % stride[0] == 0) & (ny % stride[1] == 0) & \
                        (nx % stride[2] == 0)
                    qz, qy, qx = nz // stride[0], ny // stride[1], nx // stride[2]
                    ok &= (qz < oshape[0]) & (qy < oshape[1]) & (qx < oshape[2])
                    cand.append((((b[ok] * oshape[0] + qz[ok]) * oshape[1] + qy[ok]) * oshape[2] + qx[ok]))
        okeys = np.unique(np.concatenate(cand)) if cand else np.zeros((0,), np.int64)
    ob, oz, oy, ox = _sp_decode(okeys, oshape)
    out = np.zeros((okeys.shape[0], cout), _f)
    pairs = 0
    n = keys.shape[0]
    for kz in range(kd):
        for ky in range(kh):
            for kx in range(kw):
                iz = oz * stride[0] - padding[0] + kz
                iy = oy * stride[1] - padding[1] + ky
                ix = ox * stride[2] - padding[2] + kx
                ok = (iz >= 0) & (iz < d) & (iy >= 0) & (iy < h) & (ix >= 0) & (ix < w)
                rows = np.nonzero(ok)[0]
                if rows.size == 0 or n == 0:
                    continue
                want = ((ob[rows] * d + iz[rows]) * h + iy[rows]) * w + ix[rows]
                pos = np.searchsorted(keys, want)
                hit = (pos < n) & (keys[np.minimum(pos, n - 1)] == want)
                rows, src = rows[hit], pos[hit]
                if rows.size == 0:
                    continue
                pairs += int(rows.size)
                out[rows] += feats[src] @ weight[kz, ky, kx]  # an output row occurs at most once per offset
    if bias is not None:
        out += np.asarray(bias, _f)
    return out, okeys, oshape, pairs


def sparse_to_dense_numpy(feats, keys, batch, shape):
    """to_dense + transpose([0, 4, 1, 2, 3]) + reshape [N, C * D, H, W] (sparse_resnet.py:202-205)."""
    d, h, w = (int(v) for v in shape)
    c = feats.shape[1]
    out = np.zeros((batch, c, d, h, w), _f)
    b, z, y, x = _sp_decode(keys, shape)
    out[b, :, z, y, x] = feats
    return out.reshape(batch, c * d, h, w)


def sparse_encoder_numpy(net, feats, coords, batch, trace=None):
    """SparseResNet3D.forward (sparse_resnet.py:184-206) / SparseNet3D.forward (sparsenet.py:140-182) for the module
    mirrors in paddle3d_amd/sparse.py, on sorted key sets (no dense grid).  Rows of `coords` with a negative batch
    index are padding.  Returns the [B, C * D, H, W] map; `trace` (a dict) receives, per convolution module name,
    dict(keys, shape, feats, pairs) with feats AFTER the BatchNorm / ReLU / residual add that follow the
    convolution in the reference's Sequential (the point where the device's fused kernel writes its output)."""
    import torch.nn as nn

    coords = np.asarray(coords)
    live = coords[:, 0] >= 0
    keys, order = sparse_keys_numpy(coords[live], net.sparse_shape)
    x = np.ascontiguousarray(np.asarray(feats, _f)[live][order])
    shape = tuple(net.sparse_shape)
    names = {id(m): n for n, m in net.named_modules()}

    def fold(bn):
        sc = (bn.weight / (bn.running_var + bn.eps).sqrt()).detach().cpu().numpy().astype(_f)
        sh = (bn.bias.detach().cpu().numpy().astype(_f) - bn.running_mean.detach().cpu().numpy().astype(_f) * sc)
        return sc, sh

    def conv(m, x, keys, shape):
        wt = m.weight.detach().cpu().numpy()
        b = None if m.bias is None else m.bias.detach().cpu().numpy()
        return sparse_conv_numpy(x, keys, batch, shape, wt, m.stride, m.padding, m.subm, b)

    def note(m, x, keys, shape, pairs):
        if trace is not None:
            trace[names[id(m)]] = dict(keys=keys, shape=shape, feats=x.copy(), pairs=pairs)

    def run(mod, x, keys, shape):
        if isinstance(mod, nn.Sequential):
            mods = list(mod)
            i = 0
            while i < len(mods):
                m = mods[i]
                if hasattr(m, "subm"):  # a convolution, then the BatchNorm / ReLU that follow it
                    x, keys, shape, pairs = conv(m, x, keys, shape)
                    i += 1
                    if i < len(mods) and isinstance(mods[i], nn.BatchNorm1d):
                        sc, sh = fold(mods[i])
                        x = x * sc + sh
                        i += 1
                        if i < len(mods) and isinstance(mods[i], nn.ReLU):
                            x = np.maximum(x, 0)
                            i += 1
                    note(m, x, keys, shape, pairs)
                else:
                    x, keys, shape = run(m, x, keys, shape)
                    i += 1
            return x, keys, shape
        if hasattr(mod, "conv1") and hasattr(mod, "bn2"):  # SparseBasicBlock, sparse_resnet.py:92-111
            o, _, _, p1 = conv(mod.conv1, x, keys, shape)
            sc, sh = fold(mod.bn1)
            o = np.maximum(o * sc + sh, 0)
            note(mod.conv1, o, keys, shape, p1)
            o, _, _, p2 = conv(mod.conv2, o, keys, shape)
            sc, sh = fold(mod.bn2)
            o = np.maximum(o * sc + sh + x, 0)
            note(mod.conv2, o, keys, shape, p2)
            return o, keys, shape
        raise TypeError(f"sparse_encoder_numpy: unexpected module {type(mod).__name__}")

    for stage in (net.conv_input, net.conv1, net.conv2, net.conv3, net.conv4, net.extra_conv):
        x, keys, shape = run(stage, x, keys, shape)
    return sparse_to_dense_numpy(x, keys, batch, shape)


# ---------------------------------------------------------------------------------------------------------
# NumPy restatements of the reference's Python glue around the ops (pinned to the reference's own Python by
# tests/golden/python_layers.npz, made by tests/golden/make_python_golden.py through the paddle shim).
# ---------------------------------------------------------------------------------------------------------
def rotate_nms_pcdet_numpy(boxes, scores, thresh, pre_max_size=None, post_max_size=None, kind="port"):
    """layer_libs.py:210-249: column reorder [0,1,2,4,3,5,-1], heading -> -theta - pi/2 (fp32), stable descending
    argsort, top pre_max_size, rotated NMS (nms_gpu = IoU bits + host sweep), map back, cap post_max_size."""
    b = np.asarray(boxes, np.float32)[:, [0, 1, 2, 4, 3, 5, -1]].copy()
    b[:, -1] = -b[:, -1] - np.float32(np.pi / 2)
    order = np.argsort(-np.asarray(scores, np.float32), kind="stable")
    if pre_max_size is not None:
        order = order[:pre_max_size]
    keep = nms(np.ascontiguousarray(b[order]), float(thresh), kind=kind)
    sel = order[keep]
    return sel if post_max_size is None else sel[:post_max_size]


def create_frustum_numpy(depth_cfg, input_size, downsample):
    """bevdet_transformer.py:126-140 -> [D, H, W, 3] (u, v, depth)."""
    h_in, w_in = input_size
    hf, wf = h_in // downsample, w_in // downsample
    d = np.arange(*depth_cfg, dtype=np.float32)
    x = np.linspace(0, w_in - 1, wf, dtype=np.float32)
    y = np.linspace(0, h_in - 1, hf, dtype=np.float32)
    fr = np.empty((len(d), hf, wf, 3), np.float32)
    fr[..., 0], fr[..., 1], fr[..., 2] = x[None, None, :], y[None, :, None], d[:, None, None]
    return fr


def get_lidar_coor_numpy(frustum, rots, trans, cam2imgs, post_rots, post_trans, bda):
    """bevdet_transformer.py:142-192 in float32."""
    f32 = np.float32
    B, N = trans.shape[:2]
    pts = frustum[None, None].astype(f32) - post_trans.reshape(B, N, 1, 1, 1, 3).astype(f32)
    pts = np.einsum("bnij,bndhwj->bndhwi", np.linalg.inv(post_rots.astype(f32)).astype(f32), pts).astype(f32)
    pts = np.concatenate([pts[..., :2] * pts[..., 2:3], pts[..., 2:3]], -1).astype(f32)
    comb = np.matmul(rots.astype(f32), np.linalg.inv(cam2imgs.astype(f32)).astype(f32)).astype(f32)
    pts = np.einsum("bnij,bndhwj->bndhwi", comb, pts).astype(f32) + trans.reshape(B, N, 1, 1, 1, 3).astype(f32)
    return np.einsum("bij,bndhwj->bndhwi", bda.astype(f32), pts).astype(f32)


def voxel_pooling_prepare_v2_numpy(coor, grid_lower_bound, grid_interval, grid_size):
    """bevdet_transformer.py:230-274: (ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths)."""
    B, N, D, H, W, _ = coor.shape
    num = B * N * D * H * W
    ranks_depth = np.arange(num, dtype=np.int64)
    ranks_feat = np.broadcast_to(np.arange(num // D, dtype=np.int64).reshape(B, N, 1, H, W), (B, N, D, H, W)).reshape(-1)
    c = ((coor.astype(np.float32) - np.asarray(grid_lower_bound, np.float32)) /
         np.asarray(grid_interval, np.float32)).astype(np.float32)
    c = np.trunc(c).astype(np.int64).reshape(num, 3)  # cast('int64') truncates toward zero
    batch_idx = np.repeat(np.arange(B, dtype=np.int64), num // B)
    gs = np.asarray(grid_size, np.float32)
    kept = (c[:, 0] >= 0) & (c[:, 0] < gs[0]) & (c[:, 1] >= 0) & (c[:, 1] < gs[1]) & (c[:, 2] >= 0) & (c[:, 2] < gs[2])
    c, ranks_depth, ranks_feat, batch_idx = c[kept], ranks_depth[kept], ranks_feat[kept], batch_idx[kept]
    g = gs.astype(np.int64)
    ranks_bev = batch_idx * (g[2] * g[1] * g[0]) + c[:, 2] * (g[1] * g[0]) + c[:, 1] * g[0] + c[:, 0]
    order = np.argsort(ranks_bev, kind="stable")
    ranks_bev, ranks_depth, ranks_feat = ranks_bev[order], ranks_depth[order], ranks_feat[order]
    head = np.ones(len(ranks_bev), bool)
    head[1:] = ranks_bev[1:] != ranks_bev[:-1]
    starts = np.nonzero(head)[0]
    lengths = np.diff(np.append(starts, len(ranks_bev)))
    i32 = np.int32
    return ranks_bev.astype(i32), ranks_depth.astype(i32), ranks_feat.astype(i32), starts.astype(i32), lengths.astype(i32)


# ---------------------------------------------------------------------------------------------------------
# PointPillars SSD head path (anchors, anchor mask, box decoding, per-frame post-processing), float32 NumPy;
# pinned to the reference's own Python by tests/golden/python_ssd.npz (tests/golden/make_ssd_golden.py).
# ---------------------------------------------------------------------------------------------------------
def ssd_anchors_numpy(point_cloud_range, voxel_size, anchor_configs, output_stride_factor=2):
    """anchors_generator.py:44-101 + :123-156 -> (anchors [A, 7] f32 in (y, x, config, rotation) order with
    columns (x, y, z, w, l, h, r), anchors_bv [A, 4] int64 pillar-index boxes (xmin, ymin, xmax, ymax),
    (feature_h, feature_w), grid (nx, ny))."""
    f32 = np.float32
    pr, vs = np.asarray(point_cloud_range, f32), np.asarray(voxel_size, f32)
    grid = np.round((pr[3:6] - pr[:3]) / vs).astype(np.int64)
    fw, fh = int(grid[0] // output_stride_factor), int(grid[1] // output_stride_factor)
    per_cfg = []
    for cfg in anchor_configs:
        xs_, ys_, zs_ = [f32(v) for v in cfg["anchor_strides"]]
        xo, yo, zo = [f32(v) for v in cfg["anchor_offsets"]]
        xc = np.arange(fw, dtype=f32) * xs_ + xo
        yc = np.arange(fh, dtype=f32) * ys_ + yo
        zc = np.arange(1, dtype=f32) * zs_ + zo
        rot = np.asarray(cfg["rotations"], f32)
        sizes = np.asarray(cfg["sizes"], f32).reshape(-1, 3)
        a = np.empty((1, fh, fw, sizes.shape[0], len(rot), 7), f32)   # [z, y, x, size, rotation, 7]
        a[..., 0], a[..., 1], a[..., 2] = xc[None, None, :, None, None], yc[None, :, None, None, None], zc[0]
        a[..., 3:6] = sizes[None, None, None, :, None, :]
        a[..., 6] = rot[None, None, None, None, :]
        per_cfg.append(a.reshape(1, fh, fw, -1, 7))
    anchors = np.concatenate(per_cfg, axis=-2).reshape(-1, 7)
    # rbbox2d_to_circumscribed (:158-176) on (x, y, w, l, r)
    r = anchors[:, 6]
    r = np.abs(r - np.floor(r / f32(np.pi) + f32(0.5)) * f32(np.pi))
    lying = r > f32(np.pi / 4)
    cx, cy = anchors[:, 0], anchors[:, 1]
    dx = np.where(lying, anchors[:, 4], anchors[:, 3])
    dy = np.where(lying, anchors[:, 3], anchors[:, 4])
    bv = np.stack([cx - dx / f32(2), cy - dy / f32(2), cx + dx / f32(2), cy + dy / f32(2)], 1).astype(f32)
    out = np.empty_like(bv)
    out[:, 0] = np.maximum(np.floor((bv[:, 0] - pr[0]) / vs[0]), 0)
    out[:, 1] = np.maximum(np.floor((bv[:, 1] - pr[1]) / vs[1]), 0)
    out[:, 2] = np.minimum(np.floor((bv[:, 2] - pr[0]) / vs[0]), f32(grid[0] - 1))
    out[:, 3] = np.minimum(np.floor((bv[:, 3] - pr[1]) / vs[1]), f32(grid[1] - 1))
    return anchors, out.astype(np.int64), (fh, fw), (int(grid[0]), int(grid[1]))


def ssd_anchor_mask_numpy(coords_zyx, anchors_bv, grid_xy, area_threshold=1.0):
    """anchors_generator.py:103-121, :191-210: occupancy map [ny, nx] of the frame's pillars, summed along both
    axes, four corner look-ups per anchor (the reference's corners as they are: no -1 on the lower side)."""
    nx, ny = grid_xy
    m = np.zeros((ny, nx), np.float32)
    np.add.at(m, (coords_zyx[:, 1], coords_zyx[:, 2]), np.float32(1))
    m = np.cumsum(np.cumsum(m, 0, dtype=np.float32), 1, dtype=np.float32)
    bv = anchors_bv
    area = m[bv[:, 3], bv[:, 2]] - m[bv[:, 3], bv[:, 0]] - m[bv[:, 1], bv[:, 2]] + m[bv[:, 1], bv[:, 0]]
    return area > np.float32(area_threshold)


def ssd_box_decode_numpy(encodings, anchors):
    """pointpillars_coder.py:126-148 (second_box_decode_paddle), float32."""
    f32 = np.float32
    e, a = np.asarray(encodings, f32), np.asarray(anchors, f32)
    xa, ya, za, wa, la, ha, ra = [a[..., k] for k in range(7)]
    xt, yt, zt, wt, lt, ht, rt = [e[..., k] for k in range(7)]
    diag = np.sqrt(la * la + wa * wa).astype(f32)
    out = np.stack([xt * diag + xa, yt * diag + ya, zt * ha + za, np.exp(wt).astype(f32) * wa,
                    np.exp(lt).astype(f32) * la, np.exp(ht).astype(f32) * ha, rt + ra], -1)
    return out.astype(f32)


def ssd_post_process_frame_numpy(box_preds, cls_preds, dir_preds, anchors, anchors_mask, score_threshold,
                                 center_limit_range, nms_pre_max_size, nms_post_max_size, nms_iou_threshold,
                                 kind="port", encode_background_as_zeros=True):
    """pointpillars_head.py:86-196 for one frame: decode, anchors_mask, sigmoid / max / argmax, direction argmax,
    score (>=) and centre-range filter, heading flip by the direction bit, bottom -> object centre, rotate_nms_pcdet,
    back to the bottom centre.  Returns (boxes [K, 7], scores [K], labels [K] int64); the reference's `_box_empty`
    row (zeros, -1, -1) when nothing survives."""
    f32 = np.float32
    empty = (np.zeros((1, 7), f32), -np.ones(1, f32), -np.ones(1, np.int64))
    if not anchors_mask.any():
        return empty
    box = ssd_box_decode_numpy(box_preds, anchors)[anchors_mask]
    cls = np.asarray(cls_preds, f32)[anchors_mask]
    if not encode_background_as_zeros:
        cls = cls[..., 1:]  # head.py:147-148: the background logit is dropped before the sigmoid
    conf = (f32(1) / (f32(1) + np.exp(-cls).astype(f32))).astype(f32)
    scores, labels = conf.max(-1), conf.argmax(-1).astype(np.int64)
    kept = scores >= f32(score_threshold)
    if center_limit_range is not None:
        lim = np.asarray(center_limit_range, f32)
        kept &= (box[:, :3] >= lim[:3]).all(1) & (box[:, :3] <= lim[3:]).all(1)
    if not kept.any():
        return empty
    box, scores, labels = box[kept].copy(), scores[kept], labels[kept]
    if dir_preds is not None:
        dl = np.asarray(dir_preds, f32)[anchors_mask][kept].argmax(-1).astype(bool)
        box[:, 6] += np.where((box[:, 6] > 0) ^ dl, f32(np.pi), f32(0))
    box[:, 2] = box[:, 2] + box[:, 5] * f32(0.5)
    sel = rotate_nms_pcdet_numpy(box, scores, nms_iou_threshold, nms_pre_max_size, nms_post_max_size, kind=kind)
    box = box[sel]
    box[:, 2] = box[:, 2] - box[:, 5] * f32(0.5)
    return box, scores[sel], labels[sel]
