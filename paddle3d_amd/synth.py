"""Seeded synthetic inputs shaped like the reference's datasets (there is no dataset in the container).

Shapes follow SURVEY.md section 8(d):
  * nuScenes 10-sweep LiDAR frame: float32 [300000, 5] = (x, y, z, intensity, dt), what
    ``LoadPointCloud`` produces (reference paddle3d/transforms/reader.py:91-168).
  * KITTI frame: float32 [16384, 4], camera-FOV cropped.
  * CenterHead outputs for ``centerpoint_postprocess`` (reference center_head.py:212-220).
  * box sets for ``iou3d_nms``.
  * frustum index sets for ``bev_pool_v2`` (reference bevdet_transformer.py:230-274).
Everything is NumPy on the host; callers move it to the device.
"""
from __future__ import annotations

import numpy as np

NUSC_RANGE = (-51.2, -51.2, -5.0, 51.2, 51.2, 3.0)
NUSC_PILLAR = (0.2, 0.2, 8.0)
NUSC_VOXEL = (0.075, 0.075, 0.2)
NUSC_VOXEL_RANGE = (-54.0, -54.0, -5.0, 54.0, 54.0, 3.0)
KITTI_RANGE = (0.0, -39.68, -3.0, 69.12, 39.68, 1.0)
KITTI_PILLAR = (0.16, 0.16, 4.0)


def nuscenes_sweep(seed: int, n_points: int = 300_000, dims: int = 5, sweeps: int = 10,
                   shuffle: bool = False, oob_frac: float = 0.03) -> np.ndarray:
    """One multi-sweep frame of a 32-beam spinning LiDAR over a ground plane plus box obstacles.

    Points come in FIRING ORDER, like a real HDL-32E packet stream and like ``LoadPointCloud``'s
    concatenation of sweeps (reader.py:126-164): sweep after sweep, inside a sweep azimuth step after
    azimuth step, inside a step beam 0..31.  ``shuffle=True`` permutes the frame (the ``ShufflePoint``
    training transform) -- the worst case for every gather on the path."""
    rng = np.random.default_rng(seed)
    per = n_points // sweeps
    beams = np.deg2rad(np.linspace(-30.0, 10.0, 32)).astype(np.float64)
    out = np.empty((n_points, 5), np.float32)
    sensor_h = 1.84
    # a few dozen obstacles (cars / walls) that returns cluster on
    n_obs = 60
    obs_r = rng.uniform(4.0, 55.0, n_obs)
    obs_az = rng.uniform(0, 2 * np.pi, n_obs)
    obs_w = rng.uniform(0.02, 0.12, n_obs)  # angular half-width
    for s in range(sweeps):
        lo, hi = s * per, (s + 1) * per if s < sweeps - 1 else n_points
        m = hi - lo
        step = np.arange(m) // 32
        az = (step + rng.uniform(-0.3, 0.3, m)) * (2 * np.pi / max(1, (m + 31) // 32)) + rng.uniform(0, 2 * np.pi)
        el = beams[np.arange(m) % 32] + rng.normal(0, 0.0003, m)
        # range of the ground hit for downward beams; free-space returns otherwise
        r_ground = np.where(el < -0.01, sensor_h / np.maximum(np.tan(-el), 1e-3), 1e9)
        r_free = 1.0 + 69.0 * rng.random(m) ** 2.5
        r = np.minimum(r_ground, r_free)
        # obstacle hits: a beam whose azimuth falls in an obstacle's window stops at the obstacle
        k = rng.integers(0, n_obs, m)
        d_az = np.abs(((az - obs_az[k]) + np.pi) % (2 * np.pi) - np.pi)
        hit = (d_az < obs_w[k]) & (obs_r[k] < r)
        r = np.where(hit, obs_r[k] + rng.normal(0, 0.05, m), r)
        r = np.clip(r, 1.0, 70.0)
        x = r * np.cos(el) * np.cos(az)
        y = r * np.cos(el) * np.sin(az)
        z = r * np.sin(el)  # ground sits at z ~= -sensor_h in the lidar frame
        # ego motion between sweeps smears the older sweeps a little
        x = x + 0.02 * s
        out[lo:hi, 0] = x
        out[lo:hi, 1] = y
        out[lo:hi, 2] = z
        out[lo:hi, 3] = rng.uniform(0, 255, m)
        out[lo:hi, 4] = s * 0.05
    # deliberately out-of-range points (~3 %) and a few exactly on cell / range boundaries
    n_oob = int(oob_frac * n_points)
    idx = rng.choice(n_points, n_oob, replace=False)
    out[idx, 0] = rng.uniform(52.0, 80.0, n_oob) * rng.choice([-1.0, 1.0], n_oob)
    edge = rng.choice(n_points, 64, replace=False)
    out[edge[:16], 0] = np.float32(-51.2)
    out[edge[16:32], 0] = np.float32(51.2)          # == max: must be dropped (coord == grid)
    out[edge[32:48], 1] = np.float32(0.2) * rng.integers(-200, 200, 16).astype(np.float32)
    out[edge[48:], 2] = np.float32(3.0)              # == z max: dropped
    if shuffle:
        rng.shuffle(out, axis=0)
    return np.ascontiguousarray(out[:, :dims])


def kitti_frame(seed: int, n_points: int = 16_384) -> np.ndarray:
    rng = np.random.default_rng(seed)
    az = rng.uniform(-np.pi / 4, np.pi / 4, n_points)
    beams = np.deg2rad(np.linspace(-24.8, 2.0, 64))
    el = beams[rng.integers(0, 64, n_points)]
    r_ground = np.where(el < -0.01, 1.73 / np.maximum(np.tan(-el), 1e-3), 1e9)
    r = np.minimum(r_ground, 2.0 + 75.0 * rng.random(n_points) ** 1.5)
    pts = np.empty((n_points, 4), np.float32)
    pts[:, 0] = r * np.cos(el) * np.cos(az)
    pts[:, 1] = r * np.cos(el) * np.sin(az)
    pts[:, 2] = r * np.sin(el)
    pts[:, 3] = rng.random(n_points)
    return pts


def center_head_outputs(seed: int, feat_h: int = 128, feat_w: int = 128,
                        num_classes=(1, 2, 2, 1, 2, 2), n_peaks: int = 200):
    """Per-task CenterHead maps (hm, reg, height, dim, vel, rot) as float32 [1, c, H, W]."""
    rng = np.random.default_rng(seed)
    tasks = []
    yy, xx = np.mgrid[0:feat_h, 0:feat_w]
    for ncls in num_classes:
        hm = rng.normal(-2.19 - 2.0, 0.6, (1, ncls, feat_h, feat_w))
        for _ in range(n_peaks):
            c = rng.integers(0, ncls)
            py, px = rng.integers(0, feat_h), rng.integers(0, feat_w)
            amp = rng.uniform(2.0, 8.0)
            sig = rng.uniform(0.6, 1.6)
            hm[0, c] += amp * np.exp(-((yy - py) ** 2 + (xx - px) ** 2) / (2 * sig * sig))
        t = dict(
            hm=hm.astype(np.float32),
            reg=rng.uniform(0.0, 1.0, (1, 2, feat_h, feat_w)).astype(np.float32),
            height=rng.uniform(-3.0, 1.0, (1, 1, feat_h, feat_w)).astype(np.float32),
            dim=rng.normal(0.6, 0.4, (1, 3, feat_h, feat_w)).astype(np.float32),
            vel=rng.normal(0.0, 2.0, (1, 2, feat_h, feat_w)).astype(np.float32),
            rot=rng.normal(0.0, 1.0, (1, 2, feat_h, feat_w)).astype(np.float32),
        )
        tasks.append(t)
    return tasks


def nms_boxes(seed: int, n: int = 1000, extent: float = 40.0, clusters: int = 0):
    """Boxes [n, 7] = (x, y, z, dx, dy, dz, heading) with heavy overlap, plus descending scores."""
    rng = np.random.default_rng(seed)
    clusters = clusters or max(1, n // 6)
    centres = rng.uniform(-extent, extent, (clusters, 2))
    which = rng.integers(0, clusters, n)
    xy = centres[which] + rng.normal(0, 0.9, (n, 2))
    sizes = np.array([[4.6, 1.95, 1.7], [6.9, 2.5, 2.8], [0.7, 0.7, 1.8], [2.1, 0.8, 1.5]])
    dims = sizes[rng.integers(0, len(sizes), n)] * rng.uniform(0.85, 1.15, (n, 3))
    boxes = np.concatenate(
        [xy, rng.uniform(-2, 0, (n, 1)), dims, rng.uniform(-np.pi, np.pi, (n, 1))], axis=1)
    scores = np.sort(rng.random(n))[::-1].copy()
    return boxes.astype(np.float32), scores.astype(np.float32)


def bev_pool_inputs(seed: int, n_cam: int = 6, depth_bins: int = 118, fh: int = 16, fw: int = 44,
                    channels: int = 80, bev: int = 128, keep: float = 0.7):
    """Index sets of ``voxel_pooling_prepare_v2`` (reference bevdet_transformer.py:230-274).

    A frustum point (cam, d, h, w) lands in a BEV cell; points are sorted by BEV rank and run-length
    encoded into intervals.  Geometry is synthetic (radial fan per camera) but the index structure
    (sorted ranks_bev, interval starts / lengths) is exactly the op's contract.
    """
    rng = np.random.default_rng(seed)
    n_pts = n_cam * depth_bins * fh * fw
    cam, d, h, w = np.unravel_index(np.arange(n_pts), (n_cam, depth_bins, fh, fw))
    yaw = cam * (2 * np.pi / n_cam) + (w / fw - 0.5) * (70.0 / 180.0 * np.pi)
    rr = 1.0 + d * 0.5
    x = rr * np.cos(yaw) + rng.normal(0, 0.05, n_pts)
    y = rr * np.sin(yaw) + rng.normal(0, 0.05, n_pts)
    gx = np.floor((x + 51.2) / 0.8).astype(np.int64)
    gy = np.floor((y + 51.2) / 0.8).astype(np.int64)
    kept = (gx >= 0) & (gx < bev) & (gy >= 0) & (gy < bev) & (rng.random(n_pts) < keep)
    ranks_depth = np.arange(n_pts, dtype=np.int64)[kept]
    ranks_feat = (ranks_depth // (depth_bins * fh * fw)) * (fh * fw) + ranks_depth % (fh * fw)
    ranks_bev = (gy * bev + gx)[kept]
    order = np.argsort(ranks_bev, kind="stable")
    ranks_bev, ranks_depth, ranks_feat = ranks_bev[order], ranks_depth[order], ranks_feat[order]
    start_flag = np.ones(len(ranks_bev), bool)
    start_flag[1:] = ranks_bev[1:] != ranks_bev[:-1]
    starts = np.nonzero(start_flag)[0]
    lengths = np.diff(np.append(starts, len(ranks_bev)))
    depth = rng.random((n_cam, depth_bins, fh, fw)).astype(np.float32)
    depth /= depth.sum(1, keepdims=True)
    feat = rng.normal(0, 1, (n_cam, fh, fw, channels)).astype(np.float32)
    return dict(depth=depth, feat=feat, ranks_depth=ranks_depth.astype(np.int32),
                ranks_feat=ranks_feat.astype(np.int32), ranks_bev=ranks_bev.astype(np.int32),
                interval_starts=starts.astype(np.int32), interval_lengths=lengths.astype(np.int32),
                bev_feat_shape=(1, bev, bev, channels))


def camera_rig(seed: int, n_cam: int = 6, input_size=(256, 704), batch: int = 1):
    """Calibration of a nuScenes-like ring of cameras in the form BEVDet's view transformer consumes
    (reference bevdet_transformer.py:142-192 get_lidar_coor): rots [B,N,3,3] / trans [B,N,3] camera -> ego,
    cam2imgs [B,N,3,3] intrinsics of the 1600 x 900 sensor, post_rots / post_trans the image-space resize + crop
    down to `input_size`, bda [B,3,3] the BEV augmentation (identity at test time)."""
    rng = np.random.default_rng(seed)
    h_in, w_in = input_size
    r0 = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])  # camera (x right, y down, z fwd) -> ego
    rots, trans, k, prot, ptran = [], [], [], [], []
    resize = w_in / 1600.0
    crop_h = 900.0 * resize - h_in
    for _ in range(batch):
        for c in range(n_cam):
            yaw = 2 * np.pi * c / n_cam + rng.normal(0, 0.02)
            pitch = rng.normal(0, 0.01)
            rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
            ry = np.array([[np.cos(pitch), 0, np.sin(pitch)], [0, 1, 0], [-np.sin(pitch), 0, np.cos(pitch)]])
            rots.append(rz @ ry @ r0)
            trans.append([1.5 * np.cos(yaw) + rng.normal(0, 0.05), 1.5 * np.sin(yaw) + rng.normal(0, 0.05), 1.5])
            f = 1266.0 + rng.normal(0, 5.0)
            k.append([[f, 0, 816.0 + rng.normal(0, 3.0)], [0, f, 491.0 + rng.normal(0, 3.0)], [0, 0, 1]])
            prot.append(np.diag([resize, resize, 1.0]))
            ptran.append([0.0, -crop_h, 0.0])
    sh = (batch, n_cam)
    f32 = np.float32
    return dict(rots=np.asarray(rots, f32).reshape(*sh, 3, 3), trans=np.asarray(trans, f32).reshape(*sh, 3),
                cam2imgs=np.asarray(k, f32).reshape(*sh, 3, 3), post_rots=np.asarray(prot, f32).reshape(*sh, 3, 3),
                post_trans=np.asarray(ptran, f32).reshape(*sh, 3), bda=np.tile(np.eye(3, dtype=f32), (batch, 1, 1)))


def lss_camera_rig(seed: int, n_cam: int = 6, batch: int = 1):
    """Calibration in the form BEVFusion's camera stream consumes (reference bevf_faster_rcnn.py:171-185): per view
    rots = inverse(lidar2img)[:3, :3] and trans = inverse(lidar2img)[:3, 3] of the full-size 900 x 1600 image, i.e.
    R_cam->lidar @ K^-1 and t_cam->lidar of the same nuScenes-like ring as `camera_rig`."""
    rig = camera_rig(seed, n_cam, (900, 1600), batch)
    kinv = np.linalg.inv(rig["cam2imgs"].astype(np.float64))
    rots = (rig["rots"].astype(np.float64) @ kinv).astype(np.float32)
    return dict(rots=rots, trans=rig["trans"].copy())


def lss_camera_features(seed: int, n_views: int, depth_bins: int, fh: int, fw: int, channels: int):
    """Outputs of `CamEncode.get_depth_feat` (cam_stream_lss.py:160-167) before the lift: depth [views, D, fH, fW] (a
    softmax over D) and feat [views, fH, fW, C] channels-last, float32."""
    rng = np.random.default_rng(seed)
    logit = rng.normal(0, 1.5, (n_views, depth_bins, fh, fw)).astype(np.float32)
    logit -= logit.max(1, keepdims=True)
    e = np.exp(logit)
    depth = (e / e.sum(1, keepdims=True)).astype(np.float32)
    feat = rng.normal(0, 1, (n_views, fh, fw, channels)).astype(np.float32)
    return depth, feat


def trained_like_batchnorm(model, seed: int = 0):
    """Measurement aid for RANDOM-INIT weights: BatchNorm statistics and affine parameters like a trained net's (with
    the constructor's identity statistics the activations shrink layer by layer, every head logit lands in a band 0.01
    wide, and a calibrated gain then amplifies rounding noise into score differences).  Seeded, in place; also the
    biases of sparse convolutions that have one."""
    import torch

    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            elif getattr(m, "bias", None) is not None and hasattr(m, "subm"):
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    if hasattr(model, "invalidate"):
        model.invalidate()
    return model


def trained_like_heads(model, points, fraction: float = 0.01, top_fraction: float = 0.001, top_score: float = 0.35,
                       remove_dc: bool = False):
    """Measurement aid for RANDOM-INIT weights (bench.py's mAP proxy, tests/test_model_gpu.py): gives every class of
    every task of a CenterPoint head a heat map that behaves like a trained one's, so that all tasks emit detections
    and every class is scored.  Plain random-init heads put all scores of a class into a band 0.003 wide, where the
    top-K cut and the NMS order are thousands of near-ties; one common gain and bias (30 / -3, rounds 3-4) left three
    of the six nuScenes tasks under the threshold, and with a bias per class alone the class with the heavier tail
    still took all 83 post-NMS places of its task.  So the last heat-map convolution of each task gets a gain AND a
    bias PER CLASS that put two quantiles of the class's map over the cells of `points` (a [B, N, D] device tensor,
    two frames are enough) at fixed scores: `fraction` of the cells score above the model's score threshold and
    `top_fraction` of them above `top_score`.

    remove_dc (round 6, an analysis option, off by default): a random 3x3 convolution over ReLU features answers
    mostly to the features' MEAN (the calibrated CenterPoint-Voxel maps reach |logit| 450 at the zero-padded border);
    with it every tap's weight vector is first made orthogonal to the mean first-stage feature vector (measured through
    the head itself with one-hot probe weights).  tools/prof/amp_voxel_twins.py used it to show that this offset is NOT
    what separates the AMP dense graph's detections from the fp32 graph's on that model (profiles/r06_amp_voxel_twins.txt).

    In place; the caller copies the state dict to a CPU twin afterwards."""
    import math

    import torch

    def logit(p):
        return math.log(p / (1.0 - p))

    thr = float(model.test_cfg["score_threshold"])
    with torch.no_grad():
        x = model.dense_forward(model.extract_pillars(points, dense=False))
        convs = [task.hm[-1] for task in model.bbox_head.tasks]
        if remove_dc:
            saved = [c.weight.clone() for c in convs]
            hc = int(convs[0].weight.shape[1])
            mean_f = torch.zeros(len(convs), hc, dtype=torch.float64, device=saved[0].device)
            for c0 in range(hc):  # one-hot probes: class 0's map IS first-stage channel c0 of that task's hm branch
                for c in convs:
                    c.weight.zero_()
                    c.bias.zero_()
                    c.weight[0, c0, c.weight.shape[2] // 2, c.weight.shape[3] // 2] = 1.0
                model.invalidate()
                preds, _ = model.bbox_head(x)
                for t, p in enumerate(preds):
                    mean_f[t, c0] = p["hm"][:, 0].double().mean()
            for t, (c, w) in enumerate(zip(convs, saved)):
                m = mean_f[t].to(w.dtype)
                along = (w * m.view(1, -1, 1, 1)).sum(1, keepdim=True) / (m * m).sum().clamp(min=1e-30)
                c.weight.copy_(w - along * m.view(1, -1, 1, 1))
        for c in convs:
            c.bias.zero_()
        model.invalidate()
        preds, _ = model.bbox_head(x)
        for conv, p in zip(convs, preds):
            hm = p["hm"].float().transpose(0, 1).reshape(p["hm"].shape[1], -1)  # [classes, B * H * W] logits, bias 0
            n = hm.shape[1]
            lo = hm.kthvalue(max(1, int(round(n * (1.0 - fraction)))), dim=1).values
            hi = hm.kthvalue(max(1, int(round(n * (1.0 - top_fraction)))), dim=1).values
            gain = (logit(top_score) - logit(thr)) / (hi - lo).clamp(min=1e-12)
            conv.weight.mul_(gain.to(conv.weight.dtype).view(-1, 1, 1, 1))
            conv.bias.copy_(logit(thr) - gain * lo)
        model.invalidate()
    return model
