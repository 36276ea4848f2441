"""iou3d_nms, centerpoint_postprocess and bev_pool_v2 HIP paths vs the CPU oracle."""
import numpy as np
import pytest
import torch

from paddle3d_amd import synth

pytestmark = pytest.mark.gpu

CP_CFG = dict(voxel_size=[0.2, 0.2], point_cloud_range=[-51.2, -51.2],
              post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], down_ratio=4, score_threshold=0.1,
              nms_iou_threshold=0.2, nms_pre_max_size=1000, nms_post_max_size=83)
LABEL_OFFSETS = [0, 1, 3, 5, 6, 8]


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("n", [1, 64, 65, 300, 1000])
def test_pairwise_iou_overlap(oracle, n):
    from paddle3d_amd.ops import iou3d_nms

    a, _ = synth.nms_boxes(n, n=min(n, 200))
    b, _ = synth.nms_boxes(n + 1, n=min(n, 150) + 1)
    iou = iou3d_nms.boxes_iou_bev_gpu(_cuda(a), _cuda(b)).cpu().numpy()
    ov = iou3d_nms.boxes_overlap_bev_gpu(_cuda(a), _cuda(b)).cpu().numpy()
    r_iou, r_ov = oracle.boxes_iou_bev(a, b), oracle.boxes_overlap_bev(a, b)
    # the device evaluates cos / sin / atan2 with glibc's bits (csrc/libm_exact.hpp) and everything else in the
    # reference's fp32 operation order: the matrices are the reference's bit for bit
    np.testing.assert_array_equal(ov.view(np.uint32), r_ov.view(np.uint32))
    np.testing.assert_array_equal(iou.view(np.uint32), r_iou.view(np.uint32))
    if oracle.have_ref():  # and the reference's own iou3d_cpu.cpp compiled here says the same
        np.testing.assert_array_equal(iou.view(np.uint32), oracle.boxes_iou_bev(a, b, kind="ref").view(np.uint32))


@pytest.mark.parametrize("op", ["sinf", "cosf", "expf", "atanf", "atan2f"])
def test_device_libm_is_glibc(oracle, op):
    """sinf / cosf / expf / atanf / atan2f on the device return the host libm's bits: every exponent, both signs,
    specials, and a dense sample of the ranges the path uses (headings in +-2 pi, log-dims in +-5)."""
    from paddle3d_amd.ops import iou3d_nms

    rng = np.random.default_rng(7)
    bits = rng.integers(0, 1 << 32, 1 << 20, dtype=np.uint64).astype(np.uint32)
    special = np.array([0x0, 0x80000000, 0x1, 0x7f800000, 0xff800000, 0x7fc00000, 0x3f800000, 0xbf800000, 0x40490fdb,
                        0x3fc90fdb, 0x42b00000, 0xc2ce8ed0, 0x42f00000, 0x4c000000, 0x31000000], np.uint32)
    x = np.concatenate([bits.view(np.float32), special.view(np.float32),
                        rng.uniform(-6.3, 6.3, 1 << 20).astype(np.float32),
                        rng.uniform(-130.0, 130.0, 1 << 18).astype(np.float32)])
    y = np.roll(x, 12345) if op == "atan2f" else None
    code = ["sinf", "cosf", "expf", "atanf", "atan2f"].index(op)
    want = oracle.libm_eval(code, x, y)
    got = iou3d_nms.libm_eval(op, _cuda(x), _cuda(y) if y is not None else None).cpu().numpy()
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    np.testing.assert_array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])


@pytest.mark.parametrize("block", range(8))
def test_nms_keep_fuzz(oracle, block):
    """200 box sets x 3 thresholds at n = 1000 (plain and clustered: many pairs near the threshold): keep lists equal
    the reference's (iou3d_cpu.cpp + the host sweep of iou3d_nms.cpp compiled here when /root/reference was present
    at build time, else the port, which is held to it bit for bit on the CPU)."""
    from paddle3d_amd.ops import iou3d_nms

    kind = "ref" if oracle.have_ref() else "port"
    for seed in range(1000 + 25 * block, 1000 + 25 * (block + 1)):
        boxes, _ = synth.nms_boxes(seed, n=1000, clusters=(seed % 3) * 20)
        dev = _cuda(boxes)
        for thr in (0.1, 0.2, 0.5):
            keep, num = iou3d_nms.nms_gpu(dev, thr)
            np.testing.assert_array_equal(keep[: int(num[0])].numpy(), oracle.nms(boxes, thr, kind=kind),
                                          err_msg=f"seed {seed} thr {thr}")


@pytest.mark.parametrize("n,seed", [(1, 0), (64, 1), (65, 2), (129, 3), (1000, 4), (1000, 5), (2500, 6)])
@pytest.mark.parametrize("normal", [False, True])
def test_nms_keep_exact(oracle, n, seed, normal):
    from paddle3d_amd.ops import iou3d_nms

    boxes, _ = synth.nms_boxes(seed, n=n)
    fn = iou3d_nms.nms_normal_gpu if normal else iou3d_nms.nms_gpu
    for thr in (0.2, 0.5):
        keep, num = fn(_cuda(boxes), thr)
        assert keep.dtype == torch.int32 and not keep.is_cuda and num.shape == (1,)
        got = keep[: int(num[0])].numpy()
        want = oracle.nms(boxes, thr, normal=normal)
        np.testing.assert_array_equal(got, want)


def test_nms_empty():
    from paddle3d_amd.ops import iou3d_nms

    keep, num = iou3d_nms.nms_gpu(torch.zeros((0, 7)).cuda(), 0.5)
    assert int(num[0]) == 0


def _post(oracle, tasks, with_velocity=True, **over):
    from paddle3d_amd.ops import centerpoint_postprocess as cp

    cfg = dict(CP_CFG, **over)
    lists = {k: [_cuda(t[k]) for t in tasks] for k in ("hm", "reg", "height", "dim", "vel", "rot")}
    num_classes = LABEL_OFFSETS * len(tasks)  # the len(tasks)**2 list the reference caller builds
    full_sort = cfg.pop("full_sort", False)
    if full_sort:  # the reference's own selection (full stable sort), through the explicit `selection` argument
        bb, ss, ll, nn = cp.centerpoint_postprocess_device(
            lists["hm"], lists["reg"], lists["height"], lists["dim"], lists["vel"], lists["rot"], cfg["voxel_size"],
            cfg["point_cloud_range"], cfg["post_center_range"], num_classes, cfg["down_ratio"],
            cfg["score_threshold"], cfg["nms_iou_threshold"], cfg["nms_pre_max_size"], cfg["nms_post_max_size"],
            with_velocity, full_sort=True)
        k = int(nn.item())
        b, s, l = bb[0, :k], ss[0, :k], ll[0, :k]
    else:
        b, s, l = cp.centerpoint_postprocess(lists["hm"], lists["reg"], lists["height"], lists["dim"], lists["vel"],
                                             lists["rot"], cfg["voxel_size"], cfg["point_cloud_range"],
                                             cfg["post_center_range"], num_classes, cfg["down_ratio"],
                                             cfg["score_threshold"], cfg["nms_iou_threshold"],
                                             cfg["nms_pre_max_size"], cfg["nms_post_max_size"], with_velocity)
    rb, rs, rl, margins = oracle.centerpoint_postprocess(
        tasks, cfg["voxel_size"] + [8.0], cfg["point_cloud_range"] + [0.0] * 4, cfg["post_center_range"],
        LABEL_OFFSETS[: len(tasks)], cfg["down_ratio"], cfg["score_threshold"], cfg["nms_iou_threshold"],
        cfg["nms_pre_max_size"], cfg["nms_post_max_size"], with_velocity, return_margins=True)
    return (b.cpu().numpy(), s.cpu().numpy(), l.cpu().numpy()), (rb, rs, rl), margins


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("with_velocity", [True, False])
def test_centerpoint_postprocess(oracle, seed, with_velocity):
    tasks = synth.center_head_outputs(seed)
    (b, s, l), (rb, rs, rl), margins = _post(oracle, tasks, with_velocity)
    assert l.dtype == np.int64
    assert b.shape == rb.shape, (b.shape, rb.shape, margins)
    np.testing.assert_array_equal(l, rl)
    # sigmoid / exp / atan2 carry glibc's bits on the device (csrc/libm_exact.hpp): rows equal the port's bit for bit
    np.testing.assert_array_equal(s.view(np.uint32), rs.view(np.uint32))
    np.testing.assert_array_equal(b.view(np.uint32), rb.view(np.uint32))


@pytest.mark.parametrize("pre", [1025, 4096])
def test_centerpoint_postprocess_large_cap(oracle, pre):
    """nms_pre_max_size beyond 1024 (Waymo-style configurations use 4096) with a few hundred candidates per task:
    the sweep takes its in-LDS path for the small sets although the capacity is large."""
    tasks = synth.center_head_outputs(5)
    (b, s, l), (rb, rs, rl), margins = _post(oracle, tasks, nms_pre_max_size=pre, nms_post_max_size=83)
    assert b.shape == rb.shape and b.shape[0] > 20, (b.shape, rb.shape, margins)
    np.testing.assert_array_equal(l, rl)
    np.testing.assert_array_equal(s.view(np.uint32), rs.view(np.uint32))
    np.testing.assert_array_equal(b.view(np.uint32), rb.view(np.uint32))


def test_centerpoint_postprocess_crowded(oracle):
    """Boxes 20 m long on the 0.8 m grid: every pair of the 1000 candidates per task passes the circle test, far more
    than the pair pool of the rotated NMS holds (32 per box) -- its tiles fall back to the per-tile evaluation; 6 m
    boxes fill the pool part of the way through the call (pooled and per-tile tiles side by side); 1.4 m boxes fit
    it whole.  Keep lists equal the reference's in all three."""
    for dim_bias, hm_bias in ((3.0, 6.0), (1.1, 6.0), (0.3, 6.0)):
        tasks = synth.center_head_outputs(6, feat_h=64, feat_w=64, n_peaks=0)
        for t in tasks:
            t["hm"] += hm_bias
            t["dim"] += dim_bias
        (b, s, l), (rb, rs, rl), margins = _post(oracle, tasks, nms_pre_max_size=1000, nms_post_max_size=83)
        assert b.shape == rb.shape, (b.shape, rb.shape, margins)
        np.testing.assert_array_equal(l, rl)
        np.testing.assert_array_equal(s.view(np.uint32), rs.view(np.uint32))
        np.testing.assert_array_equal(b.view(np.uint32), rb.view(np.uint32))


def test_centerpoint_postprocess_edges(oracle):
    # a task with no candidate -> the reference's fake row (zeros, -1, 0); small pre/post caps
    tasks = synth.center_head_outputs(3, feat_h=32, feat_w=48, n_peaks=20)
    tasks[1]["hm"][:] = -20.0
    (b, s, l), (rb, rs, rl), _ = _post(oracle, tasks, nms_pre_max_size=50, nms_post_max_size=7)
    assert b.shape == rb.shape
    np.testing.assert_array_equal(l, rl)
    np.testing.assert_array_equal(s.view(np.uint32), rs.view(np.uint32))
    np.testing.assert_array_equal(b.view(np.uint32), rb.view(np.uint32))
    assert (s == -1).sum() == 1
    # everything above threshold: exercises the pre-NMS cap with ties-free random scores
    tasks = synth.center_head_outputs(4, feat_h=32, feat_w=32, n_peaks=0)
    for t in tasks:
        t["hm"] += 6.0
    (b, s, l), (rb, rs, rl), _ = _post(oracle, tasks, nms_pre_max_size=200, nms_post_max_size=83)
    assert b.shape == rb.shape
    np.testing.assert_array_equal(l, rl)
    np.testing.assert_array_equal(b.view(np.uint32), rb.view(np.uint32))


@pytest.mark.parametrize("seed", [0, 1])
def test_centerpoint_postprocess_voxel_config(oracle, seed):
    """CenterPoint-Voxel's head maps: 180 x 180, down_ratio 8, 0.075 m voxels, range -54 m (the map is larger than
    the in-LDS top-K selection takes, so this is the full-sort branch) vs the oracle."""
    tasks = synth.center_head_outputs(10 + seed, feat_h=180, feat_w=180, n_peaks=300)
    (b, s, l), (rb, rs, rl), margins = _post(oracle, tasks, voxel_size=[0.075, 0.075], point_cloud_range=[-54.0, -54.0],
                                             down_ratio=8)
    assert b.shape == rb.shape and b.shape[0] > 50, (b.shape, rb.shape, margins)
    np.testing.assert_array_equal(l, rl)
    # sigmoid / exp / atan2 carry glibc's bits on the device (csrc/libm_exact.hpp): rows equal the port's bit for bit
    np.testing.assert_array_equal(s.view(np.uint32), rs.view(np.uint32))
    np.testing.assert_array_equal(b.view(np.uint32), rb.view(np.uint32))


def test_postprocess_zero_pre_nms_cap(oracle):
    """nms_pre_max_size = 0: num_bboxes_for_nms is 0 (postprocess.cu:212-216) -> no rows for tasks with candidates."""
    from paddle3d_amd.ops import centerpoint_postprocess as cp

    tasks = synth.center_head_outputs(2, feat_h=32, feat_w=32, n_peaks=30)
    lists = {k: [_cuda(t[k]) for t in tasks] for k in ("hm", "reg", "height", "dim", "vel", "rot")}
    b, s, l = cp.centerpoint_postprocess(lists["hm"], lists["reg"], lists["height"], lists["dim"], lists["vel"],
                                         lists["rot"], CP_CFG["voxel_size"], CP_CFG["point_cloud_range"],
                                         CP_CFG["post_center_range"], LABEL_OFFSETS * len(tasks), CP_CFG["down_ratio"],
                                         CP_CFG["score_threshold"], CP_CFG["nms_iou_threshold"], 0, 83, True)
    assert b.shape[0] == 0 and s.shape[0] == 0 and l.shape[0] == 0


@pytest.mark.parametrize("pre", [37, 100, 1000])
def test_postprocess_topk_select_equals_full_sort(oracle, pre):
    """The LDS top-K selection (cut-off key + ties in cell order) gives exactly the full stable sort's result,
    also when many cells share one score: quantised heat maps put hundreds of ties at the cut-off."""
    tasks = synth.center_head_outputs(7, feat_h=128, feat_w=128, n_peaks=50)
    for t in tasks:
        t["hm"] = (np.round(t["hm"] * 2.0) / 2.0 + 3.0).astype(np.float32)  # few distinct values, all selected
    (b, s, l), (rb, rs, rl), _ = _post(oracle, tasks, nms_pre_max_size=pre, nms_post_max_size=83)
    (b2, s2, l2), _, _ = _post(oracle, tasks, nms_pre_max_size=pre, nms_post_max_size=83, full_sort=True)
    assert b.shape == b2.shape
    np.testing.assert_array_equal(l, l2)
    np.testing.assert_array_equal(s, s2)
    np.testing.assert_array_equal(b, b2)
    # and the oracle (stable descending sort, ties in cell order) agrees
    assert b.shape == rb.shape
    np.testing.assert_array_equal(l, rl)
    np.testing.assert_array_equal(b.view(np.uint32), rb.view(np.uint32))


@pytest.mark.parametrize("thr", [0.1, 0.0, -1.0, 0.5, 0.9999, 1.0])
def test_postprocess_extreme_head_values_and_thresholds(oracle, thr):
    """The one-kernel selection evaluates expf in two parts (polynomial on every argument, the special arguments
    |x| >= 88 / inf / NaN afterwards) and walks only the key bits a score above the threshold can have: heat maps and
    box sizes with huge, infinite and NaN entries, thresholds from below 0 to 1, against the full-sort kernels (which
    call the whole expf per cell): same rows, bit for bit."""
    tasks = synth.center_head_outputs(11, feat_h=64, feat_w=64, n_peaks=40)
    rng = np.random.default_rng(5)
    for t in tasks:
        hm = t["hm"].reshape(-1)
        idx = rng.choice(hm.size, 400, replace=False)
        hm[idx[:80]] = rng.uniform(88.0, 200.0, 80).astype(np.float32)     # exp(-x) underflows: score 1.0
        hm[idx[80:160]] = -rng.uniform(88.0, 200.0, 80).astype(np.float32)  # exp(-x) overflows: score 0.0
        hm[idx[160:200]] = np.float32(np.inf)
        hm[idx[200:240]] = np.float32(-np.inf)
        hm[idx[240:280]] = np.float32(np.nan)
        hm[idx[280:400]] = rng.uniform(-104.0, -86.0, 120).astype(np.float32)  # around expf's overflow threshold
        dm = t["dim"].reshape(-1)
        jdx = rng.choice(dm.size, 60, replace=False)
        dm[jdx[:20]] = np.float32(89.0)    # exp overflows to inf
        dm[jdx[20:40]] = np.float32(-104.0)  # exp underflows (denormal / zero)
        dm[jdx[40:]] = np.float32(-87.5)
    from paddle3d_amd.ops import centerpoint_postprocess as cp

    lists = {k: [_cuda(t[k]) for t in tasks] for k in ("hm", "reg", "height", "dim", "vel", "rot")}

    def run(full_sort):
        bb, ss, ll, nn = cp.centerpoint_postprocess_device(
            lists["hm"], lists["reg"], lists["height"], lists["dim"], lists["vel"], lists["rot"], CP_CFG["voxel_size"],
            CP_CFG["point_cloud_range"], CP_CFG["post_center_range"], LABEL_OFFSETS * len(tasks), CP_CFG["down_ratio"],
            thr, CP_CFG["nms_iou_threshold"], 300, 83, True, full_sort=full_sort)
        k = int(nn.item())
        return bb[0, :k].cpu().numpy(), ss[0, :k].cpu().numpy(), ll[0, :k].cpu().numpy()

    b, s, l = run(False)
    b2, s2, l2 = run(True)
    assert b.shape == b2.shape and b.shape[0] > 0
    np.testing.assert_array_equal(l, l2)
    np.testing.assert_array_equal(s.view(np.uint32), s2.view(np.uint32))
    np.testing.assert_array_equal(b.view(np.uint32), b2.view(np.uint32))
    if thr < 1.0:  # scores of exactly 1.0 exist (heat-map entries beyond +88) and lead every task's list
        assert s.max() == 1.0


def test_postprocess_batch_check():
    from paddle3d_amd.ops import centerpoint_postprocess as cp

    t = synth.center_head_outputs(0, feat_h=8, feat_w=8, num_classes=(1,))[0]
    two = {k: _cuda(np.concatenate([v, v], 0)) for k, v in t.items()}
    with pytest.raises(RuntimeError, match="batch size must be 1"):
        cp.centerpoint_postprocess([two["hm"]], [two["reg"]], [two["height"]], [two["dim"]], [two["vel"]],
                                   [two["rot"]], [0.2, 0.2], [-51.2, -51.2], CP_CFG["post_center_range"], [0], 4,
                                   0.1, 0.2, 1000, 83, True)


@pytest.mark.parametrize("seed", [0, 1])
def test_bev_pool_v2_exact(oracle, seed):
    from paddle3d_amd.ops import bev_pool_v2 as bp

    d = synth.bev_pool_inputs(seed, n_cam=2, depth_bins=30, fh=8, fw=22, channels=80, bev=64)
    args = [d[k] for k in ("depth", "feat", "ranks_depth", "ranks_feat", "ranks_bev", "interval_lengths",
                           "interval_starts")]
    out = bp.bev_pool_v2(*[_cuda(a) for a in args], d["bev_feat_shape"]).cpu().numpy()
    ref = oracle.bev_pool_v2(*args, d["bev_feat_shape"])
    np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))
    # backward op: intervals by ranks_feat
    order = np.argsort(d["ranks_feat"], kind="stable")
    rb, rd, rf = d["ranks_bev"][order], d["ranks_depth"][order], d["ranks_feat"][order]
    flag = np.ones(len(rf), bool)
    flag[1:] = rf[1:] != rf[:-1]
    starts = np.nonzero(flag)[0].astype(np.int32)
    lengths = np.diff(np.append(starts, len(rf))).astype(np.int32)
    g = np.random.default_rng(seed).normal(size=d["bev_feat_shape"]).astype(np.float32)
    dg, fg = bp.bev_pool_v2_bkwd(_cuda(g), _cuda(d["depth"]), _cuda(d["feat"]), _cuda(rd), _cuda(rf), _cuda(rb),
                                 _cuda(lengths), _cuda(starts))
    rdg, rfg = oracle.bev_pool_v2_bkwd(g, d["depth"], d["feat"], rd, rf, rb, lengths, starts)
    np.testing.assert_array_equal(dg.cpu().numpy().view(np.uint32), rdg.view(np.uint32))
    np.testing.assert_array_equal(fg.cpu().numpy().view(np.uint32), rfg.view(np.uint32))


@pytest.mark.parametrize("kind", ["port", "ref"])
def test_bev_pool_v2_full_size_exact(oracle, kind):
    """BEVDet4D-sized op (6 x 118 x 16 x 44 frustum points, C = 80, 128 x 128 BEV): forward and both gradients
    bit-exact against the oracle (ref = the reference's own kernels compiled from /root/reference, run serially)."""
    from paddle3d_amd.ops import bev_pool_v2 as bp

    if kind == "ref" and not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    d = synth.bev_pool_inputs(9)
    assert d["ranks_bev"].size > 300_000 and d["feat"].shape[-1] == 80
    names = ("ranks_depth", "ranks_feat", "ranks_bev", "interval_lengths", "interval_starts")
    out = bp.bev_pool_v2(_cuda(d["depth"]), _cuda(d["feat"]), *[_cuda(d[k]) for k in names], d["bev_feat_shape"])
    ref = oracle.bev_pool_v2(d["depth"], d["feat"], *[d[k] for k in names], d["bev_feat_shape"], kind=kind)
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # backward: points re-sorted and intervals re-cut by ranks_feat, as the reference's PyLayer does
    # (bevdet_transformer.py:52-79)
    order = np.argsort(d["ranks_feat"], kind="stable")
    rd, rf, rb = d["ranks_depth"][order], d["ranks_feat"][order], d["ranks_bev"][order]
    flag = np.ones(len(rf), bool)
    flag[1:] = rf[1:] != rf[:-1]
    starts = np.nonzero(flag)[0].astype(np.int32)
    lengths = np.diff(np.append(starts, len(rf))).astype(np.int32)
    g = np.random.default_rng(3).normal(size=d["bev_feat_shape"]).astype(np.float32)
    dg, fg = bp.bev_pool_v2_bkwd(_cuda(g), _cuda(d["depth"]), _cuda(d["feat"]), _cuda(rd), _cuda(rf), _cuda(rb),
                                 _cuda(lengths), _cuda(starts))
    rdg, rfg = oracle.bev_pool_v2_bkwd(g, d["depth"], d["feat"], rd, rf, rb, lengths, starts, kind=kind)
    np.testing.assert_array_equal(dg.cpu().numpy().view(np.uint32), rdg.view(np.uint32))
    np.testing.assert_array_equal(fg.cpu().numpy().view(np.uint32), rfg.view(np.uint32))


def test_bev_pool_v2_full_size_linearity():
    """BEVDet4D-sized op: linearity in feat (a size-independent property; no oracle involved)."""
    from paddle3d_amd.ops import bev_pool_v2 as bp

    d = synth.bev_pool_inputs(9)
    idx = [_cuda(d[k]) for k in ("ranks_depth", "ranks_feat", "ranks_bev", "interval_lengths", "interval_starts")]
    depth, f1 = _cuda(d["depth"]), _cuda(d["feat"])
    f2 = torch.randn_like(f1)
    o1 = bp.bev_pool_v2(depth, f1, *idx, d["bev_feat_shape"])
    o2 = bp.bev_pool_v2(depth, f2, *idx, d["bev_feat_shape"])
    o12 = bp.bev_pool_v2(depth, f1 + f2, *idx, d["bev_feat_shape"])
    assert (o12 - (o1 + o2)).abs().max().item() < 1e-4
    # cells without an interval stay zero
    touched = torch.zeros(o1.shape[1] * o1.shape[2], dtype=torch.bool, device="cuda")
    touched[idx[2].long()] = True
    assert not o1.view(-1, o1.shape[-1])[~touched].any()


def test_lss_voxel_pooling(oracle):
    """BEVFusion camera->BEV pooling through bev_pool vs the NumPy statement of the cumsum trick."""
    from paddle3d_amd.ops import bev_pool_v2 as bp

    rng = np.random.default_rng(8)
    B, N, D, H, W, C = 2, 3, 10, 6, 8, 16
    dx, bx, nx = np.array([0.5, 0.5, 20.0], np.float32), np.array([-9.75, -9.75, 0.0], np.float32), [40, 40, 1]
    geom = rng.uniform(-12, 12, (B, N, D, H, W, 3)).astype(np.float32)
    geom[..., 2] = rng.uniform(-9, 9, geom.shape[:-1])
    x = rng.normal(size=(B, N, D, H, W, C)).astype(np.float32)
    ref = oracle.lss_voxel_pooling_numpy(geom, x, dx, bx, nx)
    out = bp.lss_voxel_pooling(_cuda(geom), _cuda(x), dx, bx, nx).cpu().numpy()
    assert out.shape == ref.shape == (B, C, 1, 40, 40)
    # the reference's cumsum trick subtracts running totals (error grows with the prefix); ours sums per cell
    assert np.abs(out - ref).max() < 5e-3
    # exact statement: per-cell sums in float64
    exact = np.zeros((B, 1, 40, 40, C))
    g = ((geom - (bx - dx / 2.0)) / dx).astype(np.int64)
    for b in range(B):
        gb, xb = g[b].reshape(-1, 3), x[b].reshape(-1, C).astype(np.float64)
        ok = (gb[:, 0] >= 0) & (gb[:, 0] < 40) & (gb[:, 1] >= 0) & (gb[:, 1] < 40) & (gb[:, 2] >= 0) & (gb[:, 2] < 1)
        np.add.at(exact[b, 0], (gb[ok, 0], gb[ok, 1]), xb[ok])
    assert np.abs(out - exact.transpose(0, 4, 1, 2, 3)).max() < 1e-4
