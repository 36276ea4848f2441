"""Size-independent properties of the ops at the full benchmark sizes (no oracle involved): what still pins
the HIP path when the CPU oracle would take too long (SURVEY.md section 8c / task brief section 3)."""
import numpy as np
import pytest
import torch

from paddle3d_amd import synth

pytestmark = pytest.mark.gpu


def test_scatter_full_size_conservation():
    """C3 size (30k pillars, 64 ch, 512x512, batch 4): every pillar lands once, nothing else is touched."""
    from paddle3d_amd.ops import pointpillars_scatter as ps

    g = torch.Generator(device="cuda").manual_seed(0)
    b, m, c = 4, 30000, 64
    feats = torch.randn(b * m, c, device="cuda", generator=g) + 3.0  # strictly non-zero rows
    cells = torch.stack([torch.randperm(512 * 512, device="cuda", generator=g)[:m] for _ in range(b)])
    co = torch.zeros(b * m, 4, dtype=torch.int32, device="cuda")
    co[:, 0] = torch.arange(b, device="cuda").repeat_interleave(m).int()
    co[:, 2] = (cells.reshape(-1) // 512).int()
    co[:, 3] = (cells.reshape(-1) % 512).int()
    out = ps.pointpillars_scatter(feats, co, b, 512, 512)
    assert out.shape == (b, c, 512, 512)
    occ = (out != 0).any(1)
    assert int(occ.sum()) == b * m
    # sum over the canvas == sum over the features (fp64 accumulate), and gathering back is the identity
    assert torch.allclose(out.double().sum(), feats.double().sum(), rtol=1e-12)
    back = out.permute(0, 2, 3, 1)[co[:, 0].long(), co[:, 2].long(), co[:, 3].long()]
    assert torch.equal(back, feats)
    # padding rows (batch -1) are ignored
    co2 = co.clone()
    co2[::2, 0] = -1
    out2 = ps.pointpillars_scatter(feats, co2, b, 512, 512)
    assert int((out2 != 0).any(1).sum()) == b * m // 2


def test_nms_properties_full_size():
    """N = 1000 / 4096: keep list ascending, unique; NMS of the kept boxes keeps all of them (idempotence);
    threshold 1.0 keeps everything; every suppressed box overlaps an earlier kept one."""
    from paddle3d_amd.ops import iou3d_nms

    for n, seed in ((1000, 11), (4096, 12)):
        boxes, _ = synth.nms_boxes(seed, n=n)
        bt = torch.from_numpy(boxes).cuda()
        keep, num = iou3d_nms.nms_gpu(bt, 0.2)
        k = keep[: int(num[0])].long()
        assert (k[1:] > k[:-1]).all() and k[0] == 0
        keep2, num2 = iou3d_nms.nms_gpu(bt[k.cuda()], 0.2)
        assert int(num2[0]) == len(k) and torch.equal(keep2[: len(k)].long(), torch.arange(len(k)))
        keep3, num3 = iou3d_nms.nms_gpu(bt, 1.0)
        assert int(num3[0]) == n
        iou = iou3d_nms.boxes_iou_bev_gpu(bt, bt[k.cuda()]).cpu()
        dead = torch.ones(n, dtype=torch.bool)
        dead[k] = False
        # a suppressed box i has IoU > thr with some kept box that precedes it
        prec = k.unsqueeze(0) < torch.arange(n).unsqueeze(1)
        assert ((iou > 0.2) & prec)[dead].any(1).all()
        # no kept box is suppressed by an earlier kept box
        assert not ((iou > 0.2) & prec)[k].any()


def test_postprocess_properties_full_size():
    """C3 head maps (6 tasks, 128x128): per task scores descending, labels inside the task's class range,
    at most nms_post_max_size rows, every score above the threshold, boxes decode inside the grid."""
    from paddle3d_amd.ops import centerpoint_postprocess as cp

    tasks = synth.center_head_outputs(21, n_peaks=400)
    lists = {k: [torch.from_numpy(t[k]).cuda() for t in tasks] for k in ("hm", "reg", "height", "dim", "vel", "rot")}
    offs = [0, 1, 3, 5, 6, 8]
    ncls = [1, 2, 2, 1, 2, 2]
    b, s, l = cp.centerpoint_postprocess(lists["hm"], lists["reg"], lists["height"], lists["dim"], lists["vel"],
                                         lists["rot"], [0.2, 0.2], [-51.2, -51.2],
                                         [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], offs * 6, 4, 0.1, 0.2, 1000, 83, True)
    s, l, b = s.cpu().numpy(), l.cpu().numpy(), b.cpu().numpy()
    assert len(s) <= 6 * 83 and (s > 0.1).all()
    start = 0
    for t in range(6):
        in_task = (l >= offs[t]) & (l < offs[t] + ncls[t])
        idx = np.nonzero(in_task)[0]
        assert len(idx) <= 83
        if len(idx):
            assert idx[0] == start and (np.diff(idx) == 1).all()  # tasks are concatenated in order
            assert (np.diff(s[idx]) <= 0).all()
            start = idx[-1] + 1
    assert (np.abs(b[:, :2]) <= 51.2 + 0.8).all() and (b[:, 3:6] > 0).all()
    assert (np.abs(b[:, 8]) <= np.pi + 1e-6).all()


def test_sparse_conv_linearity_full_grid():
    """1440x1440x41 grid, ~100k voxels: conv(a*x + y) == a*conv(x) + conv(y) on the same index set, and a
    regular conv's output set equals the dilation of the input set."""
    from paddle3d_amd.ops import sparse_conv3d as sp

    rng = np.random.default_rng(2)
    shape = (41, 1440, 1440)
    n = 100_000
    lin = rng.choice(shape[0] * shape[1] * shape[2], n, replace=False)
    z, r = np.divmod(lin, shape[1] * shape[2])
    y, x = np.divmod(r, shape[2])
    coords = torch.from_numpy(np.stack([np.zeros(n), z, y, x], 1).astype(np.int32)).cuda()
    f1 = torch.randn(n, 16, device="cuda")
    f2 = torch.randn(n, 16, device="cuda")
    w = torch.randn(3, 3, 3, 16, 32, device="cuda") / 20
    for subm, stride, pad in ((True, 1, 1), (False, 2, 1)):
        idx = sp.indices(coords, 1, shape, 3, stride, pad, subm)
        o1, o2 = sp.features(f1, idx, w), sp.features(f2, idx, w)
        o12 = sp.features(2.5 * f1 + f2, idx, w)
        assert (o12 - (2.5 * o1 + o2)).abs().max().item() < 1e-3
        if not subm:
            oc = idx.out_coords.cpu().numpy().astype(np.int64)
            want = set()
            for dz in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        q = np.stack([z + 1 - (dz + 1), y + 1 - (dy + 1), x + 1 - (dx + 1)], 1)
                        ok = (q % 2 == 0).all(1) & (q >= 0).all(1)
                        q = q[ok] // 2
                        ok2 = (q[:, 0] < idx.out_shape[0]) & (q[:, 1] < idx.out_shape[1]) & (q[:, 2] < idx.out_shape[2])
                        q = q[ok2]
                        want.update(((q[:, 0] * idx.out_shape[1] + q[:, 1]) * idx.out_shape[2] + q[:, 2]).tolist())
            got = ((oc[:, 1] * idx.out_shape[1] + oc[:, 2]) * idx.out_shape[2] + oc[:, 3]).tolist()
            assert len(got) == len(set(got)) == len(want) and set(got) == want


def test_pfn_permutation_invariance_full_size():
    """30k pillars: permuting the points inside a pillar does not change its feature (max over points, mean
    over points), and permuting pillars permutes the rows."""
    from paddle3d_amd.ops import voxel_encoder as ve
    from paddle3d_amd.ops import voxelize

    pts = torch.from_numpy(synth.nuscenes_sweep(33)).cuda()
    vox, co, npv, nv = voxelize.hard_voxelize(pts, list(synth.NUSC_PILLAR), list(synth.NUSC_RANGE), 20, 30000)
    n = int(nv.item())
    vox, co, npv = vox[:n], co[:n], npv[:n]
    c4 = torch.cat([torch.zeros(n, 1, dtype=torch.int32, device="cuda"), co], 1)
    g = torch.Generator(device="cuda").manual_seed(1)
    w1 = torch.randn(10, 32, device="cuda", generator=g) / 3
    w2 = torch.randn(64, 64, device="cuda", generator=g) / 8
    s1, b1 = torch.rand(32, device="cuda", generator=g) + 0.5, torch.randn(32, device="cuda", generator=g) * 0.1
    s2, b2 = torch.rand(64, device="cuda", generator=g) + 0.5, torch.randn(64, device="cuda", generator=g) * 0.1
    args = (0.2, 0.2, -51.1, -51.1, w1, s1, b1, w2, s2, b2)
    base = ve.pillar_feature_net(vox, npv, c4, *args)
    # reverse the valid points of every pillar
    k = torch.arange(20, device="cuda").unsqueeze(0)
    src = torch.where(k < npv.unsqueeze(1), npv.unsqueeze(1) - 1 - k, k).long()
    rev = torch.gather(vox, 1, src.unsqueeze(-1).expand(-1, -1, 5))
    out = ve.pillar_feature_net(rev.contiguous(), npv, c4, *args)
    assert (out - base).abs().max().item() < 1e-4
    perm = torch.randperm(n, device="cuda", generator=g)
    out_p = ve.pillar_feature_net(vox[perm].contiguous(), npv[perm].contiguous(), c4[perm].contiguous(), *args)
    assert torch.equal(out_p, base[perm])


def test_lds_atomic_lane_order():
    """The hardware behaviour the default path's bit-exactness rests on, asserted explicitly (it is documented
    nowhere): one wave-wide returning LDS add serves lanes that hit one word in ascending lane order, and a wave's
    LDS instructions run in program order.  4.2 M adds in four address patterns (random over 1024 words, four hot
    words, one word, a strided set), a lane in eleven skipping -- each add must return the sequential count."""
    import ctypes as C

    from paddle3d_amd._lib import lib
    from paddle3d_amd.ops._common import check, ptr, stream_ptr

    blocks, waves, rounds, table = 512, 8, 16, 1024
    rng = np.random.default_rng(1)
    n_w = blocks * waves
    mode = (np.arange(n_w) % 4)[:, None, None]
    a = np.where(mode == 0, rng.integers(0, table, (n_w, rounds, 64)),
                 np.where(mode == 1, rng.integers(0, 4, (n_w, rounds, 64)),
                          np.where(mode == 2, 7, (rng.integers(0, 37, (n_w, rounds, 64)) * 27) % table))).astype(np.uint32)
    a[rng.random(a.shape) < 1 / 11] = 0xFFFFFFFF
    dev = torch.device("cuda", 0)
    da = torch.from_numpy(a.view(np.int32)).to(dev)
    dold = torch.empty_like(da)
    check(lib().pd3_selfcheck_lds_atomic_order(ptr(da), ptr(dold), blocks, waves, rounds, table, stream_ptr(dev)),
          "selfcheck_lds_atomic_order")
    old = dold.cpu().numpy().view(np.uint32).reshape(n_w, rounds * 64)
    flat = a.reshape(n_w, rounds * 64)
    want = np.full_like(flat, 0xFFFFFFFF)
    # sequential count per wave: rank of each element among the earlier equal addresses
    for w in range(0, n_w, 64):   # vectorised over waves, sequential over the 1024 adds of a wave
        blk = flat[w:w + 64]
        cnt = np.zeros((blk.shape[0], table), np.uint32)
        rows = np.arange(blk.shape[0])
        for i in range(blk.shape[1]):
            ai = blk[:, i]
            live = ai != 0xFFFFFFFF
            idx = np.where(live, ai, 0)
            want[w:w + 64, i] = np.where(live, cnt[rows, idx], 0xFFFFFFFF)
            cnt[rows[live], idx[live]] += 1
    bad = int((old != want).sum())
    assert bad == 0, f"{bad} of {old.size} returning LDS adds out of lane order"


@pytest.mark.parametrize("n", [1, 63, 1000, 16384, 358196])
def test_stable_argsort_matches_torch(n):
    """pd3_stable_argsort (radix sort) against torch.argsort(stable=True): descending fp32 scores with many ties,
    negative values and infinities; ascending int32 ranks with runs of equal keys."""
    from paddle3d_amd.ops.sort import stable_argsort

    g = torch.Generator(device="cuda").manual_seed(n)
    s = torch.randn(n, device="cuda", generator=g)
    s = torch.round(s * 8) / 8            # ties
    if n > 10:
        s[3], s[5], s[7] = float("inf"), float("-inf"), 0.0
    got = stable_argsort(s, descending=True)
    want = torch.argsort(s, descending=True, stable=True)
    assert torch.equal(got, want)
    r = torch.randint(0, max(2, n // 7), (n,), device="cuda", generator=g, dtype=torch.int32)
    got = stable_argsort(r, descending=False, max_key=int(max(2, n // 7)))
    want = torch.argsort(r.long(), stable=True)
    assert torch.equal(got, want)
    assert torch.equal(stable_argsort(r.long()), want)
