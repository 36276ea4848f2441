"""The front half without the padded tensor (round 5): pd3_hard_voxelize_index leaves an INDEX of the points instead of
copies of them, pd3_pillar_feature_net_indexed reads the points through it.  Reference chain:
paddle3d/models/voxelizers/voxelize.py:39-58 -> paddle3d/models/voxel_encoders/pillar_encoder.py:156-210.
Everything here is bit-exact against the two full operators (which the other GPU tests pin to the oracle)."""
import numpy as np
import pytest
import torch

from paddle3d_amd import synth

pytestmark = pytest.mark.gpu


def _rows_from_index(points, span, plist, p):
    """The padded [B, V, P, D] tensor the operator would have written, rebuilt from the index on the host."""
    b, n, d = points.shape
    v = span.shape[1]
    out = np.zeros((b, v, p, d), np.float32)
    for f in range(b):
        lst = plist[f * n:(f + 1) * n]
        st, cnt = span[f, :, 0], span[f, :, 1]
        for k in range(p):
            live = np.nonzero(cnt > k)[0]
            out[f, live, k] = points[f, lst[st[live] + k]]
    return out


CONFIGS = [
    # name, voxel size, range, P, V, frames, points per frame
    ("c3_cap30k", synth.NUSC_PILLAR, synth.NUSC_RANGE, 20, 30000, 2, 300_000),
    ("c3_cap60k", synth.NUSC_PILLAR, synth.NUSC_RANGE, 20, 60000, 2, 300_000),
    ("c4_3d", synth.NUSC_VOXEL, [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], 10, 160000, 2, 300_000),
    ("c5_bevfusion", [0.25, 0.25, 8.0], [-50.0, -50.0, -5.0, 50.0, 50.0, 3.0], 64, 40000, 1, 200_000),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
@pytest.mark.parametrize("shuffle", [False, True], ids=["firing_order", "shuffled"])
def test_index_equals_the_operators_rows(cfg, shuffle):
    """(vox_span, point_list) name exactly the points the operator copies, in its order; coords, counts, num_voxels and
    the batched coors are the operator's; a ragged batch (num_points) included."""
    from paddle3d_amd.ops import voxelize

    _, vs, pr, p, v, frames, n = cfg
    rng = np.random.default_rng(7)
    pts = np.stack([synth.nuscenes_sweep(40 + i, n_points=n) for i in range(frames)])
    if shuffle:
        pts = np.stack([f[rng.permutation(n)] for f in pts])
    lens = torch.tensor([n - 1234 * i for i in range(frames)], dtype=torch.int32).cuda()
    t = torch.from_numpy(pts).cuda()
    for num_points in (None, lens):
        want = voxelize.hard_voxelize_batch(t, list(vs), list(pr), p, v, num_points, with_batch_coors=True)
        got = voxelize.hard_voxelize_index_batch(t, list(vs), list(pr), p, v, num_points)
        assert got is not None
        span, plist, coords, npv, nv, coors4 = got
        assert torch.equal(coords, want[1]) and torch.equal(npv, want[2]) and torch.equal(nv, want[3])
        assert torch.equal(coors4, want[4])
        assert torch.equal(span[..., 1], npv)
        rows = _rows_from_index(pts, span.cpu().numpy(), plist.cpu().numpy(), p)
        assert np.array_equal(rows.view(np.uint32), want[0].cpu().numpy().view(np.uint32))


def test_index_unsupported_grid_says_so():
    """A grid no wave form serves (373 M cells, above the 2^28 of the 3-D wave form) answers None (PD3_EUNSUPPORTED): the caller runs the operator."""
    from paddle3d_amd.ops import voxelize

    t = torch.from_numpy(synth.nuscenes_sweep(1, n_points=50_000)).cuda().unsqueeze(0)
    assert voxelize.hard_voxelize_index_batch(t, [0.05, 0.05, 0.1], [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], 5, 20000) is None


@pytest.mark.parametrize("d", [5, 4])
@pytest.mark.parametrize("cap", [30000, 60000])
def test_pfn_indexed_same_bytes_as_the_pair(d, cap):
    """pd3_hard_voxelize_index + pd3_pillar_feature_net_indexed against pd3_hard_voxelize + pd3_pillar_feature_net:
    the same [B * V, 64] bytes (BatchNorm scales of both signs, pillars of every fill level 0 .. 20, padding rows)."""
    from paddle3d_amd.ops import voxel_encoder as ve
    from paddle3d_amd.ops import voxelize

    g = torch.Generator(device="cuda").manual_seed(d * 1000 + cap)
    pts = np.stack([synth.nuscenes_sweep(60 + i) for i in range(3)])[:, :, :d].copy()
    t = torch.from_numpy(pts).cuda()
    vs, pr = list(synth.NUSC_PILLAR), list(synth.NUSC_RANGE)
    w1 = torch.randn(d + 5, 32, device="cuda", generator=g) * 0.3
    w2 = torch.randn(64, 64, device="cuda", generator=g) * 0.2
    s1, b1 = torch.randn(32, device="cuda", generator=g), torch.randn(32, device="cuda", generator=g) * 0.1
    s2, b2 = torch.randn(64, device="cuda", generator=g), torch.randn(64, device="cuda", generator=g) * 0.1
    s2[5] = 0.0
    vox, _, npv, nv, coors = voxelize.hard_voxelize_batch(t, vs, pr, 20, cap, with_batch_coors=True)
    b, v, p, _ = vox.shape
    args = (vs[0], vs[1], vs[0] / 2 + pr[0], vs[1] / 2 + pr[1], w1, s1, b1, w2, s2, b2)
    want = ve.pillar_feature_net(vox.view(b * v, p, d), npv.view(b * v), coors.view(b * v, 4), *args)
    span, plist, _, npv2, nv2, coors2 = voxelize.hard_voxelize_index_batch(t, vs, pr, 20, cap)
    got = ve.pillar_feature_net_indexed(t, span, plist, coors2.view(b * v, 4), 20, *args)
    assert got is not None and got.shape == want.shape
    assert int(npv.max()) == 20 and int((npv == 1).sum()) > 1000
    assert torch.equal(got, want)


def test_model_front_half_fused_equals_pair():
    """CenterPoint.extract_pillars runs the fused front (fuse_rows) and gives the same BEV features and detections as
    the pair of full operators; a list of ragged frames takes the same path."""
    from paddle3d_amd import centerpoint as cpm

    torch.manual_seed(2)
    model = cpm.centerpoint_pillars_nuscenes(max_num_voxels=(30000, 30000)).cuda().eval()
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
        for task in model.bbox_head.tasks:
            task.hm[-1].bias.fill_(-1.0)
    assert model.fuse_rows
    pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(80 + i) for i in range(3)])).cuda()
    ragged = [pts[0], pts[1][:250_000], pts[2][:123_457]]
    for inp in (pts, ragged):
        model.fuse_rows = True
        packed, lens = model._pack(inp)
        bev_f = model.extract_pillars(packed, lens)
        det_f = model.test_forward(inp)
        model.fuse_rows = False
        bev_p = model.extract_pillars(packed, lens)
        det_p = model.test_forward(inp)
        assert torch.equal(bev_f, bev_p)
        for a, c in zip(det_f, det_p):
            assert torch.equal(a["box3d_lidar"], c["box3d_lidar"]) and torch.equal(a["scores"], c["scores"])
    model.fuse_rows = True
