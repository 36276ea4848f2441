"""BEVFusion's camera -> BEV pooling (row L of SURVEY 8a) at config 5's REAL shape: 6 views x 41 depth bins x 112 x 200
feature pixels x 64 channels onto the 200 x 200 x 16 grid (configs/bevfusion/bevf_pp_2x8_1x_nusc.yaml:80-84,
cam_stream_lss.py:318-373) -- 5 510 400 frustum points, 1.41 GB of lifted features per scene.

Which statement holds the north star's 1e-3?  The map the reference DEFINES is the per-cell sum of the lifted features.
Its implementation takes differences of ONE fp32 running total over all kept points (the "cumsum trick", :111-121);
at this size that running total loses more than 1e-3 by itself (the error is printed below: 1.2e-3 on this input), so
the tolerance is held against the exact per-cell sums (float64, `oracle.lss_voxel_pooling_exact`), and the distance to
the cumsum restatement is bounded by that restatement's OWN measured error + 1e-3."""
import numpy as np
import pytest
import torch

from paddle3d_amd import synth

pytestmark = pytest.mark.gpu

ATOL = 1e-3  # north star: "within 1e-3 abs on fp32 BEV features"


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def c5():
    from paddle3d_amd.bevfusion import LiftSplatShoot

    lss = LiftSplatShoot()  # config 5's defaults
    assert (lss.D, lss.fH, lss.fW, lss.nx) == (41, 112, 200, [200, 200, 16])
    rig = synth.lss_camera_rig(0)
    geom = lss.get_geometry(_cuda(rig["rots"]), _cuda(rig["trans"]))
    depth, feat = synth.lss_camera_features(1, 6, lss.D, lss.fH, lss.fW, 64)
    return lss, rig, geom, depth, feat


def test_geometry_matches_the_reference_formula(c5):
    """get_geometry (:279-304): rots @ (u * d, v * d, d) + trans, against NumPy in float64."""
    lss, rig, geom, _, _ = c5
    fr = lss.frustum.reshape(-1, 3).astype(np.float64)
    p = np.stack([fr[:, 0] * fr[:, 2], fr[:, 1] * fr[:, 2], fr[:, 2]], -1)
    want = np.einsum("nij,pj->npi", rig["rots"][0].astype(np.float64), p) + rig["trans"][0][:, None, :]
    got = geom.cpu().numpy().reshape(6, -1, 3)
    assert np.abs(got - want).max() < 2e-4  # fp32 evaluation of coordinates up to ~60 m


def test_c5_fused_equals_materialised_and_exact_sums(oracle, c5):
    lss, _, geom, depth, feat = c5
    d, f = _cuda(depth), _cuda(feat)
    x = (d[:, :, :, :, None] * f[:, None]).reshape(1, 6, lss.D, lss.fH, lss.fW, 64)  # the reference's lifted tensor (:166)
    assert x.numel() * 4 > 1.4e9
    out_mat = lss.voxel_pooling(geom, x)
    out_fused = lss.voxel_pooling_fused(geom, d, f)
    assert out_mat.shape == out_fused.shape == (1, 64, 16, 200, 200)
    # the fused entry multiplies the same fp32 factors in the same order: identical bits, not merely within 1e-5
    assert torch.equal(out_mat, out_fused)
    # with the index sets prepared once (fixed calibration)
    lss.init_acceleration(geom)
    assert torch.equal(lss.voxel_pooling_fused(geom, d, f), out_fused)
    lss._prepared = None
    out = out_fused.cpu().numpy()
    g_cpu, x_cpu = geom.cpu().numpy(), x.cpu().numpy()
    del x, out_mat
    torch.cuda.empty_cache()
    exact = oracle.lss_voxel_pooling_exact(g_cpu, x_cpu, lss.dx, lss.bx, lss.nx)
    err = float(np.abs(out - exact).max())
    cum = oracle.lss_voxel_pooling_numpy(g_cpu, x_cpu, lss.dx, lss.bx, lss.nx)
    cum_err = float(np.abs(cum - exact).max())
    print(f"\nC5 camera->BEV pooling: device vs exact per-cell sums {err:.3e}; the reference's cumsum trick (NumPy "
          f"restatement) vs exact {cum_err:.3e}; device vs cumsum {float(np.abs(out - cum).max()):.3e}; "
          f"max |value| {float(np.abs(exact).max()):.3f}")
    assert err <= ATOL
    assert err < 1e-5  # what per-cell fp32 sums of <= 460 terms actually give
    assert np.abs(out - cum).max() <= cum_err + ATOL
    np.testing.assert_array_equal(out != 0, exact != 0)  # the same cells are written


@pytest.mark.parametrize("kind", ["port", "ref"])
def test_c5_fused_is_bev_pool_v2_bit_exact(oracle, c5, kind):
    """The fused entry IS pd3_bev_pool_v2 on split operands: its output equals the reference's own bev_pool_v2 kernel
    (compiled from /root/reference, run serially) on the same index sets, bit for bit, at 2.7 M kept points."""
    from paddle3d_amd.ops import bev_pool_v2 as bp

    if kind == "ref" and not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    lss, _, geom, depth, feat = c5
    cell, rd, rf, st, ln = bp.lss_pooling_prepare(geom, lss.dx, lss.bx, lss.nx)
    n_kept, n_int = int(cell.numel()), int(st.numel())
    assert n_kept > 2_000_000 and n_int > 200_000
    # index sets against NumPy: same kept points, cells sorted, points of a cell in index order
    g = ((geom.cpu().numpy() - (lss.bx - lss.dx / 2.0)) / lss.dx).astype(np.int64).reshape(-1, 3)
    kept = (g[:, 0] >= 0) & (g[:, 0] < 200) & (g[:, 1] >= 0) & (g[:, 1] < 200) & (g[:, 2] >= 0) & (g[:, 2] < 16)
    idx = np.nonzero(kept)[0]
    want_cell = (g[idx, 2] * 200 + g[idx, 0]) * 200 + g[idx, 1]
    order = np.argsort(want_cell, kind="stable")
    np.testing.assert_array_equal(cell.cpu().numpy(), want_cell[order])
    np.testing.assert_array_equal(rd.cpu().numpy(), idx[order])
    dhw, hw = lss.D * lss.fH * lss.fW, lss.fH * lss.fW
    np.testing.assert_array_equal(rf.cpu().numpy(), (idx[order] // dhw) * hw + idx[order] % hw)
    shape = (1, 16, 200 * 200, 64)
    out = bp.bev_pool_v2(_cuda(depth), _cuda(feat).reshape(-1, 64), rd, rf, cell, ln, st, shape).cpu().numpy()
    ref = oracle.bev_pool_v2(depth, feat.reshape(-1, 64), rd.cpu().numpy(), rf.cpu().numpy(), cell.cpu().numpy(),
                             ln.cpu().numpy(), st.cpu().numpy(), shape, kind=kind)
    np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_c5_batch_of_two_scenes(c5):
    """B = 2 with different calibrations: each scene's map equals its own single-scene map (cells carry the batch index)."""
    lss, rig, geom, depth, feat = c5
    rig2 = synth.lss_camera_rig(5)
    geom2 = lss.get_geometry(_cuda(rig2["rots"]), _cuda(rig2["trans"]))
    d2, f2 = synth.lss_camera_features(2, 6, lss.D, lss.fH, lss.fW, 64)
    one_a = lss.voxel_pooling_fused(geom, _cuda(depth), _cuda(feat))
    one_b = lss.voxel_pooling_fused(geom2, _cuda(d2), _cuda(f2))
    both = lss.voxel_pooling_fused(torch.cat([geom, geom2]), _cuda(np.concatenate([depth, d2])),
                                   _cuda(np.concatenate([feat, f2])))
    assert torch.equal(both[0:1], one_a) and torch.equal(both[1:2], one_b)
    # s2c (:386-390): [B, C, Z, X, Y] -> [B, C * Z, Y, X]
    bev = lss.s2c(both)
    assert bev.shape == (2, 64 * 16, 200, 200)
    assert torch.equal(bev[1, 5 * 16 + 3, 17, 40], both[1, 5, 3, 40, 17])
