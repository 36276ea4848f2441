"""pointpillars_scatter (exact), pillar_feature_net / voxel_mean (1e-3 abs) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from paddle3d_amd import synth

pytestmark = pytest.mark.gpu


def _pillars(oracle, seed, v=30000):
    pts = synth.nuscenes_sweep(seed)
    vox, co, npv, nv = oracle.hard_voxelize(pts, synth.NUSC_PILLAR, synth.NUSC_RANGE, 20, v)
    return vox[:nv], co[:nv], npv[:nv]


@pytest.mark.parametrize("batch", [1, 3])
def test_scatter_exact(oracle, batch):
    from paddle3d_amd.ops import pointpillars_scatter as ps

    rng = np.random.default_rng(0)
    feats, coords = [], []
    for b in range(batch):
        _, co, _ = _pillars(oracle, 30 + b, v=20000)
        f = rng.normal(size=(len(co), 64)).astype(np.float32)
        c4 = np.concatenate([np.full((len(co), 1), b, np.int32), co], 1)
        feats.append(f)
        coords.append(c4)
    f, c4 = np.concatenate(feats), np.concatenate(coords)
    out = ps.pointpillars_scatter(torch.from_numpy(f).cuda(), torch.from_numpy(c4).cuda(), batch, 512, 512)
    ref = oracle.pillar_scatter(f, c4, batch, 512, 512)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    np.testing.assert_array_equal(ref, oracle.pillar_scatter_numpy(f, c4, batch, 512, 512))


def test_scatter_odd_shapes(oracle):
    from paddle3d_amd.ops import pointpillars_scatter as ps

    rng = np.random.default_rng(1)
    ny, nx, c = 37, 41, 7  # plane and channels not multiples of 4 -> scalar path
    m = 300
    cells = rng.choice(ny * nx, m, replace=False)
    c4 = np.stack([np.zeros(m), np.zeros(m), cells // nx, cells % nx], 1).astype(np.int32)
    f = rng.normal(size=(m, c)).astype(np.float32)
    out = ps.pointpillars_scatter(torch.from_numpy(f).cuda(), torch.from_numpy(c4).cuda(), 1, ny, nx)
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.pillar_scatter(f, c4, 1, ny, nx))
    # empty input -> all-zero canvas
    out = ps.pointpillars_scatter(torch.zeros((0, 8)).cuda(), torch.zeros((0, 4), dtype=torch.int32).cuda(), 2,
                                  16, 16)
    assert not out.cpu().numpy().any()


def _pfn_params(rng, d, c1, c2, signed=False):
    def layer(i, o):
        gamma = rng.uniform(0.5, 1.5, o).astype(np.float32)
        if signed:  # trained BatchNorm scales can be negative or exactly zero: max over points must not assume a sign
            gamma *= rng.choice([-1.0, 1.0], o).astype(np.float32)
            gamma[:2] = 0.0
        return dict(weight=(rng.uniform(-1, 1, (i, o)) / np.sqrt(i)).astype(np.float32),
                    gamma=gamma, beta=rng.normal(0, 0.2, o).astype(np.float32),
                    mean=rng.normal(0, 0.2, o).astype(np.float32), var=rng.uniform(0.5, 1.5, o).astype(np.float32))

    ps = [layer(d + 5, c1)]
    if c2:
        ps.append(layer(2 * c1, c2))
    return ps


_PFN_FORMS = [(True, 0), (True, 1), (True, 2), (False, 0)]  # (two layers, kernel form: 1 per pillar, 2 packed)


@pytest.mark.parametrize("two_layers,path", _PFN_FORMS)
def test_pfn(oracle, two_layers, path):
    from paddle3d_amd.ops import voxel_encoder as ve

    rng = np.random.default_rng(2)
    vox, co, npv = _pillars(oracle, 40)
    c4 = np.concatenate([np.zeros((len(co), 1), np.int32), co], 1)
    c1, c2 = (32, 64) if two_layers else (64, 0)
    params = _pfn_params(rng, 5, c1, c2)
    ref = oracle.pfn_forward_torch(vox, npv, c4, params, synth.NUSC_PILLAR, synth.NUSC_RANGE)
    dev = torch.device("cuda")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    folded = []
    for p in params:
        s, sh = ve.fold_batchnorm(t(p["gamma"]), t(p["beta"]), t(p["mean"]), t(p["var"]), 1e-3)
        folded.append((t(p["weight"]), s, sh))
    vx, vy = synth.NUSC_PILLAR[0], synth.NUSC_PILLAR[1]
    args = [t(vox), t(npv), t(c4), vx, vy, vx / 2 + synth.NUSC_RANGE[0], vy / 2 + synth.NUSC_RANGE[1], *folded[0]]
    if two_layers:
        args += list(folded[1])
    out = ve.pillar_feature_net(*args, path=path).cpu().numpy()
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() < 1e-3, np.abs(out - ref).max()


@pytest.mark.parametrize("two_layers,path", _PFN_FORMS)
@pytest.mark.parametrize("p,d,m", [(32, 4, 500), (20, 5, 500), (7, 4, 500), (20, 5, 3), (32, 5, 1001)])
def test_pfn_random_pillars(oracle, two_layers, path, p, d, m):
    """Random pillars with every fill level 0..P (empty, partial, full) through the fast PFN kernels; BatchNorm
    scales of both signs and zero; pillar counts that are not a multiple of the packed form's chunk of 8."""
    from paddle3d_amd.ops import voxel_encoder as ve

    rng = np.random.default_rng(p * 10 + d)
    npv = rng.integers(0, p + 1, m).astype(np.int32)
    npv[:3] = [0, p, 1]
    if m > 100:
        npv[40:72] = p      # chunks of full pillars: 8 x P rows, no padded row
        npv[80:100] = 0     # a chunk without any row
        npv[100:108] = 1
    vox = rng.uniform(-3, 3, (m, p, d)).astype(np.float32)
    vox *= (np.arange(p)[None, :, None] < npv[:, None, None])
    c4 = np.concatenate([np.zeros((m, 2), np.int32), rng.integers(0, 400, (m, 2)).astype(np.int32)], 1)
    c1, c2 = (32, 64) if two_layers else (64, 0)
    params = _pfn_params(rng, d, c1, c2, signed=True)
    keep = npv > 0
    ref = oracle.pfn_forward_torch(vox[keep], npv[keep], c4[keep], params, synth.NUSC_PILLAR, synth.NUSC_RANGE)
    dev = torch.device("cuda")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    folded = []
    for q in params:
        s, sh = ve.fold_batchnorm(t(q["gamma"]), t(q["beta"]), t(q["mean"]), t(q["var"]), 1e-3)
        folded.append((t(q["weight"]), s, sh))
    vx, vy = synth.NUSC_PILLAR[0], synth.NUSC_PILLAR[1]
    args = [t(vox), t(npv), t(c4), vx, vy, vx / 2 + synth.NUSC_RANGE[0], vy / 2 + synth.NUSC_RANGE[1], *folded[0]]
    if two_layers:
        args += list(folded[1])
    out = ve.pillar_feature_net(*args, path=path).cpu().numpy()
    assert np.all(out[~keep] == 0)
    assert np.abs(out[keep] - ref).max() < 1e-3, np.abs(out[keep] - ref).max()


def test_voxel_mean(oracle):
    from paddle3d_amd.ops import voxel_encoder as ve

    pts = synth.nuscenes_sweep(50)
    vox, co, npv, nv = oracle.hard_voxelize(pts, synth.NUSC_VOXEL, synth.NUSC_VOXEL_RANGE, 10, 120000)
    out = ve.voxel_mean(torch.from_numpy(vox[:nv]).cuda(), torch.from_numpy(npv[:nv]).cuda()).cpu().numpy()
    ref = oracle.voxel_mean(vox[:nv], npv[:nv])
    assert np.abs(out - ref).max() < 1e-4


@pytest.mark.parametrize("path", [0, 1, 2])
def test_hard_vfe(oracle, path):
    """BEVFusion LiDAR-stream encoder (C5: 0.25 m pillars, P=64, D=4) vs the torch-CPU restatement; per-pillar
    form (path 1) and packed form (path 2 = what path 0 picks since round 4)."""
    from paddle3d_amd.ops import voxel_encoder as ve

    rng = np.random.default_rng(4)
    vs, pr = (0.25, 0.25, 8.0), (-50.0, -50.0, -5.0, 50.0, 50.0, 3.0)
    pts = synth.nuscenes_sweep(41, dims=4)
    vox, co, npv, nv = oracle.hard_voxelize(pts, vs, pr, 64, 30000)
    vox, co, npv = vox[:nv], co[:nv], npv[:nv]
    c4 = np.concatenate([np.zeros((nv, 1), np.int32), co], 1)

    def layer(i, o):
        return dict(weight=(rng.uniform(-1, 1, (i, o)) / np.sqrt(i)).astype(np.float32),
                    gamma=rng.uniform(0.5, 1.5, o).astype(np.float32), beta=rng.normal(0, 0.2, o).astype(np.float32),
                    mean=rng.normal(0, 0.2, o).astype(np.float32), var=rng.uniform(0.5, 1.5, o).astype(np.float32))

    params = [layer(10, 64), layer(128, 64)]
    ref = oracle.hard_vfe_forward_torch(vox, npv, c4, params, vs, pr)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = []
    for p in params:
        s, sh = ve.fold_batchnorm(t(p["gamma"]), t(p["beta"]), t(p["mean"]), t(p["var"]), 1e-3)
        args += [t(p["weight"]), s, sh]
    out = ve.hard_vfe(t(vox), t(npv), t(c4), vs, pr, *args, path=path).cpu().numpy()
    assert out.shape == ref.shape == (nv, 64)
    assert np.abs(out - ref).max() < 1e-3, np.abs(out - ref).max()


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("p,m", [(64, 1003), (32, 517), (40, 64), (64, 3), (16, 203)])
def test_hard_vfe_fill_levels(oracle, path, p, m):
    """Random HardVFE pillars with every fill level 0..P (empty, one point, partial, full: up to four 16-row blocks
    per pillar), chunks made of full pillars only, an empty chunk, pillar counts that are not a multiple of the packed
    form's chunk of 4, BatchNorm scales of both signs and zero."""
    from paddle3d_amd.ops import voxel_encoder as ve

    rng = np.random.default_rng(p * 7 + m)
    vs, pr = (0.25, 0.25, 8.0), (-50.0, -50.0, -5.0, 50.0, 50.0, 3.0)
    npv = rng.integers(0, p + 1, m).astype(np.int32)
    npv[:3] = [0, p, 1]
    if m > 100:
        npv[40:56] = p      # chunks of full pillars
        npv[80:92] = 0      # chunks without any row
        npv[100:108] = 1
        npv[200:260] = rng.integers(1, 4, len(npv[200:260]))   # the common case: a few points per pillar
    vox = rng.uniform(-3, 3, (m, p, 4)).astype(np.float32)
    vox *= (np.arange(p)[None, :, None] < npv[:, None, None])
    c4 = np.concatenate([np.zeros((m, 1), np.int32), np.zeros((m, 1), np.int32),
                         rng.integers(0, 400, (m, 2)).astype(np.int32)], 1)

    def layer(i, o):
        g = rng.uniform(-1.5, 1.5, o).astype(np.float32)
        g[:3] = 0.0
        return dict(weight=(rng.uniform(-1, 1, (i, o)) / np.sqrt(i)).astype(np.float32), gamma=g,
                    beta=rng.normal(0, 0.2, o).astype(np.float32), mean=rng.normal(0, 0.2, o).astype(np.float32),
                    var=rng.uniform(0.5, 1.5, o).astype(np.float32))

    params = [layer(10, 64), layer(128, 64)]
    keep = npv > 0
    ref = oracle.hard_vfe_forward_torch(vox[keep], npv[keep], c4[keep], params, vs, pr)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = []
    for q in params:
        s, sh = ve.fold_batchnorm(t(q["gamma"]), t(q["beta"]), t(q["mean"]), t(q["var"]), 1e-3)
        args += [t(q["weight"]), s, sh]
    out = ve.hard_vfe(t(vox), t(npv), t(c4), vs, pr, *args, path=path).cpu().numpy()
    assert np.all(out[~keep] == 0)
    assert np.abs(out[keep] - ref).max() < 1e-3, np.abs(out[keep] - ref).max()


def test_merge_sweeps(oracle):
    """nuScenes 10-sweep merge (ego-point removal, fp64 rigid transform, time-lag column) vs the NumPy statement."""
    from paddle3d_amd.ops import sweeps as sw

    rng = np.random.default_rng(6)
    frames = [rng.uniform(-40, 40, (rng.integers(20000, 35000), 5)).astype(np.float32) for _ in range(10)]
    for f in frames:
        f[rng.choice(len(f), 500, replace=False), :2] = rng.uniform(-0.99, 0.99, (500, 2))  # ego returns
    mats = []
    for i in range(9):
        a = rng.uniform(-0.1, 0.1)
        m = np.eye(4)
        m[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        m[:3, 3] = rng.uniform(-2, 2, 3)
        mats.append(m)
    lags = [0.05 * (i + 1) for i in range(9)]
    ref = oracle.merge_sweeps_numpy(frames[0], frames[1:], mats, lags)
    out = sw.merge_sweeps(torch.from_numpy(frames[0]).cuda(), [torch.from_numpy(f).cuda() for f in frames[1:]],
                          mats, lags).cpu().numpy()
    assert out.shape == ref.shape and out.shape[1] == 6
    # bit for bit, 250 k points: the oracle is pinned to the reference's executed reader.py (tests/test_reader_golden.py)
    np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))
    # and the merged cloud feeds hard_voxelize unchanged
    from paddle3d_amd.ops import voxelize
    v = voxelize.hard_voxelize(torch.from_numpy(np.ascontiguousarray(out[:, :5])).cuda(), list(synth.NUSC_PILLAR),
                               list(synth.NUSC_RANGE), 20, 30000)
    assert int(v[3].item()) > 1000
