"""hard_voxelize HIP path vs the CPU oracle (bit-exact: voxels, coords, num_points, num_voxels)."""
import numpy as np
import pytest
import torch

from paddle3d_amd import synth

pytestmark = pytest.mark.gpu


PATH = 0  # pd3_hard_voxelize_path selector of the running parametrisation (0 automatic, 1 generic sort path, ...)


@pytest.fixture(params=["auto", "sort", "tiled", "gather", "wave", "wave_s1", "wave_s2", "wave_prio_split", "wave3d",
                        "wave_payload"], autouse=True)
def vox_path(request):
    """Every test runs on the automatic path choice, with the generic sort path forced, and with each form of the
    tiled path forced (2 = payload copied into a compact array, 3 = rows gathered through an index list, 5 = the
    wave form, 7 / 8 = the wave form with 8192- / 10240-point route tiles) -- explicit
    `path` argument of the C ABI, no process-wide switches.  A forced tiled form on a grid it does not take (the
    82.9 M-cell 0.075 m grid) must answer "unsupported configuration", which skips the parametrisation."""
    global PATH
    if request.param == "wave3d" and request.function.__name__ in (
            "test_batch_and_ragged", "test_properties_full_size", "test_batched_coors_output",
            "test_host_points_are_staged"):
        pytest.skip("pillar-grid test: the 3-D wave form takes grids above 2^20 cells (its own batch / ragged / "
                    "coors test is test_wave3d_batch_ragged_coors)")
    # 13 = the wave form with both round-4 measurement variants on: heavy waves of the group kernel at raised issue
    # priority, the batch as two half batches on two streams (the batch index of coors_batched continues across them)
    PATH = {"auto": 0, "sort": 1, "tiled": 2, "gather": 3, "wave": 5, "wave_s1": 7, "wave_s2": 8,
            "wave_prio_split": 13, "wave3d": 14,  # 14: the wave form for 3-D grids (hash tables per group)
            "wave_payload": 17}[request.param]    # 17: the payload carried through the route kernel (measurement form)
    yield request.param
    PATH = 0


def _unsupported_ok(fn, *a, **k):
    from paddle3d_amd._lib import Paddle3DAmdError

    try:
        return fn(*a, **k)
    except Paddle3DAmdError as e:
        if PATH >= 2 and "status -3" in str(e):
            pytest.skip("forced tiled form: grid beyond the tiled path (PD3_EUNSUPPORTED, as documented)")
        raise


def _run(points, voxel_size, pc_range, p, v):
    from paddle3d_amd.ops import voxelize

    t = torch.from_numpy(points).cuda()
    out = _unsupported_ok(voxelize.hard_voxelize, t, list(voxel_size), list(pc_range), p, v, path=PATH)
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in out]


def _check(oracle, points, voxel_size, pc_range, p, v, kind="port"):
    vox, co, npv, nv = _run(points, voxel_size, pc_range, p, v)
    rv, rc, rn, rnv = oracle.hard_voxelize(points, voxel_size, pc_range, p, v, kind)
    assert int(nv[0]) == rnv
    np.testing.assert_array_equal(co, rc)
    np.testing.assert_array_equal(npv, rn)
    # bit-exact payload (compare as uint32 so that -0.0 / NaN payloads would be caught too)
    np.testing.assert_array_equal(vox.view(np.uint32), rv.view(np.uint32))
    return rnv


CONFIGS = {
    # name: (generator, voxel_size, range, P, V)
    "c1_kitti": (lambda s: synth.kitti_frame(s), synth.KITTI_PILLAR, synth.KITTI_RANGE, 32, 16000),
    "c3_nusc_train_cap": (lambda s: synth.nuscenes_sweep(s), synth.NUSC_PILLAR, synth.NUSC_RANGE, 20, 30000),
    "c3_nusc_test_cap": (lambda s: synth.nuscenes_sweep(s), synth.NUSC_PILLAR, synth.NUSC_RANGE, 20, 60000),
    "c3_nusc_d4_shuffled": (lambda s: synth.nuscenes_sweep(s, dims=4, shuffle=True), synth.NUSC_PILLAR,
                            synth.NUSC_RANGE, 20, 30000),
    "c4_voxel": (lambda s: synth.nuscenes_sweep(s), synth.NUSC_VOXEL, synth.NUSC_VOXEL_RANGE, 10, 120000),
    "c5_bevfusion": (lambda s: synth.nuscenes_sweep(s, dims=4), (0.25, 0.25, 8.0),
                     (-50.0, -50.0, -5.0, 50.0, 50.0, 3.0), 64, 30000),
}


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("seed", [0, 1])
def test_matches_oracle(oracle, name, seed):
    gen, vs, pr, p, v = CONFIGS[name]
    nv = _check(oracle, gen(seed), vs, pr, p, v)
    assert nv > 0


def test_matches_reference_code(oracle):
    """Same check against the reference's own compiled kernel (oracle/_ref), when it travelled."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    gen, vs, pr, p, v = CONFIGS["c3_nusc_train_cap"]
    _check(oracle, gen(7), vs, pr, p, v, kind="ref")


def test_edge_cases(oracle):
    vs, pr = synth.NUSC_PILLAR, synth.NUSC_RANGE
    rng = np.random.default_rng(5)
    # all points outside the range -> zero voxels
    far = np.full((1000, 5), 500.0, np.float32)
    _check(oracle, far, vs, pr, 20, 100)
    # one cell hammered by many points (in-voxel overflow), tiny cap
    one = np.zeros((5000, 5), np.float32)
    one[:, :2] = 0.05
    one[:, 3] = np.arange(5000)
    _check(oracle, one, vs, pr, 20, 10)
    # max_voxels = 1: only the first cell survives, later points of that cell still join
    pts = synth.nuscenes_sweep(3, n_points=20000)
    _check(oracle, pts, vs, pr, 20, 1)
    # single point, odd sizes around tile boundaries of the sort (2048) and scan (4096)
    for n in (1, 2047, 2048, 2049, 4095, 4097):
        _check(oracle, synth.nuscenes_sweep(11, n_points=6000)[:n], vs, pr, 5, 50)
    # NaN / inf coordinates are dropped like the x86 reference drops them
    bad = synth.nuscenes_sweep(4, n_points=5000)
    bad[rng.choice(5000, 50, replace=False), 0] = np.nan
    bad[rng.choice(5000, 50, replace=False), 1] = np.inf
    bad[rng.choice(5000, 50, replace=False), 2] = -np.inf
    _check(oracle, bad, vs, pr, 20, 3000)


def test_batch_and_ragged(oracle):
    from paddle3d_amd.ops import voxelize

    frames = [synth.nuscenes_sweep(20 + i, n_points=50000) for i in range(3)]
    lens = [50000, 31234, 1]
    pts = torch.from_numpy(np.stack(frames)).cuda()
    num = torch.tensor(lens, dtype=torch.int32).cuda()
    vox, co, npv, nv = voxelize.hard_voxelize_batch(pts, list(synth.NUSC_PILLAR), list(synth.NUSC_RANGE), 20,
                                                    30000, num_points=num, path=PATH)
    torch.cuda.synchronize()
    for b in range(3):
        rv, rc, rn, rnv = oracle.hard_voxelize(frames[b][: lens[b]], synth.NUSC_PILLAR, synth.NUSC_RANGE, 20,
                                               30000)
        assert int(nv[b]) == rnv
        np.testing.assert_array_equal(co[b].cpu().numpy(), rc)
        np.testing.assert_array_equal(npv[b].cpu().numpy(), rn)
        np.testing.assert_array_equal(vox[b].cpu().numpy().view(np.uint32), rv.view(np.uint32))


def test_properties_full_size():
    """Size-independent properties at the full benchmark size (no oracle involved)."""
    from paddle3d_amd.ops import voxelize

    pts_np = synth.nuscenes_sweep(42)
    pts = torch.from_numpy(pts_np).cuda()
    vox, co, npv, nv = voxelize.hard_voxelize(pts, list(synth.NUSC_PILLAR), list(synth.NUSC_RANGE), 20, 60000,
                                              path=PATH)
    n = int(nv.item())
    co, npv, vox = co.cpu().numpy(), npv.cpu().numpy(), vox.cpu().numpy()
    # coords unique and inside the grid, counts in [1, P], padding is zero
    lin = co[:n, 1].astype(np.int64) * 512 + co[:n, 2]
    assert len(np.unique(lin)) == n
    assert (co[:n, 0] == 0).all() and (co[:n, 1:] >= 0).all() and (co[:n, 1:] < 512).all()
    assert (npv[:n] >= 1).all() and (npv[:n] <= 20).all() and (npv[n:] == 0).all()
    assert not vox[n:].any() and not co[n:].any()
    pad = np.arange(20)[None, :] >= npv[:, None]
    assert not vox[pad].any()
    # every stored point lies in its voxel's cell, and is an input point
    k = np.arange(20)[None, :] < npv[:n, None]
    cx = np.floor((vox[:n, :, 0] - np.float32(-51.2)) / np.float32(0.2)).astype(np.int64)
    cy = np.floor((vox[:n, :, 1] - np.float32(-51.2)) / np.float32(0.2)).astype(np.int64)
    assert (cx[k] == np.broadcast_to(co[:n, 2:3], cx.shape)[k]).all()
    assert (cy[k] == np.broadcast_to(co[:n, 1:2], cy.shape)[k]).all()
    # idempotence: voxelising the stored points again reproduces the same voxel set and counts
    stored = torch.from_numpy(np.ascontiguousarray(vox[:n][k])).cuda()
    vox2, co2, npv2, nv2 = voxelize.hard_voxelize(stored, list(synth.NUSC_PILLAR), list(synth.NUSC_RANGE), 20,
                                                  60000, path=PATH)
    assert int(nv2.item()) == n
    np.testing.assert_array_equal(co2.cpu().numpy()[:n], co[:n])
    np.testing.assert_array_equal(npv2.cpu().numpy()[:n], npv[:n])


def test_batched_coors_output(oracle):
    """coors_batched = (batch, z, y, x) with -1 on padding rows, from both paths."""
    from paddle3d_amd.ops import voxelize

    frames = np.stack([synth.nuscenes_sweep(30 + i, n_points=40000) for i in range(2)])
    out = voxelize.hard_voxelize_batch(torch.from_numpy(frames).cuda(), list(synth.NUSC_PILLAR),
                                       list(synth.NUSC_RANGE), 20, 12000, with_batch_coors=True, path=PATH)
    vox, co, npv, nv, c4 = [o.cpu().numpy() for o in out]
    for b in range(2):
        n = int(nv[b])
        np.testing.assert_array_equal(c4[b, :, 1:], co[b])
        assert (c4[b, :n, 0] == b).all() and (c4[b, n:, 0] == -1).all()


def test_dynamic_voxelize_agrees_with_hard_voxelize(oracle):
    """dynamic_voxelize's per-point cells = the cells hard_voxelize (and the reference CPU kernel) assigns."""
    from paddle3d_amd.ops import voxelize as V

    pts = synth.nuscenes_sweep(11, n_points=50_000)
    co = V.dynamic_voxelize(torch.from_numpy(pts).cuda(), list(synth.NUSC_PILLAR), list(synth.NUSC_RANGE)).cpu().numpy()
    vs, pr = np.asarray(synth.NUSC_PILLAR, np.float32), np.asarray(synth.NUSC_RANGE, np.float32)
    grid = np.round((pr[3:] - pr[:3]) / vs).astype(np.int64)
    c = np.floor((pts[:, :3] - pr[:3]) / vs)  # fp32 subtract / divide / floor, as voxelize_op.cc:37-45
    inside = np.all((c >= 0) & (c < grid), axis=1)
    want = np.where(inside[:, None], c[:, ::-1], -1).astype(np.int32)
    assert np.array_equal(co, want)
    # and the set of occupied cells equals the coords hard_voxelize reports when nothing is capped
    rv, rc, rn, rnv = oracle.hard_voxelize(pts, synth.NUSC_PILLAR, synth.NUSC_RANGE, 20, 200_000)
    assert set(map(tuple, co[inside])) == set(map(tuple, rc[:rnv]))


def test_host_points_are_staged(oracle):
    """voxelize_op.cc:149-166: CPU points -> hard_voxelize_cpu, results on the CPU; GPU-pinned points -> the device
    kernel, results on the GPU.  Here both run the device kernel (bit-identical to the CPU kernel)."""
    from paddle3d_amd.ops import voxelize

    pts = synth.nuscenes_sweep(31, n_points=50_000)
    args = (list(synth.NUSC_PILLAR), list(synth.NUSC_RANGE), 20, 9000)
    rv, rc, rn, rnv = oracle.hard_voxelize(pts, synth.NUSC_PILLAR, synth.NUSC_RANGE, 20, 9000)
    cpu_out = voxelize.hard_voxelize(torch.from_numpy(pts), *args, path=PATH)
    pin_out = voxelize.hard_voxelize(torch.from_numpy(pts).pin_memory(), *args, path=PATH)
    assert all(not o.is_cuda for o in cpu_out) and all(o.is_cuda for o in pin_out)
    for out in (cpu_out, pin_out):
        vox, co, npv, nv = [o.cpu().numpy() for o in out]
        assert int(nv[0]) == rnv
        np.testing.assert_array_equal(co, rc)
        np.testing.assert_array_equal(npv, rn)
        np.testing.assert_array_equal(vox.view(np.uint32), rv.view(np.uint32))
    with pytest.raises(RuntimeError, match="PD_DISPATCH_FLOATING_TYPES"):
        voxelize.hard_voxelize(torch.from_numpy(pts.astype(np.float16)).cuda(), *args)


@pytest.mark.parametrize("name", ["c1_kitti", "c3_nusc_train_cap"])
def test_float64_points(oracle, name):
    """PD_DISPATCH_FLOATING_TYPES (voxelize_op.cc:128): the reference's CPU kernel also exists for double points --
    cell = floor((p - (double)min) / (double)size) in double, voxels in double.  Bit-exact against that instantiation
    (compiled from /root/reference when it was present at build time, else the port, which test_oracle.py holds to
    it); points are placed on cell boundaries of the fp32 rule and a hair beside them, where fp32 and fp64 disagree."""
    from paddle3d_amd.ops import voxelize

    if PATH not in (0, 1):
        pytest.skip("float64 points take the generic sort path")
    gen, vs, pr, p, v = CONFIGS[name]
    pts = gen(11).astype(np.float64)
    rng = np.random.default_rng(3)
    k = rng.choice(len(pts), 2000, replace=False)
    pts[k[:1000], 0] = pr[0] + np.float64(vs[0]) * rng.integers(0, 300, 1000)          # on the fp64 boundary
    pts[k[1000:], 1] = np.float32(pr[1]) + np.float32(vs[1]) * rng.integers(0, 300, 1000).astype(np.float32)
    pts[k[1000:], 1] += rng.choice([-1e-12, 0.0, 1e-12], 1000)                           # a hair beside the fp32 one
    kind = "ref" if oracle.have_ref() else "port"
    rv, rc, rn, rnv = oracle.hard_voxelize(pts, vs, pr, p, v, kind)
    vox, co, npv, nv = voxelize.hard_voxelize(torch.from_numpy(pts).cuda(), list(vs), list(pr), p, v, path=PATH)
    assert vox.dtype == torch.float64 and int(nv[0]) == rnv and rnv > 1000
    np.testing.assert_array_equal(co.cpu().numpy(), rc)
    np.testing.assert_array_equal(npv.cpu().numpy(), rn)
    np.testing.assert_array_equal(vox.cpu().numpy().view(np.uint64), rv.view(np.uint64))
    cpu_out = voxelize.hard_voxelize(torch.from_numpy(pts), list(vs), list(pr), p, v, path=PATH)  # CPU in -> CPU out
    assert not cpu_out[0].is_cuda and torch.equal(cpu_out[0], vox.cpu())



def test_wave3d_heavy_group(oracle):
    """The multi-pass branch of the 3-D wave form (voxelize_wave3d.hpp): a 512 x 512 x 16 grid (2^22 cells, 4096
    possible cells per group) with two groups far over the 768 cells a hash pass takes -- one with 2500 occupied cells
    of 1 .. 14 points (three passes of eight fit), one with 1530 single-point cells (765 per pass expected, so a pass
    overflows and the pass count doubles from the start) -- inside 20 000 background points, shuffled.  Bit-exact
    against the oracle like every other input; paths that do not take the grid skip."""
    rng = np.random.default_rng(77)
    gx, gy, gz, vs = 512, 512, 16, 0.25
    pr = (-64.0, -64.0, -2.0, 64.0, 64.0, 2.0)
    keys = []
    for g0, ncell, reps in ((5, 2500, (1, 15)), (9, 1530, (1, 2))):
        L = rng.choice(4096, ncell, replace=False).astype(np.int64)
        lo = (g0 - 37 * L) & 1023
        k = L * 1024 + lo
        keys.append(np.repeat(k, rng.integers(reps[0], reps[1], ncell)))
    keys = np.concatenate(keys)
    cx, cy, cz = keys % gx, (keys // gx) % gy, keys // (gx * gy)
    jit = rng.uniform(0.2, 0.8, (len(keys), 3))
    heavy = np.stack([pr[0] + (cx + jit[:, 0]) * vs, pr[1] + (cy + jit[:, 1]) * vs, pr[2] + (cz + jit[:, 2]) * vs], 1)
    bg = np.stack([rng.uniform(-70, 70, 20000), rng.uniform(-70, 70, 20000), rng.uniform(-2.5, 2.5, 20000)], 1)
    pts = np.concatenate([heavy, bg]).astype(np.float32)
    pts = np.concatenate([pts, rng.uniform(0, 1, (len(pts), 2)).astype(np.float32)], 1)
    rng.shuffle(pts, axis=0)
    nv = _check(oracle, np.ascontiguousarray(pts), (vs, vs, vs), pr, 10, 30000)
    assert nv > 10000


def test_wave3d_batch_ragged_coors(oracle):
    """The 3-D wave form on a ragged batch of three frames of the 0.075 m grid (1440 x 1440 x 40) with the batched
    coors output: every frame equals the oracle's run on its own points, padding rows carry batch -1."""
    from paddle3d_amd.ops import voxelize

    frames = [synth.nuscenes_sweep(30 + i, n_points=60000) for i in range(3)]
    lens = [60000, 41234, 1]
    pts = torch.from_numpy(np.stack(frames)).cuda()
    num = torch.tensor(lens, dtype=torch.int32).cuda()
    out = _unsupported_ok(voxelize.hard_voxelize_batch, pts, list(synth.NUSC_VOXEL), list(synth.NUSC_VOXEL_RANGE), 10,
                          40000, num_points=num, with_batch_coors=True, path=PATH)
    vox, co, npv, nv, c4 = out
    torch.cuda.synchronize()
    c4 = c4.cpu().numpy().reshape(3, 40000, 4)
    for b in range(3):
        rv, rc, rn, rnv = oracle.hard_voxelize(frames[b][: lens[b]], synth.NUSC_VOXEL, synth.NUSC_VOXEL_RANGE, 10, 40000)
        assert int(nv[b]) == rnv
        np.testing.assert_array_equal(co[b].cpu().numpy(), rc)
        np.testing.assert_array_equal(npv[b].cpu().numpy(), rn)
        np.testing.assert_array_equal(vox[b].cpu().numpy().view(np.uint32), rv.view(np.uint32))
        assert (c4[b, :rnv, 0] == b).all() and (c4[b, rnv:, 0] == -1).all()
        np.testing.assert_array_equal(c4[b, :rnv, 1:], rc[:rnv])
