"""sparse_conv3d (submanifold + regular) vs a dense torch conv3d oracle on densified inputs (CPU fp32).

The reference's sparse arithmetic is Paddle core (not vendored): parity is pinned on the public definition.
Oracle: dense conv3d of the densified input, restricted to the active output set -- for a submanifold conv the
input set, for a regular conv every position some active input reaches (= conv of the occupancy mask > 0)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _random_sparse(rng, batch, shape, n, c):
    d, h, w = shape
    lin = rng.choice(batch * d * h * w, n, replace=False)
    b, r = np.divmod(lin, d * h * w)
    z, r = np.divmod(r, h * w)
    y, x = np.divmod(r, w)
    coords = np.stack([b, z, y, x], 1).astype(np.int32)
    feats = rng.normal(size=(n, c)).astype(np.float32)
    return coords, feats


def _densify(coords, feats, batch, shape):
    d, h, w = shape
    dense = torch.zeros(batch, feats.shape[1], d, h, w)
    co = torch.from_numpy(coords).long()
    dense[co[:, 0], :, co[:, 1], co[:, 2], co[:, 3]] = torch.from_numpy(feats)
    mask = torch.zeros(batch, 1, d, h, w)
    mask[co[:, 0], 0, co[:, 1], co[:, 2], co[:, 3]] = 1
    return dense, mask


def _dense_oracle(coords, feats, batch, shape, weight, ks, stride, pad, subm):
    dense, mask = _densify(coords, feats, batch, shape)
    w = torch.from_numpy(weight).permute(4, 3, 0, 1, 2).contiguous()  # [kd,kh,kw,ci,co] -> [co,ci,kd,kh,kw]
    out = F.conv3d(dense, w, stride=stride, padding=pad)
    if subm:
        active = mask
    else:
        active = (F.conv3d(mask, torch.ones(1, 1, *ks), stride=stride, padding=pad) > 0).float()
    return out * active, active


CASES = [
    # name, batch, shape, n, cin, cout, ks, stride, pad, subm
    ("subm3", 2, (9, 20, 24), 600, 5, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    ("subm3_c128", 1, (5, 16, 16), 300, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    ("down_s2_p1", 2, (9, 20, 24), 500, 16, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1), False),
    ("down_s2_p011", 1, (11, 20, 18), 400, 64, 128, (3, 3, 3), (2, 2, 2), (0, 1, 1), False),
    ("extra_311", 2, (5, 12, 10), 200, 128, 128, (3, 1, 1), (2, 1, 1), (0, 0, 0), False),
    ("k1", 1, (4, 8, 8), 60, 8, 8, (1, 1, 1), (1, 1, 1), (0, 0, 0), False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_matches_dense_oracle(case):
    from paddle3d_amd.ops import sparse_conv3d as sp

    _, batch, shape, n, cin, cout, ks, stride, pad, subm = case
    rng = np.random.default_rng(hash(case[0]) % 1000)
    coords, feats = _random_sparse(rng, batch, shape, n, cin)
    weight = (rng.normal(size=(*ks, cin, cout)) / np.sqrt(cin * np.prod(ks))).astype(np.float32)
    idx = sp.indices(torch.from_numpy(coords).cuda(), batch, shape, ks, stride, pad, subm)
    out = sp.features(torch.from_numpy(feats).cuda(), idx, torch.from_numpy(weight).cuda())
    ref, active = _dense_oracle(coords, feats, batch, shape, weight, ks, stride, pad, subm)
    # output index set is exactly the active set
    oc = idx.out_coords.cpu().long()
    assert idx.n_out == int(active.sum().item())
    assert active[oc[:, 0], 0, oc[:, 1], oc[:, 2], oc[:, 3]].all()
    if not subm:  # regular conv rows come out sorted by (b, z, y, x)
        lin = ((oc[:, 0] * idx.out_shape[0] + oc[:, 1]) * idx.out_shape[1] + oc[:, 2]) * idx.out_shape[2] + oc[:, 3]
        assert (lin[1:] > lin[:-1]).all()
    dense = sp.to_dense(out, idx.out_coords, batch, idx.out_shape).cpu()
    want = ref.reshape(batch, cout * ref.shape[2], ref.shape[3], ref.shape[4])
    assert dense.shape == want.shape
    assert (dense - want).abs().max().item() < 2e-4


def test_fused_epilogue_and_rulebook_reuse():
    from paddle3d_amd import sparse as S

    torch.manual_seed(0)
    rng = np.random.default_rng(3)
    batch, shape = 2, (7, 16, 16)
    coords, feats = _random_sparse(rng, batch, shape, 500, 16)
    blk = S.SparseBasicBlock(16, 16, "k").cuda().eval()
    with torch.no_grad():
        for bn in (blk.bn1, blk.bn2):
            bn.running_mean.normal_(0, 0.1)
            bn.running_var.uniform_(0.5, 1.5)
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.1)
        blk.conv1.bias.normal_(0, 0.1)
        blk.conv2.bias.normal_(0, 0.1)
    x = S.SparseConvTensor(torch.from_numpy(feats).cuda(), torch.from_numpy(coords).cuda(), shape, batch)
    y = blk(x)
    assert "k" in x.cache  # both convs of the block share one rulebook
    # dense statement of the block (sparse_resnet.py:92-111), BatchNorm acting on active sites only
    dense, mask = _densify(coords, feats, batch, shape)

    def conv(m, t):
        w = m.weight.detach().cpu().permute(4, 3, 0, 1, 2).contiguous()
        return F.conv3d(t, w, m.bias.detach().cpu(), padding=1) * mask

    def bn(b, t):
        s = (b.weight / torch.sqrt(b.running_var + b.eps)).detach().cpu().view(1, -1, 1, 1, 1)
        sh = (b.bias - b.running_mean * b.weight / torch.sqrt(b.running_var + b.eps)).detach().cpu().view(1, -1, 1, 1, 1)
        return (t * s + sh) * mask

    o = torch.relu(bn(blk.bn1, conv(blk.conv1, dense)))
    o = torch.relu(bn(blk.bn2, conv(blk.conv2, o)) + dense)
    got = y.dense().cpu().reshape(o.shape)
    assert (got - o).abs().max().item() < 2e-4


def _randomise(net):
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            if getattr(m, "bias", None) is not None and hasattr(m, "subm"):
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)


@pytest.mark.parametrize("which", ["SparseResNet3D", "SparseNet3D"])
def test_sparse_encoder_matches_dense_stack(oracle, which):
    """Whole middle encoder on a shrunken grid (z 41 -> 21 -> 11 -> 5 -> 2 as in the reference comments,
    sparse_resnet.py:136-163) against the dense conv3d + BatchNorm + ReLU + add stack built from the same
    parameters (oracle.sparse_encoder_dense_torch)."""
    from paddle3d_amd import sparse as S

    torch.manual_seed(1)
    net = getattr(S, which)(5, voxel_size=(0.5, 0.5, 0.2), point_cloud_range=(-8, -8, -5, 8, 8, 3)).cuda().eval()
    _randomise(net)
    assert net.sparse_shape == (41, 32, 32)
    rng = np.random.default_rng(5)
    coords, feats = _random_sparse(rng, 2, net.sparse_shape, 3000, 5)
    out = net(torch.from_numpy(feats).cuda(), torch.from_numpy(coords).cuda(), 2)
    if which == "SparseNet3D":
        assert out["spatial_features_stride"] == 8 and set(out["multi_scale_3d_features"]) == {
            "x_conv1", "x_conv2", "x_conv3", "x_conv4"}
        assert out["multi_scale_3d_features"]["x_conv3"].features.shape[1] == 64
        out = out["spatial_features"]
    assert out.shape == (2, 128 * 2, 4, 4)
    ref = oracle.sparse_encoder_dense_torch(net.cpu(), feats, coords, 2)
    net.cuda()
    assert out.shape == ref.shape
    err = (out.cpu() - ref).abs().max().item()
    assert err < 1e-3 * max(1.0, ref.abs().max().item()), err
    # determinism: identical result on a second run (sorted output rows, fixed summation order)
    out2 = net(torch.from_numpy(feats).cuda(), torch.from_numpy(coords).cuda(), 2)
    out2 = out2["spatial_features"] if isinstance(out2, dict) else out2
    assert torch.equal(out, out2)


def test_sparse_encoder_empty_input():
    """An empty voxel set gives an all-zero map (the reference's layers accept nnz == 0)."""
    from paddle3d_amd import sparse as S

    net = S.SparseResNet3D(5, voxel_size=(0.5, 0.5, 0.2), point_cloud_range=(-8, -8, -5, 8, 8, 3)).cuda().eval()
    out = net(torch.zeros(0, 5, device="cuda"), torch.zeros(0, 4, dtype=torch.int32, device="cuda"), 1)
    assert out.shape == (1, 256, 4, 4) and not out.any()


def test_sparse_parameter_names_follow_the_reference():
    """State-dict keys = the reference's (sparse_resnet.py:126-164 Sequential indices), so a converted checkpoint
    places every middle-encoder key."""
    from paddle3d_amd import sparse as S

    keys = set(S.SparseResNet3D(5).state_dict())
    for k in ("conv_input.0.weight", "conv_input.1.running_mean", "conv1.1.conv2.bias", "conv2.0.weight",
              "conv2.1.running_var", "conv2.3.conv1.weight", "conv2.4.bn2.weight", "conv4.0.weight",
              "extra_conv.0.weight", "extra_conv.1.bias"):
        assert k in keys, k
    assert "conv2.0.bias" not in keys  # bias_attr=False on the strided convolutions
    keys = set(S.SparseNet3D(4).state_dict())
    for k in ("conv_input.0.weight", "conv1.0.0.weight", "conv1.0.1.running_mean", "conv2.0.0.weight", "conv2.2.1.bias",
              "conv4.2.0.weight", "extra_conv.0.weight"):
        assert k in keys, k


PLAN_CHAINS = [
    # the CenterPoint-Voxel encoder's chain on a small grid; a chain with an unpadded stride-3 kernel-3 convolution
    ("resnet_like", 2, (21, 40, 36), 3000,
     [((3, 3, 3), (1, 1, 1), (1, 1, 1), True, "a"), ((3, 3, 3), (1, 1, 1), (1, 1, 1), True, "a"),
      ((3, 3, 3), (2, 2, 2), (1, 1, 1), False, None), ((3, 3, 3), (1, 1, 1), (1, 1, 1), True, "b"),
      ((3, 3, 3), (2, 2, 2), (0, 1, 1), False, None), ((3, 3, 3), (1, 1, 1), (1, 1, 1), True, "c"),
      ((3, 1, 1), (2, 1, 1), (0, 0, 0), False, None)]),
    ("odd_paddings", 1, (9, 30, 31), 1500,
     [((3, 3, 3), (3, 3, 3), (0, 0, 0), False, None), ((1, 3, 3), (1, 1, 1), (0, 1, 1), True, None),
      ((2, 2, 2), (2, 2, 2), (1, 0, 1), False, None)]),
    ("dense_line", 1, (1, 1, 700), 650, [((1, 1, 3), (1, 1, 1), (0, 0, 1), True, None),
                                          ((1, 1, 3), (1, 1, 2), (0, 0, 1), False, None)]),
]


@pytest.mark.parametrize("chain", PLAN_CHAINS, ids=[c[0] for c in PLAN_CHAINS])
def test_plan_equals_per_conv_indices(chain):
    """plan() (sorted key sets, device-side counts, LDS-staged hash rulebooks) against indices() applied conv by
    conv (global hash table): identical output coordinate sets, identical neighbour tables.  The input carries
    padding rows (batch = -1) in shuffled order, like the voxelizer's fixed-shape output."""
    from paddle3d_amd.ops import sparse_conv3d as sp
    _, batch, shape, n, convs = chain
    rng = np.random.default_rng(11)
    coords, _ = _random_sparse(rng, batch, shape, n, 1)
    pad = np.full((n // 3, 4), -1, np.int32)
    mixed = np.concatenate([coords, pad])
    perm = rng.permutation(len(mixed))
    mixed = mixed[perm]
    specs = [sp.ConvSpec(*c) for c in convs]
    pl = sp.plan(torch.from_numpy(mixed).cuda(), batch, shape, specs)
    assert pl.n_in == n
    got_rows = mixed[pl.order.cpu().numpy()]
    assert (got_rows == pl.coords.cpu().numpy()).all()
    lin = ((got_rows[:, 0].astype(np.int64) * shape[0] + got_rows[:, 1]) * shape[1] + got_rows[:, 2]) * shape[2] + \
        got_rows[:, 3]
    assert (np.diff(lin) > 0).all()  # raster order, every real row exactly once
    cur, cur_shape = pl.coords, tuple(shape)
    for spec, idx in zip(specs, pl.indices):
        ref = sp.indices(cur, batch, cur_shape, spec.kernel_size, spec.stride, spec.padding, spec.subm)
        assert ref.n_out == idx.n_out and tuple(ref.out_shape) == tuple(idx.out_shape)
        assert torch.equal(ref.out_coords, idx.out_coords)
        assert torch.equal(ref.nbr, idx.nbr)
        cur, cur_shape = idx.out_coords, idx.out_shape
    # rulebooks are shared where the key says so
    assert pl.indices[0] is pl.indices[1] if chain[0] == "resnet_like" else True


def test_plan_large_tiles_and_empty():
    """more rows than one rulebook tile and more neighbours per (kz, ky) piece than one LDS chunk; empty input"""
    from paddle3d_amd.ops import sparse_conv3d as sp
    rng = np.random.default_rng(5)
    shape = (3, 64, 2048)
    coords, _ = _random_sparse(rng, 1, shape, 150000, 1)
    specs = [sp.ConvSpec((3, 3, 3), (1, 1, 1), (1, 1, 1), True), sp.ConvSpec((3, 3, 3), (2, 2, 2), (1, 1, 1))]
    c = torch.from_numpy(coords).cuda()
    pl = sp.plan(c, 1, shape, specs)
    cur = pl.coords
    ref0 = sp.indices(cur, 1, shape, (3, 3, 3), 1, 1, True)
    assert torch.equal(ref0.nbr, pl.indices[0].nbr)
    ref1 = sp.indices(cur, 1, shape, (3, 3, 3), 2, 1, False)
    assert torch.equal(ref1.out_coords, pl.indices[1].out_coords) and torch.equal(ref1.nbr, pl.indices[1].nbr)
    empty = sp.plan(torch.full((10, 4), -1, dtype=torch.int32).cuda(), 1, shape, specs)
    assert empty.n_in == 0 and all(i.n_out == 0 for i in empty.indices)


def _keys_of(indices, shape):
    c = indices.cpu().numpy().astype(np.int64)
    d, h, w = shape
    return ((c[:, 0] * d + c[:, 1]) * h + c[:, 2]) * w + c[:, 3]


def test_sparse_encoder_full_c4_grid_vs_keyset_oracle(oracle):
    """Config 4 at the size the bench runs (configs/centerpoint/centerpoint_voxels_0075voxel_nuscenes_10sweep.yml:
    111-173: 41 x 1440 x 1440 cells of 0.075 m, two real synth.nuscenes_sweep frames, ~130 k voxels each, LiDAR-shaped
    occupancy: dense near-sensor neighbourhoods and sparse far rings) against oracle.sparse_encoder_numpy, which
    needs no dense grid.  EVERY convolution of SparseResNet3D (sparse_resnet.py:115-206) -- the input layer (5 -> 16),
    submanifold 16 / 32 / 64 / 128, the three strided 3x3x3 layers, the final (3, 1, 1) / (2, 1, 1) layer -- is
    compared where the device's fused kernel writes: index set exact and in the same (raster) order, features within
    1e-3 of the largest magnitude of the layer; then the [2, 256, 180, 180] map."""
    from paddle3d_amd import centerpoint as cpm
    from paddle3d_amd import sparse as S
    from paddle3d_amd import synth

    torch.manual_seed(3)
    model = cpm.centerpoint_voxels_nuscenes().cuda().eval()
    net = model.middle_encoder
    _randomise(net)
    assert net.sparse_shape == (41, 1440, 1440)
    pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(93), synth.nuscenes_sweep(94)])).cuda()
    voxels, coors, npv, nv = model.voxelizer(pts)
    b, v, p, d = voxels.shape
    coors = coors.view(b * v, 4)
    feats = model.voxel_encoder(voxels.view(b * v, p, d), npv.view(b * v), coors)
    assert int((coors[:, 0] >= 0).sum()) == int(nv.sum()) > 200_000  # padding rows travel along (batch -1)

    got = {}
    names = {id(m): n for n, m in net.named_modules()}
    hooks = [m.register_forward_hook(lambda mod, inp, out: got.__setitem__(names[id(mod)], out))
             for m in net.modules() if isinstance(m, S._SparseConv)]
    try:
        bev = net(feats, coors, b)
    finally:
        for h in hooks:
            h.remove()
    trace = {}
    ref_bev = oracle.sparse_encoder_numpy(net, feats.cpu().numpy(), coors.cpu().numpy(), b, trace=trace)
    assert sorted(trace) == sorted(got) and len(got) == 21
    kinds = set()
    for name, want in trace.items():
        t = got[name]
        mod = dict(net.named_modules())[name]
        kinds.add((mod.subm, tuple(mod.weight.shape[-2:]), mod.ks))
        assert tuple(t.spatial_shape) == tuple(want["shape"]), name
        np.testing.assert_array_equal(_keys_of(t.indices, t.spatial_shape), want["keys"], err_msg=name)
        f = t.features.cpu().numpy()
        scale = max(1.0, float(np.abs(want["feats"]).max()))
        err = float(np.abs(f - want["feats"]).max())
        assert err < 1e-3 * scale, (name, err, scale)
    # every layer type of the encoder was in the comparison
    assert {(True, (5, 16), (3, 3, 3)), (True, (16, 16), (3, 3, 3)), (True, (32, 32), (3, 3, 3)),
            (True, (64, 64), (3, 3, 3)), (True, (128, 128), (3, 3, 3)), (False, (16, 32), (3, 3, 3)),
            (False, (32, 64), (3, 3, 3)), (False, (64, 128), (3, 3, 3)), (False, (128, 128), (3, 1, 1))} <= kinds
    assert bev.shape == ref_bev.shape == (2, 256, 180, 180)
    err = float(np.abs(bev.cpu().numpy() - ref_bev).max())
    assert err < 1e-3 * max(1.0, float(np.abs(ref_bev).max())), err
    # the same map with the rows in raster order (no tile order) and from the unsynced plan: identical bytes
    from paddle3d_amd.ops import sparse_conv3d as sp
    net.remember_capacities = True
    assert torch.equal(net(feats, coors, b), bev)   # synced plan, capacities remembered
    assert torch.equal(net(feats, coors, b), bev) and net._overflow is not None and not net.take_overflow()
    sp.TILE_ORDER = False
    try:
        net.remember_capacities = False
        assert torch.equal(net(feats, coors, b), bev)
    finally:
        sp.TILE_ORDER = True
        net.remember_capacities = None
    # mixed precision (net.amp: the layers from 16 -> 32 on run on the fp16 matrix cores, fp16 rows between them): the
    # same index sets, every layer and the map within fp16's resolution of the oracle (3 % of the layer's magnitude --
    # twenty fp16 layers deep; the fp32 form above holds 1e-3)
    got.clear()
    net.amp = True
    net.remember_capacities = False  # exact-size arrays: the hooks compare whole index sets
    hooks = [m.register_forward_hook(lambda mod, inp, out: got.__setitem__(names[id(mod)], out))
             for m in net.modules() if isinstance(m, S._SparseConv)]
    try:
        bev16 = net(feats, coors, b)
    finally:
        net.amp = False
        net.remember_capacities = None
        for h in hooks:
            h.remove()
    assert bev16.dtype == torch.float32 and got["conv3.3.conv1"].features.dtype == torch.float16
    assert got["conv1.1.conv2"].features.dtype == torch.float32  # the 16-channel layers stay fp32
    for name, want in trace.items():
        t = got[name]
        np.testing.assert_array_equal(_keys_of(t.indices, t.spatial_shape), want["keys"], err_msg=name)
        scale = max(1.0, float(np.abs(want["feats"]).max()))
        err = float(np.abs(t.features.float().cpu().numpy() - want["feats"]).max())
        print("amp", name, "err", err, "of", scale)
        assert err < 3e-2 * scale, ("amp", name, err, scale)
    err16 = float(np.abs(bev16.cpu().numpy() - ref_bev).max())
    assert 0 < err16 < 3e-2 * max(1.0, float(np.abs(ref_bev).max())), err16
    # the dense neighbourhoods the bench's tiles see are in this input: > 10 existing pairs per row on the 32+ layers
    assert trace["conv2.3.conv1"]["pairs"] > 10 * trace["conv2.3.conv1"]["keys"].shape[0]


@pytest.mark.parametrize("layer", [(16, 16, True), (32, 64, False), (64, 64, True), (128, 128, True)],
                         ids=lambda l: f"{l[0]}to{l[1]}{'subm' if l[2] else 'down'}")
def test_tile_order_changes_no_byte(layer):
    """The tile order (rows of a window sorted by neighbour mask, pd3_sparse_tile_order) only decides which rows share
    a 16-row block of the gather-GEMM: the order array is a permutation of every window's rows (-1 past the row
    count) and the features are the same bytes with and without it."""
    from paddle3d_amd.ops import sparse_conv3d as sp

    cin, cout, subm = layer
    rng = np.random.default_rng(cin + cout)
    shape = (9, 120, 130)
    n = 20000  # 2.4 windows of 8192 rows: a full window, a partial one, padding
    coords, feats = _random_sparse(rng, 2, shape, n, cin)
    # clustered occupancy as well: half of the rows in a dense slab, so that masks range from 1 to 27 neighbours
    slab = np.stack(np.meshgrid(np.arange(2), np.arange(3, 6), np.arange(40, 90), np.arange(30, 60), indexing="ij"),
                    -1).reshape(-1, 4).astype(np.int32)
    coords = np.unique(np.concatenate([coords, slab]), axis=0)
    feats = rng.normal(size=(len(coords), cin)).astype(np.float32)
    spec = sp.ConvSpec((3, 3, 3), (1, 1, 1), (1, 1, 1), True) if subm else sp.ConvSpec((3, 3, 3), (2, 2, 2), (1, 1, 1))
    w = torch.from_numpy((rng.normal(size=(3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)).cuda()
    res = torch.from_numpy(rng.normal(size=(len(coords) * 8, cout)).astype(np.float32)).cuda()
    outs = []
    for flag in (True, False):
        sp.TILE_ORDER = flag
        try:
            pl = sp.plan(torch.from_numpy(coords).cuda(), 2, shape, [spec])
            idx = pl.indices[0]
            f = torch.from_numpy(feats).cuda().index_select(0, pl.order)
            outs.append(sp.features(f, idx, w, None, None, None, res[: idx.n_out].contiguous(), True))
            if flag:
                order = idx.order.cpu().numpy()
                assert len(order) % 8192 == 0 and len(order) >= idx.n_out
                live = order[order >= 0]
                assert len(live) == idx.n_out and np.array_equal(np.sort(live), np.arange(idx.n_out))
                for w0 in range(0, len(order), 8192):  # every window holds ITS rows, the padding last
                    piece = order[w0:w0 + 8192]
                    k = int((piece >= 0).sum())
                    assert (piece[:k] >= w0).all() and (piece[:k] < w0 + 8192).all() and (piece[k:] == -1).all()
                # rows of one block are neighbours in mask order: fewer (block, offset) steps than in raster order
                nb = idx.nbr.cpu().numpy() >= 0
                def steps(rows):
                    pad = (-len(rows)) % 16
                    m = np.concatenate([nb[rows], np.zeros((pad, nb.shape[1]), bool)]).reshape(-1, 16, nb.shape[1])
                    return int(m.any(1).sum())
                assert steps(live) < steps(np.arange(idx.n_out))
            else:
                assert idx.order is None
        finally:
            sp.TILE_ORDER = True
    assert torch.equal(outs[0], outs[1])


def test_bf16x3_inf_input_becomes_nan_only_where_it_is_gathered():
    """The statement at csrc/sparse_conv_x3.hip:15-16, pinned: an Inf in an input row turns into NaN in the bf16x3
    kernel (x - bf16(x) = Inf - Inf), where the fp32 kernel carries the Inf; output rows that do not gather the row are
    untouched (identical to the run without the Inf).  Activations of a network are finite; a caller that cannot
    promise that uses `ops.sparse_conv3d.SPLIT_BF16 = False`."""
    from paddle3d_amd.ops import sparse_conv3d as sp

    rng = np.random.default_rng(3)
    shape = (9, 40, 40)
    coords, feats = _random_sparse(rng, 1, shape, 6000, 64)
    w = (rng.normal(size=(3, 3, 3, 64, 64)) / np.sqrt(27 * 64)).astype(np.float32)
    w = np.abs(w)  # one sign: Inf * w never meets -Inf in the fp32 kernel's sum
    pl = sp.plan(torch.from_numpy(coords).cuda(), 1, shape, [sp.ConvSpec((3, 3, 3), (1, 1, 1), (1, 1, 1), True)])
    idx = pl.indices[0]
    f = torch.from_numpy(feats).cuda().index_select(0, pl.order)
    wt = torch.from_numpy(w).cuda()
    packed = sp.pack_weight_bf16x3(wt)
    clean = sp.features_bf16x3(f, idx, packed, 64, 64)
    bad_row = 1234
    f_inf = f.clone()
    f_inf[bad_row, 5] = float("inf")
    got = sp.features_bf16x3(f_inf, idx, packed, 64, 64)
    ref = sp.features(f_inf, idx, wt)
    touched = (idx.nbr[: idx.n_out] == bad_row).any(1)
    assert int(touched.sum()) >= 1 and bool(touched[bad_row])
    assert torch.isinf(ref[touched]).any(1).all() and not torch.isnan(ref).any()   # the fp32 kernel: Inf stays Inf
    assert torch.isnan(got[touched]).any(1).all()                                  # bf16x3: NaN, as documented
    assert torch.equal(got[~touched], clean[~touched]) and torch.isfinite(got[~touched]).all()


def test_plan_without_host_sync_and_overflow(oracle):
    """SparseResNet3D plans its first forward of a shape with the one host sync and remembers the index sets' sizes;
    the next forward of that shape plans from the remembered capacities with NO host round trip (device row counts,
    arrays at capacity) and gives the same bytes.  A capacity that is too small is reported by take_overflow() and
    CenterPoint.test_forward then recomputes the frame with exact sizes."""
    from paddle3d_amd import centerpoint as cpm
    from paddle3d_amd import synth

    torch.manual_seed(5)
    pcr = [-9.6, -9.6, -5.0, 9.6, 9.6, 3.0]
    model = cpm.centerpoint_voxels_nuscenes(max_num_voxels=(40000, 40000), point_cloud_range=pcr).cuda().eval()
    net = model.middle_encoder
    _randomise(net)
    with torch.no_grad():
        for task in model.bbox_head.tasks:
            task.hm[-1].bias.fill_(-1.0)
    pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(93, n_points=120_000),
                                     synth.nuscenes_sweep(94, n_points=120_000)])).cuda()
    voxels, coors, npv, nv = model.voxelizer(pts)
    b, v, p, d = voxels.shape
    coors = coors.view(b * v, 4)
    feats = model.voxel_encoder(voxels.view(b * v, p, d), npv.view(b * v), coors)
    # the default is the exact plan: a direct call never remembers anything and can never truncate
    assert net.remember_capacities is None
    plain = net(feats, coors, b)
    assert not hasattr(net, "_caps") or net._caps == {}
    assert torch.equal(model.extract_pillars(pts), plain) and getattr(net, "_overflow", None) is None
    net.remember_capacities = True        # opt in: this test reads take_overflow() itself
    first = net(feats, coors, b)          # synced plan, capacities remembered
    assert torch.equal(first, plain)
    assert len(net._caps) == 1 and not net.take_overflow()
    caps = next(iter(net._caps.values()))
    second = net(feats, coors, b)         # planned from the capacities
    assert net._overflow is not None and torch.equal(first, second)
    assert not net.take_overflow()
    ref = oracle.sparse_encoder_numpy(net, feats.cpu().numpy(), coors.cpu().numpy(), b)
    assert np.abs(second.cpu().numpy() - ref).max() < 1e-3 * max(1.0, np.abs(ref).max())
    # another frame of the same shape (other row counts) through the unsynced plan
    pts2 = torch.from_numpy(np.stack([synth.nuscenes_sweep(95, n_points=120_000),
                                      synth.nuscenes_sweep(96, n_points=120_000)])).cuda()
    dets_a = model.test_forward(pts2)
    net.remember_capacities = False
    dets_b = model.test_forward(pts2)
    net.remember_capacities = None        # the default: test_forward opts in by itself (it checks the overflow word)
    object.__setattr__(net, "_caps", {})
    dets_c = model.test_forward(pts2)
    assert len(net._caps) == 1            # ... and remembered the capacities of this shape
    dets_d = model.test_forward(pts2, device_only=True)   # no check possible here: exact plan, nothing pending
    assert getattr(net, "_overflow", None) is None and dets_d[3].shape[0] == 2
    net.remember_capacities = True
    for a, c, e in zip(dets_a, dets_b, dets_c):
        assert torch.equal(a["box3d_lidar"], e["box3d_lidar"]) and torch.equal(a["scores"], e["scores"])
        assert torch.equal(a["box3d_lidar"], c["box3d_lidar"]) and torch.equal(a["scores"], c["scores"])
    # capacities far too small: the unsynced forward truncates, says so, and test_forward recomputes
    key = next(iter(net._caps))
    net._caps[key] = [caps[0]] + [max(8192, c // 8) for c in caps[1:]]
    truncated = net(feats, coors, b)
    assert net.take_overflow() and net._caps == {}
    assert not torch.equal(truncated, first)
    net(feats, coors, b)
    key = next(iter(net._caps))
    net._caps[key] = [net._caps[key][0]] + [max(8192, c // 8) for c in net._caps[key][1:]]
    dets_c = model.test_forward(pts2)     # first attempt overflows, second attempt plans with the sync
    for a, c in zip(dets_a, dets_c):
        assert torch.equal(a["box3d_lidar"], c["box3d_lidar"]) and torch.equal(a["scores"], c["scores"])


F16_CASES = [
    # cin, cout, subm, kernel, stride, padding, epilogue (bias, bn, residual, relu), out_f32
    (16, 32, False, (3, 3, 3), (2, 2, 2), (1, 1, 1), (False, True, False, True), False),
    (32, 32, True, (3, 3, 3), (1, 1, 1), (1, 1, 1), (True, True, True, True), False),
    (32, 64, False, (3, 3, 3), (2, 2, 2), (1, 1, 1), (False, True, False, True), False),
    (64, 64, True, (3, 3, 3), (1, 1, 1), (1, 1, 1), (True, True, True, True), False),
    (64, 128, False, (3, 3, 3), (2, 2, 2), (0, 1, 1), (False, False, False, False), False),
    (128, 128, True, (3, 3, 3), (1, 1, 1), (1, 1, 1), (True, True, True, True), False),
    (128, 128, False, (3, 1, 1), (2, 1, 1), (0, 0, 0), (False, True, False, True), True),
    (48, 32, True, (3, 3, 3), (1, 1, 1), (1, 1, 1), (True, False, False, False), True),  # a 16-channel chunk, 3 chunks
]


@pytest.mark.parametrize("case", F16_CASES, ids=lambda c: f"{c[0]}to{c[1]}{'subm' if c[2] else 'down'}")
@pytest.mark.parametrize("order", [True, False], ids=["tile_order", "raster"])
def test_features_f16_matches_fp32_math_on_fp16_operands(case, order):
    """pd3_sparse_conv3d_features_f16 against the fp32 kernel fed the SAME fp16-rounded rows and weights: what is left
    is the accumulation order and the fp16 rounding of the result (fp32 out: 2e-4; fp16 out: one fp16 ulp of the
    magnitude on top).  Every chunk width (16 / 32 / 64 channels), every output width, both epilogue forms, rows in
    tile order and in raster order, a partial last tile."""
    from paddle3d_amd.ops import sparse_conv3d as sp

    cin, cout, subm, ks, stride, pad, (with_bias, with_bn, with_res, relu), out_f32 = case
    rng = np.random.default_rng(cin * 7 + cout)
    shape = (9, 60, 70)
    coords, feats = _random_sparse(rng, 2, shape, 9000, cin)
    slab = np.stack(np.meshgrid(np.arange(2), np.arange(3, 6), np.arange(20, 45), np.arange(30, 60), indexing="ij"),
                    -1).reshape(-1, 4).astype(np.int32)
    coords = np.unique(np.concatenate([coords, slab]), axis=0)
    feats = rng.normal(size=(len(coords), cin)).astype(np.float32)
    w = (rng.normal(size=(*ks, cin, cout)) / np.sqrt(np.prod(ks) * cin)).astype(np.float32)
    sp.TILE_ORDER = order
    try:
        pl = sp.plan(torch.from_numpy(coords).cuda(), 2, shape, [sp.ConvSpec(ks, stride, pad, subm)])
        idx = pl.indices[0]
        f16 = torch.from_numpy(feats).cuda().index_select(0, pl.order).half()
        w16 = torch.from_numpy(w).cuda().half()
        bias = torch.from_numpy(rng.normal(size=cout).astype(np.float32)).cuda() if with_bias else None
        sc = torch.from_numpy(rng.normal(size=cout).astype(np.float32)).cuda() if with_bn else None
        sh = torch.from_numpy(rng.normal(size=cout).astype(np.float32)).cuda() if with_bn else None
        res = torch.from_numpy(rng.normal(size=(idx.n_out, cout)).astype(np.float32)).cuda().half() if with_res else None
        want = sp.features(f16.float(), idx, w16.float(), bias, sc, sh, None if res is None else res.float(), relu)
        got = sp.features_f16(f16, idx, sp.pack_weight_f16(w16.float()), cin, cout, bias, sc, sh, res, relu,
                              out_f32=out_f32)
    finally:
        sp.TILE_ORDER = True
    assert got.dtype == (torch.float32 if out_f32 else torch.float16) and got.shape == want.shape
    assert idx.n_out % 256 != 0 and idx.n_out > 2000
    mag = float(want.abs().max())
    tol = 2e-4 * max(1.0, mag) + (0.0 if out_f32 else 1e-3 * mag)
    assert float((got.float() - want).abs().max()) < tol


@pytest.mark.parametrize("case", F16_CASES, ids=lambda c: f"{c[0]}to{c[1]}{'subm' if c[2] else 'down'}")
@pytest.mark.parametrize("order", [True, False], ids=["tile_order", "raster"])
def test_features_bf16x3_is_fp32_arithmetic(case, order):
    """pd3_sparse_conv3d_features_bf16x3 (fp32 operands as three bf16 pieces, six piece products, fp32 accumulation)
    against an fp64 gather-GEMM of the same fp32 rows and weights: its error is that of the fp32 matrix-core kernel
    (pd3_sparse_conv3d_features_ordered) -- both a few 1e-7 of the magnitude -- and 1000 times below what a 16-bit
    operand format leaves (the fp16 form's test: 2e-4).  Values with a wide spread of magnitudes, every chunk width,
    every output width, all epilogue forms, tile order and raster order, a partial last tile."""
    from paddle3d_amd.ops import sparse_conv3d as sp

    cin, cout, subm, ks, stride, pad, (with_bias, with_bn, with_res, relu), _ = case
    rng = np.random.default_rng(cin * 11 + cout)
    shape = (9, 60, 70)
    coords, feats = _random_sparse(rng, 2, shape, 9000, cin)
    slab = np.stack(np.meshgrid(np.arange(2), np.arange(3, 6), np.arange(20, 45), np.arange(30, 60), indexing="ij"),
                    -1).reshape(-1, 4).astype(np.int32)
    coords = np.unique(np.concatenate([coords, slab]), axis=0)
    feats = (rng.normal(size=(len(coords), cin)) * np.exp(rng.normal(size=(len(coords), cin)))).astype(np.float32)
    w = (rng.normal(size=(*ks, cin, cout)) * np.exp(rng.normal(size=(*ks, cin, cout))) /
         np.sqrt(np.prod(ks) * cin)).astype(np.float32)
    sp.TILE_ORDER = order
    try:
        pl = sp.plan(torch.from_numpy(coords).cuda(), 2, shape, [sp.ConvSpec(ks, stride, pad, subm)])
        idx = pl.indices[0]
        f = torch.from_numpy(feats).cuda().index_select(0, pl.order)
        wt = torch.from_numpy(w).cuda()
        bias = torch.from_numpy(rng.normal(size=cout).astype(np.float32)).cuda() if with_bias else None
        sc = torch.from_numpy(rng.normal(size=cout).astype(np.float32)).cuda() if with_bn else None
        sh = torch.from_numpy(rng.normal(size=cout).astype(np.float32)).cuda() if with_bn else None
        res = torch.from_numpy(rng.normal(size=(idx.n_out, cout)).astype(np.float32)).cuda() if with_res else None
        assert sp.bf16x3_supported(cin, cout, idx.kernel_volume)
        fp32 = sp.features(f, idx, wt, bias, sc, sh, res, relu)
        got = sp.features_bf16x3(f, idx, sp.pack_weight_bf16x3(wt), cin, cout, bias, sc, sh, res, relu)
        again = sp.features_bf16x3(f, idx, sp.pack_weight_bf16x3(wt), cin, cout, bias, sc, sh, res, relu)
    finally:
        sp.TILE_ORDER = True
    # the fp64 reference: gathered rows (a zero row for a missing neighbour) times the offset's matrix
    f64 = torch.cat([f.double(), torch.zeros(1, cin, dtype=torch.float64, device="cuda")])
    w64 = wt.double().reshape(-1, cin, cout)
    nbr = idx.nbr[: idx.n_out].long()
    nbr = torch.where(nbr < 0, torch.full_like(nbr, f.shape[0]), nbr)
    want = torch.zeros(idx.n_out, cout, dtype=torch.float64, device="cuda")
    for k in range(idx.kernel_volume):
        want += f64.index_select(0, nbr[:, k]) @ w64[k]
    if bias is not None:
        want += bias.double()
    if sc is not None:
        want = want * sc.double() + sh.double()
    if res is not None:
        want += res.double()
    if relu:
        want = want.clamp_min(0)
    assert got.dtype == torch.float32 and got.shape == want.shape and torch.equal(got, again)
    assert idx.n_out % 256 != 0 and idx.n_out > 2000
    mag = float(want.abs().max())
    e_x3, e_32 = float((got.double() - want).abs().max()), float((fp32.double() - want).abs().max())
    print(f"bf16x3 error {e_x3:.3g}, fp32 kernel error {e_32:.3g}, magnitude {mag:.3g}")
    assert e_x3 <= max(2.0 * e_32, 2e-7 * mag), (e_x3, e_32, mag)
    assert e_x3 < 2e-6 * mag


def test_sparse_encoder_amp_close_to_fp32_and_voxel_model(oracle):
    """CenterPoint-Voxel under mixed precision on a half-range copy of config 4 (0.075 m voxels on +-28.8 m: 41 x 768 x
    768 cells, 96 x 96 head maps), BatchNorm statistics and heads like a trained net's (synth.trained_like_batchnorm /
    trained_like_heads), detections compared box for box (same frame and class, centre within 0.5 m, score within 0.02):

    * the SPARSE ENCODER in fp16 (this model's own part; dense graph fp32): the map within 2 % of the fp32 map's
      magnitude and at least 99 % of the fp32 graph's boxes have an AMP twin, and the other way round (measured: all);
    * the WHOLE graph in fp16 (`set_amp(True)`: the dense backbone / FPN / head too): the head's regression maps stay
      and the heat maps within 5e-3 of their magnitude, but on THIS model's calibrated random
      heads 11 % of the boxes change cell or vanish: the BEV map is 90 % empty, the top 1 % of the cells the
      calibration stretches over the score range is a slice 1 / 25 .. 1 / 75 of the active cells' logit range, and
      fp16's 2.5e-3 of that range is 6 % of the slice (tools/prof/amp_voxel_twins.py, profiles/r06_amp_voxel_twins.txt;
      the pillar model's 128 x 128 maps hold 99.7 % under the same fp16 kernels, test_amp_graph_close_to_fp32).  The bar
      here is therefore on the maps, with the twin fraction printed and held above 0.75."""
    from paddle3d_amd import centerpoint as cpm
    from paddle3d_amd import nuscenes_bridge as nb
    from paddle3d_amd import synth

    torch.manual_seed(9)
    pcr = [-28.8, -28.8, -5.0, 28.8, 28.8, 3.0]
    model = cpm.centerpoint_voxels_nuscenes(max_num_voxels=(120000, 120000), point_cloud_range=pcr).cuda().eval()
    synth.trained_like_batchnorm(model, 7)
    pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(93 + i) for i in range(4)])).cuda()
    synth.trained_like_heads(model, pts[:2])  # (plain random-init heads: one narrow score band, the comparison is tie-breaking)

    def run():
        bev = model.extract_pillars(pts)
        preds, _ = model.bbox_head(model.dense_forward(bev))
        return bev, preds, model.test_forward(pts)

    bev32, p32, d32 = run()
    model.middle_encoder.amp = True       # the encoder alone
    assert not model.backbone.amp and not model.bbox_head.amp
    bev16, _, d16e = run()
    model.set_amp(True)                   # the whole graph
    assert model.middle_encoder.amp and model.backbone.amp and model.bbox_head.amp
    _, p16, d16 = run()
    model.set_amp(False)
    assert bev16.dtype == torch.float32 and bev16.shape == bev32.shape
    rel = float((bev16 - bev32).abs().max() / bev32.abs().max())
    assert 0 < rel < 2e-2, rel
    miss = nb.unmatched_detections(d16e, d32, score_tol=2e-2)
    back = nb.unmatched_detections(d32, d16e, score_tol=2e-2)
    print("voxel model, fp16 sparse encoder against fp32: map", rel, "fp32 boxes without a twin", miss,
          "AMP boxes without a twin", back)
    assert miss["total"] > 400 and back["total"] > 400 and len(d16e) == len(d32) == 4
    assert miss["unmatched"] <= 0.01 * miss["total"], miss
    assert back["unmatched"] <= 0.01 * back["total"], back
    # the whole graph: the maps
    for a, r in zip(p16, p32):
        for k in r:
            err = float((a[k].float() - r[k].float()).abs().max())
            mag = float(r[k].float().abs().max())
            assert err <= 5e-3 * mag, (k, err, mag)
    miss = nb.unmatched_detections(d16, d32, score_tol=2e-2)
    back = nb.unmatched_detections(d32, d16, score_tol=2e-2)
    print("voxel model, whole graph fp16 against fp32: fp32 boxes without a twin", miss, "AMP boxes without a twin", back)
    assert miss["unmatched"] <= 0.25 * miss["total"] and back["unmatched"] <= 0.25 * back["total"], (miss, back)
