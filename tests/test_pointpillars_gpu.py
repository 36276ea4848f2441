"""PointPillars (KITTI) path on the GPU: the transposed-convolution-by-4 FPN level, the fused 1x1 SSD head, the
ssd_postprocess op against the reference-Python goldens and the oracle, and the whole model end to end."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_ssd_golden as G  # noqa: E402
from state_util import rebuild_state  # noqa: E402

from paddle3d_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sg():
    return np.load(os.path.join(HERE, "golden", "python_ssd.npz"))


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("cin,cout,h,wv", [(256, 128, 62, 54), (32, 16, 5, 8), (64, 4, 9, 3)])
def test_patch_deconv4_matches_torch(cin, cout, h, wv):
    """mode 3 = Conv2DTranspose kernel 4 stride 4 (+ bias + ReLU) at a channel offset, input rows zero-padded to a
    multiple of 4 (62 x 54 is PointPillars-KITTI's third backbone stage)."""
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(cin + wv)
    pitch = conv.pitch4(wv)
    x = torch.zeros(2, cin, h, pitch)
    x[..., :wv] = torch.randn(2, cin, h, wv, generator=g)
    w = torch.randn(cin, cout, 4, 4, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv_transpose2d(x[..., :wv], w, b, stride=4))
    assert conv.patch_mode(w, 4, True) == 3 and conv.patch_supported(3, cin, cout, h, pitch)
    out = torch.full((2, cout + 24, 4 * h, 4 * wv), 7.0, device="cuda")
    conv.patch_conv_bias_relu(x.cuda(), conv.pack_patch_weight(w, 3, True).cuda(), b.cuda(), 3, cout, out, 16, relu=True,
                              w_valid=wv)
    assert torch.all(out[:, :16] == 7.0) and torch.all(out[:, 16 + cout:] == 7.0)
    assert (out[:, 16:16 + cout].cpu() - ref).abs().max() < 5e-5


@pytest.mark.parametrize("cin,cout,h,w", [(384, 20, 248, 216), (64, 70, 6, 10), (32, 1, 4, 4)])
def test_patch_conv1x1_any_cout(cin, cout, h, w):
    """mode 1 with a row count that is not a multiple of 64 (the fused SSD head: 2 + 14 + 4 maps), no activation."""
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(cout)
    x = torch.randn(2, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b)
    out = torch.full((2, cout + 2, h, w), 7.0, device="cuda")
    conv.patch_conv_bias_relu(x.cuda(), conv.pack_patch_weight(wt, 1, False).cuda(), b.cuda(), 1, cout, out, 1, relu=False)
    assert torch.all(out[:, 0] == 7.0) and torch.all(out[:, cout + 1] == 7.0)
    assert (out[:, 1:cout + 1].cpu() - ref).abs().max() < 5e-5


def _head(c, tag=None, sg=None):
    from paddle3d_amd import checkpoint
    from paddle3d_amd.pointpillars import AnchorGenerator, SSDHead

    head = SSDHead(num_classes=c["num_classes"], feature_channels=c["channels"],
                   num_anchor_per_loc=2 * len(c["anchor_configs"]), **c["head"])
    if sg is not None:
        state = rebuild_state(sg[f"{tag}_head_keys"], sg[f"{tag}_head_shapes"], 40 + ord(tag))
        assert checkpoint.load_paddle_state_dict(head, state) == []
    gen = AnchorGenerator(2, c["pcr"], c["vs"], c["anchor_configs"], 1)
    return head.cuda().eval(), gen.cuda()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_ssd_head_forward_vs_reference_python(sg, tag):
    """SSDHead.forward (three 1x1 convolutions as one GEMM, NCHW map viewed as [B, A, width]) vs the reference's own
    forward; parameters through the .pdparams loader."""
    c = G.CASES[tag]
    head, gen = _head(c, tag, sg)
    feats = G.features(tag, c, gen.grid_size[0], gen.grid_size[1])
    out = head(_cuda(feats))
    for name in ("cls", "box", "dir"):
        got = out[f"{name}_preds"].cpu().numpy()
        assert got.shape == sg[f"{tag}_fw_{name}"].shape
        assert np.abs(got - sg[f"{tag}_fw_{name}"]).max() < 2e-5


def _map_from_preds(cls, box, dirp, fh, fw, apl):
    """[B, A, width] predictions -> the NCHW head map the convolutions would have written (cls | box | dir)."""
    def group(p):
        b, _, width = p.shape
        return p.reshape(b, fh, fw, apl * width).transpose(0, 3, 1, 2)

    return np.ascontiguousarray(np.concatenate([group(cls), group(box), group(dirp)], 1))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_ssd_postprocess_vs_reference_python(oracle, sg, tag):
    """The device op on the golden inputs: same labels and rows as the reference's post_process per frame (boxes /
    scores to an ulp of exp), zero rows where the reference returns its `_box_empty` marker; and bit-for-bit the
    selection of the oracle restatement."""
    c = G.CASES[tag]
    head, gen = _head(c)
    a = gen.anchors.shape[0]
    cls, box, dirp = G.head_outputs(tag, c, a)
    fh, fw = gen.feature_map_size
    m = _cuda(_map_from_preds(cls, box, dirp, fh, fw, gen.num_anchors_per_loc))
    # padding rows (batch -1) and a shuffled order must not matter
    co = sg[f"{tag}_coords"]
    rng = np.random.default_rng(5)
    co = np.concatenate([co, np.full((7, 4), -1, np.int32)])[rng.permutation(co.shape[0] + 7)]
    dets = head.post_process(m, gen, _cuda(co.astype(np.int32)))
    sorted_dets = head.post_process(m, gen, _cuda(co.astype(np.int32)), full_sort=True)  # the reference's own selection
    assert len(dets) == c["batch"]
    for d, e in zip(dets, sorted_dets):
        for k in ("box3d_lidar", "scores", "label_preds"):
            assert torch.equal(d[k], e[k])
    for b, d in enumerate(dets):
        gb, gs, gl = sg[f"{tag}_out_boxes_{b}"], sg[f"{tag}_out_scores_{b}"], sg[f"{tag}_out_labels_{b}"]
        if gs.tolist() == [-1.0]:
            assert d["scores"].numel() == 0
            continue
        np.testing.assert_array_equal(d["label_preds"].cpu().numpy(), gl)
        np.testing.assert_allclose(d["box3d_lidar"].cpu().numpy(), gb, rtol=1e-6, atol=4e-6)
        np.testing.assert_allclose(d["scores"].cpu().numpy(), gs, rtol=0, atol=3e-7)
    # device-only form: counts on the device, the marker row in row 0 of an empty frame
    bb, ss, ll, nn = head.post_process(m, gen, _cuda(co.astype(np.int32)), device_only=True)
    assert nn.dtype == torch.int32 and nn.is_cuda
    if tag == "b":
        assert int(nn[1]) == 0 and float(ss[1, 0]) == -1.0 and int(ll[1, 0]) == -1 and float(bb[1, 0].abs().sum()) == 0


@pytest.mark.parametrize("batch,seed,thr,pre", [(3, 1, 0.05, 1000), (2, 2, 0.6, 1000), (2, 3, 0.02, 1500),
                                                 (2, 5, 0.6, 4096)])
def test_ssd_postprocess_kitti_size_vs_oracle(oracle, batch, seed, thr, pre):
    """Full KITTI head map (248 x 216 x 2 = 107 136 anchors per frame) against the oracle restatement on the same
    inputs: identical rows, labels and order; boxes / scores within an ulp of exp.  Top-K selection kernel and full
    sort; a pre-NMS cap beyond the selection kernel's 1024 (which takes the sort by itself); a cap of 4096 with a
    few hundred candidates per frame (the NMS sweep's in-LDS path under a capacity it was not sized for once)."""
    from paddle3d_amd.pointpillars import KITTI_CAR_ANCHORS, AnchorGenerator, SSDHead

    pcr, vs = list(synth.KITTI_RANGE), list(synth.KITTI_PILLAR)
    gen = AnchorGenerator(2, pcr, vs, KITTI_CAR_ANCHORS, 1).cuda()
    lim = [0.0, -39.68, -5.0, 69.12, 39.68, 5.0]
    head = SSDHead(1, 64, 2, nms_score_threshold=thr, nms_pre_max_size=pre, nms_post_max_size=300,
                   nms_iou_threshold=0.5, prediction_center_limit_range=lim).cuda().eval()
    fh, fw = gen.feature_map_size
    assert (fh, fw) == (248, 216) and gen.anchors.shape[0] == 107136
    rng = np.random.default_rng(seed)
    a = gen.anchors.shape[0]
    cls = rng.normal(-4.0, 2.0, (batch, a, 1)).astype(np.float32)
    box = rng.normal(0, 0.3, (batch, a, 7)).astype(np.float32)
    dirp = rng.normal(0, 1, (batch, a, 2)).astype(np.float32)
    coords = []
    for b in range(batch):
        vox, co, npv, nv = oracle.hard_voxelize(synth.kitti_frame(300 + seed * 10 + b), vs, pcr, 32, 40000)
        coords.append(np.concatenate([np.full((nv, 1), b, np.int32), co[:nv]], 1))
    co = np.concatenate(coords).astype(np.int32)
    m = _cuda(_map_from_preds(cls, box, dirp, fh, fw, 2))
    dets = head.post_process(m, gen, _cuda(co))
    for d, e in zip(dets, head.post_process(m, gen, _cuda(co), full_sort=True)):  # both selections, same bytes
        for k in ("box3d_lidar", "scores", "label_preds"):
            assert torch.equal(d[k], e[k])
    an, bv = gen.anchors.cpu().numpy(), gen.anchors_bv.cpu().numpy().astype(np.int64)
    for b in range(batch):
        mask = oracle.ssd_anchor_mask_numpy(co[co[:, 0] == b][:, 1:], bv, gen.grid_size, 1.0)
        assert 0 < mask.sum() < a
        rb, rs, rl = oracle.ssd_post_process_frame_numpy(box[b], cls[b], dirp[b], an, mask, thr, lim, pre, 300, 0.5)
        assert rs.shape[0] > 10
        np.testing.assert_array_equal(dets[b]["label_preds"].cpu().numpy(), rl)
        np.testing.assert_allclose(dets[b]["scores"].cpu().numpy(), rs, rtol=0, atol=3e-7)
        np.testing.assert_allclose(dets[b]["box3d_lidar"].cpu().numpy(), rb, rtol=1e-6, atol=4e-6)


def test_ssd_postprocess_variants_and_edges(oracle):
    """Head variants the KITTI config does not use (background logit in front of the classes, no direction
    classifier, no centre range) against the oracle, and the degenerate inputs: no pillars at all, only padding rows,
    a zero pre-NMS cap."""
    from paddle3d_amd.pointpillars import AnchorGenerator, SSDHead

    c = G.CASES["b"]
    gen = AnchorGenerator(2, c["pcr"], c["vs"], c["anchor_configs"], 1).cuda()
    fh, fw = gen.feature_map_size
    apl, a = gen.num_anchors_per_loc, gen.anchors.shape[0]
    rng = np.random.default_rng(9)
    nx, ny = gen.grid_size
    cells = rng.choice(nx * ny, 400, replace=False)
    co = np.zeros((400, 4), np.int32)
    co[:, 0], co[:, 2], co[:, 3] = rng.integers(0, 2, 400), cells // nx, cells % nx
    an, bv = gen.anchors.cpu().numpy(), gen.anchors_bv.cpu().numpy().astype(np.int64)
    for bg_zero, use_dir, lim in ((False, True, None), (True, False, c["head"]["prediction_center_limit_range"]),
                                  (False, False, None)):
        ncls = c["num_classes"]
        width = ncls if bg_zero else ncls + 1
        head = SSDHead(ncls, 32, apl, encode_background_as_zeros=bg_zero, use_direction_classifier=use_dir,
                       nms_score_threshold=0.3, nms_pre_max_size=150, nms_post_max_size=25, nms_iou_threshold=0.3,
                       prediction_center_limit_range=lim).cuda().eval()
        cls = rng.normal(-1.0, 2.0, (2, a, width)).astype(np.float32)
        box = rng.normal(0, 0.35, (2, a, 7)).astype(np.float32)
        dirp = rng.normal(0, 1, (2, a, 2)).astype(np.float32)
        groups = [cls, box] + ([dirp] if use_dir else [])
        m = _cuda(np.ascontiguousarray(np.concatenate(
            [g.reshape(2, fh, fw, apl * g.shape[2]).transpose(0, 3, 1, 2) for g in groups], 1)))
        dets = head.post_process(m, gen, _cuda(co))
        for b in range(2):
            mask = oracle.ssd_anchor_mask_numpy(co[co[:, 0] == b][:, 1:], bv, gen.grid_size, 1.0)
            rb, rs, rl = oracle.ssd_post_process_frame_numpy(box[b], cls[b], dirp[b] if use_dir else None, an, mask, 0.3,
                                                             lim, 150, 25, 0.3, encode_background_as_zeros=bg_zero)
            assert rs.shape[0] > 3 and rs[0] >= 0
            np.testing.assert_array_equal(dets[b]["label_preds"].cpu().numpy(), rl)
            np.testing.assert_allclose(dets[b]["scores"].cpu().numpy(), rs, rtol=0, atol=3e-7)
            np.testing.assert_allclose(dets[b]["box3d_lidar"].cpu().numpy(), rb, rtol=1e-6, atol=4e-6)
    head = SSDHead(c["num_classes"], 32, apl, nms_pre_max_size=150, nms_post_max_size=25).cuda().eval()
    cls = rng.normal(2.0, 1.0, (2, a, c["num_classes"])).astype(np.float32)
    m = _cuda(_map_from_preds(cls, rng.normal(0, 0.3, (2, a, 7)).astype(np.float32),
                              rng.normal(0, 1, (2, a, 2)).astype(np.float32), fh, fw, apl))
    none = torch.zeros((0, 4), dtype=torch.int32, device="cuda")
    pad = torch.full((5, 4), -1, dtype=torch.int32, device="cuda")
    for coors in (none, pad):  # no pillar: no anchor passes the area test, every frame is empty
        dets = head.post_process(m, gen, coors)
        assert [d["scores"].numel() for d in dets] == [0, 0]
    assert all(d["scores"].numel() > 0 for d in head.post_process(m, gen, _cuda(co)))
    head0 = SSDHead(c["num_classes"], 32, apl, nms_pre_max_size=0, nms_post_max_size=25).cuda().eval()
    assert [d["scores"].numel() for d in head0.post_process(m, gen, _cuda(co))] == [0, 0]  # order[:0]: nothing to keep


def _randomise_bn(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)


@pytest.mark.parametrize("config", ["car", "cyclist_pedestrian"])
def test_pointpillars_kitti_end_to_end_vs_oracle(oracle, config):
    """Configs 1 / 2 end to end on two frames: device voxelize -> PFN -> scatter -> SECOND -> FPN (1 / 2 / 4 transposed
    convolutions) -> fused SSD head -> ssd_postprocess, against the oracle pipeline (reference voxelizer, torch-CPU
    statement of the layers, NumPy post-processing).  The cyclist / pedestrian config has 100 points per pillar, a
    stride-1 first backbone block (head map = pillar grid, 248 x 296, third stage 62 x 74 in rows of pitch 76), two
    classes and four anchors per location (a 44-map head: both MFMA row blocks of the 1x1 GEMM)."""
    from paddle3d_amd import pointpillars as ppm

    torch.manual_seed(4)
    make = ppm.pointpillars_kitti_car if config == "car" else ppm.pointpillars_kitti_cyclist_pedestrian
    model = make().cuda().eval()
    _randomise_bn(model)
    with torch.no_grad():
        model.head.cls_head.bias.fill_(-1.5)
    pts = np.stack([synth.kitti_frame(500), synth.kitti_frame(501)])
    dets = model.test_forward(torch.from_numpy(pts).cuda())
    assert len(dets) == 2
    cpu = make().eval()
    cpu.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    gen, h = cpu.anchor_generator, cpu.head
    an, bv = gen.anchors.numpy(), gen.anchors_bv.numpy().astype(np.int64)
    vs, pcr = cpu.voxelizer.voxel_size, cpu.voxelizer.point_cloud_range
    p_max, v_max = cpu.voxelizer.max_num_points_in_voxel, cpu.voxelizer.max_num_voxels[1]
    nx, ny = gen.grid_size
    fh, fw = gen.feature_map_size
    apl, ncls = h.num_anchor_per_loc, h.num_classes
    c_cls, c_box = apl * ncls, apl * 7
    for b in range(2):
        vox, co, npv, nv = oracle.hard_voxelize(pts[b], vs, pcr, p_max, v_max)
        c4 = np.concatenate([np.zeros((nv, 1), np.int32), co[:nv]], 1)
        params = [dict(weight=l.linear.weight.t().detach().numpy(), gamma=l.norm.weight.detach().numpy(),
                       beta=l.norm.bias.detach().numpy(), mean=l.norm.running_mean.numpy(),
                       var=l.norm.running_var.numpy()) for l in cpu.pillar_encoder.pfn_layers]
        feats = oracle.pfn_forward_torch(vox[:nv], npv[:nv], c4, params, vs, pcr)
        bev = torch.from_numpy(oracle.pillar_scatter(feats, c4, 1, ny, nx))
        with torch.no_grad():
            x = oracle.second_fpn_torch(cpu.neck, oracle.second_backbone_torch(cpu.backbone, bev))
            assert x.shape == (1, 384, fh, fw)
            ref_map = torch.cat([h.cls_head(x), h.box_head(x), h.dir_head(x)], 1)[0].numpy()
        # the device graph's head map for this frame
        voxels, coors, npv_d, _ = model.voxelizer(torch.from_numpy(pts[b:b + 1]).cuda())
        v = voxels.shape[1]
        f = model.pillar_encoder(voxels.view(v, p_max, 4), npv_d.view(v), coors.view(v, 4))
        gx = model.neck(model.backbone(model.scatter(f, coors.view(v, 4), 1)))
        got_map = model.head.head_map(gx)[0].cpu().numpy()
        assert got_map.shape == ref_map.shape == (c_cls + c_box + apl * 2, fh, fw)
        assert np.abs(got_map - ref_map).max() < 1e-3  # the north star's bar on fp32 features
        # post-processing of the CPU map by the oracle vs the device detections (maps differ by ~1e-5: compare as
        # sets, every strong reference detection has a twin)
        mask = oracle.ssd_anchor_mask_numpy(co[:nv], bv, gen.grid_size, 1.0)
        pr = ref_map.reshape(ref_map.shape[0], -1).T  # [hw, channels]
        cls = pr[:, :c_cls].reshape(-1, ncls)
        box = pr[:, c_cls:c_cls + c_box].reshape(-1, 7)
        dirp = pr[:, c_cls + c_box:].reshape(-1, 2)
        rb, rs, rl = oracle.ssd_post_process_frame_numpy(box, cls, dirp, an, mask, h.nms_score_threshold,
                                                         h.pred_center_limit_range, h.nms_pre_max_size,
                                                         h.nms_post_max_size, h.nms_iou_threshold)
        gb, gs, gl = (dets[b][k].cpu().numpy() for k in ("box3d_lidar", "scores", "label_preds"))
        assert rs.shape[0] > 20 and gs.shape[0] > 20
        strong = rs > h.nms_score_threshold + 1e-3
        matched = 0
        for i in np.nonzero(strong)[0]:
            d = np.abs(gb[:, :2] - rb[i, :2]).sum(1) + (gl != rl[i]) * 1e3
            j = int(np.argmin(d))
            if d[j] < 1e-2 and abs(gs[j] - rs[i]) < 1e-3:
                matched += 1
        assert matched >= 0.95 * strong.sum(), (matched, int(strong.sum()), len(gs))


def test_pointpillars_batched_equals_single_and_prevoxelized():
    """Frames are independent; the reference's pre-voxelized entry gives the same detections as the device path."""
    from paddle3d_amd import pointpillars as ppm

    torch.manual_seed(5)
    model = ppm.pointpillars_kitti_car().cuda().eval()
    with torch.no_grad():
        model.head.cls_head.bias.fill_(-1.5)
    pts = torch.from_numpy(np.stack([synth.kitti_frame(510 + i) for i in range(3)])).cuda()
    both = model.test_forward(pts)
    for b in range(3):
        one = model.test_forward(pts[b:b + 1])[0]
        for k in ("box3d_lidar", "scores", "label_preds"):
            assert torch.equal(both[b][k], one[k])
    voxels, coors, npv, nv = model.voxelizer(pts)
    keep = coors.view(-1, 4)[:, 0] >= 0  # the reference's samples hold the occupied pillars only
    pre = model.test_forward_voxels(voxels.view(-1, 32, 4)[keep], coors.view(-1, 4)[keep].contiguous(),
                                    npv.view(-1)[keep], 3)
    for b in range(3):
        for k in ("box3d_lidar", "scores", "label_preds"):
            assert torch.equal(both[b][k], pre[b][k])
    with pytest.raises(RuntimeError):
        model.train()
        model.test_forward(pts)
