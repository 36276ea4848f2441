"""The oracle's restatements of the reference's PYTHON layers (oracle/pyoracle.py) against golden vectors made by
executing the reference's own source through a paddle->torch shim (tests/golden/make_python_golden.py): pins
E1 / E2 / E3 / S1 / L / R1 / B (index prep, frustum geometry) / D1 of SURVEY.md section 8 to executed reference
code.  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from state_util import rebuild_state  # noqa: E402


@pytest.fixture(scope="module")
def pg():
    return np.load(os.path.join(HERE, "golden", "python_layers.npz"))


def _pfn_params(state, prefix, n):
    out = []
    for i in range(n):
        p = f"{prefix}.{i}"
        out.append(dict(weight=state[f"{p}.linear.weight"], gamma=state[f"{p}.norm.weight"], beta=state[f"{p}.norm.bias"],
                        mean=state[f"{p}.norm._mean"], var=state[f"{p}.norm._variance"]))
    return out


def test_shim_param_generator_in_sync():
    import paddle_shim

    rng_a, rng_b = np.random.default_rng(1), np.random.default_rng(1)
    from state_util import synth_param

    for key, shape in (("a.weight", (3, 4)), ("a._variance", (5,)), ("b.bias", (2,)), ("c.weight", (8, 4, 3, 3))):
        np.testing.assert_array_equal(paddle_shim.synth_param(key, shape, rng_a), synth_param(key, shape, rng_b))


@pytest.mark.parametrize("tag,layers,vs,pcr", [
    ("pfn2", 2, (0.2, 0.2, 8.0), (-51.2, -51.2, -5.0, 51.2, 51.2, 3.0)),
    ("pfn1", 1, (0.2, 0.2, 8.0), (-51.2, -51.2, -5.0, 51.2, 51.2, 3.0))])
def test_pfn_restatement(oracle, pg, tag, layers, vs, pcr):
    state = rebuild_state(pg[f"{tag}_keys"], pg[f"{tag}_shapes"], 11)
    out = oracle.pfn_forward_torch(pg[f"{tag}_voxels"], pg[f"{tag}_num_points"], pg[f"{tag}_coors"],
                                   _pfn_params(state, "pfn_layers", layers), vs, pcr)
    np.testing.assert_allclose(out, pg[f"{tag}_out"], rtol=1e-5, atol=2e-5)


def test_hard_vfe_and_voxel_mean_restatements(oracle, pg):
    state = rebuild_state(pg["vfe_keys"], pg["vfe_shapes"], 12)
    out = oracle.hard_vfe_forward_torch(pg["vfe_voxels"], pg["vfe_num_points"], pg["vfe_coors"],
                                        _pfn_params(state, "vfe_layers", 2), (0.25, 0.25, 8.0),
                                        (-50.0, -50.0, -5.0, 50.0, 50.0, 3.0))
    np.testing.assert_allclose(out, pg["vfe_out"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(oracle.voxel_mean(pg["vmean_voxels"], pg["vmean_num_points"]), pg["vmean_out"],
                               rtol=1e-6, atol=1e-6)


def test_scatter_restatements(oracle, pg):
    for fn in (oracle.pillar_scatter, oracle.pillar_scatter_numpy):
        np.testing.assert_array_equal(fn(pg["scatter_feats"], pg["scatter_coors"], 2, 32, 48), pg["scatter_out"])


def test_lss_voxel_pooling_restatement(oracle, pg):
    out = oracle.lss_voxel_pooling_numpy(pg["lss_geom"], pg["lss_x"], np.array([0.5, 0.5, 20.0], np.float32),
                                         np.array([-9.75, -9.75, 0.0], np.float32), [40, 40, 1])
    # the cumsum trick's result depends on the summation order of one global running total: fp32 noise only
    np.testing.assert_allclose(out, pg["lss_out"], rtol=0, atol=2e-4)
    np.testing.assert_array_equal(out != 0, pg["lss_out"] != 0)


def test_lss_exact_sums_against_the_reference_output(oracle, pg):
    """The float64 per-cell sums (the yardstick of tests/test_lss_c5_gpu.py) against the output of the reference's own
    voxel_pooling executed through the shim: same cells written, values within the cumsum trick's fp32 noise."""
    dx, bx = np.array([0.5, 0.5, 20.0], np.float32), np.array([-9.75, -9.75, 0.0], np.float32)
    exact = oracle.lss_voxel_pooling_exact(pg["lss_geom"], pg["lss_x"], dx, bx, [40, 40, 1])
    assert exact.dtype == np.float64 and exact.shape == pg["lss_out"].shape
    np.testing.assert_allclose(exact, pg["lss_out"], rtol=0, atol=2e-4)
    np.testing.assert_array_equal(exact != 0, pg["lss_out"] != 0)
    # and brute force on a sub-sample of cells: the reduceat grouping is the per-cell sum
    B, C = pg["lss_x"].shape[0], pg["lss_x"].shape[-1]
    g = ((pg["lss_geom"] - (bx - dx / 2.0)) / dx).astype(np.int64)
    for b in range(B):
        gb, xb = g[b].reshape(-1, 3), pg["lss_x"][b].reshape(-1, C).astype(np.float64)
        for (cx, cy) in [(3, 7), (20, 20), (39, 0)]:
            m = (gb[:, 0] == cx) & (gb[:, 1] == cy) & (gb[:, 2] == 0)
            np.testing.assert_allclose(exact[b, :, 0, cx, cy], xb[m].sum(0), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("tag,pre,post", [("a", 300, 80), ("b", None, None), ("c", 50, 500)])
def test_rotate_nms_pcdet_restatement(oracle, pg, tag, pre, post):
    for kind in (["port", "ref"] if oracle.have_ref() else ["port"]):
        sel = oracle.rotate_nms_pcdet_numpy(pg["rnms_boxes"], pg["rnms_scores"], 0.2, pre, post, kind=kind)
        np.testing.assert_array_equal(sel, pg[f"rnms_sel_{tag}"])


def test_frustum_and_index_prep_restatements(oracle, pg):
    grid = dict(x=[-51.2, 51.2, 0.8], y=[-51.2, 51.2, 0.8], z=[-5, 3, 8], depth=[1.0, 60.0, 0.5])
    fr = oracle.create_frustum_numpy(grid["depth"], (64, 176), 16)
    assert fr.shape == (118, 4, 11, 3)
    cams = {k[len("prep_cam_"):]: pg[k] for k in pg.files if k.startswith("prep_cam_")}
    coor = oracle.get_lidar_coor_numpy(fr, cams["rots"], cams["trans"], cams["cam2imgs"], cams["post_rots"],
                                       cams["post_trans"], cams["bda"])
    np.testing.assert_allclose(coor, pg["prep_coor"], rtol=2e-5, atol=2e-4)
    lower = np.array([-51.2, -51.2, -5.0], np.float32)
    step = np.array([0.8, 0.8, 8.0], np.float32)
    size = np.array([(51.2 + 51.2) / 0.8, (51.2 + 51.2) / 0.8, (3 + 5) / 8], np.float32)
    got = oracle.voxel_pooling_prepare_v2_numpy(pg["prep_coor"], lower, step, size)
    for a, name in zip(got, ("ranks_bev", "ranks_depth", "ranks_feat", "interval_starts", "interval_lengths")):
        np.testing.assert_array_equal(a, pg[f"prep_{name}"], err_msg=name)


def _dense_models():
    from paddle3d_amd import centerpoint as cpm

    tasks = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "construction_vehicle"])]
    backbone = cpm.SecondBackbone(16, (64, 64, 128), (1, 2, 1), (2, 2, 2))
    neck = cpm.SecondFPN((64, 64, 128), (64, 64, 64), (0.5, 1, 2), use_conv_for_no_stride=True)
    head = cpm.CenterHead(192, tasks, dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)))
    return backbone, neck, head


def _load_dense(pg, tmp_path=None):
    from paddle3d_amd import checkpoint

    mods = _dense_models()
    for name, mod, seed in zip(("backbone", "neck", "head"), mods, (21, 22, 23)):
        state = rebuild_state(pg[f"dense_{name}_keys"], pg[f"dense_{name}_shapes"], seed)
        if tmp_path is not None:  # through the .pdparams wire format
            path = os.path.join(str(tmp_path), f"{name}.pdparams")
            checkpoint.save_pdparams(state, path)
            state = path
        assert checkpoint.load_paddle_state_dict(mod, state, strict=True) == []
        mod.eval()
    return mods


SUB = (slice(None), slice(None), slice(3, None, 8), slice(5, None, 8))


def test_dense_graph_structure_and_checkpoint_loader(oracle, pg, tmp_path):
    """Our SecondBackbone / SecondFPN / CenterHead containers take the reference's state dict (names, layouts,
    `.pdparams` pickle) and, run as plain torch layers, reproduce the reference's forward."""
    backbone, neck, head = _load_dense(pg, tmp_path)
    x = torch.from_numpy(np.random.default_rng(31).normal(size=(1, 16, 512, 512)).astype(np.float32))
    with torch.no_grad():
        feats = oracle.second_fpn_torch(neck, oracle.second_backbone_torch(backbone, x))
        preds, shared = oracle.center_head_torch(head, feats)
    np.testing.assert_allclose(feats.numpy()[SUB], pg["dense_neck_out_sub"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(shared.numpy()[SUB], pg["dense_shared_sub"], rtol=1e-4, atol=1e-4)
    for t, pd in enumerate(preds):
        for name, v in pd.items():
            np.testing.assert_allclose(v.numpy()[SUB], pg[f"dense_task{t}_{name}_sub"], rtol=1e-4, atol=1e-4,
                                       err_msg=f"task {t} {name}")


def test_loader_is_strict():
    from paddle3d_amd import centerpoint as cpm
    from paddle3d_amd import checkpoint

    m = cpm.SecondBackbone(16, (64,), (1,), (2,))
    good = {k.replace("running_mean", "_mean").replace("running_var", "_variance"): v.numpy()
            for k, v in m.state_dict().items() if not k.endswith("num_batches_tracked")}
    assert checkpoint.load_paddle_state_dict(m, good) == []
    with pytest.raises(RuntimeError, match="without a place"):
        checkpoint.load_paddle_state_dict(m, dict(good, **{"blocks.0.9.weight": np.zeros(3, np.float32)}))
    bad = dict(good)
    bad.pop("blocks.0.1._mean")
    with pytest.raises(RuntimeError, match="without a value"):
        checkpoint.load_paddle_state_dict(m, bad)
    with pytest.warns(UserWarning):
        assert checkpoint.load_paddle_state_dict(m, bad, strict=False) == []


def test_sparse_encoder_takes_reference_names():
    """CenterPoint-Voxel: a synthetic state dict with the reference's names / layouts (sparse_resnet.py:126-164;
    Conv3D [kd, kh, kw, in, out], BatchNorm _mean / _variance) places every entry."""
    from paddle3d_amd import checkpoint
    from paddle3d_amd import sparse as S

    net = S.SparseResNet3D(5)
    rng = np.random.default_rng(0)
    state = {}
    for k, v in net.state_dict().items():
        if k.endswith("num_batches_tracked"):
            continue
        state[k.replace("running_mean", "_mean").replace("running_var", "_variance")] = \
            rng.normal(size=tuple(v.shape)).astype(np.float32)
    assert state["conv2.0.weight"].shape == (3, 3, 3, 16, 32) and "conv2.3.bn1._variance" in state
    assert checkpoint.load_paddle_state_dict(net, state) == []
    np.testing.assert_array_equal(net.conv2[0].weight.detach().numpy(), state["conv2.0.weight"])
    np.testing.assert_array_equal(net.conv2[3].bn1.running_var.numpy(), state["conv2.3.bn1._variance"])


def test_centerpoint_postprocess_orchestration_vs_reference_python(oracle):
    """P / P0: the oracle's statement of the centerpoint_postprocess operator against the reference's own Python
    post-processing (CenterHead.predict -> single_post_processing, center_head.py:341-510, executed through the
    shim by tests/golden/make_predict_golden.py): same rows in the same order, same labels (per-task offsets), boxes
    and scores to an ulp of exp / atan2.  (The centre-range test is the one step the two reference paths apply to
    different values; the fixture's range keeps it always true.)"""
    import make_predict_golden as G

    gold = np.load(os.path.join(HERE, "golden", "python_predict.npz"))
    maps, cfg = G.head_maps(), G.CFG
    offsets = np.concatenate([[0], np.cumsum([t["num_class"] for t in G.TASKS])[:-1]]).astype(int).tolist()
    for b in range(G.BATCH):
        tasks = [{k: v[b:b + 1] for k, v in t.items()} for t in maps]
        rb, rs, rl = oracle.centerpoint_postprocess(
            tasks, cfg["voxel_size"] + [8.0], cfg["point_cloud_range"] + [0.0] * 4, cfg["post_center_limit_range"],
            offsets, cfg["down_ratio"], cfg["score_threshold"], cfg["nms"]["nms_iou_threshold"],
            cfg["nms"]["nms_pre_max_size"], cfg["nms"]["nms_post_max_size"], True)
        np.testing.assert_array_equal(rl, gold[f"labels_{b}"])
        np.testing.assert_allclose(rs, gold[f"scores_{b}"], rtol=0, atol=2e-7)
        np.testing.assert_allclose(rb, gold[f"boxes_{b}"], rtol=2e-6, atol=2e-6)


def test_pdparams_loader_refuses_code(tmp_path):
    """A .pdparams is a pickle; the loader unpickles numpy arrays only (a checkpoint from the net cannot run code)."""
    import pickle

    import numpy as np
    import pytest

    from paddle3d_amd import checkpoint

    good = tmp_path / "good.pdparams"
    checkpoint.save_pdparams({"a.weight": np.arange(6, dtype=np.float32).reshape(2, 3)}, str(good))
    assert checkpoint.load_pdparams(str(good))["a.weight"].shape == (2, 3)

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > /dev/null",))

    bad = tmp_path / "bad.pdparams"
    with open(bad, "wb") as f:
        pickle.dump({"a.weight": Evil()}, f, protocol=2)
    with pytest.raises(pickle.UnpicklingError, match="refusing to load"):
        checkpoint.load_pdparams(str(bad))


def test_packed_weights_follow_in_place_and_child_updates():
    """The inference caches (folded BatchNorm, packed kernel layouts) are keyed on (storage, version, device) of the
    module's parameters: an in-place write or a load_state_dict on a CHILD module must not leave stale packed weights
    behind (round-2 advice: they were only dropped by train() / _apply() / load on the caching module itself)."""
    import torch

    from paddle3d_amd import centerpoint as cpm

    head = cpm.centerpoint_pillars_nuscenes(max_num_voxels=(100, 100)).bbox_head.eval()
    first = head._plan()["bf"].clone()
    assert head._plan() is head._cache  # unchanged parameters: the same plan object
    task = head.tasks[0]
    name = list(task.heads)[0]
    with torch.no_grad():
        getattr(task, name)[1].bias.add_(1.0)  # in-place write on a grandchild
    second = head._plan()["bf"]
    assert not torch.equal(first, second)
    sd = {k: v + 0.5 if k.endswith("1.bias") else v for k, v in task.state_dict().items()}
    task.load_state_dict(sd)  # reload of a child, not of the head
    assert not torch.equal(second, head._plan()["bf"])
