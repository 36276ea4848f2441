"""PointPillars SSD head path on the CPU: the oracle's NumPy restatement (oracle/pyoracle.py ssd_*) and the product's
host-side AnchorGenerator against golden vectors recorded from the reference's own Python
(tests/golden/python_ssd.npz, made by tests/golden/make_ssd_golden.py through the paddle shim)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_ssd_golden as G  # noqa: E402  (case table + seeded input builders; nothing runs at import)


@pytest.fixture(scope="module")
def sg():
    return np.load(os.path.join(HERE, "golden", "python_ssd.npz"))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_anchor_restatements(oracle, sg, tag):
    c = G.CASES[tag]
    an, bv, (fh, fw), grid = oracle.ssd_anchors_numpy(c["pcr"], c["vs"], c["anchor_configs"], 2)
    np.testing.assert_array_equal(an, sg[f"{tag}_anchors"])
    np.testing.assert_array_equal(bv, sg[f"{tag}_anchors_bv"])
    # the product's host-side generator (paddle3d_amd/pointpillars.py) builds the same arrays
    from paddle3d_amd.pointpillars import AnchorGenerator

    gen = AnchorGenerator(2, c["pcr"], c["vs"], c["anchor_configs"], 1)
    np.testing.assert_array_equal(gen.anchors.numpy(), sg[f"{tag}_anchors"])
    np.testing.assert_array_equal(gen.anchors_bv.numpy(), sg[f"{tag}_anchors_bv"])
    assert gen.feature_map_size == (fh, fw) and gen.grid_size == grid
    assert gen.num_anchors_per_loc == 2 * len(c["anchor_configs"])
    co = sg[f"{tag}_coords"]
    for b in range(c["batch"]):
        m = oracle.ssd_anchor_mask_numpy(co[co[:, 0] == b][:, 1:], bv, grid, 1.0)
        np.testing.assert_array_equal(m, sg[f"{tag}_mask_{b}"])


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("kind", ["port", "ref"])
def test_post_process_restatement(oracle, sg, tag, kind):
    if kind == "ref" and not oracle.have_ref():
        pytest.skip("oracle/_ref not built here")
    c = G.CASES[tag]
    an, bv, _, grid = oracle.ssd_anchors_numpy(c["pcr"], c["vs"], c["anchor_configs"], 2)
    cls, box, dirp = G.head_outputs(tag, c, an.shape[0])
    np.testing.assert_allclose(oracle.ssd_box_decode_numpy(box[0], an), sg[f"{tag}_decoded_0"], rtol=2e-7, atol=1e-6)
    co, h = sg[f"{tag}_coords"], c["head"]
    for b in range(c["batch"]):
        mask = oracle.ssd_anchor_mask_numpy(co[co[:, 0] == b][:, 1:], bv, grid, 1.0)
        bb, ss, ll = oracle.ssd_post_process_frame_numpy(
            box[b], cls[b], dirp[b], an, mask, h["nms_score_threshold"], h["prediction_center_limit_range"],
            h["nms_pre_max_size"], h["nms_post_max_size"], h["nms_iou_threshold"], kind=kind)
        np.testing.assert_array_equal(ll, sg[f"{tag}_out_labels_{b}"])
        # exp / sigmoid differ by an ulp between torch (the shim) and NumPy
        np.testing.assert_allclose(bb, sg[f"{tag}_out_boxes_{b}"], rtol=1e-6, atol=4e-6)
        np.testing.assert_allclose(ss, sg[f"{tag}_out_scores_{b}"], rtol=0, atol=3e-7)
    assert sg["b_out_scores_1"].tolist() == [-1.0] and sg["b_out_labels_1"].tolist() == [-1]  # the `_box_empty` frame
