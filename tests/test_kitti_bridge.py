"""Detections -> KITTI evaluation records (paddle3d_amd/kitti_bridge.py) against records made by the reference's own
NumPy code (tests/golden/make_kitti_golden.py: _parse_result_to_sample, filter_fake_result, box_lidar_to_camera,
corners_3d, project_to_image, _parse_predictions_to_eval_format).  CPU only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_kitti_golden as G  # noqa: E402


def test_kitti_annos_match_reference_code(tmp_path):
    from paddle3d_amd import kitti_bridge as kb

    gold = np.load(os.path.join(HERE, "golden", "python_kitti.npz"))
    frames = [0, 1, 2]
    dets = [dict(box3d_lidar=gold[f"in_boxes_{i}"], scores=gold[f"in_scores_{i}"], label_preds=gold[f"in_labels_{i}"])
            for i in frames]
    calibs = [G.calibration(40 + i) for i in frames]
    annos = kb.detections_to_kitti_annos(dets, calibs, {0: "Car", 1: "Cyclist", 2: "Pedestrian"})
    for i, a in zip(frames, annos):
        for k, v in a.items():
            ref = gold[f"det_{k}_{i}"]
            assert np.asarray(v).shape == ref.shape, (i, k)
            if k == "name":
                assert [str(x) for x in v] == [str(x) for x in ref]
            elif k == "bbox":  # float64 projection rounded once; einsum order may differ by an ulp
                np.testing.assert_allclose(v, ref, rtol=1e-6, atol=1e-3)
            else:
                np.testing.assert_allclose(np.asarray(v, np.float64), ref.astype(np.float64), rtol=1e-6, atol=1e-6)
    assert len(annos[2]["name"]) == 0  # the head's `_box_empty` marker row makes a frame without boxes
    np.testing.assert_allclose(annos[0]["alpha"], gold["alpha_0"], rtol=0, atol=1e-6)
    # result files: one line per box, KITTI column order (h w l from dimensions (l, h, w))
    kb.write_label_files(annos, ["000000", "000001", "000002"], str(tmp_path))
    lines = open(tmp_path / "000000.txt").read().strip().split("\n")
    assert len(lines) == 25 and open(tmp_path / "000002.txt").read() == ""
    f = lines[0].split()
    assert f[0] == annos[0]["name"][0] and len(f) == 16
    l, h, w = annos[0]["dimensions"][0]
    assert abs(float(f[8]) - h) < 0.006 and abs(float(f[9]) - w) < 0.006 and abs(float(f[10]) - l) < 0.006
    # tensors as input (what PointPillars.test_forward returns) give the same records
    import torch

    tdets = [{k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()} for d in dets[:1]]
    again = kb.detections_to_kitti_annos(tdets, calibs[:1], {0: "Car", 1: "Cyclist", 2: "Pedestrian"})
    np.testing.assert_array_equal(again[0]["bbox"], annos[0]["bbox"])
