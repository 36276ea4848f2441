"""Detections -> nuScenes records (paddle3d_amd/nuscenes_bridge.py): against the records the reference's own code
produces (tests/golden/make_nuscenes_golden.py executes centerpoint.py:180-201, nuscenes_utils.py:140-208 and
nuscenes_metric.py:58-123), against hand-computable geometry, and the devkit-free AP restatement on cases whose
answer is known.  CPU only."""
import json
import os

import numpy as np
import pytest

from paddle3d_amd import nuscenes_bridge as nb

HERE = os.path.dirname(os.path.abspath(__file__))


def _q(yaw):
    return [np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)]


@pytest.mark.parametrize("i", [0, 1, 2])
def test_records_equal_the_reference(i):
    g = np.load(os.path.join(HERE, "golden", "python_nuscenes.npz"))
    det = dict(box3d_lidar=g[f"in_boxes_{i}"], scores=g[f"in_scores_{i}"], label_preds=g[f"in_labels_{i}"])
    sp = dict(rotation=g[f"sensor_{i}"][:4], translation=g[f"sensor_{i}"][4:])
    ep = dict(rotation=g[f"ego_{i}"][:4], translation=g[f"ego_{i}"][4:])
    recs = nb.detections_to_results([det], [f"tok{i}"], [sp], [ep])[f"tok{i}"]
    assert len(recs) == int(g[f"n_{i}"][0]) < int((g[f"in_scores_{i}"] >= 0).sum())  # the range filter dropped some
    assert [r["detection_name"] for r in recs] == g[f"rec_detection_name_{i}"].tolist()
    assert [r["attribute_name"] for r in recs] == g[f"rec_attribute_name_{i}"].tolist()
    assert [r["sample_token"] for r in recs] == g[f"rec_sample_token_{i}"].tolist()
    np.testing.assert_array_equal([r["size"] for r in recs], g[f"rec_size_{i}"])
    np.testing.assert_array_equal([r["detection_score"] for r in recs], g[f"rec_detection_score_{i}"])
    np.testing.assert_allclose([r["translation"] for r in recs], g[f"rec_translation_{i}"], rtol=0, atol=1e-9)
    np.testing.assert_allclose([r["velocity"] for r in recs], g[f"rec_velocity_{i}"], rtol=0, atol=1e-9)
    np.testing.assert_allclose([r["rotation"] for r in recs], g[f"rec_rotation_{i}"], rtol=0, atol=1e-7)


def test_default_attribute_table_is_the_references_argmax():
    g = np.load(os.path.join(HERE, "golden", "python_nuscenes.npz"))
    for name, attr in zip(g["attr_argmax_names"].tolist(), g["attr_argmax"].tolist()):
        assert nb.DEFAULT_ATTRIBUTE[name] == attr, name


def test_conversion_geometry(tmp_path):
    det = dict(box3d_lidar=np.array([[10.0, 0.0, -1.0, 4.0, 2.0, 1.5, 3.0, 0.0, 0.3],     # car, moving
                                     [0.0, 45.0, 0.0, 0.8, 0.8, 1.8, 0.0, 0.0, 0.0],       # pedestrian beyond 40 m
                                     [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],        # fake row
                                     [5.0, 5.0, 0.0, 10.0, 3.0, 3.5, 0.0, 0.05, 1.0]]),    # bus, standing
               scores=np.array([0.9, 0.8, -1.0, 0.5]), label_preds=np.array([0, 8, 0, 3]))
    sensor = dict(rotation=_q(np.pi / 2), translation=[1.0, 0.0, 2.0])    # LiDAR yawed 90 degrees on the ego
    ego = dict(rotation=_q(np.pi), translation=[100.0, 200.0, 0.0])
    res = nb.detections_to_results([det], ["tok"], [sensor], [ego])
    recs = res["tok"]
    assert [r["detection_name"] for r in recs] == ["car", "bus"]  # fake row and far pedestrian dropped
    car = recs[0]
    # (10, 0, -1) --sensor--> (0, 10, -1) + (1, 0, 2) = (1, 10, 1) --ego (yaw pi)--> (-1, -10, 1) + (100, 200, 0)
    np.testing.assert_allclose(car["translation"], [99.0, 190.0, 1.0], atol=1e-9)
    np.testing.assert_allclose(car["size"], [4.0, 2.0, 1.5])
    # heading: -0.3 - pi/2, then + pi/2 (sensor) + pi (ego)
    yaw = -0.3 - np.pi / 2 + np.pi / 2 + np.pi
    q = np.array(car["rotation"])
    np.testing.assert_allclose(np.abs(q), np.abs(_q(yaw)), atol=1e-7)
    # velocity (3, 0) rotated by 90 then 180 degrees = (0, -3)
    np.testing.assert_allclose(car["velocity"], [0.0, -3.0], atol=1e-9)
    assert car["attribute_name"] == "vehicle.moving" and car["detection_score"] == pytest.approx(0.9)
    assert recs[1]["attribute_name"] == "vehicle.stopped"
    path = nb.results_to_json(res, str(tmp_path / "pred.json"))
    blob = json.load(open(path))
    assert blob["meta"]["use_lidar"] is True and blob["meta"]["use_camera"] is False
    assert len(blob["results"]["tok"]) == 2


def test_attribute_rule():
    assert nb.box_attribute([0.3, 0.0], "bicycle") == "cycle.with_rider"
    assert nb.box_attribute([0.0, 0.0], "pedestrian") == "pedestrian.standing"
    assert nb.box_attribute([1.0, 0.0], "pedestrian") == "pedestrian.moving"     # the distribution's arg-max
    assert nb.box_attribute([0.0, 0.0], "car") == "vehicle.parked"
    assert nb.box_attribute([0.0, 0.0], "barrier") == "cycle.with_rider"          # the reference's all-zero row


def _dets(rng, n, frames=4):
    out = []
    for _ in range(frames):
        b = np.zeros((n, 9))
        b[:, :2] = rng.uniform(-50, 50, (n, 2))
        out.append(dict(box3d_lidar=b, scores=rng.uniform(0.1, 1.0, n), label_preds=rng.integers(0, 10, n)))
    return out


def test_map_proxy_known_answers():
    rng = np.random.default_rng(3)
    truth = _dets(rng, 200)
    assert nb.nuscenes_style_map(truth, truth)["mAP"] == pytest.approx(1.0)
    # shifted by 0.7 m: missed at the 0.5 m threshold only -> 3/4 per class
    moved = [dict(d, box3d_lidar=d["box3d_lidar"] + np.array([0.7] + [0.0] * 8)) for d in truth]
    assert nb.nuscenes_style_map(moved, truth)["mAP"] == pytest.approx(0.75, abs=0.02)
    # the lower-scored half of the predictions removed: recall stops at ~0.5, precision 1 up to there
    half = []
    for d in truth:
        keep = d["scores"] >= np.median(d["scores"])
        half.append({k: v[keep] for k, v in d.items()})
    m = nb.nuscenes_style_map(half, truth)["mAP"]
    assert 0.35 < m < 0.55
    # wrong labels everywhere: nothing matches
    wrong = [dict(d, label_preds=(d["label_preds"] + 1) % 10) for d in truth]
    assert nb.nuscenes_style_map(wrong, truth)["mAP"] < 0.05
    # a single AP by hand: 2 truths, predictions (tp, fp, tp) by descending score -> precision 1, 1/2, 2/3 at recall 1/2, 1/2, 1
    ap = nb.average_precision([[0, 0], [9, 9], [5, 5]], [0.9, 0.8, 0.7], [0, 0, 0], [[0, 0.1], [5, 5.1]], [0, 0], 0.5)
    rec_i = np.linspace(0, 1, 101)
    want = np.interp(rec_i, [0.5, 0.5, 1.0], [1.0, 0.5, 2 / 3], right=0)[11:] - 0.1
    assert ap == pytest.approx(float(np.mean(np.clip(want, 0, None))) / 0.9)
