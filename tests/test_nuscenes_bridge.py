"""Detections -> nuScenes records (paddle3d_amd/nuscenes_bridge.py) against hand-computable geometry.  CPU only."""
import json

import numpy as np

from paddle3d_amd import nuscenes_bridge as nb


def _q(yaw):
    return [np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)]


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
    np.testing.assert_allclose(np.abs(q), np.abs(_q(yaw)), atol=1e-9)
    # velocity (3, 0) rotated by 90 then 180 degrees = (0, -3)
    np.testing.assert_allclose(car["velocity"], [0.0, -3.0], atol=1e-9)
    assert car["attribute_name"] == "vehicle.moving" and car["detection_score"] == 0.9
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
