"""Detections -> nuScenes result records: the step after the hot path that closes the loop to box mAP
(reference paddle3d/datasets/nuscenes/nuscenes_metric.py:58-170, nuscenes_utils.py:140-208,
models/detection/centerpoint/centerpoint.py:180-201).

Pure NumPy (quaternion algebra included: the reference leans on nuscenes-devkit's Box / pyquaternion, which are
not installed here); `evaluate()` hands the JSON to nuscenes-devkit's NuScenesEval when that package and a dataset
are present.  mAP parity itself cannot be measured offline (no data, no weights) -- what is pinned by the tests is
the geometry of the conversion."""
from __future__ import annotations

import json
import os
import tempfile

import numpy as np

__all__ = ["NUSC_CLASS_NAMES", "CLASS_RANGE_CVPR_2019", "DEFAULT_ATTRIBUTE", "box_attribute", "detections_to_results",
           "results_to_json", "evaluate"]

# class order of the CenterPoint nuScenes configs (tasks concatenated: centerpoint_pillars_02voxel_nuscenes_10sweep.yml)
NUSC_CLASS_NAMES = ["car", "truck", "construction_vehicle", "bus", "trailer", "barrier", "motorcycle", "bicycle",
                    "pedestrian", "traffic_cone"]
# nuscenes-devkit detection_cvpr_2019 config, class_range (metres in the ego frame)
CLASS_RANGE_CVPR_2019 = dict(car=50, truck=50, bus=50, trailer=50, construction_vehicle=50, pedestrian=40,
                             motorcycle=40, bicycle=40, traffic_cone=30, barrier=30)
# arg-max of the reference's cls_attr_dist per class (nuscenes_utils.py:205-207; barrier / traffic_cone have an
# all-zero row there, so the first key wins -- kept as is)
DEFAULT_ATTRIBUTE = dict(barrier="cycle.with_rider", bicycle="cycle.without_rider", bus="vehicle.moving",
                         car="vehicle.parked", construction_vehicle="vehicle.parked", ignore="vehicle.parked",
                         motorcycle="cycle.without_rider", pedestrian="pedestrian.moving",
                         traffic_cone="cycle.with_rider", trailer="vehicle.parked", truck="vehicle.parked")


def _qmul(a, b):
    """Hamilton product of quaternions (w, x, y, z)."""
    w0, x0, y0, z0 = a
    w1, x1, y1, z1 = b
    return np.array([w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1, w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
                     w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1, w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1], np.float64)


def _qrot(q, v):
    """Rotate vector v by unit quaternion q."""
    w, x, y, z = q
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float64)
    return r @ np.asarray(v, np.float64)


def box_attribute(velocity_xy, label_name: str) -> str:
    """nuscenes_utils.py:183-208."""
    attr = None
    if float(np.hypot(velocity_xy[0], velocity_xy[1])) > 0.2:
        if label_name in ("car", "construction_vehicle", "bus", "truck", "trailer"):
            attr = "vehicle.moving"
        elif label_name in ("bicycle", "motorcycle"):
            attr = "cycle.with_rider"
    else:
        if label_name == "pedestrian":
            attr = "pedestrian.standing"
        elif label_name == "bus":
            attr = "vehicle.stopped"
    return DEFAULT_ATTRIBUTE[label_name] if attr is None else attr


def detections_to_results(detections, sample_tokens, sensor_poses, ego_poses, class_names=None, class_range=None):
    """`CenterPoint.test_forward` output -> {sample_token: [nuScenes detection records]}.

    detections     per frame dict(box3d_lidar [K, 9] (x, y, z, l, w, h, vx, vy, theta), scores [K], label_preds [K])
                   (tensors or arrays)
    sensor_poses   per frame dict(rotation=(w, x, y, z), translation=(x, y, z)): LiDAR -> ego (`calibrated_sensor`)
    ego_poses      per frame dict(rotation, translation): ego -> global (`ego_pose`)
    Steps of _parse_predictions_to_eval_format (nuscenes_metric.py:58-126): drop the fake rows (score < 0), heading
    -> -theta - pi/2 about z (second_bbox_to_nuscenes_box), LiDAR -> ego, drop boxes beyond their class range in the
    ego frame, ego -> global."""
    class_names = NUSC_CLASS_NAMES if class_names is None else class_names
    class_range = CLASS_RANGE_CVPR_2019 if class_range is None else class_range
    res = {}
    for det, token, sp, ep in zip(detections, sample_tokens, sensor_poses, ego_poses):
        boxes = np.asarray(det["box3d_lidar"].cpu() if hasattr(det["box3d_lidar"], "cpu") else det["box3d_lidar"], np.float64)
        scores = np.asarray(det["scores"].cpu() if hasattr(det["scores"], "cpu") else det["scores"], np.float64)
        labels = np.asarray(det["label_preds"].cpu() if hasattr(det["label_preds"], "cpu") else det["label_preds"])
        q_s, t_s = np.asarray(sp["rotation"], np.float64), np.asarray(sp["translation"], np.float64)
        q_e, t_e = np.asarray(ep["rotation"], np.float64), np.asarray(ep["translation"], np.float64)
        out = []
        for i in range(boxes.shape[0]):
            if scores[i] < 0:  # filter_fake_result
                continue
            name = class_names[int(labels[i])]
            yaw = -boxes[i, -1] - np.pi / 2
            q = np.array([np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)])
            center = boxes[i, :3].copy()
            vel = np.array([boxes[i, 6], boxes[i, 7], 0.0]) if boxes.shape[1] == 9 else np.zeros(3)
            # LiDAR -> ego
            center, q, vel = _qrot(q_s, center) + t_s, _qmul(q_s, q), _qrot(q_s, vel)
            if float(np.linalg.norm(center[:2])) > class_range[name]:
                continue
            # ego -> global
            center, q, vel = _qrot(q_e, center) + t_e, _qmul(q_e, q), _qrot(q_e, vel)
            out.append(dict(sample_token=token, translation=center.tolist(), size=boxes[i, 3:6].tolist(),
                            rotation=q.tolist(), detection_name=name, detection_score=float(scores[i]),
                            velocity=vel[:2].tolist(), attribute_name=box_attribute(vel[:2], name)))
        res[token] = out
    return res


def results_to_json(results: dict, path: str, channel: str = "LIDAR_TOP") -> str:
    """The submission file of NuScenesMetric.compute (nuscenes_metric.py:136-151)."""
    blob = dict(meta=dict(use_camera=channel.startswith("CAM"), use_lidar=channel == "LIDAR_TOP", use_radar=False,
                          use_map=False, use_external=False), results=results)
    with open(path, "w") as f:
        json.dump(blob, f)
    return path


def evaluate(results: dict, nusc, eval_set: str, channel: str = "LIDAR_TOP", eval_version: str = "detection_cvpr_2019"):
    """NuScenesMetric.compute (nuscenes_metric.py:130-170): needs nuscenes-devkit and the dataset behind `nusc`."""
    try:
        from nuscenes.eval.detection.config import config_factory
        from nuscenes.eval.detection.evaluate import NuScenesEval
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("evaluate() needs the nuscenes-devkit package (not installed in this image)") from e
    with tempfile.TemporaryDirectory() as tmp:
        path = results_to_json(results, os.path.join(tmp, "nuscenes_pred.json"), channel)
        ev = NuScenesEval(nusc, config=config_factory(eval_version), result_path=path, eval_set=eval_set,
                          output_dir=tmp, verbose=False)
        ev.main(plot_examples=0, render_curves=False)
        with open(os.path.join(tmp, "metrics_summary.json")) as f:
            return json.load(f)
