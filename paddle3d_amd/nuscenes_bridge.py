"""Detections -> nuScenes result records: the step after the hot path that closes the loop to box mAP
(reference paddle3d/datasets/nuscenes/nuscenes_metric.py:58-170, nuscenes_utils.py:140-208,
models/detection/centerpoint/centerpoint.py:180-201).

Pure NumPy (quaternion algebra included: the reference leans on nuscenes-devkit's Box / pyquaternion, which are
not installed here); `evaluate()` hands the JSON to nuscenes-devkit's NuScenesEval when that package and a dataset
are present.  mAP parity itself cannot be measured offline (no data, no weights).  What IS pinned
(tests/test_nuscenes_bridge.py against tests/golden/python_nuscenes.npz): the records equal the ones the reference's
own `_parse_results_to_sample` -> `filter_fake_result` -> `second_bbox_to_nuscenes_box` ->
`_parse_predictions_to_eval_format` -> `get_nuscenes_box_attribute` produce when executed on the same detections and
poses (which rows survive, their order, names, attributes, sizes and scores exactly; centres / velocities to 1e-9,
quaternions to 1e-7: the reference halves the float32 heading in whatever precision its NumPy version promotes to).
`average_precision` / `nuscenes_style_map` below are a devkit-free restatement of the detection benchmark's matching
(centre distance 0.5 / 1 / 2 / 4 m, greedy by descending score, AP over recall > 0.1 with precision floored at 0.1) used
for the mAP PROXY of the bench and tests: the HIP pipeline's detections scored against the oracle pipeline's."""
from __future__ import annotations

import json
import os
import tempfile

import numpy as np

__all__ = ["NUSC_CLASS_NAMES", "CLASS_RANGE_CVPR_2019", "DEFAULT_ATTRIBUTE", "box_attribute", "detections_to_results",
           "results_to_json", "evaluate", "average_precision", "nuscenes_style_map", "DIST_THRESHOLDS"]

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
        boxes32 = np.asarray(det["box3d_lidar"].cpu() if hasattr(det["box3d_lidar"], "cpu") else det["box3d_lidar"], np.float32)
        boxes = boxes32.astype(np.float64)  # BBoxes3D is a float32 array (geometries/structure.py:35); what is read from it widens exactly
        scores = np.asarray(det["scores"].cpu() if hasattr(det["scores"], "cpu") else det["scores"], np.float64)
        labels = np.asarray(det["label_preds"].cpu() if hasattr(det["label_preds"], "cpu") else det["label_preds"])
        q_s, t_s = np.asarray(sp["rotation"], np.float64), np.asarray(sp["translation"], np.float64)
        q_e, t_e = np.asarray(ep["rotation"], np.float64), np.asarray(ep["translation"], np.float64)
        out = []
        for i in range(boxes.shape[0]):
            if scores[i] < 0:  # filter_fake_result
                continue
            name = class_names[int(labels[i])]
            # second_bbox_to_nuscenes_box rewrites the heading IN the float32 box array (nuscenes_utils.py:162)
            yaw = float(-boxes32[i, -1] - np.float32(np.pi / 2))
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


# ----------------------------------------------------------------------------------------------------------------
# mAP proxy: the detection benchmark's matching and AP, restated without the devkit (nuscenes-devkit
# eval/detection/algo.py `accumulate` + `calc_ap`, detection_cvpr_2019: dist_ths 0.5 / 1 / 2 / 4 m, dist_fcn
# center_distance, min_recall 0.1, min_precision 0.1, 101 recall points).  Used to score one pipeline's detections
# against another's on the same frames -- NOT a dataset mAP (there is no ground truth offline).
DIST_THRESHOLDS = (0.5, 1.0, 2.0, 4.0)


def average_precision(pred_xy, pred_score, pred_frame, gt_xy, gt_frame, dist_th, min_recall=0.1, min_precision=0.1):
    """AP of one class at one centre-distance threshold.  pred_* / gt_*: centres [n, 2], scores [n], frame ids [n].
    Predictions are visited by descending score (stable); each takes the closest not-yet-matched truth of its frame
    if that is closer than `dist_th`."""
    pred_xy, gt_xy = np.asarray(pred_xy, np.float64).reshape(-1, 2), np.asarray(gt_xy, np.float64).reshape(-1, 2)
    pred_score, pred_frame = np.asarray(pred_score, np.float64), np.asarray(pred_frame)
    gt_frame = np.asarray(gt_frame)
    npos = len(gt_xy)
    if npos == 0 or len(pred_xy) == 0:
        return 0.0
    by_frame = {}
    for j, f in enumerate(gt_frame.tolist()):
        by_frame.setdefault(f, []).append(j)
    taken = np.zeros(npos, bool)
    order = np.argsort(-pred_score, kind="stable")
    tp = np.zeros(len(order))
    for r, i in enumerate(order):
        cand = [j for j in by_frame.get(pred_frame[i].item() if hasattr(pred_frame[i], "item") else pred_frame[i], [])
                if not taken[j]]
        if not cand:
            continue
        d = np.hypot(gt_xy[cand, 0] - pred_xy[i, 0], gt_xy[cand, 1] - pred_xy[i, 1])
        k = int(np.argmin(d))
        if d[k] < dist_th:
            taken[cand[k]] = True
            tp[r] = 1.0
    if tp.sum() == 0:
        return 0.0
    ctp, cfp = np.cumsum(tp), np.cumsum(1.0 - tp)
    prec, rec = ctp / (ctp + cfp), ctp / float(npos)
    rec_interp = np.linspace(0, 1, 101)
    prec = np.interp(rec_interp, rec, prec, right=0)
    prec = prec[int(round(100 * min_recall)) + 1:] - min_precision
    prec[prec < 0] = 0
    return float(np.mean(prec)) / (1.0 - min_precision)


def nuscenes_style_map(pred, truth, num_classes=10, dist_ths=DIST_THRESHOLDS):
    """mAP of `pred` against `truth`: both lists (one entry per frame) of dict(box3d_lidar [K, >= 2], scores [K],
    label_preds [K]) as `CenterPoint.test_forward` returns them.  Rows with score < 0 (the fake row of an empty
    frame) are ignored.  Returns dict(mAP, per_class {label: mean AP over thresholds}, classes_scored).  Classes
    without a truth box are left out of the mean (the devkit scores them 0 against real annotations; here there is
    nothing to find)."""
    def flat(dets):
        xy, sc, lab, fr = [], [], [], []
        for f, d in enumerate(dets):
            b = np.asarray(d["box3d_lidar"].cpu() if hasattr(d["box3d_lidar"], "cpu") else d["box3d_lidar"], np.float64)
            s = np.asarray(d["scores"].cpu() if hasattr(d["scores"], "cpu") else d["scores"], np.float64)
            l = np.asarray(d["label_preds"].cpu() if hasattr(d["label_preds"], "cpu") else d["label_preds"])
            keep = s >= 0
            xy.append(b[keep, :2])
            sc.append(s[keep])
            lab.append(l[keep])
            fr.append(np.full(int(keep.sum()), f))
        return np.concatenate(xy), np.concatenate(sc), np.concatenate(lab), np.concatenate(fr)

    pxy, ps, pl, pf = flat(pred)
    txy, _, tl, tf = flat(truth)
    per_class = {}
    for c in range(num_classes):
        if not (tl == c).any():
            continue
        pm, tm = pl == c, tl == c
        per_class[c] = float(np.mean([average_precision(pxy[pm], ps[pm], pf[pm], txy[tm], tf[tm], th) for th in dist_ths]))
    m = float(np.mean(list(per_class.values()))) if per_class else float("nan")
    return dict(mAP=m, per_class=per_class, classes_scored=len(per_class))


def unmatched_detections(pred, truth, dist_th: float = 0.5, score_tol: float = 1e-3):
    """How many detections of `truth` have no twin in `pred` (same frame, same class, centre closer than `dist_th`,
    score within `score_tol`; one-to-one) -- the count behind an mAP-proxy figure: the nuScenes AP is quantised (one
    missing box of a class costs one of its 90 recall bins at every threshold, 1 / 900 of a ten-class mAP whatever
    the number of boxes), so "one box of 31 872 differs" and "0.99889" are the same statement.  Returns
    dict(unmatched, total)."""
    missing = total = 0
    for p, t in zip(pred, truth):
        pb = np.asarray(p["box3d_lidar"].cpu() if hasattr(p["box3d_lidar"], "cpu") else p["box3d_lidar"], np.float64)
        ps = np.asarray(p["scores"].cpu() if hasattr(p["scores"], "cpu") else p["scores"], np.float64)
        pl = np.asarray(p["label_preds"].cpu() if hasattr(p["label_preds"], "cpu") else p["label_preds"])
        tb = np.asarray(t["box3d_lidar"].cpu() if hasattr(t["box3d_lidar"], "cpu") else t["box3d_lidar"], np.float64)
        ts = np.asarray(t["scores"].cpu() if hasattr(t["scores"], "cpu") else t["scores"], np.float64)
        tl = np.asarray(t["label_preds"].cpu() if hasattr(t["label_preds"], "cpu") else t["label_preds"])
        free = ps >= 0
        for j in np.nonzero(ts >= 0)[0]:
            total += 1
            cand = np.nonzero(free & (pl == tl[j]) & (np.abs(ps - ts[j]) <= score_tol))[0]
            if len(cand):
                d = np.hypot(pb[cand, 0] - tb[j, 0], pb[cand, 1] - tb[j, 1])
                k = int(np.argmin(d))
                if d[k] < dist_th:
                    free[cand[k]] = False
                    continue
            missing += 1
    return dict(unmatched=missing, total=total)
