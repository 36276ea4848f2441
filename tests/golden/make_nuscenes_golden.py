"""Golden vectors for the detections -> nuScenes result records conversion, from the reference's own code:
`CenterPoint._parse_results_to_sample` (models/detection/centerpoint/centerpoint.py:180-201), `filter_fake_result`,
`second_bbox_to_nuscenes_box`, `get_nuscenes_box_attribute` + `cls_attr_dist` (datasets/nuscenes/nuscenes_utils.py:27-208,
imported from where it lies) and `NuScenesMetric._parse_predictions_to_eval_format`
(datasets/nuscenes/nuscenes_metric.py:58-123, exec'd by line range into a bare class: the file's import list needs the
devkit).

    python tests/golden/make_nuscenes_golden.py        # needs /root/reference; writes python_nuscenes.npz

THIRD-PARTY CODE THAT IS NOT IN THE REFERENCE TREE (requirements.txt names `nuscenes-devkit` and `pyquaternion`
without versions; neither is installed here): the two classes the reference's functions construct are restated below
from their published algorithms --
  * `pyquaternion.Quaternion` (0.9.9): construction from `axis=, radians=` (`_from_axis_angle`) and from a 4-sequence,
    `elements`, Hamilton product through `_q_matrix`, `rotation_matrix` = `_q_matrix . _q_bar_matrix^H` [1:, 1:] after
    `_normalise()`;
  * `nuscenes.utils.data_classes.Box` (devkit 1.1.x): fields, `rotate` (centre and velocity by the rotation matrix,
    orientation by the left product), `translate`.
So what this fixture pins is the REFERENCE's own glue (fake-row filter, the heading convention `-theta - pi/2` in
float32, velocity padding, sensor -> ego -> global order, the class-range filter in the ego frame, the record fields,
the attribute rule and its arg-max table); the quaternion algebra is pinned only to this restatement.  `NuScenesEval`
(the mAP itself) needs the devkit and the dataset: unmeasurable offline, stated in DESIGN.md.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import paddle_shim as ps  # noqa: E402

REF = "/root/reference"


class Quaternion:
    """pyquaternion 0.9.9, the members the reference's path touches."""

    def __init__(self, *args, **kwargs):
        if "axis" in kwargs:
            axis = np.array(kwargs["axis"], dtype=float)
            angle = kwargs.get("radians", kwargs.get("angle"))
            mag_sq = np.dot(axis, axis)
            if abs(1.0 - mag_sq) > 1e-12:
                axis = axis / np.sqrt(mag_sq)
            theta = angle / 2.0
            r = np.cos(theta)
            i = axis * np.sin(theta)
            self.q = np.array([r, i[0], i[1], i[2]], dtype=float)
        elif "array" in kwargs:
            self.q = np.array(kwargs["array"], dtype=float)
        elif len(args) == 1:
            self.q = np.array(args[0], dtype=float)
            assert self.q.shape == (4,)
        else:
            self.q = np.array(args, dtype=float)
            assert self.q.shape == (4,)

    @property
    def elements(self):
        return self.q

    def _q_matrix(self):
        w, x, y, z = self.q
        return np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])

    def _q_bar_matrix(self):
        w, x, y, z = self.q
        return np.array([[w, -x, -y, -z], [x, w, z, -y], [y, -z, w, x], [z, y, -x, w]])

    def _normalise(self):
        ss = np.dot(self.q, self.q)
        if not abs(1.0 - ss) < 1e-14:
            n = np.sqrt(ss)
            if n > 0:
                self.q = self.q / n

    @property
    def rotation_matrix(self):
        self._normalise()
        return np.dot(self._q_matrix(), self._q_bar_matrix().conj().transpose())[1:][:, 1:]

    def __mul__(self, other):
        return Quaternion(array=np.dot(self._q_matrix(), other.q))


class Box:
    """nuscenes-devkit 1.1.x `nuscenes.utils.data_classes.Box`, the members the reference's path touches."""

    def __init__(self, center, size, orientation, label=np.nan, score=np.nan, velocity=(np.nan, np.nan, np.nan),
                 name=None, token=None):
        assert len(center) == 3 and len(size) == 3
        self.center = np.array(center)
        self.wlh = np.array(size)
        self.orientation = orientation
        self.label = int(label) if not np.isnan(label) else label
        self.score = float(score) if not np.isnan(score) else score
        self.velocity = np.array(velocity)
        self.name = name
        self.token = token

    def rotate(self, quaternion):
        self.center = np.dot(quaternion.rotation_matrix, self.center)
        self.orientation = quaternion * self.orientation
        self.velocity = np.dot(quaternion.rotation_matrix, self.velocity)

    def translate(self, x):
        self.center += x


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


NAMES = ["car", "truck", "construction_vehicle", "bus", "trailer", "barrier", "motorcycle", "bicycle", "pedestrian",
         "traffic_cone"]
CLASS_RANGE = dict(car=50, truck=50, bus=50, trailer=50, construction_vehicle=50, pedestrian=40, motorcycle=40,
                   bicycle=40, traffic_cone=30, barrier=30)  # nuscenes-devkit detection_cvpr_2019.json


def unit_quat(rng, max_tilt):
    yaw = rng.uniform(-np.pi, np.pi)
    a = np.array([rng.normal(0, max_tilt), rng.normal(0, max_tilt), 1.0])
    a /= np.linalg.norm(a)
    q = np.array([np.cos(yaw / 2), *(a * np.sin(yaw / 2))])
    return (q / np.linalg.norm(q)).tolist()


def detections(seed, n):
    """test_forward-style rows: box3d_lidar [n, 9] (x, y, z, dx, dy, dz, vx, vy, theta) fp32, scores, labels."""
    rng = np.random.default_rng(seed)
    b = np.zeros((n, 9), np.float32)
    r = rng.uniform(2, 58, n)
    az = rng.uniform(-np.pi, np.pi, n)
    b[:, 0], b[:, 1], b[:, 2] = r * np.cos(az), r * np.sin(az), rng.uniform(-3, 1, n)
    b[:, 3:6] = rng.uniform(0.4, 11.0, (n, 3))
    b[:, 6:8] = rng.normal(0, 1.5, (n, 2)) * (rng.random((n, 1)) < 0.6)   # 40 % standing still
    b[:, 8] = rng.uniform(-3.5, 3.5, n)
    s = rng.uniform(0.05, 0.99, n).astype(np.float32)
    s[rng.choice(n, max(1, n // 10), replace=False)] = -1.0                  # fake rows (score < 0)
    lab = rng.integers(0, 10, n).astype(np.int64)
    return b, s, lab


def main():
    ps.install(REF)
    for name, attrs in (("numba", {}), ("pyquaternion", dict(Quaternion=Quaternion)), ("nuscenes", {}),
                        ("nuscenes.utils", {}), ("nuscenes.utils.data_classes", dict(Box=Box))):
        m = types.ModuleType(name)
        m.jit = m.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    _load("paddle3d.geometries.structure", "paddle3d/geometries/structure.py")
    bbox = _load("paddle3d.geometries.bbox", "paddle3d/geometries/bbox.py")
    geo = sys.modules["paddle3d.geometries"]
    geo.BBoxes2D, geo.BBoxes3D, geo.CoordMode = bbox.BBoxes2D, bbox.BBoxes3D, bbox.CoordMode
    sample = _load("paddle3d.sample", "paddle3d/sample.py")
    nu = _load("paddle3d_nuscenes_utils_for_golden", "paddle3d/datasets/nuscenes/nuscenes_utils.py")
    # CenterPoint._parse_results_to_sample (centerpoint.py:180-201) into a bare class
    ns_m = dict(Sample=sample.Sample, SampleMeta=sample.SampleMeta, BBoxes3D=bbox.BBoxes3D)
    src = open(os.path.join(REF, "paddle3d/models/detection/centerpoint/centerpoint.py")).read().split("\n")
    assert src[179].strip().startswith("def _parse_results_to_sample"), src[179]
    exec(compile("class _C:\n" + "\n".join(src[179:201]), "centerpoint.py:180-201", "exec"), ns_m)
    # NuScenesMetric._parse_predictions_to_eval_format (nuscenes_metric.py:58-123) into a bare class
    ns = dict(np=np, List=list, Sample=sample.Sample, Quaternion=Quaternion, filter_fake_result=nu.filter_fake_result,
              second_bbox_to_nuscenes_box=nu.second_bbox_to_nuscenes_box,
              get_nuscenes_box_attribute=nu.get_nuscenes_box_attribute)
    src = open(os.path.join(REF, "paddle3d/datasets/nuscenes/nuscenes_metric.py")).read().split("\n")
    assert src[57].strip().startswith("def _parse_predictions_to_eval_format"), src[57]
    exec(compile("class _M:\n" + "\n".join(src[57:123]), "nuscenes_metric.py:58-123", "exec"), ns)

    rng = np.random.default_rng(77)
    tables = dict(sample={}, sample_data={}, ego_pose={}, calibrated_sensor={})

    class FakeNusc:
        def get(self, table, token):
            return tables[table][token]

    metric = ns["_M"]()
    metric.nusc, metric.channel, metric.class_names = FakeNusc(), "LIDAR_TOP", NAMES
    metric.eval_detection_configs = types.SimpleNamespace(class_range=CLASS_RANGE)
    T = ps.tensor
    out, results, frames = {}, [], [(0, 60), (1, 7), (2, 200)]
    for i, n in frames:
        tok = f"tok{i}"
        tables["sample"][tok] = dict(data=dict(LIDAR_TOP=f"sd{i}"))
        tables["sample_data"][f"sd{i}"] = dict(ego_pose_token=f"ep{i}", calibrated_sensor_token=f"cs{i}")
        sp = dict(rotation=unit_quat(rng, 0.01), translation=[0.94 + rng.normal(0, 0.01), rng.normal(0, 0.01), 1.84])
        ep = dict(rotation=unit_quat(rng, 0.02), translation=[rng.uniform(300, 1800), rng.uniform(800, 1700), 0.0])
        tables["calibrated_sensor"][f"cs{i}"], tables["ego_pose"][f"ep{i}"] = sp, ep
        b, s, lab = detections(90 + i, n)
        results.append(dict(box3d_lidar=T(b), scores=T(s), label_preds=T(lab), meta=tok))
        out[f"in_boxes_{i}"], out[f"in_scores_{i}"], out[f"in_labels_{i}"] = b, s, lab
        out[f"sensor_{i}"] = np.asarray(sp["rotation"] + sp["translation"], np.float64)
        out[f"ego_{i}"] = np.asarray(ep["rotation"] + ep["translation"], np.float64)
    samples = ns_m["_C"]()._parse_results_to_sample(results, dict(path=["a.bin"] * 3, modality=["lidar"] * 3))
    res = metric._parse_predictions_to_eval_format(samples)
    for i, _ in frames:
        recs = res[f"tok{i}"]
        out[f"n_{i}"] = np.asarray([len(recs)])
        for k in ("translation", "size", "rotation", "velocity", "detection_score"):
            out[f"rec_{k}_{i}"] = np.asarray([r[k] for r in recs], np.float64)
        for k in ("detection_name", "attribute_name", "sample_token"):
            out[f"rec_{k}_{i}"] = np.asarray([r[k] for r in recs]).astype(str)
        print(i, len(recs), "records of", int((out[f"in_scores_{i}"] >= 0).sum()), "real rows")
    out["attr_argmax_names"] = np.asarray(sorted(nu.cls_attr_dist)).astype(str)
    out["attr_argmax"] = np.asarray([nu.get_nuscenes_box_attribute(types.SimpleNamespace(velocity=np.zeros(3)), k)
                                     if k not in ("pedestrian", "bus") else
                                     max(nu.cls_attr_dist[k].items(), key=lambda kv: kv[1])[0]
                                     for k in sorted(nu.cls_attr_dist)]).astype(str)
    path = os.path.join(HERE, "python_nuscenes.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} KB")


if __name__ == "__main__":
    main()
