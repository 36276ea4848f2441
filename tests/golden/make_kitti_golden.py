"""Golden vectors for the detections -> KITTI evaluation records conversion, from the reference's own (NumPy) code:
`SSDHead._parse_result_to_sample` (models/detection/pointpillars/pointpillars_head.py:198-221), `BBoxes3D.corners_3d`,
`rotation_3d_in_axis`, `project_to_image` (geometries/bbox.py), `filter_fake_result`, `box_lidar_to_camera`,
`coord_velodyne_to_camera` (datasets/kitti/kitti_utils.py:101-150, 245-272) and
`KittiMetric._parse_predictions_to_eval_format` / `get_camera_box2d` (datasets/kitti/kitti_metric.py:71-141).

    python tests/golden/make_kitti_golden.py        # needs /root/reference; writes python_kitti.npz

The modules are loaded from where they lie (numba / pyquaternion, which bbox.py imports for other functions, are
stubbed; kitti_metric.py's import list drags in the evaluation third-party code, so its two methods are exec'd by
line range into a bare class).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import paddle_shim as ps  # noqa: E402

REF = "/root/reference"


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def calibration(seed):
    """A KITTI-like calibration tuple (P0, P1, P2, P3, R0_rect, V2C, I2V), kitti_det.py:130-176."""
    rng = np.random.default_rng(seed)
    f, cx, cy = 721.5 + rng.normal(0, 2), 609.6 + rng.normal(0, 2), 172.9 + rng.normal(0, 2)

    def proj(bx):
        return np.array([[f, 0, cx, bx], [0, f, cy, rng.normal(0, 0.2)], [0, 0, 1, rng.normal(0, 0.003)]], np.float32)

    a = rng.normal(0, 0.01, 3)
    r0 = (np.eye(3) + np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])).astype(np.float32)
    v2c = np.array([[0, -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0]], np.float32) + rng.normal(0, 0.01, (3, 4)).astype(np.float32)
    v2c[:, 3] = [-0.004, -0.076, -0.272]
    return (proj(0.0), proj(-386.0), proj(44.86), proj(-339.5), r0, v2c, np.zeros((3, 4), np.float32))


def detections(seed, n):
    """SSDHead-style results: boxes (x, y, z bottom, w, l, h, r) in the KITTI lidar frame, scores, labels."""
    rng = np.random.default_rng(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, 0], b[:, 1], b[:, 2] = rng.uniform(5, 60, n), rng.uniform(-20, 20, n), rng.uniform(-2.2, -1.0, n)
    b[:, 3], b[:, 4], b[:, 5] = rng.uniform(1.4, 1.9, n), rng.uniform(3.2, 4.6, n), rng.uniform(1.3, 1.8, n)
    b[:, 6] = rng.uniform(-3.5, 3.5, n)
    return b, rng.uniform(0.05, 0.99, n).astype(np.float32), rng.integers(0, 3, n).astype(np.int64)


def main():
    paddle = ps.install(REF)
    for name in ("numba", "pyquaternion"):
        m = types.ModuleType(name)
        m.jit = m.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
        m.Quaternion = object
        sys.modules[name] = m
    _load("paddle3d.geometries.structure", "paddle3d/geometries/structure.py")
    bbox = _load("paddle3d.geometries.bbox", "paddle3d/geometries/bbox.py")
    geo = sys.modules["paddle3d.geometries"]
    geo.BBoxes2D, geo.BBoxes3D, geo.CoordMode = bbox.BBoxes2D, bbox.BBoxes3D, bbox.CoordMode
    sample = _load("paddle3d.sample", "paddle3d/sample.py")
    ku = _load("paddle3d.datasets.kitti.kitti_utils", "paddle3d/datasets/kitti/kitti_utils.py")
    hd = ps.load("paddle3d.models.detection.pointpillars.pointpillars_head")
    hd.Sample, hd.BBoxes3D, hd.CoordMode = sample.Sample, bbox.BBoxes3D, bbox.CoordMode
    # KittiMetric.get_camera_box2d + _parse_predictions_to_eval_format into a bare class
    ns = dict(np=np, List=list, Sample=sample.Sample, BBoxes3D=bbox.BBoxes3D, BBoxes2D=bbox.BBoxes2D,
              CoordMode=bbox.CoordMode, project_to_image=bbox.project_to_image,
              box_lidar_to_camera=ku.box_lidar_to_camera, filter_fake_result=ku.filter_fake_result)
    src = open(os.path.join(REF, "paddle3d/datasets/kitti/kitti_metric.py")).read().split("\n")
    body = "\n".join(src[70:141])  # lines 71-141: the two methods, indented as class members
    exec(compile("class _M:\n" + body, "kitti_metric.py:71-141", "exec"), ns)
    metric = ns["_M"]()
    metric.classmap = {0: "Car", 1: "Cyclist", 2: "Pedestrian"}
    out, preds = {}, []
    T = ps.tensor
    cases = [(0, 25), (1, 1), (2, 0)]  # frame 2: the head's `_box_empty` marker row -> a Sample without boxes
    metric.indexes = [f"{i:06d}" for i, _ in cases]
    for i, n in cases:
        calibs = calibration(40 + i)
        if n:
            b, s, l = detections(50 + i, n)
        else:
            b, s, l = np.zeros((1, 7), np.float32), -np.ones(1, np.float32), -np.ones(1, np.int64)
        res = dict(box3d_lidar=T(b), scores=T(s), label_preds=T(l))
        smp = hd.SSDHead._parse_result_to_sample(res, f"{i:06d}.bin", [T(c) for c in calibs], dict(id=f"{i:06d}"))
        preds.append(smp)
        out[f"in_boxes_{i}"], out[f"in_scores_{i}"], out[f"in_labels_{i}"] = b, s, l
        if n:
            out[f"alpha_{i}"] = np.asarray(smp.alpha)
    dets = metric._parse_predictions_to_eval_format(preds)
    for (i, n), det in zip(cases, dets):
        for k, v in det.items():
            out[f"det_{k}_{i}"] = np.asarray(v) if k != "name" else np.asarray(v).astype(str)
        print(i, {k: np.asarray(v).shape for k, v in det.items()})
    path = os.path.join(HERE, "python_kitti.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} KB")


if __name__ == "__main__":
    main()
