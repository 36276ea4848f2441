"""Detections -> KITTI evaluation records: the step after PointPillars' hot path that closes the loop to KITTI AP
(reference: SSDHead._parse_result_to_sample, models/detection/pointpillars/pointpillars_head.py:198-221;
datasets/kitti/kitti_utils.py:101-150 box_lidar_to_camera / coord_velodyne_to_camera, :245-272 filter_fake_result;
datasets/kitti/kitti_metric.py:71-141 get_camera_box2d / _parse_predictions_to_eval_format; geometries/bbox.py:132-160
corners_3d, :696-722 rotation_3d_in_axis / project_to_image).

Pure NumPy, with the reference's dtype flow (boxes live in float32 containers, the lidar -> camera transform and the
image projection run in float64 and are rounded back) so that the records agree with the reference's to the last bits;
tests/golden/python_kitti.npz holds records made by the reference's own code.  AP itself needs the KITTI evaluation
code and data (`kitti_eval`, third party in the reference); `write_label_files` produces the standard result files any
KITTI evaluator reads.
"""
from __future__ import annotations

import os

import numpy as np

__all__ = ["detections_to_kitti_annos", "anno_to_label_lines", "write_label_files"]


def _empty_anno():
    z = np.zeros
    return dict(truncated=z([0]), occluded=z([0]), alpha=z([0]), name=z([0]), bbox=z([0, 4]), dimensions=z([0, 3]),
                location=z([0, 3]), rotation_y=z([0]), score=z([0]))


def _lidar_to_camera(xyz, calibs):
    """kitti_utils.py:136-150: [x, y, z, 1] @ (R0_rect @ V2C)^T in float64."""
    r0 = np.eye(4)
    r0[:3, :3] = calibs[4]
    v2c = np.eye(4)
    v2c[:3, :4] = calibs[5]
    pts = np.concatenate([xyz, np.ones([xyz.shape[0], 1])], axis=1)
    return (pts @ (r0 @ v2c).T)[:, :3]


def _camera_box2d(cam, proj):
    """kitti_metric.py:71-78 over bbox.py:132-160 (origin (.5, 1, .5), rotation about the camera y axis) and
    bbox.py:716-722 (the 4th homogeneous coordinate the reference appends is ZERO: the projection's translation
    column does not enter)."""
    f32 = np.float32
    n = cam.shape[0]
    dx, dy, dz = cam[:, 3], cam[:, 4], cam[:, 5]
    xc = np.array([[0., 0., 0., 0., 1., 1., 1., 1.]], f32).repeat(n, axis=0)
    yc = np.array([[0., 0., 1., 1., 0., 0., 1., 1.]], f32).repeat(n, axis=0)
    zc = np.array([[0., 1., 1., 0., 0., 1., 1., 0.]], f32).repeat(n, axis=0)
    corners = np.concatenate([(dx[:, None] * (xc - 0.5))[:, :, None], (dy[:, None] * (yc - 1.0))[:, :, None],
                              (dz[:, None] * (zc - 0.5))[:, :, None]], axis=-1)
    ang = cam[:, 6]
    s, c = np.sin(ang), np.cos(ang)
    one, zero = np.ones_like(c), np.zeros_like(c)
    rot_t = np.stack([[c, zero, -s], [zero, one, zero], [s, zero, c]])  # axis = 1
    corners = np.einsum("aij,jka->aik", corners, rot_t) + cam[:, None, 0:3]
    p4 = np.concatenate([corners, np.zeros(list(corners.shape[:-1]) + [1])], axis=-1)
    p2 = p4 @ np.asarray(proj).T
    uv = p2[..., :2] / p2[..., 2:3]
    return np.concatenate([uv.min(axis=1), uv.max(axis=1)], axis=1).astype(f32)


def detections_to_kitti_annos(detections, calibs, class_names):
    """detections: per frame dict(box3d_lidar [K, 7] (x, y, z bottom, w, l, h, r; KITTI lidar frame), scores [K],
    label_preds [K]) as PointPillars.test_forward returns them (tensors or arrays; a frame without detections has
    K = 0, or the reference's marker row with score -1); calibs: per frame the reference's calibration tuple
    (P0, P1, P2, P3, R0_rect, V2C, ...), kitti_det.py:130-176; class_names: label -> name.
    -> per frame the record kitti_eval takes: name, truncated, occluded, alpha, bbox (image, x1 y1 x2 y2),
    dimensions (l, h, w), location (camera frame, bottom centre), rotation_y, score."""
    f32 = np.float32
    out = []
    for det, cal in zip(detections, calibs):
        def host(v, dt):
            return np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dt)

        box, score = host(det["box3d_lidar"], f32).reshape(-1, 7), host(det["scores"], f32).reshape(-1)
        label = host(det["label_preds"], np.int64).reshape(-1)
        # alpha on ALL rows, then the marker rows go (head.py:217-219, kitti_utils.py:245-272)
        alpha = (-np.arctan2(-box[:, 1], box[:, 0]) + box[:, 6]).astype(f32)
        keep = score >= 0
        box, score, label, alpha = box[keep], score[keep], label[keep], alpha[keep]
        if box.shape[0] == 0:
            out.append(_empty_anno())
            continue
        xyz = _lidar_to_camera(box[:, 0:3], cal)
        # (x, y, z)_cam, l, h, w, r in a float32 container, origin (.5, 1, .5): kitti_utils.py:101-114
        cam = np.concatenate([xyz, box[:, 4:5], box[:, 5:6], box[:, 3:4], box[:, 6:7]], axis=-1).astype(f32)
        n = box.shape[0]
        out.append(dict(truncated=np.zeros([n]), occluded=np.zeros([n]), alpha=alpha,
                        name=np.array([class_names[int(k)] for k in label]), bbox=_camera_box2d(cam, cal[2]),
                        dimensions=cam[:, 3:6], location=cam[:, 0:3], rotation_y=cam[:, 6], score=score))
    return out


def anno_to_label_lines(anno):
    """One record -> lines of a KITTI result file: type truncated occluded alpha x1 y1 x2 y2 h w l x y z ry score."""
    lines = []
    for i in range(len(anno["name"])):
        l, h, w = (float(v) for v in anno["dimensions"][i])
        x1, y1, x2, y2 = (float(v) for v in anno["bbox"][i])
        x, y, z = (float(v) for v in anno["location"][i])
        lines.append("%s %.2f %d %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.4f" % (
            anno["name"][i], float(anno["truncated"][i]), int(anno["occluded"][i]), float(anno["alpha"][i]), x1, y1,
            x2, y2, h, w, l, x, y, z, float(anno["rotation_y"][i]), float(anno["score"][i])))
    return lines


def write_label_files(annos, sample_ids, directory):
    """<directory>/<id>.txt per frame (empty file for a frame without detections)."""
    os.makedirs(directory, exist_ok=True)
    for anno, sid in zip(annos, sample_ids):
        with open(os.path.join(directory, f"{sid}.txt"), "w") as f:
            f.write("\n".join(anno_to_label_lines(anno)) + ("\n" if len(anno["name"]) else ""))
