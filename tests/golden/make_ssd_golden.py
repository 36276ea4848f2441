"""Golden vectors for the PointPillars SSD head path, from the reference's OWN Python executed through
tests/golden/paddle_shim.py (same method as make_python_golden.py; separate file so that the other fixture stays
byte-identical).

    python tests/golden/make_ssd_golden.py        # needs /root/reference; writes python_ssd.npz

What runs is the reference source itself:
  AnchorGenerator / AnchorGeneratorStride / generate_anchors_mask   models/detection/pointpillars/anchors_generator.py:21-210
  PointPillarsCoder.decode (second_box_decode_paddle)                 models/detection/pointpillars/pointpillars_coder.py:126-148
  SSDHead.forward / post_process / _single_post_process / _box_not_empty / _box_empty
                                                                       models/detection/pointpillars/pointpillars_head.py:31-196
  rotate_nms_pcdet                                                     models/layers/layer_libs.py:210-249 (iou3d_nms.nms_gpu =
                                                                       the reference's own IoU + sweep from oracle/_ref)
Two reduced-size cases (the anchor arithmetic does not depend on the map size):
  a  the KITTI-car head: 1 class, 2 anchors per location, 64 x 64 pillars -> 32 x 32 map, batch 2
  b  a three-class head: 3 anchor configs (6 anchors per location), 48 x 64 pillars -> 24 x 32 map, batch 3 with one
     frame whose pillars are too few for any anchor (the `_box_empty` branch) and a tight centre range
"""
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
from make_python_golden import shapes_blob  # noqa: E402

REF = "/root/reference"

CASES = {
    "a": dict(pcr=[0.0, -5.12, -3.0, 10.24, 5.12, 1.0], vs=[0.16, 0.16, 4.0], num_classes=1, channels=64, batch=2,
              anchor_configs=[dict(sizes=[1.6, 3.9, 1.56], anchor_strides=[0.32, 0.32, 0.0],
                                   anchor_offsets=[0.16, -4.96, -1.78], rotations=[0, 1.57], matched_threshold=0.6,
                                   unmatched_threshold=0.45)],
              head=dict(nms_score_threshold=0.05, nms_pre_max_size=1000, nms_post_max_size=300, nms_iou_threshold=0.5,
                        prediction_center_limit_range=[0.0, -5.12, -5.0, 10.24, 5.12, 5.0]),
              pillars=[700, 25]),
    # case b: the middle frame has ONE pillar -> no anchor passes the area test -> the `_box_empty` branch
    "b": dict(pcr=[0.0, -5.12, -3.0, 7.68, 5.12, 1.0], vs=[0.16, 0.16, 4.0], num_classes=3, channels=32, batch=3,
              anchor_configs=[dict(sizes=[1.6, 3.9, 1.56], anchor_strides=[0.32, 0.32, 0.0],
                                   anchor_offsets=[0.16, -4.96, -1.78], rotations=[0, 1.57], matched_threshold=0.6,
                                   unmatched_threshold=0.45),
                              dict(sizes=[0.6, 0.8, 1.73], anchor_strides=[0.32, 0.32, 0.0],
                                   anchor_offsets=[0.16, -4.96, -1.465], rotations=[0, 1.57], matched_threshold=0.5,
                                   unmatched_threshold=0.35),
                              dict(sizes=[0.6, 1.76, 1.73], anchor_strides=[0.32, 0.32, 0.0],
                                   anchor_offsets=[0.16, -4.96, -1.465], rotations=[0, 1.57], matched_threshold=0.5,
                                   unmatched_threshold=0.35)],
              head=dict(nms_score_threshold=0.3, nms_pre_max_size=200, nms_post_max_size=40, nms_iou_threshold=0.3,
                        prediction_center_limit_range=[0.5, -4.0, -2.2, 7.0, 4.0, -0.6]),
              pillars=[500, 1, 30]),
}


def features(tag, c, nx, ny):
    """The head's input map of case `tag` (rebuilt from the seed by the tests; not stored)."""
    rng = np.random.default_rng(170 + ord(tag))
    return rng.normal(size=(c["batch"], c["channels"], ny // 2, nx // 2)).astype(np.float32)


def head_outputs(tag, c, num_anchors):
    """post_process inputs (cls / box / dir predictions) of case `tag`, rebuilt from the seed by the tests."""
    rng = np.random.default_rng(270 + ord(tag))
    cls = rng.normal(-1.0, 2.0, (c["batch"], num_anchors, c["num_classes"])).astype(np.float32)
    box = rng.normal(0, 0.35, (c["batch"], num_anchors, 7)).astype(np.float32)
    dirp = rng.normal(0, 1, (c["batch"], num_anchors, 2)).astype(np.float32)
    return cls, box, dirp


def main():
    paddle = ps.install(REF)
    from oracle import pyoracle as O

    O.build(ref=True)
    T = ps.tensor

    def nms_gpu(boxes, thresh):
        keep = O.nms(boxes.numpy(), float(thresh), kind="ref" if O.have_ref() else "port")
        full = np.zeros(boxes.shape[0], np.int32)
        full[: len(keep)] = keep
        return T(full), T(np.array([len(keep)], np.int64))

    sys.modules["paddle3d.ops"].iou3d_nms = types.SimpleNamespace(nms_gpu=nms_gpu)
    ag = ps.load("paddle3d.models.detection.pointpillars.anchors_generator")
    hd = ps.load("paddle3d.models.detection.pointpillars.pointpillars_head")
    out = {}
    for tag, c in CASES.items():
        rng = np.random.default_rng({"a": 71, "b": 72}[tag])
        gen = ag.AnchorGenerator(output_stride_factor=2, point_cloud_range=c["pcr"], voxel_size=c["vs"],
                                 anchor_configs=c["anchor_configs"], anchor_area_threshold=1)
        nx, ny = int(gen.grid_size[0]), int(gen.grid_size[1])
        apl = 2 * len(c["anchor_configs"])
        head = hd.SSDHead(num_classes=c["num_classes"], feature_channels=c["channels"], num_anchor_per_loc=apl,
                          encode_background_as_zeros=True, use_direction_classifier=True, box_code_size=7, **c["head"])
        head.eval()
        k, f = shapes_blob(ps.fill_state(head, 40 + ord(tag)))
        feats = features(tag, c, nx, ny)
        with torch.no_grad():
            fw = head(T(feats))
        out.update({f"{tag}_head_keys": k, f"{tag}_head_shapes": f,
                    f"{tag}_fw_cls": fw["cls_preds"].numpy(), f"{tag}_fw_box": fw["box_preds"].numpy(),
                    f"{tag}_fw_dir": fw["dir_preds"].numpy()})
        out[f"{tag}_anchors"] = gen.anchors.numpy()
        out[f"{tag}_anchors_bv"] = gen.anchors_bv.numpy()
        a = gen.anchors.shape[0]
        # post_process inputs with a wide score spread (the random-weight head above gives sigmoid ~ 0.5 everywhere)
        cls, box, dirp = head_outputs(tag, c, a)
        coords = []
        for b, m in enumerate(c["pillars"]):
            cells = rng.choice(nx * ny, m, replace=False)
            co = np.zeros((m, 4), np.int32)
            co[:, 0], co[:, 2], co[:, 3] = b, cells // nx, cells % nx
            coords.append(co)
        coords = np.concatenate(coords)
        out[f"{tag}_coords"] = coords
        head.in_export_mode = True  # post_process on one frame, without the Sample containers (:93-107)
        for b in range(c["batch"]):
            this = T(coords[coords[:, 0] == b][:, 1:])
            mask = gen(this)
            out[f"{tag}_mask_{b}"] = mask.numpy()
            preds = dict(cls_preds=T(cls[b:b + 1]), box_preds=T(box[b:b + 1]), dir_preds=T(dirp[b:b + 1]))
            with torch.no_grad():
                res = head.post_process(None, preds, gen.anchors, mask)
            if b == 0:
                out[f"{tag}_decoded_0"] = preds["box_preds"].numpy()[0]
            out[f"{tag}_out_boxes_{b}"] = res["box3d_lidar"].numpy()
            out[f"{tag}_out_scores_{b}"] = res["scores"].numpy()
            out[f"{tag}_out_labels_{b}"] = res["label_preds"].numpy().astype(np.int64)
            print(tag, b, "mask", int(mask.sum()), "rows", res["scores"].shape[0])
    path = os.path.join(HERE, "python_ssd.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
