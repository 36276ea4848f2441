"""Golden vectors for the CenterHead post-processing ORCHESTRATION from the reference's own Python:
`CenterHead.predict` -> `post_processing` -> `single_post_processing` (paddle3d/models/detection/centerpoint/
center_head.py:341-441, :443-510, :568-585), the non-custom-op form of what `predict_by_custom_op` hands to the
`centerpoint_postprocess` CUDA operator, executed through tests/golden/paddle_shim.py.

    python tests/golden/make_predict_golden.py        # needs /root/reference; writes python_predict.npz

What it pins (the CUDA operator has no CPU or Python form of its own): sigmoid -> max / argmax over classes, exp(dim),
atan2(rot), the centre decode `(cell + reg) * down_ratio * voxel_size + range`, the `>` score test, the column
order handed to `rotate_nms_pcdet` (descending stable sort, top nms_pre_max_size, rotated NMS over the reference's
IoU + sweep from oracle/_ref, nms_post_max_size), the per-task label offsets and the task-order concatenation.
What it does NOT pin: the centre-range test, which the two reference paths apply to different values (this Python on
the decoded centre, the CUDA operator on the raw reg / height maps, postprocess.cu:72-77) -- the range here is wide
enough that both are always true.  Every task keeps at least one cell (the two paths also disagree on the label of
the fake row of an empty task).
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

REF = "/root/reference"
TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "construction_vehicle"]),
         dict(num_class=2, class_names=["pedestrian", "traffic_cone"])]
CFG = dict(post_center_limit_range=[-100.0, -100.0, -100.0, 100.0, 100.0, 100.0], score_threshold=0.1, down_ratio=4,
           voxel_size=[0.2, 0.2], point_cloud_range=[-12.8, -12.8],
           nms=dict(nms_iou_threshold=0.2, nms_pre_max_size=300, nms_post_max_size=40))
H = W = 32
BATCH = 2


class Cfg(dict):
    """config node: attribute access + dict.get, like the reference's test_cfg"""

    def __getattr__(self, k):
        v = self[k]
        return Cfg(v) if isinstance(v, dict) else v


def head_maps(seed=911):
    """The head outputs of every task (rebuilt from the seed by the tests; not stored)."""
    rng = np.random.default_rng(seed)
    out = []
    for t in TASKS:
        out.append(dict(hm=rng.normal(-2.5, 1.6, (BATCH, t["num_class"], H, W)).astype(np.float32),
                        reg=rng.random((BATCH, 2, H, W)).astype(np.float32),
                        height=rng.normal(-1.0, 0.5, (BATCH, 1, H, W)).astype(np.float32),
                        dim=rng.normal(0.6, 0.3, (BATCH, 3, H, W)).astype(np.float32),
                        vel=rng.normal(0, 1, (BATCH, 2, H, W)).astype(np.float32),
                        rot=rng.normal(0, 1, (BATCH, 2, H, W)).astype(np.float32)))
    return out


def main():
    ps.install(REF)
    from oracle import pyoracle as O

    O.build(ref=True)
    T = ps.tensor

    def nms_gpu(boxes, thresh):
        keep = O.nms(boxes.numpy(), float(thresh), kind="ref" if O.have_ref() else "port")
        full = np.zeros(boxes.shape[0], np.int32)
        full[: len(keep)] = keep
        return T(full), T(np.array([len(keep)], np.int64))

    sys.modules["paddle3d.ops"].iou3d_nms = types.SimpleNamespace(nms_gpu=nms_gpu)
    ch = ps.load("paddle3d.models.detection.centerpoint.center_head")
    head = ch.CenterHead(in_channels=64, tasks=TASKS, common_heads=dict(reg=(2, 2), height=(1, 2), dim=(3, 2),
                                                                        rot=(2, 2), vel=(2, 2)),
                         share_conv_channel=64, num_hm_conv=2)
    head.eval()
    preds = [{k: T(v) for k, v in t.items()} for t in head_maps()]
    with torch.no_grad():
        rets = head.predict({}, preds, Cfg(CFG))
    out = {}
    for b, r in enumerate(rets):
        out[f"boxes_{b}"] = r["box3d_lidar"].numpy()
        out[f"scores_{b}"] = r["scores"].numpy()
        out[f"labels_{b}"] = r["label_preds"].numpy().astype(np.int64)
        assert (out[f"scores_{b}"] >= 0).all(), "a task came back empty: pick another seed"
        print(b, out[f"boxes_{b}"].shape, np.bincount(out[f"labels_{b}"]))
    path = os.path.join(HERE, "python_predict.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} KB")


if __name__ == "__main__":
    main()
