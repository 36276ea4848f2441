"""Generate the committed golden vectors from the REFERENCE's own code (oracle/_ref).

    python tests/golden/make_golden.py          # run in the dev container (needs /root/reference)

The reference ships no golden vectors, known-answer tests or op tests for this path (SURVEY.md
section 4 / 8c), so these fixtures are produced by compiling the reference's arithmetic line ranges
(oracle/extract_ref.sh, oracle/ref_wrap.cpp) and running them on small seeded inputs.  Inputs are stored
next to the outputs so the fixtures are self-contained on the GPU box, where /root/reference is absent.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from paddle3d_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    O.build(ref=True)
    assert O.have_ref(), "oracle/_ref could not be built (is /root/reference mounted?)"
    out = {}
    # ---- hard_voxelize: nuScenes pillars with the cap hit, KITTI, 0.075 m voxels -------------------
    cases = {
        "vox_nusc": (synth.nuscenes_sweep(0, n_points=6000), synth.NUSC_PILLAR, synth.NUSC_RANGE, 20, 1500),
        "vox_kitti": (synth.kitti_frame(1, n_points=3000), synth.KITTI_PILLAR, synth.KITTI_RANGE, 32, 16000),
        "vox_fine": (synth.nuscenes_sweep(2, n_points=4000), synth.NUSC_VOXEL, synth.NUSC_VOXEL_RANGE, 10, 2500),
    }
    for name, (pts, vs, pr, p, v) in cases.items():
        vox, co, npv, nv = O.hard_voxelize(pts, vs, pr, p, v, kind="ref")
        out[f"{name}_points"] = pts
        out[f"{name}_cfg"] = np.array(list(vs) + list(pr) + [p, v], np.float64)
        out[f"{name}_coords"] = co[:nv]
        out[f"{name}_num_points"] = npv[:nv]
        out[f"{name}_num_voxels"] = np.array([nv], np.int32)
        # voxels are mostly padding: store only the occupied slots, row-major
        k = np.arange(p)[None, :] < npv[:nv, None]
        out[f"{name}_stored_points"] = vox[:nv][k]
    # ---- rotated IoU / overlap / NMS ---------------------------------------------------------------
    boxes, _ = synth.nms_boxes(3, n=96)
    out["iou_boxes"] = boxes
    out["iou_matrix"] = O.boxes_iou_bev(boxes, boxes, kind="ref")
    out["overlap_matrix"] = O.boxes_overlap_bev(boxes, boxes, kind="ref")
    for thr in (0.2, 0.5):
        out[f"nms_keep_{int(thr * 10)}"] = O.nms(boxes, thr, kind="ref")
        out[f"nms_normal_keep_{int(thr * 10)}"] = O.nms(boxes, thr, normal=True, kind="ref")
    # ---- CenterPoint decode kernel -----------------------------------------------------------------
    t = synth.center_head_outputs(5, feat_h=16, feat_w=24, num_classes=(2,), n_peaks=12)[0]
    sig = (1.0 / (1.0 + np.exp(-t["hm"][0].astype(np.float64)))).astype(np.float32)
    score = sig.max(0).reshape(-1)
    expdim = np.exp(t["dim"][0].astype(np.float64)).astype(np.float32)
    bx, mask, sidx = O.centerpoint_decode_ref(score, t["reg"][0], t["height"][0], expdim, t["vel"][0], t["rot"][0],
                                              0.1, 24, 4.0, [0.2, 0.2], [-51.2, -51.2],
                                              [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0])
    for k in ("reg", "height", "vel", "rot"):
        out[f"decode_{k}"] = t[k][0]
    out["decode_score"], out["decode_expdim"] = score, expdim
    out["decode_boxes"], out["decode_mask"] = bx, mask
    # ---- bev_pool_v2 forward / backward ------------------------------------------------------------
    d = synth.bev_pool_inputs(6, n_cam=1, depth_bins=12, fh=4, fw=8, channels=8, bev=16)
    args = [d[k] for k in ("depth", "feat", "ranks_depth", "ranks_feat", "ranks_bev", "interval_lengths",
                           "interval_starts")]
    for k, a in zip(("depth", "feat", "ranks_depth", "ranks_feat", "ranks_bev", "interval_lengths",
                     "interval_starts"), args):
        out[f"bev_{k}"] = a
    out["bev_shape"] = np.array(d["bev_feat_shape"], np.int32)
    out["bev_out"] = O.bev_pool_v2(*args, d["bev_feat_shape"], kind="ref")
    g = np.random.default_rng(7).normal(size=d["bev_feat_shape"]).astype(np.float32)
    order = np.argsort(d["ranks_feat"], kind="stable")
    rb, rd, rf = d["ranks_bev"][order], d["ranks_depth"][order], d["ranks_feat"][order]
    flag = np.ones(len(rf), bool)
    flag[1:] = rf[1:] != rf[:-1]
    starts = np.nonzero(flag)[0].astype(np.int32)
    lengths = np.diff(np.append(starts, len(rf))).astype(np.int32)
    dg, fg = O.bev_pool_v2_bkwd(g, d["depth"], d["feat"], rd, rf, rb, lengths, starts, kind="ref")
    out.update(bevb_out_grad=g, bevb_ranks_depth=rd, bevb_ranks_feat=rf, bevb_ranks_bev=rb,
               bevb_interval_starts=starts, bevb_interval_lengths=lengths, bevb_depth_grad=dg, bevb_feat_grad=fg)
    path = os.path.join(HERE, "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
