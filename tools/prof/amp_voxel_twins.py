"""Where do the CenterPoint-Voxel AMP graph's detections part from the fp32 graph's?  Half-range copy of config 4,
BatchNorm statistics and heads like a trained net's; per AMP scope (sparse encoder only / dense graph only / both): BEV
map error, raw heat-map logit error against the logits' own spread, twin fractions at several score tolerances."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from paddle3d_amd import centerpoint as cpm, nuscenes_bridge as nb, synth

torch.manual_seed(9)
pcr = [-28.8, -28.8, -5.0, 28.8, 28.8, 3.0]
model = cpm.centerpoint_voxels_nuscenes(max_num_voxels=(120000, 120000), point_cloud_range=pcr).cuda().eval()
synth.trained_like_batchnorm(model, 7)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(93 + i) for i in range(frames)])).cuda()
DC = len(sys.argv) > 2 and sys.argv[2] == "dc"
synth.trained_like_heads(model, pts[:2], remove_dc=DC)
print("# heads calibrated", "with the mean-feature response removed (remove_dc)" if DC else "as the tests do")


def run(enc, dense):
    model.set_amp(False)
    model.middle_encoder.amp = enc
    model.backbone.amp = dense
    model.bbox_head.amp = dense
    model.backbone.amp_out_f16 = dense and model.neck.amp_ok(None)
    with torch.no_grad():
        bev = model.extract_pillars(pts)
        preds, _ = model.bbox_head(model.dense_forward(bev))
        hm = [p["hm"].float() for p in preds]
        dets = model.bbox_head.predict_by_custom_op(preds, model.test_cfg)
    return bev, hm, [{k: d[k].cpu().numpy() for k in ("box3d_lidar", "scores", "label_preds")} for d in dets]


b32, h32, d32 = run(False, False)
for name, enc, dense in (("encoder only", True, False), ("dense only", False, True), ("both", True, True)):
    b, h, d = run(enc, dense)
    rel = float((b - b32).abs().max() / b32.abs().max())
    rms = float((b - b32).pow(2).mean().sqrt() / b32.pow(2).mean().sqrt())
    # logit error against the distance between the 99 % and 99.9 % quantiles of the fp32 logits (what the calibrated
    # heads stretch over the score range 0.1 .. 0.35)
    worst = 0.0
    for a, r in zip(h, h32):
        for c in range(r.shape[1]):
            v = r[:, c].reshape(-1)
            n = v.numel()
            lo = v.kthvalue(int(n * 0.99)).values
            hi = v.kthvalue(int(n * 0.999)).values
            top = v >= lo
            e = (a[:, c].reshape(-1) - v)[top].abs()
            worst = max(worst, float(e.mean() / (hi - lo)))
    line = f"{name:13s} map max {rel:.2e} rms {rms:.2e}  mean |dlogit| of the top 1 % cells / (q99.9 - q99) {worst:.3f}  twins:"
    for tol in (0.02, 0.05, 0.1):
        m = nb.unmatched_detections(d, d32, score_tol=tol)
        line += f"  tol {tol}: {1 - m['unmatched'] / max(1, m['total']):.4f}"
    line += f"  ({m['total']} boxes)"
    print(line)

# --- the dense-only case in detail: per head output error, and what the unmatched boxes look like
model.set_amp(False)
with torch.no_grad():
    bev = model.extract_pillars(pts)
    p32, _ = model.bbox_head(model.dense_forward(bev))
    model.backbone.amp = model.bbox_head.amp = True
    model.backbone.amp_out_f16 = model.neck.amp_ok(None)
    p16, _ = model.bbox_head(model.dense_forward(bev))
    d16 = model.bbox_head.predict_by_custom_op(p16, model.test_cfg)
    d32 = model.bbox_head.predict_by_custom_op(p32, model.test_cfg)
for t, (a, r) in enumerate(zip(p16, p32)):
    print("task", t, {k: f"{float((a[k].float() - r[k].float()).abs().max()):.3e}/{float(r[k].float().abs().max()):.2e}" for k in r})
hist = {"none_within_2m": 0, "far": 0, "score": 0, "taken": 0}
for a, r in zip(d16, d32):
    pb, ps, pl = a["box3d_lidar"].cpu().numpy(), a["scores"].cpu().numpy(), a["label_preds"].cpu().numpy()
    tb, ts, tl = r["box3d_lidar"].cpu().numpy(), r["scores"].cpu().numpy(), r["label_preds"].cpu().numpy()
    for j in range(len(ts)):
        cand = np.nonzero(pl == tl[j])[0]
        if not len(cand):
            hist["none_within_2m"] += 1
            continue
        dd = np.hypot(pb[cand, 0] - tb[j, 0], pb[cand, 1] - tb[j, 1])
        k = int(np.argmin(dd))
        if dd[k] > 2.0:
            hist["none_within_2m"] += 1
        elif dd[k] > 0.5:
            hist["far"] += 1
        elif abs(ps[cand[k]] - ts[j]) > 0.02:
            hist["score"] += 1
        else:
            hist["taken"] += 1
print("fp32 boxes by nearest AMP box of the class:", hist, "per frame counts", [len(r["scores"]) for r in d32],
      [len(a["scores"]) for a in d16])
