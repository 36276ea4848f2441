#!/usr/bin/env python
"""bench.py -- scenes/s of the CenterPoint-Pillars nuScenes hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the whole hot path (hard_voxelize -> PFN -> pointpillars_scatter -> SECOND backbone
+ FPN -> CenterHead -> centerpoint_postprocess, + for N > 1 the RCCL all-gather of the per-frame box
records) over one batch of `--batch` synthetic nuScenes-shaped sweeps per GPU (300k points x 5, 0.2 m
pillars, 30k-voxel cap; BASELINE.json configs[2]).  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.  Weak scaling: every rank processes its own batch, so value = N*B*K / time.

The line also carries
  roofline      hard_voxelize's launch sequence (the op the north star sets the >=50 % HBM target on):
                algorithmic bytes per launch / its HIP-event duration inside the timed region;
  rooflines     the same for the other ops (scatter / PFN on HBM, the dense graph on fp32 MFMA);
  cpu_baseline  the oracle pipeline (reference CPU voxelizer compiled from /root/reference when present,
                otherwise the port; torch-CPU dense graph) on a bounded sample, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec (guides: ~6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32-input MFMA peak (MI355X_MICROARCH.md)

N_POINTS, DIMS, P = 300_000, 5, 20


def algorithmic_bytes(v):
    """SURVEY.md section 8(d)."""
    vox = 4 * N_POINTS * DIMS + 4 * v * P * DIMS + 12 * v + 4 * v + 4
    scatter = 4 * v * 64 + 16 * v + 4 * 64 * 512 * 512
    pfn = 4 * v * P * DIMS + 4 * v + 16 * v + 4 * v * 64
    post = 4 * 128 * 128 * 70
    return dict(hard_voxelize=vox, pointpillars_scatter=scatter, pillar_feature_net=pfn,
                centerpoint_postprocess=post)


def dense_flops():
    """2*Cin*Cout*k*k*Hout*Wout per conv of backbone + FPN + CenterHead at 512x512 input."""
    fl = 0

    def conv(cin, cout, k, h):
        return 2 * cin * cout * k * k * h * h

    fl += conv(64, 64, 3, 256) + 3 * conv(64, 64, 3, 256)
    fl += conv(64, 128, 3, 128) + 5 * conv(128, 128, 3, 128)
    fl += conv(128, 256, 3, 64) + 5 * conv(256, 256, 3, 64)
    fl += conv(64, 128, 2, 128) + conv(128, 128, 1, 128) + 2 * 256 * 128 * 128 * 128  # deconv k2 s2: 1 tap/output
    fl += conv(384, 64, 3, 128)
    fl += 36 * conv(64, 64, 3, 128) + conv(64, 70, 3, 128)
    return fl


def make_batch(batch, seed0, device):
    from paddle3d_amd import synth

    uniq = min(batch, 4)
    frames = [synth.nuscenes_sweep(seed0 + i) for i in range(uniq)]
    arr = np.stack([frames[i % uniq] for i in range(batch)])
    return torch.from_numpy(arr).to(device)


def cpu_baseline(model_cpu, max_voxels, frames=8):
    """Oracle pipeline on the host cores (bounded sample)."""
    from oracle import pyoracle as O
    from paddle3d_amd import synth

    kind = "ref" if O.have_ref() else "port"
    threads = torch.get_num_threads()
    cfg = model_cpu.test_cfg
    t0 = time.perf_counter()
    for i in range(frames):
        pts = synth.nuscenes_sweep(100 + i)
        vox, co, npv, nv = O.hard_voxelize(pts, synth.NUSC_PILLAR, synth.NUSC_RANGE, P, max_voxels, kind)
        c4 = np.concatenate([np.zeros((nv, 1), np.int32), co[:nv]], 1)
        params = []
        for l in model_cpu.voxel_encoder.pfn_layers:
            params.append(dict(weight=l.linear.weight.t().detach().numpy(), gamma=l.norm.weight.detach().numpy(),
                               beta=l.norm.bias.detach().numpy(), mean=l.norm.running_mean.numpy(),
                               var=l.norm.running_var.numpy()))
        feats = O.pfn_forward_torch(vox[:nv], npv[:nv], c4, params, synth.NUSC_PILLAR, synth.NUSC_RANGE)
        canvas = O.pillar_scatter(feats, c4, 1, 512, 512)
        with torch.no_grad():
            x = model_cpu.dense_forward(torch.from_numpy(canvas))
            preds, _ = model_cpu.bbox_head(x)
        tasks = [{k: v.numpy() for k, v in p.items()} for p in preds]
        O.centerpoint_postprocess(tasks, cfg["voxel_size"] + [8.0], cfg["point_cloud_range"] + [0.0] * 4,
                                  cfg["post_center_limit_range"], [0, 1, 3, 5, 6, 8], cfg["down_ratio"],
                                  cfg["score_threshold"], cfg["nms"]["nms_iou_threshold"],
                                  cfg["nms"]["nms_pre_max_size"], cfg["nms"]["nms_post_max_size"], True)
    dt = time.perf_counter() - t0
    return dict(value=frames / dt, unit="scenes/s", cores=threads, kind="reference" if kind == "ref" else "port",
                sample=f"{frames} frames of the same workload: hard_voxelize = "
                       f"{'reference voxelize_op.cc:19-82 compiled from /root/reference' if kind == 'ref' else 'C port'}"
                       f" (1 thread), PFN/dense graph = torch CPU fp32 ({threads} threads), scatter/postprocess = C port")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="frames per GPU per step")
    ap.add_argument("--max-voxels", type=int, default=30000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from paddle3d_amd import centerpoint as cpm
    from paddle3d_amd import dist as pdist
    from paddle3d_amd._lib import lib

    rank, world, local = pdist.init_from_env()
    if world != args.gpus and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP ops have no CPU path")
    lib()  # fail loudly if libpaddle3d_amd.so is missing
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.backends.cudnn.benchmark = True  # only matters for PD3_DENSE_BACKEND=miopen (MIOpen find mode)
    torch.manual_seed(0)
    V, B = args.max_voxels, args.batch

    model = cpm.centerpoint_pillars_nuscenes(max_num_voxels=(V, V)).to(dev).eval()
    pts = make_batch(B, 100 + 16 * rank, dev)
    cfg = model.test_cfg
    max_per_img = cfg["max_per_img"]

    ev_names = ["start", "hard_voxelize", "pillar_feature_net", "pointpillars_scatter", "dense", "postprocess",
                "gather"]

    def step(events=None):
        def mark(i):
            if events is not None:
                events[i].record()

        mark(0)
        voxels, coors, npv, nv = model.voxelizer(pts)
        mark(1)
        b, v, p, d = voxels.shape
        feats = model.voxel_encoder(voxels.view(b * v, p, d), npv.view(b * v), coors.view(b * v, 4))
        mark(2)
        canvas = model.middle_encoder(feats, coors.view(b * v, 4), b)
        mark(3)
        x = model.dense_forward(canvas)
        preds, _ = model.bbox_head(x)
        mark(4)
        bx, sc, lb, cnt = model.bbox_head.predict_by_custom_op(preds, cfg, device_only=True)
        mark(5)
        rec = pdist.pack_records(bx, sc, lb, cnt, max_per_img)
        all_rec, all_cnt = pdist.gather_detections(rec, cnt)
        mark(6)
        return all_rec, all_cnt

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        barrier()
        events = [[torch.cuda.Event(enable_timing=True) for _ in ev_names] for _ in range(args.steps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            out = step(events[k])
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        per_op_ms = {}
        for i in range(1, len(ev_names)):
            per_op_ms[ev_names[i]] = float(np.mean([events[k][i - 1].elapsed_time(events[k][i])
                                                    for k in range(args.steps)]))
        alg = algorithmic_bytes(V)

        # HBM traffic per launch from the PMC passes (tools/gpu_traffic.sh -> profiles/*_traffic.json), when a
        # profile of this exact configuration is committed; collected offline because rocprofv3 --pmc cannot
        # wrap the timed run itself
        traffic = {}
        try:
            import glob
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
                t = json.load(open(path))
                if t.get("batch") == B and t.get("max_voxels") == V:
                    traffic = t
        except Exception:  # noqa: BLE001
            traffic = {}

        def hbm(name, key):
            a = alg[key] * B / (per_op_ms[name] * 1e-3) / 1e9
            tr = traffic.get(key, {}).get("bytes_per_launch")
            return dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBPS, unit="GB/s", frac=a / HBM_PEAK_GBPS,
                        traffic=tr, ms_per_launch=per_op_ms[name], units_per_launch=B,
                        algorithmic_bytes_per_unit=alg[key])

        tf = dense_flops() * B / (per_op_ms["dense"] * 1e-3) / 1e12
        rooflines = dict(
            hard_voxelize=hbm("hard_voxelize", "hard_voxelize"),
            pillar_feature_net=hbm("pillar_feature_net", "pillar_feature_net"),
            pointpillars_scatter=hbm("pointpillars_scatter", "pointpillars_scatter"),
            centerpoint_postprocess=hbm("postprocess", "centerpoint_postprocess"),
            dense_backbone_fpn_head=dict(bound="mfma", achieved=tf, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                                         frac=tf / MFMA_F32_PEAK_TFLOPS, traffic=None,
                                         ms_per_launch=per_op_ms["dense"], units_per_launch=B,
                                         flops_per_unit=dense_flops(),
                                         note="direct-form flops of the graph / time; the stride-1 3x3 layers run "
                                              "Winograd F(2x2,3x3) (2.25x fewer MFMA flops), all fp32"))
        line = {
            "metric": "scenes/sec CenterPoint-Pillars nuScenes 300k-pt sweeps",
            "value": world * B * args.steps / dt, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "CenterPoint-Pillars nuScenes 10-sweep: 300000 pts x 5 per scene, 0.2 m pillars "
                                   f"(512x512), P=20, max_voxels={V}, batch {B} scenes/GPU/step, random-init weights, "
                                   "full graph voxelize->PFN->scatter->SECOND+FPN->CenterHead->postprocess"
                                   + ("->RCCL all-gather" if world > 1 else ""),
                       "frames_per_gpu_per_step": B, "max_voxels": V, "parallelism": f"dp{world} (frames)"},
            "roofline": dict(rooflines["hard_voxelize"], kernel=(
                "hard_voxelize launch sequence (vt_route + vt_group + vt_count + vt_assign + vt_write)"
                if os.environ.get("PD3_VOXELIZE_PATH", "tiled") != "sort" else
                "hard_voxelize launch sequence (cell_key + radix sort + seg_head + scan + gather)")),
            "rooflines": rooflines,
            # `roofline` is the kernel the north star puts the HBM target on; by time the step is dominated by
            # the dense graph (rooflines["dense_backbone_fpn_head"], MFMA bound)
            "dominant_by_time": "dense_backbone_fpn_head",
            "per_op_ms": per_op_ms,
            "detections_first_frame": int(out[1][0].item()),
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                model_cpu = cpm.centerpoint_pillars_nuscenes(max_num_voxels=(V, V)).eval()
                model_cpu.load_state_dict(model.state_dict())
                line["cpu_baseline"] = cpu_baseline(model_cpu, V)
            except Exception as e:  # the baseline is reported, never required
                line["cpu_baseline"] = dict(value=None, unit="scenes/s", cores=0, kind="port", sample=f"failed: {e}")
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
