#!/usr/bin/env python
"""bench.py -- scenes/s of the CenterPoint-Pillars nuScenes hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads (the default is BASELINE.json's headline; the others give the driver a clock on configs 4 / 5):
  centerpoint_pillars  hard_voxelize -> PFN -> pointpillars_scatter -> SECOND backbone + FPN -> CenterHead ->
                       centerpoint_postprocess (+ for N > 1 the RCCL all-gather of the per-frame box records) over
                       one batch of `--batch` synthetic nuScenes-shaped sweeps per GPU (300k points x 5, 0.2 m
                       pillars, 30k-voxel cap; BASELINE.json configs[2])
  centerpoint_voxel    the same graph with 0.075 m voxels, VoxelMean and the sparse-conv middle encoder (configs[3])
  bev_pool_v2          the camera->BEV pooling op at BEVDet4D size (configs[4]'s bev_pool)
A "step" is ONE pass of the whole path over one batch.  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.  Weak scaling: every rank processes its own batch, so value = N*B*K / time.

The line also carries
  roofline      hard_voxelize's launch sequence (the op the north star sets the >=50 % HBM target on):
                algorithmic bytes per launch / its HIP-event duration inside the timed region;
  rooflines     the same for the other ops: scatter / postprocess on HBM; PFN and the dense graph on the fp32
                MFMA peak, both as EXECUTED flops (what the matrix cores do) and as direct-form flops;
  extras        h2d_inclusive scenes/s (the batch copied from pinned host memory inside every step), batch-1
                latency, the copy / fill ceilings measured on this device;
  cpu_baseline  the oracle pipeline (reference CPU voxelizer compiled from /root/reference when present,
                otherwise the port; torch-CPU dense graph) on a bounded sample, rank 0 at N=1 only: all host
                threads, plus 1 thread and P processes x 1 thread.
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


def algorithmic_bytes(v, n=N_POINTS, d=DIMS, p=P):
    """SURVEY.md section 8(d)."""
    vox = 4 * n * d + 4 * v * p * d + 12 * v + 4 * v + 4
    scatter = 4 * v * 64 + 16 * v + 4 * 64 * 512 * 512
    pfn = 4 * v * p * d + 4 * v + 16 * v + 4 * v * 64
    post = 4 * 128 * 128 * 70
    return dict(hard_voxelize=vox, pointpillars_scatter=scatter, pillar_feature_net=pfn,
                centerpoint_postprocess=post)


def dense_flops():
    """(direct-form, executed) flops per scene of backbone + FPN + CenterHead at 512x512 input:
    direct = 2*Cin*Cout*k*k*Hout*Wout per conv; executed = what the kernels issue: the stride-1 3x3 layers run
    Winograd F(4x4,3x3) = 36 multiplies per 4x4 outputs x 9 taps -> a quarter of the direct flops."""
    def conv(cin, cout, k, h):
        return 2 * cin * cout * k * k * h * h

    s1 = 3 * conv(64, 64, 3, 256) + 5 * conv(128, 128, 3, 128) + 5 * conv(256, 256, 3, 64)   # backbone, stride 1
    s1 += conv(384, 64, 3, 128) + 36 * conv(64, 64, 3, 128)                                   # head, Winograd
    s2 = conv(64, 64, 3, 256) + conv(64, 128, 3, 128) + conv(128, 256, 3, 64)                 # stride 2, direct
    other = conv(64, 128, 2, 128) + conv(128, 128, 1, 128) + 2 * 256 * 128 * 128 * 128        # FPN patch GEMMs
    other += conv(64, 70, 3, 128)                                                             # final grouped (VALU)
    return s1 + s2 + other, s1 / 4 + s2 + other


def pfn_flops(v, mfma_per_scene=None):
    """(direct-form, executed) flops per scene of the two-layer PFN: direct = per real-or-padded point
    2*(10*32 + 64*64); executed = the v_mfma_f32_16x16x4_f32 (2048 flops each) the kernel issues.
    The packed form (round 3) packs the stored points of 8 consecutive pillars into 16-row blocks, so the count depends
    on the fill levels and is taken from the batch (`pfn_packed_mfma`); without it, the per-pillar
    form's 38 per pillar slot are assumed."""
    mf = mfma_per_scene if mfma_per_scene is not None else v * 38
    return v * P * 2 * (10 * 32 + 64 * 64), mf * 2048


def pfn_packed_mfma(npv, p, chunk=8):
    """MFMA instructions the packed PFN kernel issues for num_points_per_voxel `npv` [B, V] (csrc/pfn.hip): 38 per
    16-row block of a chunk's stored points (6 layer 1 + 32 layer 2) and 32 per chunk that holds a pillar (the
    row-independent half of layer 2 for the chunk's 8 pillars)."""
    n = npv.reshape(-1).to(torch.int64).clamp(min=0, max=p)
    pad = (-n.numel()) % chunk
    if pad:
        n = torch.cat([n, n.new_zeros(pad)])
    rows = n.reshape(-1, chunk).sum(1)
    return int((((rows + 15) // 16) * 38 + (rows > 0).to(torch.int64) * 32).sum().item())


def make_batch(batch, seed0, device=None, pin=False):
    from paddle3d_amd import synth

    arr = np.stack([synth.nuscenes_sweep(seed0 + i) for i in range(batch)])  # `batch` DISTINCT frames
    t = torch.from_numpy(arr)
    if pin:
        return t.pin_memory()
    return t.to(device)


def _oracle_scene(model_cpu, max_voxels, seed):
    """One scene through the oracle pipeline (the CPU statement of the whole path)."""
    from oracle import pyoracle as O
    from paddle3d_amd import synth

    kind = "ref" if O.have_ref() else "port"
    O.centerpoint_pillars_pipeline(model_cpu, [synth.nuscenes_sweep(seed)], P, max_voxels, kind, dense_batch=1)
    return kind


def map_proxy(model, model_cpu, max_voxels, frames, dev):
    """mAP-shaped evidence without a dataset: `frames` synthetic scenes through the oracle pipeline (CPU) and through
    the device pipeline with the same weights; nuScenes-style AP (centre distance 0.5 / 1 / 2 / 4 m,
    paddle3d_amd.nuscenes_bridge) of the device's detections scored against the oracle's, and the other way round.
    The heads' last heat-map convolutions are scaled by 30 (bias per class: synth.trained_like_heads) on BOTH sides: plain
    random-init heads put all scores of a class into a band 0.003 wide, where the top-K cut and the NMS order are
    thousands of near-ties and the figure measures tie-breaking of 1e-6 noise (0.996 CPU against CPU), not the
    pipelines; spread like a trained head's (0.10 .. 0.77) it is insensitive to such noise (1.0 CPU against CPU)."""
    import copy

    from oracle import pyoracle as O
    from paddle3d_amd import nuscenes_bridge as nb
    from paddle3d_amd import synth

    model, model_cpu = copy.deepcopy(model), copy.deepcopy(model_cpu)
    with torch.no_grad():
        for m in (model, model_cpu):
            # BatchNorm statistics like a trained net's (with the constructor's identity statistics the activations
            # shrink layer by layer and no cell reaches the score threshold at all), the same values on both sides
            g = torch.Generator().manual_seed(0)
            for mod in m.modules():
                if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                    mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
                    mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
                    mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                    mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
    pts = np.stack([synth.nuscenes_sweep(700 + i) for i in range(frames)])
    # heads like a trained net's: gain 30, the bias per class from the heat maps of two frames (every class of every
    # task crosses the score threshold in 1 % of the cells); the CPU twin takes the device model's parameters
    synth.trained_like_heads(model, torch.from_numpy(pts[:2]).to(dev))
    model_cpu.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    if hasattr(model_cpu, "invalidate"):
        model_cpu.invalidate()
    t0 = time.perf_counter()
    ref = O.centerpoint_pillars_pipeline(model_cpu, pts, P, max_voxels)
    t_cpu = time.perf_counter() - t0
    got = []
    with torch.no_grad():
        for b0 in range(0, frames, 16):
            for d in model.test_forward(torch.from_numpy(pts[b0:b0 + 16]).to(dev)):
                got.append({k: d[k].cpu().numpy() for k in ("box3d_lidar", "scores", "label_preds")})
    fwd, back = nb.nuscenes_style_map(got, ref), nb.nuscenes_style_map(ref, got)
    return dict(value=fwd["mAP"], reverse=back["mAP"], frames=frames, classes_scored=fwd["classes_scored"],
                oracle_boxes_without_device_twin=nb.unmatched_detections(got, ref),
                device_boxes_without_oracle_twin=nb.unmatched_detections(ref, got),
                per_class={str(c): round(v, 5) for c, v in fwd["per_class"].items()},
                oracle_detections=int(sum(len(r["scores"]) for r in ref)),
                device_detections=int(sum(int((g["scores"] >= 0).sum()) for g in got)), cpu_seconds=t_cpu,
                note="AP of the HIP pipeline's detections against the oracle pipeline's (as if those were the "
                     "annotations), mean over classes and the four centre-distance thresholds; random-init weights "
                     "(heat-map heads scaled so that scores spread like a trained head's), "
                     "so the absolute detections mean nothing -- the figure says how far the two pipelines' outputs "
                     "are apart on the mAP scale (1.0 = identical detection sets; the north star's 0.1 mAP = 0.001 "
                     "here, and the AP is quantised: one box without a twin costs its class one of 90 recall bins at every "
                     "threshold = 1/900 of the mean, see *_without_*_twin for the counts); "
                     "tests/test_model_gpu.py::test_map_proxy_64_frames runs 64 frames")


def _oracle_worker(args):
    """Process-pool worker of the P x 1-thread leg: builds its own model, times one scene with one thread."""
    state_path, max_voxels, seed = args
    torch.set_num_threads(1)
    from paddle3d_amd import centerpoint as cpm

    m = cpm.centerpoint_pillars_nuscenes(max_num_voxels=(max_voxels, max_voxels)).eval()
    m.load_state_dict(torch.load(state_path))
    t0 = time.perf_counter()
    _oracle_scene(m, max_voxels, seed)
    return time.perf_counter() - t0


def cpu_baseline(model_cpu, max_voxels, frames=6):
    """Oracle pipeline on the host cores (bounded samples, ~25 s in all)."""
    import multiprocessing as mp
    import tempfile

    threads = torch.get_num_threads()
    t0 = time.perf_counter()
    for i in range(frames):
        kind = _oracle_scene(model_cpu, max_voxels, 100 + i)
    dt_all = time.perf_counter() - t0
    src = ("reference voxelize_op.cc:19-82 compiled from /root/reference" if kind == "ref" else "C port")
    out = dict(value=frames / dt_all, unit="scenes/s", cores=threads, kind="reference" if kind == "ref" else "port",
               sample=f"{frames} frames of the same workload: hard_voxelize = {src} (1 thread), PFN / dense graph = "
                      f"torch CPU fp32 ({threads} threads), scatter / postprocess = C port")
    # (i) one thread, one frame at a time -- how the reference's CPU path runs a frame (voxelize_op.cc:36)
    torch.set_num_threads(1)
    try:
        t0 = time.perf_counter()
        _oracle_scene(model_cpu, max_voxels, 100)
        dt1 = time.perf_counter() - t0
    finally:
        torch.set_num_threads(threads)
    out["one_thread"] = dict(value=1.0 / dt1, unit="scenes/s", cores=1, sample="1 frame, every stage on 1 thread")
    # (ii) P processes x 1 thread, one frame each
    procs = max(1, min(os.cpu_count() or 1, 32))
    try:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "state.pt")
            torch.save(model_cpu.state_dict(), path)
            ctx = mp.get_context("spawn")
            t0 = time.perf_counter()
            with ctx.Pool(procs) as pool:
                pool.map(_oracle_worker, [(path, max_voxels, 100 + i) for i in range(procs)])
            dtp = time.perf_counter() - t0
        out["procs_x_1thread"] = dict(value=procs / dtp, unit="scenes/s", cores=procs,
                                      sample=f"{procs} processes x 1 thread, one frame each (wall time incl. process "
                                             "start and model construction)")
    except Exception as e:  # noqa: BLE001 -- a reported extra, never required
        out["procs_x_1thread"] = dict(value=None, unit="scenes/s", cores=procs, sample=f"failed: {e}")
    return out


class NodeSampler:
    """What else the node is doing while the bench runs: the pool's boxes are 8-GPU nodes shared with other jobs, and
    the ops next to the step boundary have run 1.2x slower on some of them (round 2 called it the "slow box").  A
    thread reads the amdgpu sysfs files every 20 ms: every card's gpu_busy_percent and current sclk level.  Recorded
    in `extras.node_state`, so that a slow line can be told from a loaded node by data instead of by guess."""

    def __init__(self, period=0.02):
        import glob
        import threading

        self.cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/gpu_busy_percent"))
        self.period, self.samples, self._stop = period, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read(path):
        try:
            return open(path).read()
        except OSError:
            return ""

    def _run(self):
        while not self._stop.is_set():
            row = []
            for c in self.cards:
                busy = self._read(c).strip()
                cur = [l for l in self._read(c.replace("gpu_busy_percent", "pp_dpm_sclk")).splitlines() if "*" in l]
                mhz = "".join(ch for ch in (cur[0].split(":")[1] if cur else "") if ch.isdigit())
                row.append((int(busy) if busy.isdigit() else -1, int(mhz) if mhz else -1))
            self.samples.append(row)
            self._stop.wait(self.period)

    def __enter__(self):
        if self.cards:
            self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.cards:
            self._t.join(timeout=1.0)

    def summary(self):
        if not self.samples:
            return dict(note="no amdgpu sysfs on this host")
        a = np.asarray(self.samples)  # [samples, cards, (busy, sclk MHz)]
        busy_mean = a[:, :, 0].mean(0)
        return dict(cards=len(self.cards), samples=int(a.shape[0]), period_s=self.period,
                    gpu_busy_percent_mean=[round(float(v), 1) for v in busy_mean],
                    sclk_mhz_median=[int(np.median(a[:, k, 1])) for k in range(a.shape[1])],
                    cards_busy_over_50_percent=int((busy_mean > 50).sum()),
                    note="all cards of the node, sampled while the repeated blocks ran (this process drives one of "
                         "them; the others belong to other jobs)")


def measured_ceilings(dev, mb=384):
    """Device copy / fill rates of this box (GB/s), the practical ceilings next to the 8 TB/s spec."""
    n = mb * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)

    def t(fn, it=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e-3

    fill = n * 4 / t(lambda: a.fill_(1.0)) / 1e9
    copy = 2 * n * 4 / t(lambda: b.copy_(a)) / 1e9
    return dict(fill_GBps=fill, copy_GBps_read_plus_write=copy, buffer_MB=mb)


class _HostEvent:
    """CPU stand-in for torch.cuda.Event (the --stub-ops launch-path test runs without a GPU)."""

    def __init__(self):
        self.t = 0.0

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def _events(names, steps, dev):
    if dev.type != "cuda":
        return [[_HostEvent() for _ in names] for _ in range(steps)]
    return [[torch.cuda.Event(enable_timing=True) for _ in names] for _ in range(steps)]


def _timed_loop(step, args, world, dev, names, finish=None):
    """`finish` (optional): called once after the K-th step INSIDE the timed region, before the closing synchronize +
    barrier -- the overlapped result hand-off (dist.GatherPipeline) completes the batch still in flight there, so
    every collective of the K steps is inside the K steps' time; its return value replaces the last output.
    The contract: W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides, MAX
    over ranks.  Returns (seconds, per-op milliseconds (median over the K steps of the HIP-event intervals), last
    output, info).  info["repeats_s"]: the same K steps timed `--repeats` more times after the contract block (the
    0.2 s region of a 20-step run moves by a few per cent from box to box; the spread is reported, `value` is always
    the first block); info["ranks_seen"]: ranks that answered an all-gather after the timed region."""
    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def block(events):
        sync()
        barrier()
        t0 = time.perf_counter()
        out = None
        for k in range(args.steps):
            out = step(events[k] if events is not None else None)
        if finish is not None:
            out = finish(out)
        sync()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    with torch.no_grad():
        for _ in range(args.warmup):
            step(None)
        if finish is not None:
            finish(None)
        events = _events(names, args.steps, dev)
        dt, out = block(events)
        sync()
        per_op_ms = {names[i]: float(np.median([events[k][i - 1].elapsed_time(events[k][i])
                                                for k in range(args.steps)])) for i in range(1, len(names))}
        with NodeSampler() as sampler:
            repeats = [block(None)[0] for _ in range(max(0, args.repeats))]
        node = sampler.summary() if args.repeats > 0 else None
    seen = 1
    if world > 1:
        mine = torch.tensor([torch.distributed.get_rank()], dtype=torch.int64, device=dev)
        allr = torch.empty(world, dtype=torch.int64, device=dev)
        torch.distributed.all_gather_into_tensor(allr, mine)
        seen = int(torch.unique(allr).numel())
    info = dict(dt=dt, repeats_s=repeats, ranks_seen=seen, node=node)
    _LAST_LOOP.clear()
    _LAST_LOOP.update(info)
    return dt, per_op_ms, out, info



def _rank_max_seconds(dt, world, dev):
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def _timed_region(fn, world, dev):
    """barrier + synchronize, fn(), synchronize + barrier; seconds, MAX over ranks (the contract's bracket)."""
    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    sync()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    fn()
    sync()
    if world > 1:
        torch.distributed.barrier()
    return _rank_max_seconds(time.perf_counter() - t0, world, dev)


def strong_scaling(run_batch, flush, make_frames, frames, batch, rank, world, dev, passes=3):
    """STRONG scaling beside the contract's weak-scaling line: a FIXED set of `frames` scenes (the same scenes for any
    number of ranks), sharded round-robin by dist.shard_frames (frame i -> rank i % world), every rank walking its
    shard in batches of `batch` with the per-batch all-gather of the result records; one pass = every frame of the set
    once.  value = frames * passes / seconds (MAX over ranks).  Every rank must call this (collectives inside)."""
    from paddle3d_amd import dist as pdist

    if frames % world != 0:
        return dict(value=None, note=f"skipped: {frames} frames do not split evenly over {world} ranks")
    mine = pdist.shard_frames(frames, rank, world)
    shard = make_frames(mine)
    batches = [shard[i:i + batch] for i in range(0, len(mine), batch)]

    def one_pass():
        for b in batches:
            run_batch(b)
        if flush is not None:
            flush()

    one_pass()  # warm-up (batch shapes of the shard may differ from the weak-scaling batch)
    dt = _timed_region(lambda: [one_pass() for _ in range(passes)], world, dev)
    return dict(value=frames * passes / dt, unit="scenes/s", scaling="strong", frames=frames, passes=passes,
                frames_per_rank=len(mine), batches_per_rank_per_pass=len(batches), ms_per_pass=dt / passes * 1e3,
                note="a fixed frame set sharded by dist.shard_frames over the ranks (total work constant as N grows), "
                     "inputs resident in HBM, one all-gather of the result records per batch; the >= 6x target at 8 "
                     "GPUs is read from the weak-scaling `value` of the driver's N = 1, 2, 4, 8 lines (the "
                     "contract), this figure shows what the same node does on a fixed job")


def h2d_inclusive(run_batch, flush, host_batch, stage, steps, world, dev):
    """The same steps with the batch copied from PINNED host memory inside every step, not overlapped with compute,
    on every rank at once (N ranks share the host's PCIe root complexes and memory channels): scenes/s over all
    ranks, MAX over ranks."""
    def step():
        stage.copy_(host_batch, non_blocking=True)
        run_batch(stage)

    for _ in range(2):
        step()
    if flush is not None:
        flush()

    def region():
        for _ in range(steps):
            step()
        if flush is not None:
            flush()

    dt = _timed_region(region, world, dev)
    b = host_batch.shape[0]
    return dict(value=world * b * steps / dt, unit="scenes/s",
                note=f"{host_batch[0].numel() * 4 / 1e6:.1f} MB per scene over PCIe from pinned memory inside every "
                     f"step on each of the {world} rank(s), not overlapped with compute; MAX over ranks")


_LAST_LOOP = {}  # what the last _timed_loop saw (contract-block seconds, repeated blocks, ranks): main() adds it to the line


def h2d_overlapped(run_batch, flush, host_batches, steps, world, dev):
    """The same steps with every batch uploaded from PINNED host memory, double buffered: batch k + 1's copy travels on
    a copy stream while batch k is computed (paddle3d_amd.dist.H2DStage).  scenes/s over all ranks, MAX over ranks."""
    from paddle3d_amd import dist as pdist

    stage = pdist.H2DStage(host_batches[0].shape, host_batches[0].dtype, dev)
    nb = len(host_batches)

    def region(k_steps):
        stage.submit(host_batches[0])
        for k in range(k_steps):
            if k + 1 < k_steps:
                stage.submit(host_batches[(k + 1) % nb])
            x = stage.acquire()
            run_batch(x)
            stage.release()
        if flush is not None:
            flush()

    region(3)
    dt = _timed_region(lambda: region(steps), world, dev)
    b = host_batches[0].shape[0]
    return dict(value=world * b * steps / dt, unit="scenes/s",
                note=f"{host_batches[0][0].numel() * 4 / 1e6:.1f} MB per scene over PCIe from pinned memory, two "
                     f"alternating host batches, the copy of batch k + 1 on a copy stream beside the compute of batch k "
                     f"(dist.H2DStage) on each of the {world} rank(s); MAX over ranks")


def _dist_fields(line, args, world):
    """Fields every workload's line carries about the launch: ranks that took part, spread over repeated blocks."""
    info = _LAST_LOOP
    line["ranks_seen"] = info.get("ranks_seen", 1)
    line["collective_backend"] = (torch.distributed.get_backend() if world > 1 else None)
    if info.get("repeats_s"):
        vals = sorted(line["value"] * info["dt"] / t for t in info["repeats_s"])
        line.setdefault("extras", {})["repeat_blocks"] = dict(
            blocks=len(vals), steps_each=args.steps, unit=line["unit"], min=vals[0], median=vals[len(vals) // 2],
            max=vals[-1], note="the same K steps timed again after the contract block; `value` is the contract block")
    if info.get("node"):
        line.setdefault("extras", {})["node_state"] = info["node"]
    return line


def _traffic(batch, v):
    """HBM traffic per launch from the PMC passes (tools/gpu_traffic.sh -> profiles/*_traffic.json), when a profile
    of this exact configuration is committed; collected offline because rocprofv3 --pmc cannot wrap the timed run."""
    import glob

    found = {}
    try:
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
            t = json.load(open(path))
            if t.get("batch") == batch and t.get("max_voxels") == v and t.get("front", "pair") == "pair":
                found = t
    except Exception:  # noqa: BLE001
        found = {}
    return found


def _vox_floor():
    """The measured floor of hard_voxelize's MEMORY ACCESSES at C3 x 16 frames (tools/hwcheck/voxfloor: a program with
    no ranking logic that only streams the points, gathers 2.16 M kept 20-byte records into the fixed-shape output and
    performs the first-point stores / loads), from the newest profiles/r*_voxfloor.txt: what the access set costs on
    this machine warm / after a cache flush, next to what the operator achieves."""
    import glob
    import re

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_voxfloor.txt")))
    if not paths:
        return None
    try:
        text = open(paths[-1]).read().split("# tools/hwcheck/voxfloor --random")[0]
        warm = re.search(r"warm:.*?sum ([0-9.]+) us = ([0-9.]+) of", text)
        cold = re.search(r"cold:.*?sum ([0-9.]+) us = ([0-9.]+) of", text)
        return dict(warm_us=float(warm.group(1)), warm_frac=float(warm.group(2)), cold_us=float(cold.group(1)),
                    cold_frac=float(cold.group(2)), source="profiles/" + os.path.basename(paths[-1]),
                    note="tools/hwcheck/voxfloor: the operator's memory accesses alone (no ranking logic), 16 frames of "
                         "config 3, index lists with a 10-sweep frame's locality; warm = points in the last-level "
                         "cache, cold = after a 1 GiB flush (the in-step state lies between)")
    except Exception:  # noqa: BLE001
        return None


def bench_pillars(args, rank, world, dev):
    from paddle3d_amd import centerpoint as cpm
    from paddle3d_amd import dist as pdist

    V, B = args.max_voxels, args.batch
    model = cpm.centerpoint_pillars_nuscenes(max_num_voxels=(V, V)).to(dev).eval()
    model.voxelizer.path = args.vox_path
    amp = args.workload == "centerpoint_pillars_amp"
    model.set_amp(amp)
    pts = make_batch(B, 100 + B * rank, dev)
    cfg = model.test_cfg
    max_per_img = cfg["max_per_img"]
    names = ["start", "hard_voxelize", "pillar_feature_net", "pointpillars_scatter", "dense", "postprocess", "gather"]
    # the result hand-off: batch k's all-gather travels on RCCL's stream while batch k + 1 is computed (--gather sync:
    # the collective inside the step, on the compute stream's critical path)
    pipe = pdist.GatherPipeline() if args.gather == "overlap" else None
    # --front fused (default): voxelizer -> PFN through the index of the points, as CenterPoint.test_forward runs it;
    # --front pair: pd3_hard_voxelize (the full operator, padded tensor written) + pd3_pillar_feature_net
    fused_front = args.front == "fused" and args.vox_path == 0 and getattr(model, "fuse_rows", False)
    if fused_front:
        with torch.no_grad():
            probe = model.voxelizer.index(pts[:1])
            fused_front = probe is not None and model.voxel_encoder.forward_indexed(
                pts[:1], probe[0], probe[1], probe[2].view(-1, 4)) is not None

    def hand_off(rec, cnt):
        if pipe is None:
            return pdist.gather_detections(rec, cnt)
        prev = pipe.submit(rec, cnt)
        return prev if prev is not None else (rec, cnt)

    def finish(out):
        return pipe.flush() if pipe is not None else out

    def compute(points, events):
        """One step up to the operator's own record: (rec [B, max_per_img, 11], cnt [B]) of THIS batch."""
        def mark(i):
            if events is not None:
                events[i].record()

        mark(0)
        feats = None
        if fused_front:
            # the model path: the voxelizer leaves an INDEX of the points (no padded [V, P, D] tensor), the PFN reads
            # the points through it (pd3_hard_voxelize_index + pd3_pillar_feature_net_indexed)
            idx = model.voxelizer.index(points)
            if idx is not None:
                span, plist, coors, npv, nv = idx
                mark(1)
                b, v = int(coors.shape[0]), int(coors.shape[1])
                feats = model.voxel_encoder.forward_indexed(points, span, plist, coors.view(b * v, 4))
        if feats is None:
            voxels, coors, npv, nv = model.voxelizer(points)
            mark(1)
            b, v, p, d = voxels.shape
            feats = model.voxel_encoder(voxels.view(b * v, p, d), npv.view(b * v), coors.view(b * v, 4))
        mark(2)
        canvas = model.scatter(feats, coors.view(b * v, 4), b)
        mark(3)
        x = model.dense_forward(canvas)
        preds, _ = model.bbox_head(x)
        mark(4)
        _bx, _sc, _lb, cnt, rec = model.bbox_head.predict_by_custom_op(preds, cfg, device_only=True,
                                                                      records=max_per_img)
        mark(5)
        return rec, cnt

    def run(points, events):
        rec, cnt = compute(points, events)
        all_rec, all_cnt = hand_off(rec, cnt)  # the record comes out of the operator itself
        if events is not None:
            events[6].record()
        return all_rec, all_cnt

    # --graph: the step as five HIP graphs (one per op, so that the per-op HIP events stay between them): ~60 kernel
    # launches and their Python / allocator work become five graph launches.  Same kernels, same order, same buffers
    # every replay; the collective stays outside.  Measured: no difference on this path (the host needs 0.8-1.0 ms to
    # enqueue a 10 ms step, the GPU never waits for it), so the default stays the eager step.
    launch = "eager"
    step = lambda ev: run(pts, ev)  # noqa: E731
    cpu_ms = None
    with torch.no_grad():
        for _ in range(2):  # packs weights, sizes workspaces: nothing of that may happen inside a capture
            run(pts, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(pts, None)
        cpu_ms = (time.perf_counter() - t0) * 1e3  # host time to enqueue one eager step (no sync)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        gpu_ms_est = cpu_ms + (time.perf_counter() - t1) * 1e3  # enqueue + drain of that one step
        # --graph forces replay; otherwise it is turned on only where the host would hold the GPU up (dist.choose_launch:
        # enqueue time above half of the step, measured under the node's real contention)
        want_graph = pdist.choose_launch(cpu_ms, gpu_ms_est, "graph" if args.graph else "auto") == "graph"
        if want_graph:
            fused_front = False  # (the captured segments are the pair form's)
            try:
                st = {}

                def seg_vox():
                    st["vox"] = model.voxelizer(pts)

                def seg_pfn():
                    voxels, coors, npv, _nv = st["vox"]
                    b, v, p, d = voxels.shape
                    st["b"], st["c4"] = b, coors.view(b * v, 4)
                    st["feats"] = model.voxel_encoder(voxels.view(b * v, p, d), npv.view(b * v), st["c4"])

                def seg_scatter():
                    st["canvas"] = model.scatter(st["feats"], st["c4"], st["b"])

                def seg_dense():
                    st["preds"] = model.bbox_head(model.dense_forward(st["canvas"]))[0]

                def seg_post():
                    st["post"] = model.bbox_head.predict_by_custom_op(st["preds"], cfg, device_only=True,
                                                                      records=max_per_img)

                segs = [seg_vox, seg_pfn, seg_scatter, seg_dense, seg_post]
                pool = torch.cuda.graph_pool_handle()
                graphs = []
                for f in segs:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pool):
                        f()
                    graphs.append(g)
                torch.cuda.synchronize()

                def run_graphs(events):
                    if events is not None:
                        events[0].record()
                    for i, g in enumerate(graphs):
                        g.replay()
                        if events is not None:
                            events[i + 1].record()
                    _bx, _sc, _lb, cnt, rec = st["post"]
                    res = hand_off(rec, cnt)
                    if events is not None:
                        events[6].record()
                    return res

                # the guard compares THIS batch's records of the two launch paths (not what hand_off returns: with
                # --gather overlap that is the previous batch's result, which would compare eager with eager)
                ref = [t.clone() for t in compute(pts, None)]
                for g in graphs:
                    g.replay()
                torch.cuda.synchronize()
                got = (st["post"][4], st["post"][3])
                if not (torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])):
                    raise RuntimeError("graph replay and eager step disagree")
                step, launch = run_graphs, f"hip graphs ({len(graphs)} per step, one per op) + eager result hand-off"
            except Exception as e:  # noqa: BLE001  (capture is an optimisation of the launch path, never a requirement)
                torch.cuda.synchronize()
                print(f"bench: HIP-graph capture failed ({type(e).__name__}: {e}); running the eager step", file=sys.stderr)
                step, launch = (lambda ev: run(pts, ev)), "eager (graph capture failed)"
    # The north star's roofline is that of the OPERATOR pd3_hard_voxelize (the padded [V, P, D] tensor written).  With
    # the fused front the contract block's step no longer contains it, so it is timed in a block of its own, in the
    # contract's own shape (the same _timed_loop: barrier, W warm-up steps, K timed steps of the PAIR form of the graph
    # with the result hand-off, HIP events around the operator, the points evicted by the rest of the step) in front of
    # the contract block.  `value` / `ms_per_step` are the contract block's (fused front).
    pair_ms = None
    if fused_front:
        import copy as _copy

        a2 = _copy.copy(args)
        a2.repeats = 0
        fused_front = False  # (compute() reads the flag when it runs)
        _dt2, pair_ms, _out2, _info2 = _timed_loop(step, a2, world, dev, names, finish=finish)
        fused_front = True
    dt, per_op_ms, out, info = _timed_loop(step, args, world, dev, names, finish=finish)
    op_ms = per_op_ms if pair_ms is None else pair_ms
    multi = {}
    if args.strong_frames > 0 and not args.no_extras:
        # every rank takes part (collectives inside); the line is rank 0's
        loop = dict(_LAST_LOOP)
        with torch.no_grad():
            from paddle3d_amd import synth

            def shard(ids):
                return torch.from_numpy(np.stack([synth.nuscenes_sweep(1000 + i) for i in ids])).to(dev)

            flush = (lambda: pipe.flush()) if pipe is not None else None
            multi["strong_scaling"] = strong_scaling(lambda b: run(b, None), flush, shard, args.strong_frames, B, rank,
                                                     world, dev)
            if world > 1:
                host = make_batch(B, 100 + B * rank, pin=True)
                multi["h2d_inclusive"] = h2d_inclusive(lambda b: run(b, None), flush, host, torch.empty_like(pts),
                                                       args.steps, world, dev)
                multi["h2d_overlapped"] = h2d_overlapped(lambda b: run(b, None), flush,
                                                         [host, make_batch(B, 900 + B * rank, pin=True)], args.steps,
                                                         world, dev)
        _LAST_LOOP.clear()
        _LAST_LOOP.update(loop)
    if rank != 0:
        return None
    alg = algorithmic_bytes(V)
    traffic = _traffic(B, V)

    def hbm(name, key, src=None):
        src = per_op_ms if src is None else src
        a = alg[key] * B / (src[name] * 1e-3) / 1e9
        tr = traffic.get(key, {}).get("bytes_per_launch")
        return dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBPS, unit="GB/s", frac=a / HBM_PEAK_GBPS, traffic=tr,
                    ms_per_launch=src[name], units_per_launch=B, algorithmic_bytes_per_unit=alg[key])

    def mfma(ms, direct, executed, note):
        ex = executed * B / (ms * 1e-3) / 1e12
        return dict(bound="mfma", achieved=ex, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=ex / MFMA_F32_PEAK_TFLOPS, traffic=None, ms_per_launch=ms, units_per_launch=B,
                    executed_flops_per_unit=executed, direct_form_flops_per_unit=direct,
                    direct_form_tflops=direct * B / (ms * 1e-3) / 1e12, note=note)

    d_direct, d_exec = dense_flops()
    with torch.no_grad():
        pfn_mfma = pfn_packed_mfma(model.voxelizer(pts)[2], P) / B
    p_direct, p_exec = pfn_flops(V, pfn_mfma)
    rooflines = dict(
        hard_voxelize=dict(hbm("hard_voxelize", "hard_voxelize", op_ms),
                           measured_in=("W + K steps of the pair form of the graph (pd3_hard_voxelize + pd3_pillar_feature_net), "
                                        "run in front of the contract block: the contract block's step holds "
                                        "pd3_hard_voxelize_index instead, see front_half") if fused_front
                           else "the contract block's steps",
                           input_state=("every step reads the same 16-frame batch (96 MB: it would fit the 256 MB Infinity "
                                        "Cache), but the 8 ms of convolutions between two voxelizer runs evict it -- the "
                                        "route kernel takes ~30 us inside the step against ~24 us when the operator is "
                                        "looped alone (profiles/r05_vox_paths.txt), so the in-step figure is the cold-input "
                                        "one; roofline.floor gives both states of the bare memory accesses")),
        pointpillars_scatter=(dict(bound="hbm", fused_into="dense_backbone_fpn_head", achieved=None,
                                   peak=HBM_PEAK_GBPS, unit="GB/s", frac=None,
                                   traffic=traffic.get("pointpillars_scatter", {}).get("bytes_per_launch"),
                                   ms_per_launch=per_op_ms["pointpillars_scatter"], units_per_launch=B,
                                   note="PointPillarsScatter is fused into the first backbone convolution (round 3): "
                                        "this interval holds the inverse-map kernels only, the canvas is never "
                                        "written; `pd3_pointpillars_scatter` alone runs at 0.49-0.51 of the HBM "
                                        "roofline (DESIGN 4.2)")
                              if getattr(model, "fuse_scatter", False)
                              else hbm("pointpillars_scatter", "pointpillars_scatter")),
        centerpoint_postprocess=dict(hbm("postprocess", "centerpoint_postprocess"),
                                     us_per_frame=per_op_ms["postprocess"] * 1e3 / B,
                                     note="latency bound (SURVEY 8(d)): us_per_frame is the figure, the HBM "
                                          "fraction is for completeness (6 tasks x up to 1000 candidates per "
                                          "frame: random-init heads fill the NMS cap)"),
        pillar_feature_net=mfma(per_op_ms["pillar_feature_net"], p_direct, p_exec,
                                "achieved / frac = executed MFMA flops: the packed form issues 38 "
                                "v_mfma_f32_16x16x4_f32 per 16-row block of stored points (packed per 8 pillars) + 32 "
                                f"per chunk, {pfn_mfma:.0f} per scene counted on this batch; the kernel is bound by "
                                "VALU / LDS instruction issue next to the MFMAs, not by the matrix pipe (DESIGN 4.3); "
                                "direct_form_tflops = the layer's own multiply-adds over all P slots / time"),
        dense_backbone_fpn_head=mfma(per_op_ms["dense"], d_direct, d_exec,
                                     "achieved / frac = executed flops: the 52 stride-1 3x3 layers run Winograd "
                                     "F(4x4,3x3) (a quarter of the direct multiplies), the 3 stride-2 layers and the "
                                     "FPN levels run direct GEMMs, all fp32; direct_form_tflops = 127.2 GFLOP/scene "
                                     "/ time (may exceed the peak: fewer multiplies are issued than counted)"))
    line = {
        "metric": ("scenes/sec CenterPoint-Pillars nuScenes 300k-pt sweeps" +
                   (" (AMP O2: fp16 matrix cores in the stride-1 convolutions)" if amp else "")),
        "value": world * B * args.steps / dt, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16 x f16 -> f32 (dense 3x3 stride 1), f32 elsewhere" if amp else "f32",
        "data": "synthetic",
        "config": {"workload": "CenterPoint-Pillars nuScenes 10-sweep: 300000 pts x 5 per scene, 0.2 m pillars "
                               f"(512x512), P=20, max_voxels={V}, batch {B} distinct scenes/GPU/step, random-init "
                               "weights, full graph voxelize->PFN->scatter->SECOND+FPN->CenterHead->postprocess"
                               + ("->RCCL all-gather" if world > 1 else ""),
                   "frames_per_gpu_per_step": B, "max_voxels": V, "parallelism": f"dp{world} (frames)",
                   "launch": launch, "host_ms_to_enqueue_one_eager_step": cpu_ms,
                   "launch_policy": "graph replay is turned on when enqueueing a step takes the host more than half of "
                                    "the step's GPU time (dist.choose_launch); --graph forces it",
                   "result_hand_off": ("all-gather of batch k overlapped with batch k + 1 (dist.GatherPipeline)"
                                       if pipe is not None else "all-gather inside the step")},
        "roofline": dict(rooflines["hard_voxelize"],
                         kernel="hard_voxelize launch sequence (vw_route + vw_group + vw_assign + vw_rows: the wave form, "
                                "voxelize_wave.hpp; --vox-path picks another form)",
                         target=0.5, floor=_vox_floor()),
        "rooflines": rooflines,
        "front_half": dict(
            form=("fused: pd3_hard_voxelize_index (no padded [V, P, D] tensor) + pd3_pillar_feature_net_indexed"
                  if fused_front else "pair: pd3_hard_voxelize + pd3_pillar_feature_net"),
            ms_in_step=per_op_ms["hard_voxelize"] + per_op_ms["pillar_feature_net"],
            pair_ms=dict(hard_voxelize=op_ms["hard_voxelize"], pillar_feature_net=op_ms["pillar_feature_net"]),
            note="per_op_ms.hard_voxelize / .pillar_feature_net are the intervals of the form the step runs; pair_ms are "
                 "the two full operators inside K steps of the pair form"),
        # `roofline` is the kernel the north star puts the HBM target on; by time the step is dominated by
        # the dense graph (rooflines["dense_backbone_fpn_head"], MFMA bound)
        "dominant_by_time": "dense_backbone_fpn_head",
        "per_op_ms": per_op_ms,
        "detections_first_frame": int(out[1][0].item()),
    }
    if amp:
        # what the mixed-precision graph costs in accuracy on this batch: head maps against the fp32 graph's, and the
        # detections of the two graphs scored against each other on the mAP scale (the fp32 graph as the annotations)
        from paddle3d_amd import nuscenes_bridge as nb

        import copy

        from paddle3d_amd import synth

        with torch.no_grad():
            # a copy with heads like a trained net's (synth.trained_like_heads: every class of every task fires), so
            # that the detections compared are not thousands of near-ties of one score band
            m2 = copy.deepcopy(model)
            m2.set_amp(False)
            synth.trained_like_heads(m2, pts[:2])

            def maps_and_dets(flag):
                m2.set_amp(flag)
                canvas = m2.extract_pillars(pts, dense=False)
                preds, _ = m2.bbox_head(m2.dense_forward(canvas))
                dets = m2.bbox_head.predict_by_custom_op(preds, cfg)
                return preds, [{k: d[k].cpu().numpy() for k in ("box3d_lidar", "scores", "label_preds")} for d in dets]

            p16, d16 = maps_and_dets(True)
            p32, d32 = maps_and_dets(False)
            err = max(float((a[k].float() - b[k].float()).abs().max()) for a, b in zip(p16, p32) for k in a)
            mag = max(float(b[k].float().abs().max()) for b in p32 for k in b)
            del m2
        res = nb.nuscenes_style_map(d16, d32)
        line["amp_error"] = dict(head_maps_max_abs=err, head_maps_max_magnitude=mag,
                                 map_proxy_vs_fp32=res["mAP"], classes_scored=res["classes_scored"],
                                 per_class_ap_vs_fp32={str(c): round(v, 5) for c, v in res["per_class"].items()},
                                 fp32_boxes_without_amp_twin=nb.unmatched_detections(d16, d32, score_tol=2e-2),
                                 amp_boxes_without_fp32_twin=nb.unmatched_detections(d32, d16, score_tol=2e-2),
                                 frames=int(pts.shape[0]),
                                 note="fp16 activations and weights, fp32 accumulation; random-init weights with "
                                      "heads calibrated like a trained net's (synth.trained_like_heads); the AP is "
                                      "quantised (one box of a class without a twin = one of 90 recall bins = 0.011 of "
                                      "that class's AP, whatever the number of boxes): the *_without_*_twin counts "
                                      "(same frame and class, centre within 0.5 m, score within 0.02) say how many "
                                      "boxes that is; "
                                      "tests/test_model_gpu.py::test_amp_graph_close_to_fp32 runs 64 frames")
        for k in ("dense_backbone_fpn_head",):
            rooflines[k]["note"] = ("AMP: the stride-1 3x3 layers run direct-form on the fp16 matrix cores (peak 2.5 "
                                    "PFLOP/s); achieved / frac here are still priced against the fp32 peak with the "
                                    "fp32 graph's executed-flop count and are not a utilisation figure for this mode")
    if multi:
        line["extras"] = dict(multi)
    if world == 1 and not args.no_extras:
        extras = line.setdefault("extras", {})
        with torch.no_grad():
            # (a) the same steps with the batch copied from pinned host memory inside every step (not overlapped)
            host = make_batch(B, 100, pin=True)
            stage = torch.empty_like(pts)

            def h2d_step():
                stage.copy_(host, non_blocking=True)
                return run(stage, None)

            for _ in range(2):
                h2d_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                h2d_step()
            torch.cuda.synchronize()
            dth = time.perf_counter() - t0
            extras["h2d_inclusive"] = dict(value=B * args.steps / dth, unit="scenes/s",
                                           note=f"{host.numel() * 4 / B / 1e6:.1f} MB per scene over PCIe from pinned "
                                                "memory inside every step, not overlapped with compute")
            ov = h2d_overlapped(lambda b: run(b, None), (lambda: pipe.flush()) if pipe is not None else None,
                                [host, make_batch(B, 900, pin=True)], args.steps, 1, dev)
            ov["fraction_of_resident"] = ov["value"] / (world * B * args.steps / dt)
            extras["h2d_overlapped"] = ov
            # (b) per-frame latency: batch 1, one frame in flight
            one = pts[:1].contiguous()
            for _ in range(3):
                run(one, None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                run(one, None)
            torch.cuda.synchronize()
            extras["latency_batch1_ms"] = (time.perf_counter() - t0) / 20 * 1e3
            # (c) two batches in flight: batch k + 1's front half (voxelize -> PFN -> scatter: instruction / latency
            # bound) on a second stream beside batch k's dense graph + postprocess (matrix-core bound).  Same kernels,
            # same work per batch; a different schedule, so it is reported under its own name, never as `value`.
            try:
                main_s, side_s = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)

                def front():
                    voxels, coors, npv, _nv = model.voxelizer(pts)
                    b, v, p, d = voxels.shape
                    feats = model.voxel_encoder(voxels.view(b * v, p, d), npv.view(b * v), coors.view(b * v, 4))
                    return model.scatter(feats, coors.view(b * v, 4), b)

                def back(canvas):
                    preds, _ = model.bbox_head(model.dense_forward(canvas))
                    out = model.bbox_head.predict_by_custom_op(preds, cfg, device_only=True, records=max_per_img)
                    return pdist.gather_detections(out[4], out[3])

                def pipelined(steps):
                    done = [None] * (steps + 1)
                    canvas = None
                    for k in range(steps + 1):
                        nxt = None
                        if k < steps:
                            if k >= 2 and done[k - 2] is not None:
                                side_s.wait_event(done[k - 2])  # at most two batches in flight
                            with torch.cuda.stream(side_s):
                                nxt = front()
                                ready = torch.cuda.Event()
                                ready.record(side_s)
                        if canvas is not None:
                            main_s.wait_event(canvas[1])
                            cv = canvas[0]
                            for t in ([cv.features, cv.coords, cv.inv] if hasattr(cv, "inv") else [cv]):
                                t.record_stream(main_s)
                            res = back(canvas[0])
                            done[k - 1] = torch.cuda.Event()
                            done[k - 1].record(main_s)
                        canvas = (nxt, ready) if nxt is not None else None
                    return res

                side_s.wait_stream(main_s)
                pipelined(3)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = pipelined(args.steps)
                torch.cuda.synchronize()
                dtp = time.perf_counter() - t0
                same = bool(torch.equal(res[0], out[0]) and torch.equal(res[1], out[1]))
                extras["pipelined_two_streams"] = dict(
                    value=B * args.steps / dtp, unit="scenes/s", identical_results=same,
                    note="batch k+1's voxelize + PFN + scatter on a second HIP stream beside batch k's dense graph + "
                         "postprocess; same kernels and work per batch, two batches in flight; not the headline")
            except Exception as e:  # noqa: BLE001 -- an extra
                torch.cuda.synchronize()
                extras["pipelined_two_streams"] = dict(value=None, note=f"failed: {type(e).__name__}: {e}")
        extras["measured_ceilings"] = measured_ceilings(dev)
    if world == 1:
        if not args.no_cpu_baseline:
            try:
                model_cpu = cpm.centerpoint_pillars_nuscenes(max_num_voxels=(V, V)).eval()
                model_cpu.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
                line["cpu_baseline"] = cpu_baseline(model_cpu, V)
            except Exception as e:  # the baseline is reported, never required
                line["cpu_baseline"] = dict(value=None, unit="scenes/s", cores=0, kind="port", sample=f"failed: {e}")
                model_cpu = None
            if model_cpu is not None and args.map_frames > 0 and "extras" in line:
                try:
                    line["extras"]["map_proxy"] = map_proxy(model, model_cpu, V, args.map_frames, dev)
                except Exception as e:  # noqa: BLE001 -- an extra
                    line["extras"]["map_proxy"] = dict(value=None, note=f"failed: {type(e).__name__}: {e}")
    if world == 1 and not args.no_extras:
        del model
        torch.cuda.empty_cache()
        loop = dict(_LAST_LOOP)  # the headline's launch facts, not the last extra workload's
        line["extras"]["other_workloads"] = other_workloads(args, rank, world, dev)
        _LAST_LOOP.clear()
        _LAST_LOOP.update(loop)
    return line


def c4_cpu_baseline():
    """CenterPoint-Voxel on the host cores, bounded: the dense conv3d statement of the sparse encoder (the only CPU
    statement there is: Paddle's sparse kernels are not vendored) fits a CPU only on a cropped grid, so this is ONE
    frame of a quarter-range copy of config 4 (41 x 256 x 256 cells, 120 k points): reference voxelizer (oracle/_ref
    when present) -> voxel mean -> dense conv3d stack -> torch dense graph -> C postprocess."""
    from oracle import pyoracle as O
    from paddle3d_amd import centerpoint as cpm
    from paddle3d_amd import synth

    pcr = [-9.6, -9.6, -5.0, 9.6, 9.6, 3.0]
    torch.manual_seed(8)
    cpu = cpm.centerpoint_voxels_nuscenes(max_num_voxels=(40000, 40000), point_cloud_range=pcr).eval()
    kind = "ref" if O.have_ref() else "port"
    cfg = cpu.test_cfg
    pts = synth.nuscenes_sweep(93, n_points=120_000)
    t0 = time.perf_counter()
    vox, co, npv, nv = O.hard_voxelize(pts, synth.NUSC_VOXEL, pcr, 10, 40000, kind)
    mean = O.voxel_mean(vox[:nv], npv[:nv])
    c4 = np.concatenate([np.zeros((nv, 1), np.int32), co[:nv]], 1)
    bev = O.sparse_encoder_dense_torch(cpu.middle_encoder, mean, c4, 1)
    with torch.no_grad():
        preds, _ = O.center_head_torch(cpu.bbox_head, O.dense_forward_torch(cpu, bev))
    tasks = [{k: v.numpy() for k, v in p.items()} for p in preds]
    O.centerpoint_postprocess(tasks, cfg["voxel_size"] + [8.0], cfg["point_cloud_range"] + [0.0] * 4,
                              cfg["post_center_limit_range"], [0, 1, 3, 5, 6, 8], cfg["down_ratio"],
                              cfg["score_threshold"], cfg["nms"]["nms_iou_threshold"], cfg["nms"]["nms_pre_max_size"],
                              cfg["nms"]["nms_post_max_size"], True)
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit="cropped scenes/s", cores=torch.get_num_threads(),
                kind="reference" if kind == "ref" else "port",
                sample="1 frame of a QUARTER-RANGE crop of config 4 (0.075 m voxels on +-9.6 m: 41 x 256 x 256 cells, "
                       f"120000 points, {int(nv)} voxels): reference voxelizer, sparse encoder as dense torch conv3d "
                       "(the full 41 x 1440 x 1440 grid has no dense CPU statement that finishes), torch dense graph, "
                       "C postprocess; 1/32 of the full grid's cells, so not comparable with `value` one to one")


def c1_cpu_baseline(frames=2):
    """PointPillars-KITTI (BASELINE config 1, "on the Paddle CPU reference path") on the host cores: reference
    voxelizer (oracle/_ref when present), torch-CPU PFN / SECOND / FPN / head, NumPy anchor mask + decode + NMS (the
    oracle's statement of SSDHead.post_process), `frames` frames of the same synthetic KITTI clouds."""
    from oracle import pyoracle as O
    from paddle3d_amd import pointpillars as ppm
    from paddle3d_amd import synth

    torch.manual_seed(4)
    cpu = ppm.pointpillars_kitti_car().eval()
    with torch.no_grad():
        cpu.head.cls_head.bias.fill_(-2.0)
    kind = "ref" if O.have_ref() else "port"
    gen, h = cpu.anchor_generator, cpu.head
    an, bv = gen.anchors.numpy(), gen.anchors_bv.numpy().astype(np.int64)
    vs, pcr = cpu.voxelizer.voxel_size, cpu.voxelizer.point_cloud_range
    p_max, v_max = cpu.voxelizer.max_num_points_in_voxel, cpu.voxelizer.max_num_voxels[1]
    nx, ny = gen.grid_size
    apl, ncls = h.num_anchor_per_loc, h.num_classes
    c_cls, c_box = apl * ncls, apl * 7
    params = [dict(weight=l.linear.weight.t().detach().numpy(), gamma=l.norm.weight.detach().numpy(),
                   beta=l.norm.bias.detach().numpy(), mean=l.norm.running_mean.numpy(), var=l.norm.running_var.numpy())
              for l in cpu.pillar_encoder.pfn_layers]
    t0 = time.perf_counter()
    for i in range(frames):
        pts = synth.kitti_frame(100 + i, 16384)
        vox, co, npv, nv = O.hard_voxelize(pts, vs, pcr, p_max, v_max, kind)
        c4 = np.concatenate([np.zeros((nv, 1), np.int32), co[:nv]], 1)
        feats = O.pfn_forward_torch(vox[:nv], npv[:nv], c4, params, vs, pcr)
        bev = torch.from_numpy(O.pillar_scatter(feats, c4, 1, ny, nx))
        with torch.no_grad():
            x = O.second_fpn_torch(cpu.neck, O.second_backbone_torch(cpu.backbone, bev))
            m = torch.cat([h.cls_head(x), h.box_head(x), h.dir_head(x)], 1)[0].numpy()
        pr = m.reshape(m.shape[0], -1).T
        mask = O.ssd_anchor_mask_numpy(co[:nv], bv, gen.grid_size, 1.0)
        O.ssd_post_process_frame_numpy(pr[:, c_cls:c_cls + c_box].reshape(-1, 7), pr[:, :c_cls].reshape(-1, ncls),
                                       pr[:, c_cls + c_box:].reshape(-1, 2), an, mask, h.nms_score_threshold,
                                       h.pred_center_limit_range, h.nms_pre_max_size, h.nms_post_max_size,
                                       h.nms_iou_threshold)
    dt = time.perf_counter() - t0
    return dict(value=frames / dt, unit="frames/s", cores=torch.get_num_threads(),
                kind="reference" if kind == "ref" else "port",
                sample=f"{frames} frames of the same workload: hard_voxelize = "
                       f"{'reference voxelize_op.cc:19-82 compiled from /root/reference' if kind == 'ref' else 'C port'} "
                       "(1 thread), PFN / SECOND / FPN / head = torch CPU fp32, anchor mask / decode / NMS = NumPy + C port")


def other_workloads(args, rank, world, dev):
    """Short runs of BASELINE.json's other single-GPU configurations inside the default invocation, so that one
    driver-run line carries every config that fits one GPU (value, ms per step, roofline fraction each)."""
    import copy

    out = {}
    todo = [("centerpoint_pillars_amp", bench_pillars, 16),
            ("pointpillars_kitti", bench_pointpillars_kitti, 16), ("centerpoint_voxel", bench_voxel, 8),
            ("centerpoint_voxel_amp", bench_voxel, 8),
            ("bevfusion_lidar", bench_bevfusion_lidar, 16), ("bev_pool_v2", bench_bev_pool, 1)]
    for name, fn, batch in todo:
        a = copy.copy(args)
        a.batch, a.steps, a.warmup, a.repeats = batch, 5, 2, 0
        if fn is bench_voxel:  # (a 35-45 ms step whose first few runs still grow the allocator's pools)
            a.steps, a.warmup = 10, 4
        a.no_extras = a.no_cpu_baseline = True
        a.workload = name
        try:
            with torch.no_grad():
                line = fn(a, rank, world, dev)
            rf = line["roofline"]
            out[name] = dict(metric=line["metric"], value=line["value"], unit=line["unit"], steps=a.steps,
                             warmup=a.warmup, ms_per_step=line["ms_per_step"], workload=line["config"]["workload"],
                             roofline=dict(kernel=rf.get("kernel"), bound=rf["bound"], frac=rf["frac"],
                                           achieved=rf["achieved"], unit=rf["unit"]),
                             rooflines={k: dict(bound=v["bound"], frac=v.get("frac")) for k, v in
                                        line.get("rooflines", {}).items()},
                             per_op_ms=line["per_op_ms"])
            for extra in ("amp_error", "dtype"):
                if extra in line:
                    out[name][extra] = line[extra]
        except Exception as e:  # noqa: BLE001 -- reported extras, never required for the headline
            out[name] = dict(error=f"{type(e).__name__}: {e}")
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    for name, leg in (("centerpoint_voxel", c4_cpu_baseline), ("pointpillars_kitti", c1_cpu_baseline)):
        if not args.no_cpu_baseline and name in out and "error" not in out[name]:
            try:
                out[name]["cpu_baseline"] = leg()
            except Exception as e:  # noqa: BLE001
                out[name]["cpu_baseline"] = dict(value=None, sample=f"failed: {type(e).__name__}: {e}")
    return out


def bench_voxel(args, rank, world, dev):
    """CenterPoint-Voxel (config 4): 0.075 m voxels, sort-path hard_voxelize, VoxelMean, SparseResNet3D, dense
    graph at 180 x 180, postprocess."""
    from paddle3d_amd import centerpoint as cpm
    from paddle3d_amd import dist as pdist

    B = args.batch
    V = 160000  # the reference's test-time cap (max_num_voxels: [120000, 160000])
    model = cpm.centerpoint_voxels_nuscenes(max_num_voxels=(120000, V)).to(dev).eval()
    amp = args.workload == "centerpoint_voxel_amp"
    model.set_amp(amp)  # the sparse encoder from 16 -> 32 on (the 180-wide dense maps are not the fp16 kernel's shape)
    pts = make_batch(B, 100 + B * rank, dev)
    cfg = model.test_cfg
    names = ["start", "hard_voxelize", "voxel_mean_sparse_encoder", "dense", "postprocess", "gather"]
    stats = {}

    def run(events):
        def mark(i):
            if events is not None:
                events[i].record()

        mark(0)
        voxels, coors, npv, nv = model.voxelizer(pts)
        mark(1)
        b, v, p, d = voxels.shape
        voxels, coors, npv = voxels.view(b * v, p, d), coors.view(b * v, 4), npv.view(b * v)
        feats = model.voxel_encoder(voxels, npv, coors)  # padding rows included: the encoder skips them
        x = model.middle_encoder(feats, coors, b)
        mark(2)
        x = model.dense_forward(x)
        preds, _ = model.bbox_head(x)
        mark(3)
        _bx, _sc, _lb, cnt, rec = model.bbox_head.predict_by_custom_op(preds, cfg, device_only=True,
                                                                      records=cfg["max_per_img"])
        mark(4)
        all_rec, all_cnt = pdist.gather_detections(rec, cnt)
        mark(5)
        return all_rec, all_cnt

    dt, per_op_ms, out, info = _timed_loop(run, args, world, dev, names)
    # the encoder planned every timed step from remembered capacities (no host round trip inside the step); did a set
    # outgrow its capacity?  (one read-back, after the timed region)
    overflow = bool(model.middle_encoder.take_overflow())
    if rank != 0:
        return None
    alg = 4 * N_POINTS * DIMS + 4 * V * 10 * DIMS + 16 * V + 4
    a = alg * B / (per_op_ms["hard_voxelize"] * 1e-3) / 1e9
    with torch.no_grad():  # untimed: how many multiply-adds the encoder's rulebooks hold for this batch
        from paddle3d_amd import sparse as _sparse

        voxels, coors, npv, nv = model.voxelizer(pts)
        b, v, p, d = voxels.shape
        keep = coors.view(b * v, 4)[:, 0] >= 0
        cs = coors.view(b * v, 4)[keep].contiguous()
        stats["active_voxels"] = int(cs.shape[0])
        sp = _sparse.count_flops(model.middle_encoder, model.voxel_encoder(voxels.view(b * v, p, d)[keep],
                                                                          npv.view(b * v)[keep], cs), cs, b)
    line = {
        "metric": "scenes/sec CenterPoint-Voxel nuScenes 300k-pt sweeps" + (
            " (AMP O2: the sparse encoder's convolutions on the fp16 matrix cores)" if amp else ""),
        "value": world * B * args.steps / dt, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16 x f16 -> f32 (sparse convolutions from 16 -> 32 channels on), f32 elsewhere" if amp else "f32",
        "data": "synthetic",
        "config": {"workload": "CenterPoint-Voxel nuScenes 10-sweep: 300000 pts x 5 per scene, 0.075 m voxels "
                               f"(1440x1440x40), P=10, max_voxels={V}, batch {B} distinct scenes/GPU/step, random-init "
                               "weights, voxelize->VoxelMean->SparseResNet3D->SECOND+FPN->CenterHead->postprocess",
                   "frames_per_gpu_per_step": B, "max_voxels": V, "parallelism": f"dp{world} (frames)"},
        "roofline": dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBPS, unit="GB/s", frac=a / HBM_PEAK_GBPS, traffic=None,
                         ms_per_launch=per_op_ms["hard_voxelize"], units_per_launch=B, algorithmic_bytes_per_unit=alg,
                         kernel="hard_voxelize launch sequence, 3-D wave form (voxelize_wave3d.hpp: route + group with an "
                                "LDS hash table per wave + assign + rows) on the 82.9 M-cell grid"),
        "per_op_ms": per_op_ms, "active_voxels_per_batch": stats.get("active_voxels"),
        "detections_first_frame": int(out[1][0].item()),
        "sparse_plan": dict(host_syncs_per_step=0, capacity_overflow=overflow,
                            note="index sets planned from remembered capacities (first forward of the shape: one "
                                 "sync); an overflow would make the timed steps invalid"),
    }
    if overflow:
        line["error"] = "sparse plan: an index set outgrew its remembered capacity during the timed steps"
    if not amp:
        from paddle3d_amd.ops import sparse_conv3d as _sp3

        with torch.no_grad():  # (untimed) the encoder's map by the fp32 matrix-core kernel in every layer, for comparison
            bev_x3 = model.extract_pillars(pts)
            _sp3.SPLIT_BF16 = False
            try:
                bev_32 = model.extract_pillars(pts)
            finally:
                _sp3.SPLIT_BF16 = True
        line["sparse_arithmetic"] = dict(
            form="fp32; the layers with >= 64 output channels multiply on the bf16 matrix cores with every fp32 operand "
                 "cut into three bf16 pieces (hi + mid + lo = the value exactly) and six of the nine piece products "
                 "accumulated in fp32 (csrc/sparse_conv_x3.hip): the error against exact arithmetic is that of the fp32 "
                 "matrix-core kernel (tests/test_sparse_conv_gpu.py::test_features_bf16x3_is_fp32_arithmetic)",
            encoder_map_max_abs_diff_vs_fp32_kernel=float((bev_x3 - bev_32).abs().max()),
            encoder_map_max_abs=float(bev_32.abs().max()))
    if amp:
        from paddle3d_amd import nuscenes_bridge as nb

        with torch.no_grad():  # what the mode costs in accuracy on this batch (untimed)
            bev16 = model.extract_pillars(pts)
            d16 = model.test_forward(pts)
            model.set_amp(False)
            bev32 = model.extract_pillars(pts)
            d32 = model.test_forward(pts)
            model.set_amp(True)
        line["amp_error"] = dict(
            bev_map_max_abs=float((bev16 - bev32).abs().max()), bev_map_max_magnitude=float(bev32.abs().max()),
            fp32_boxes_without_amp_twin=nb.unmatched_detections(d16, d32, score_tol=2e-2),
            amp_boxes_without_fp32_twin=nb.unmatched_detections(d32, d16, score_tol=2e-2),
            note="the encoder's [B, 256, 180, 180] map and the detections of the AMP graph against the fp32 graph's; fp16 "
                 "feature rows and weights, fp32 accumulation, random-init weights")
    if sp:
        ms = per_op_ms["voxel_mean_sparse_encoder"]
        line["rooflines"] = {"sparse_encoder": dict(
            bound="mfma", achieved=sp["pairs"] / (ms * 1e-3) / 1e12, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
            frac=sp["pairs"] / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, traffic=None, ms_per_launch=ms,
            units_per_launch=B, flops_existing_pairs=sp["pairs"], flops_dense_equivalent=sp["dense"],
            note="flops of the (output row, kernel offset) pairs that exist, 2*Cin*Cout each, over the whole encoder "
                 "stage time (index building included); dense-equivalent counts all 27 offsets" +
                 ("; AMP: priced against the fp32 matrix peak for comparison with the fp32 line, not a utilisation of "
                  "the fp16 pipe (2.5 PFLOP/s) -- this mode is bound by the gather" if amp else ""))}
    return line


def bench_bevfusion_lidar(args, rank, world, dev):
    """BEVFusion LiDAR stream front half (config 5, configs/bevfusion/bevf_pp_2x8_1x_nusc.yaml:87-116): 0.25 m pillars
    on +-50 m (400 x 400), P = 64, V = 40 000, D = 4: hard_voxelize -> HardVFE (64, 64) -> PointPillarsScatter."""
    from paddle3d_amd import centerpoint as cpm
    from paddle3d_amd import synth

    B, V, PV, D4 = args.batch, 40000, 64, 4
    vs, pr = (0.25, 0.25, 8.0), (-50.0, -50.0, -5.0, 50.0, 50.0, 3.0)
    voxelizer = cpm.HardVoxelizer(vs, pr, PV, [30000, V]).eval()
    vfe = cpm.HardVFE(D4, (64, 64), False, True, True, vs, pr).to(dev).eval()
    scatter = cpm.PointPillarsScatter(64, vs, pr)
    pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(100 + B * rank + i, dims=D4) for i in range(B)])).to(dev)
    names = ["start", "hard_voxelize", "hard_vfe", "pointpillars_scatter"]

    def run(events):
        def mark(i):
            if events is not None:
                events[i].record()

        mark(0)
        voxels, coors, npv, nv = voxelizer(pts)
        mark(1)
        b, v, p, d = voxels.shape
        feats = vfe(voxels.view(b * v, p, d), npv.view(b * v), coors.view(b * v, 4))
        mark(2)
        canvas = scatter(feats, coors.view(b * v, 4), b)
        mark(3)
        return canvas, nv

    with torch.no_grad():
        dt, per_op_ms, out, info = _timed_loop(run, args, world, dev, names)
    if rank != 0:
        return None
    alg_v = 4 * N_POINTS * D4 + 4 * V * PV * D4 + 16 * V + 4
    a = alg_v * B / (per_op_ms["hard_voxelize"] * 1e-3) / 1e9
    alg_s = 4 * V * 64 + 16 * V + 4 * 64 * 400 * 400
    a_s = alg_s * B / (per_op_ms["pointpillars_scatter"] * 1e-3) / 1e9
    return {
        "metric": "frames/sec BEVFusion LiDAR stream front half (voxelize + HardVFE + scatter)",
        "value": world * B * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BEVFusion LiDAR stream: {N_POINTS} pts x {D4} per scene, 0.25 m pillars (400x400), "
                               f"P={PV}, max_voxels={V}, batch {B} distinct scenes/GPU/step, random-init weights, "
                               "hard_voxelize->HardVFE->PointPillarsScatter",
                   "frames_per_gpu_per_step": B, "max_voxels": V, "parallelism": f"dp{world} (frames)"},
        "roofline": dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBPS, unit="GB/s", frac=a / HBM_PEAK_GBPS, traffic=None,
                         ms_per_launch=per_op_ms["hard_voxelize"], units_per_launch=B,
                         algorithmic_bytes_per_unit=alg_v,
                         kernel="hard_voxelize launch sequence, wave form (vw_route + vw_group + vw_assign + vw_rows); "
                                "the fixed-shape [V, 64, 4] output dominates the bytes"),
        "rooflines": {"pointpillars_scatter": dict(bound="hbm", achieved=a_s, peak=HBM_PEAK_GBPS, unit="GB/s",
                                                   frac=a_s / HBM_PEAK_GBPS, traffic=None,
                                                   ms_per_launch=per_op_ms["pointpillars_scatter"],
                                                   units_per_launch=B, algorithmic_bytes_per_unit=alg_s)},
        "per_op_ms": per_op_ms, "voxels_first_frame": int(out[1][0]),
    }


def bench_pointpillars_kitti(args, rank, world, dev):
    """PointPillars-KITTI, the whole inference graph (config 1, configs/pointpillars/pointpillars_xyres16_kitti_car.yml:
    86-146): 16 384 camera-FOV points x 4, 0.16 m pillars (432 x 496), P = 32, V = 40 000: hard_voxelize ->
    PillarFeatureNet (64) -> PointPillarsScatter -> SECOND backbone -> FPN (transposed convolutions 1 / 2 / 4) -> SSD
    head (one 1x1 GEMM) -> anchor masks + decode + rotated NMS (ssd_postprocess)."""
    from paddle3d_amd import pointpillars as ppm
    from paddle3d_amd import synth

    B, V, PV, D4, NK = args.batch, 40000, 32, 4, 16384
    model = ppm.pointpillars_kitti_car((16000, V)).to(dev).eval()
    with torch.no_grad():
        model.head.cls_head.bias.fill_(-2.0)  # random weights: a few hundred anchors per frame pass the 0.05 threshold
    pts = torch.from_numpy(np.stack([synth.kitti_frame(100 + B * rank + i, NK) for i in range(B)])).to(dev)
    names = ["start", "hard_voxelize", "pillar_feature_net", "pointpillars_scatter", "dense", "ssd_head_postprocess"]

    def run(events):
        def mark(i):
            if events is not None:
                events[i].record()

        mark(0)
        voxels, coors, npv, nv = model.voxelizer(pts)
        mark(1)
        b, v, p, d = voxels.shape
        c4 = coors.view(b * v, 4)
        feats = model.pillar_encoder(voxels.view(b * v, p, d), npv.view(b * v), c4)
        mark(2)
        canvas = model.scatter(feats, c4, b)
        mark(3)
        x = model.neck(model.backbone(canvas))
        mark(4)
        out = model.head.post_process(model.head.head_map(x), model.anchor_generator, c4, device_only=True)
        mark(5)
        return out, nv

    with torch.no_grad():
        dt, per_op_ms, out, info = _timed_loop(run, args, world, dev, names)
    if rank != 0:
        return None
    alg_v = 4 * NK * D4 + 4 * V * PV * D4 + 16 * V + 4
    a = alg_v * B / (per_op_ms["hard_voxelize"] * 1e-3) / 1e9
    alg_s = 4 * V * 64 + 16 * V + 4 * 64 * 432 * 496
    a_s = alg_s * B / (per_op_ms["pointpillars_scatter"] * 1e-3) / 1e9

    def conv(cin, cout, k, h, w):
        return 2 * cin * cout * k * k * h * w

    s1 = 3 * conv(64, 64, 3, 248, 216) + 5 * conv(128, 128, 3, 124, 108) + 5 * conv(256, 256, 3, 62, 54)
    s2 = conv(64, 64, 3, 248, 216) + conv(64, 128, 3, 124, 108) + conv(128, 256, 3, 62, 54)
    other = conv(64, 128, 1, 248, 216) + conv(128, 128, 2, 124, 108) + conv(256, 128, 4, 62, 54)
    direct, executed = s1 + s2 + other, s1 / 4 + s2 + other
    tf = executed * B / (per_op_ms["dense"] * 1e-3) / 1e12
    return {
        "metric": "frames/sec PointPillars-KITTI (whole inference graph)",
        "value": world * B * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"PointPillars-KITTI: {NK} pts x {D4} per frame, 0.16 m pillars (432x496), P={PV}, "
                               f"max_voxels={V}, batch {B} distinct frames/GPU/step, random-init weights, "
                               "hard_voxelize->PillarFeatureNet(64)->PointPillarsScatter->SECOND+FPN->SSDHead->"
                               "anchor mask + decode + rotated NMS",
                   "frames_per_gpu_per_step": B, "max_voxels": V, "parallelism": f"dp{world} (frames)"},
        "roofline": dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBPS, unit="GB/s", frac=a / HBM_PEAK_GBPS, traffic=None,
                         ms_per_launch=per_op_ms["hard_voxelize"], units_per_launch=B,
                         algorithmic_bytes_per_unit=alg_v,
                         kernel="hard_voxelize launch sequence, tiled path; the fixed-shape [V, 32, 4] output is "
                                "20.5 of the 20.8 MB per frame"),
        "rooflines": {"pointpillars_scatter": (dict(bound="hbm", fused_into="dense_backbone_fpn", achieved=None,
                                                    peak=HBM_PEAK_GBPS, unit="GB/s", frac=None, traffic=None,
                                                    ms_per_launch=per_op_ms["pointpillars_scatter"], units_per_launch=B,
                                                    note="fused into the first backbone convolution: inverse-map "
                                                         "kernels only, no canvas written")
                                               if getattr(model, "fuse_scatter", False) else
                                               dict(bound="hbm", achieved=a_s, peak=HBM_PEAK_GBPS, unit="GB/s",
                                                    frac=a_s / HBM_PEAK_GBPS, traffic=None,
                                                    ms_per_launch=per_op_ms["pointpillars_scatter"],
                                                    units_per_launch=B, algorithmic_bytes_per_unit=alg_s)),
                      "dense_backbone_fpn": dict(bound="mfma", achieved=tf, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                                                 frac=tf / MFMA_F32_PEAK_TFLOPS, traffic=None,
                                                 ms_per_launch=per_op_ms["dense"], units_per_launch=B,
                                                 executed_flops_per_unit=executed, direct_form_flops_per_unit=direct,
                                                 note="executed flops: stride-1 3x3 layers by Winograd F(4x4,3x3) (a "
                                                      "quarter of the direct multiplies), the rest direct GEMMs")},
        "per_op_ms": per_op_ms, "voxels_first_frame": int(out[1][0]),
        "detections_first_frame": int(out[0][3][0]),
    }


def bench_bev_pool(args, rank, world, dev):
    """bev_pool_v2 forward at BEVDet4D size: 6 cameras x 118 depth bins x 16 x 44, C = 80, 128 x 128 BEV."""
    from paddle3d_amd import synth
    from paddle3d_amd.bevdet import LSSViewTransformer
    from paddle3d_amd.ops import bev_pool_v2 as bp

    # index sets from the real frustum geometry of bevdet4d_r50_depth_nuscenes.yml:174-186 (6 cameras of a synthetic
    # nuScenes-like rig), built on the device by pd3_frustum_to_lidar + pd3_voxel_pooling_prepare
    vt = LSSViewTransformer()
    cams = synth.camera_rig(0)
    coor = vt.get_lidar_coor(*[torch.from_numpy(cams[k]).to(dev) for k in ("rots", "trans", "cam2imgs", "post_rots",
                                                                           "post_trans", "bda")])
    rb, rd, rf, st, ln = vt.voxel_pooling_prepare_v2(coor)
    rng = np.random.default_rng(0)
    t = dict(depth=torch.from_numpy(rng.random((6, 118, 16, 44)).astype(np.float32)).to(dev),
             feat=torch.from_numpy(rng.normal(size=(6, 16, 44, 80)).astype(np.float32)).to(dev),
             ranks_depth=rd, ranks_feat=rf, ranks_bev=rb, interval_lengths=ln, interval_starts=st)
    shape = (1, 128, 128, 80)
    names = ["start", "bev_pool_v2"]

    def run(events):
        if events is not None:
            events[0].record()
        out = bp.bev_pool_v2(t["depth"], t["feat"], t["ranks_depth"], t["ranks_feat"], t["ranks_bev"],
                             t["interval_lengths"], t["interval_starts"], shape)
        if events is not None:
            events[1].record()
        return out

    dt, per_op_ms, out, info = _timed_loop(run, args, world, dev, names)
    if rank != 0:
        return None
    n_pts, n_int, c = int(t["ranks_bev"].numel()), int(t["interval_lengths"].numel()), int(t["feat"].shape[-1])
    alg = 4 * (n_pts * (1 + c) + 3 * n_pts + 2 * n_int) + 4 * int(out.numel())
    a = alg / (per_op_ms["bev_pool_v2"] * 1e-3) / 1e9
    return {
        "metric": "bev_pool_v2 forward frames/sec (BEVDet4D shapes)", "value": world * args.steps / dt, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"bev_pool_v2 forward: {n_pts} frustum points in {n_int} intervals, C={c}, BEV "
                               f"{tuple(shape)}", "parallelism": f"dp{world} (frames)"},
        "roofline": dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBPS, unit="GB/s", frac=a / HBM_PEAK_GBPS, traffic=None,
                         ms_per_launch=per_op_ms["bev_pool_v2"], units_per_launch=1, algorithmic_bytes_per_unit=alg,
                         kernel="bev_pool_v2_kernel (gathered operands counted once per use, SURVEY 8(d))"),
        "per_op_ms": per_op_ms,
    }


def bench_stub(args, rank, world, dev):
    """--stub-ops: the launch / timing / collective path of the default workload with the GPU ops replaced by a fixed
    synthetic detection set on the CPU, so that `bench.py --gpus N` (self-launch, gloo, barriers, MAX over ranks,
    all-gather inside the step) can be exercised without a GPU (tests/test_bench_dist_cpu.py).  The line it prints is
    marked `stub` and is not a measurement."""
    from paddle3d_amd import dist as pdist

    B, max_per_img = args.batch, 500
    g = torch.Generator().manual_seed(1234 + rank)
    bx = torch.randn(B, 498, 9, generator=g)
    sc = torch.rand(B, 498, generator=g)
    lb = torch.randint(0, 10, (B, 498), generator=g)
    cnt = torch.randint(1, 498, (B,), generator=g, dtype=torch.int32)
    names = ["start", "ops_stub", "gather"]
    pipe = pdist.GatherPipeline() if args.gather == "overlap" else None

    def hand_off(rec, c):
        if pipe is None:
            return pdist.gather_detections(rec, c)
        prev = pipe.submit(rec, c)
        return prev if prev is not None else (rec, c)

    def run_batch(frames, events=None):
        """`frames`: a [b, ...] tensor standing for a batch of scenes (only its length is used)."""
        b = frames.shape[0]
        if events is not None:
            events[0].record()
        time.sleep(0.002)  # stands for the device work of a step
        if events is not None:
            events[1].record()
        rec = pdist.pack_records(bx[:b], sc[:b], lb[:b], cnt[:b], max_per_img)
        out = hand_off(rec, cnt[:b])
        if events is not None:
            events[2].record()
        return out

    fake = torch.zeros(B, 4)
    finish = (lambda out: pipe.flush()) if pipe is not None else None
    dt, per_op_ms, out, _info = _timed_loop(lambda ev: run_batch(fake, ev), args, world, dev, names, finish=finish)
    assert out[0].shape[0] == world * B and out[1].shape[0] == world * B
    # what arrived is every rank's own record, in rank order (rank r's generator seed is 1234 + r)
    for r in range(world):
        gr = torch.Generator().manual_seed(1234 + r)
        want = pdist.pack_records(torch.randn(B, 498, 9, generator=gr), torch.rand(B, 498, generator=gr),
                                  torch.randint(0, 10, (B, 498), generator=gr),
                                  torch.randint(1, 498, (B,), generator=gr, dtype=torch.int32), max_per_img)
        assert torch.equal(out[0][r * B:(r + 1) * B], want), f"rank {r}'s records did not arrive intact"
    multi = {}
    if args.strong_frames > 0:
        loop = dict(_LAST_LOOP)
        flush = (lambda: pipe.flush()) if pipe is not None else None
        multi["strong_scaling"] = strong_scaling(lambda b: run_batch(b), flush, lambda ids: torch.zeros(len(ids), 4),
                                                 args.strong_frames, B, rank, world, dev, passes=2)
        multi["h2d_inclusive"] = h2d_inclusive(lambda b: run_batch(b), flush, torch.zeros(B, 4), torch.zeros(B, 4),
                                               args.steps, world, dev)
        multi["h2d_overlapped"] = h2d_overlapped(lambda b: run_batch(b), flush, [torch.zeros(B, 4), torch.ones(B, 4)],
                                                 args.steps, world, dev)
        _LAST_LOOP.clear()
        _LAST_LOOP.update(loop)
    if rank != 0:
        return None
    return {"metric": "scenes/sec CenterPoint-Pillars nuScenes 300k-pt sweeps", "stub": True,
            "value": world * B * args.steps / dt, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "STUB: no device ops; launch / collective path only", "frames_per_gpu_per_step": B,
                       "parallelism": f"dp{world} (frames)"},
            "per_op_ms": per_op_ms, "frames_gathered": int(out[1].shape[0]), "extras": multi,
            "result_hand_off": "overlap" if pipe is not None else "sync"}


def _self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script through torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1) and hand back its exit status."""
    import socket
    import subprocess

    if not args.stub_ops:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node; refusing to "
                             f"report a {args.gpus}-GPU number from fewer devices")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default 16; 8 for centerpoint_voxel; 32 gives the headline workload +2 %% scenes/s)")
    ap.add_argument("--max-voxels", type=int, default=30000)
    ap.add_argument("--workload", default="centerpoint_pillars",
                    choices=["centerpoint_pillars", "centerpoint_pillars_amp", "centerpoint_voxel",
                             "centerpoint_voxel_amp", "bev_pool_v2",
                             "bevfusion_lidar", "pointpillars_kitti"])
    ap.add_argument("--vox-path", type=int, default=0, help="pd3_hard_voxelize_path selector (0 = library default, "
                    "1 generic sort, 2 tiled with a compact payload array, 3 tiled with gathered rows, 5 wave form)")
    ap.add_argument("--front", choices=["fused", "pair"], default="fused", help="front half of the pillar graphs: fused = "
                    "voxelizer -> PFN through an index of the points (the model path), pair = the two full operators")
    ap.add_argument("--graph", action="store_true", help="replay the step as five captured HIP graphs (one per op) "
                    "instead of launching every kernel from the host (centerpoint_pillars; same kernels and buffers)")
    ap.add_argument("--repeats", type=int, default=None, help="time the same K steps this many more times after the "
                    "contract block and report min / median / max (default 4 at N=1, 0 otherwise)")
    ap.add_argument("--gather", choices=["overlap", "sync"], default="overlap", help="result hand-off: the all-gather "
                    "of batch k overlapped with batch k + 1 (default) or inside the step")
    ap.add_argument("--strong-frames", type=int, default=128, help="frames of the fixed set of the strong-scaling extra "
                    "(sharded over the ranks by dist.shard_frames; 0 = off)")
    ap.add_argument("--no-affinity", action="store_true", help="do not pin the ranks of an N > 1 run to the cores next "
                    "to their GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--map-frames", type=int, default=16, help="frames of the mAP proxy in extras (device detections "
                    "scored against the oracle pipeline's; ~1 s of CPU per frame; 0 = off)")
    ap.add_argument("--no-extras", action="store_true", help="skip the h2d-inclusive / batch-1 / ceiling measurements "
                    "and the other workloads (profiling runs: only warm-up + timed steps are launched)")
    ap.add_argument("--stub-ops", action="store_true", help="test hook: no device ops, launch / collective path only "
                    "(CPU, gloo); the line is marked stub")
    args = ap.parse_args(argv)
    if args.batch is None:
        args.batch = {"centerpoint_voxel": 8, "centerpoint_voxel_amp": 8}.get(args.workload, 16)
    if args.repeats is None:
        args.repeats = 4 if args.gpus == 1 else 0

    # `python bench.py --gpus N` with no launcher around it: become the launcher
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_self_launch(args, argv))

    from paddle3d_amd import dist as pdist

    rank, world, local = pdist.init_from_env("gloo" if args.stub_ops else None)
    if world != args.gpus:
        if world > 1:
            torch.distributed.destroy_process_group()
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing "
                         "to print a line whose n_gpus would not be what was asked for")
    affinity = None
    if world > 1 and not args.no_affinity:
        affinity = pdist.set_cpu_affinity(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if args.stub_ops:
        dev = torch.device("cpu")
        line = bench_stub(args, rank, world, dev)
    else:
        from paddle3d_amd._lib import lib

        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP ops have no CPU path")
        lib()  # fail loudly if libpaddle3d_amd.so is missing
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        torch.manual_seed(0)
        fn = dict(centerpoint_pillars=bench_pillars, centerpoint_pillars_amp=bench_pillars,
                  centerpoint_voxel=bench_voxel, centerpoint_voxel_amp=bench_voxel, bev_pool_v2=bench_bev_pool,
                  bevfusion_lidar=bench_bevfusion_lidar, pointpillars_kitti=bench_pointpillars_kitti)[args.workload]
        line = fn(args, rank, world, dev)
    if rank == 0:
        if world > 1:
            line.setdefault("config", {})["cpu_affinity_rank0"] = (
                f"{len(affinity)} cpus {affinity[0]}-{affinity[-1]}" if affinity else "not pinned")
        print(json.dumps(_dist_fields(line, args, world)))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
