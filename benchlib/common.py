"""bench.py's shared parts: machine peaks, the contract's timed loop (W warm-up steps, EXACTLY K steps between barrier +
synchronize on both sides, MAX over ranks), HIP-event intervals per op, the roofline objects, the multi-rank extras."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec (guides: ~6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32-input MFMA peak (MI355X_MICROARCH.md)
# Dense matrix-core peak PER PIPE (MI355X_MICROARCH.md, "Matrix cores"): a roofline is priced against the pipe the
# kernel that ran issues on, never against another one.  "bf16x3" = fp32 arithmetic carried by SIX bf16 products per
# fp32 product (csrc/sparse_conv_x3.hip, csrc/conv_x3.hip): counted in fp32-equivalent flops, its ceiling is a sixth of
# the bf16 pipe's.
MFMA_PEAK_TFLOPS = {"f32": 157.3, "f16": 2500.0, "bf16": 2500.0, "bf16x3": 2500.0 / 6.0}

N_POINTS, DIMS, P = 300_000, 5, 20


def algorithmic_bytes(v, n=N_POINTS, d=DIMS, p=P):
    """SURVEY.md section 8(d)."""
    vox = 4 * n * d + 4 * v * p * d + 12 * v + 4 * v + 4
    scatter = 4 * v * 64 + 16 * v + 4 * 64 * 512 * 512
    pfn = 4 * v * p * d + 4 * v + 16 * v + 4 * v * 64
    post = 4 * 128 * 128 * 70
    return dict(hard_voxelize=vox, pointpillars_scatter=scatter, pillar_feature_net=pfn,
                centerpoint_postprocess=post)


def hbm_roofline(bytes_per_launch, ms, units_per_launch=1, traffic=None, traffic_source=None, **more):
    """`roofline` object of an HBM-bound kernel (sequence): algorithmic bytes per launch / its HIP-event duration."""
    a = bytes_per_launch / (ms * 1e-3) / 1e9
    out = dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBPS, unit="GB/s", frac=a / HBM_PEAK_GBPS, traffic=traffic,
               ms_per_launch=ms, units_per_launch=units_per_launch,
               algorithmic_bytes_per_unit=bytes_per_launch / max(1, units_per_launch))
    if traffic is not None or traffic_source is not None:
        out["traffic_source"] = traffic_source
    out.update(more)
    return out


def mfma_roofline(flops_by_pipe, ms, units_per_launch=1, **more):
    """`roofline` object of a matrix-core-bound kernel (sequence).  flops_by_pipe: {"f32" | "f16" | "bf16" | "bf16x3":
    EXECUTED flops per launch on that pipe} (bf16x3 in fp32-equivalent flops).  achieved = all flops / time; peak = the
    rate of the same mix with every part at ITS pipe's dense peak (sum of flops / sum of flops_i / peak_i), so frac =
    (time at peak) / (time measured) and cannot exceed 1 unless the flop count is wrong."""
    flops_by_pipe = {k: float(v) for k, v in flops_by_pipe.items() if v}
    total = sum(flops_by_pipe.values())
    t_peak = sum(v / (MFMA_PEAK_TFLOPS[k] * 1e12) for k, v in flops_by_pipe.items())
    peak = total / t_peak / 1e12 if t_peak > 0 else MFMA_PEAK_TFLOPS["f32"]
    ach = total / (ms * 1e-3) / 1e12
    out = dict(bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak, traffic=None, ms_per_launch=ms,
               units_per_launch=units_per_launch, pipes={k: dict(executed_flops_per_launch=v, peak=MFMA_PEAK_TFLOPS[k])
                                                         for k, v in flops_by_pipe.items()})
    out.update(more)
    return out


def make_batch(batch, seed0, device=None, pin=False):
    from paddle3d_amd import synth

    arr = np.stack([synth.nuscenes_sweep(seed0 + i) for i in range(batch)])  # `batch` DISTINCT frames
    t = torch.from_numpy(arr)
    if pin:
        return t.pin_memory()
    return t.to(device)


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
