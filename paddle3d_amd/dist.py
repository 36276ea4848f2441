"""Frame-parallel inference across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference evaluates on a single device (paddle3d/apis/trainer.py:47-51,402) and has no collective
on the inference path; frames are independent, so the path shards with NO data-path collective: rank r
takes frames r, r + world, ... (or its own slice of a batch).  The only exchange is the result hand-off
the north star asks for: ONE all-gather per batch of frames of a fixed-shape record
    float32 [frames, max_per_img, 11] = 9 box values (7 without velocity, zero padded), score, label
plus int32 [frames] row counts.  At ~22 KB per frame the collective is latency bound, so it is issued
once per batch, never per frame or per task.
`backend="nccl"` is RCCL on ROCm; tests run the same code over gloo on CPU with world_size 2 and 8.

`GatherPipeline` issues that all-gather asynchronously (RCCL runs it on its own stream): batch k's records travel
while batch k + 1 is computed, the gathered result of batch k is handed out when batch k + 1 is submitted (or by
`flush()`).  `set_cpu_affinity` pins a rank's host threads to the cores next to its GPU (eight ranks of one node
otherwise migrate across both sockets while they enqueue ~60 launches per step).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

RECORD_WIDTH = 11


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world, local_rank); a plain single-process run returns (0, 1, 0) without a process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_frames(num_frames: int, rank: int, world: int):
    """Round-robin frame ownership: frame i -> rank i % world."""
    return list(range(rank, num_frames, world))


def pack_records(boxes: torch.Tensor, scores: torch.Tensor, labels: torch.Tensor, counts: torch.Tensor,
                 max_per_img: int = 500) -> torch.Tensor:
    """[F, R, dims] / [F, R] / [F, R] (+ counts [F]) -> float32 [F, max_per_img, 11], rows >= count zeroed."""
    f, r, dims = boxes.shape
    rec = torch.zeros((f, max_per_img, RECORD_WIDTH), dtype=torch.float32, device=boxes.device)
    k = min(r, max_per_img)
    rec[:, :k, :dims] = boxes[:, :k]
    rec[:, :k, 9] = scores[:, :k]
    rec[:, :k, 10] = labels[:, :k].to(torch.float32)
    valid = torch.arange(max_per_img, device=boxes.device).unsqueeze(0) < counts.clamp(max=max_per_img).unsqueeze(1)
    # where(), not a 0/1 multiply: rows >= count may hold anything (NaN * 0 = NaN)
    return torch.where(valid.unsqueeze(-1), rec, torch.zeros((), dtype=rec.dtype, device=rec.device))


def gather_detections(records: torch.Tensor, counts: torch.Tensor):
    """All-gather the per-rank records [F, M, 11] and counts [F] -> ([world*F, M, 11], [world*F]) in rank order.
    Every rank ends up with every frame's boxes (what a metric aggregation needs).  Two collectives of a
    fixed, equal size per rank: RCCL all_gather_into_tensor."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return records, counts
    world = dist.get_world_size()
    out_r = torch.empty((world * records.shape[0],) + tuple(records.shape[1:]), dtype=records.dtype,
                        device=records.device)
    out_c = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(out_r, records.contiguous())
    dist.all_gather_into_tensor(out_c, counts.contiguous())
    return out_r, out_c


def unpack_records(records: torch.Tensor, counts: torch.Tensor, with_velocity: bool = True):
    """Inverse of pack_records for host-side consumers: list of dict(box3d_lidar, scores, label_preds)."""
    dims = 9 if with_velocity else 7
    out = []
    for rec, k in zip(records.cpu(), counts.cpu().tolist()):
        out.append(dict(box3d_lidar=rec[:k, :dims], scores=rec[:k, 9], label_preds=rec[:k, 10].to(torch.int64)))
    return out


class GatherPipeline:
    """The per-batch result hand-off, one batch deep: `submit(records, counts)` starts the all-gather of THIS batch
    without waiting for it (async_op: RCCL's own stream, ordered after the producing kernels of the current stream)
    and returns the gathered (records, counts) of the PREVIOUS batch (None the first time); `flush()` waits for the
    batch in flight and returns it.  With one rank it degenerates to handing the result through one batch late, so
    the calling code is the same for any world size.  `submit` takes a private copy of its inputs (22 KB per frame,
    on the current stream): the caller may overwrite `records` / `counts` as soon as submit returns -- a captured HIP
    graph replays into the same static buffers every step, and a preallocated record buffer is reused the same way."""

    def __init__(self):
        self._pending = None

    @staticmethod
    def _start(records, counts):
        records, counts = records.clone(memory_format=torch.contiguous_format), counts.clone(
            memory_format=torch.contiguous_format)
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return (None, None, records, counts, records, counts)
        world = dist.get_world_size()
        out_r = torch.empty((world * records.shape[0],) + tuple(records.shape[1:]), dtype=records.dtype,
                            device=records.device)
        out_c = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
        w_r = dist.all_gather_into_tensor(out_r, records, async_op=True)
        w_c = dist.all_gather_into_tensor(out_c, counts, async_op=True)
        return (w_r, w_c, out_r, out_c, records, counts)

    @staticmethod
    def _finish(p):
        if p is None:
            return None
        w_r, w_c, out_r, out_c, _r, _c = p
        if w_r is not None:
            w_r.wait()  # makes the CURRENT stream wait for the collective (no host block on RCCL)
            w_c.wait()
        return out_r, out_c

    def submit(self, records: torch.Tensor, counts: torch.Tensor):
        prev = self._finish(self._pending)
        self._pending = self._start(records, counts)
        return prev

    def flush(self):
        prev, self._pending = self._finish(self._pending), None
        return prev


class H2DStage:
    """Double-buffered host -> device staging of the input batches (the reference's deploy loop copies every frame from
    the host before it runs it, deploy/centerpoint/cpp/main.cc:177-181; here the copy of batch k + 1 travels on its own
    stream while batch k is computed).

        stage = H2DStage(host_batch.shape, torch.float32, device)
        stage.submit(host[0])
        for k in range(K):
            if k + 1 < K:
                stage.submit(host[k + 1])     # copy stream; waits only for the buffer's previous consumer
            x = stage.acquire()               # the compute stream waits for THIS batch's copy, not for the host
            run(x)
            stage.release()                   # the buffer may be overwritten once the kernels enqueued so far are done

    `submit` takes pinned host memory for a truly asynchronous copy (pageable memory works, synchronously).  No host
    synchronisation anywhere: ordering is by events between the two streams.  On a CPU device (the gloo tests) the
    copies are plain synchronous copies."""

    def __init__(self, shape, dtype, device, depth: int = 2):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.buffers = [torch.empty(tuple(shape), dtype=dtype, device=self.device) for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.ready = [None] * depth      # copy finished (recorded on the copy stream)
        self.consumed = [None] * depth   # consumer finished (recorded on the compute stream)
        self._head = self._tail = self._inflight = 0

    def submit(self, host: torch.Tensor):
        if self._inflight == len(self.buffers):
            raise RuntimeError("H2DStage: every buffer holds a batch that has not been released")
        i = self._head
        self._head = (i + 1) % len(self.buffers)
        self._inflight += 1
        if not self.cuda:
            self.buffers[i].copy_(host)
            return
        with torch.cuda.stream(self.copy_stream):
            if self.consumed[i] is not None:
                self.copy_stream.wait_event(self.consumed[i])
            self.buffers[i].copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
            self.ready[i] = ev

    def acquire(self) -> torch.Tensor:
        if self._inflight == 0:
            raise RuntimeError("H2DStage: acquire() without a submitted batch")
        i = self._tail
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(self.ready[i])
        return self.buffers[i]

    def release(self):
        i = self._tail
        self._tail = (i + 1) % len(self.buffers)
        self._inflight -= 1
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.consumed[i] = ev


def choose_launch(host_ms_per_step: float, gpu_ms_per_step: float, requested: str = "auto") -> str:
    """"eager" or "graph" for the step's launch path.  `requested` = "eager" / "graph" pins it; "auto" turns HIP-graph
    replay on when the host needs more than half of the GPU's step time to enqueue a step -- measured on THIS rank while
    the node's other ranks run their own warm-up, i.e. under the contention that will be there (ranks are pinned to
    disjoint cores by set_cpu_affinity, so their enqueue times do not add up; what can happen on a loaded host is that
    each rank's own enqueue time grows).  Below that ratio the GPU never waits for the host and replay changes nothing
    (measured in rounds 2 and 4: 0.8-1.3 ms of enqueue for a 8.4-10 ms step)."""
    if requested in ("eager", "graph"):
        return requested
    if gpu_ms_per_step <= 0:
        return "eager"
    return "graph" if host_ms_per_step > 0.5 * gpu_ms_per_step else "eager"


def _cpulist(text: str):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_node(local_rank: int):
    """NUMA node of the GPU a rank drives (sysfs, by PCI address), or None when it cannot be told."""
    try:
        prop = torch.cuda.get_device_properties(local_rank)
        addr = f"{getattr(prop, 'pci_domain_id', 0):04x}:{prop.pci_bus_id:02x}:{prop.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{addr}/numa_node") as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except Exception:  # noqa: BLE001 -- no GPU, an older torch, a container without sysfs
        return None


def plan_cpu_affinity(local_rank: int, local_world: int, allowed=None, numa_of_rank=None, node_cpus=None):
    """CPUs for one of `local_world` ranks of a node: the cores of its GPU's NUMA node (shared evenly among the ranks
    whose GPUs sit on that node) when the topology is known, else a contiguous 1 / local_world slice of the allowed
    set.  Pure function of its arguments (the test feeds it a made-up topology)."""
    allowed = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(allowed)
    if local_world <= 1 or not allowed:
        return allowed
    if numa_of_rank is not None and node_cpus is not None and numa_of_rank[local_rank] is not None:
        node = numa_of_rank[local_rank]
        mates = [r for r in range(local_world) if numa_of_rank[r] == node]
        cpus = [c for c in node_cpus.get(node, []) if c in set(allowed)]
        if len(cpus) >= len(mates):
            k = mates.index(local_rank)
            per = len(cpus) // len(mates)
            return cpus[k * per:(k + 1) * per]
    per = max(1, len(allowed) // local_world)
    lo = min(local_rank * per, len(allowed) - per)
    return allowed[lo:lo + per]


def set_cpu_affinity(local_rank: int, local_world: int):
    """Pin this process (and the threads it starts later) per plan_cpu_affinity; returns the CPU list it chose, or
    None when the platform has no sched_setaffinity / the call is refused.  torch's intra-op thread count follows."""
    try:
        numa = [gpu_numa_node(r) for r in range(local_world)] if torch.cuda.is_available() else None
        node_cpus = None
        if numa and any(n is not None for n in numa):
            node_cpus = {}
            for n in set(x for x in numa if x is not None):
                with open(f"/sys/devices/system/node/node{n}/cpulist") as f:
                    node_cpus[n] = _cpulist(f.read())
        cpus = plan_cpu_affinity(local_rank, local_world, None, numa, node_cpus)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        torch.set_num_threads(max(1, min(torch.get_num_threads(), len(cpus))))
        return cpus
    except Exception:  # noqa: BLE001 -- affinity is an optimisation, never a requirement
        return None
