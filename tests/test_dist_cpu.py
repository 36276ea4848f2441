"""World-size-2 gloo test of the frame sharding + result all-gather (the N>1 path of bench.py)."""
import os
import socket

import pytest
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from paddle3d_amd import dist as pdist

    r, w, _ = pdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    frames = pdist.shard_frames(7, r, w)
    # per-frame fake detections whose content encodes (frame id, row)
    f = len(frames)
    rows = 12
    boxes = torch.zeros(f, rows, 9)
    scores = torch.zeros(f, rows)
    labels = torch.zeros(f, rows, dtype=torch.int64)
    counts = torch.zeros(f, dtype=torch.int32)
    for i, fid in enumerate(frames):
        k = fid % 5 + 1
        counts[i] = k
        boxes[i, :, 0] = fid
        boxes[i, :, 1] = torch.arange(rows)
        scores[i] = 1.0 / (1 + torch.arange(rows))
        labels[i] = fid % 10
    # equal-size records per rank: pad the shorter shard with an empty frame
    per = (7 + w - 1) // w
    if f < per:
        pad = per - f
        boxes = torch.cat([boxes, torch.zeros(pad, rows, 9)])
        scores = torch.cat([scores, torch.zeros(pad, rows)])
        labels = torch.cat([labels, torch.zeros(pad, rows, dtype=torch.int64)])
        counts = torch.cat([counts, torch.zeros(pad, dtype=torch.int32)])
    rec = pdist.pack_records(boxes, scores, labels, counts, max_per_img=10)
    all_rec, all_cnt = pdist.gather_detections(rec, counts)
    dist.barrier()
    # numpy arrays travel by value (tensors would go through shared-memory handles that die with this process)
    q.put((rank, all_rec.numpy().copy(), all_cnt.numpy().copy(), frames))
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort(key=lambda t: t[0])
    results = [(r, torch.from_numpy(a), torch.from_numpy(b), f) for r, a, b, f in results]
    (_, rec0, cnt0, fr0), (_, rec1, cnt1, fr1) = results
    # both ranks hold identical gathered data
    assert torch.equal(rec0, rec1) and torch.equal(cnt0, cnt1)
    assert fr0 == [0, 2, 4, 6] and fr1 == [1, 3, 5]
    assert rec0.shape == (8, 10, 11) and cnt0.tolist() == [1, 3, 5, 2, 2, 4, 1, 0]
    # frame ids come back in rank-major order and rows beyond the count are zero
    order = fr0 + fr1
    for slot, fid in enumerate(order):
        k = int(cnt0[slot])
        assert (rec0[slot, :k, 0] == fid).all() and (rec0[slot, :k, 10] == fid % 10).all()
        assert not rec0[slot, k:].any()
    from paddle3d_amd import dist as pdist

    dets = pdist.unpack_records(rec0, cnt0)
    assert dets[1]["box3d_lidar"].shape == (3, 9) and dets[1]["label_preds"].dtype == torch.int64


def test_single_process_is_passthrough():
    from paddle3d_amd import dist as pdist

    assert pdist.shard_frames(5, 0, 1) == [0, 1, 2, 3, 4]
    rec = torch.zeros(2, 4, 11)
    cnt = torch.zeros(2, dtype=torch.int32)
    a, b = pdist.gather_detections(rec, cnt)
    assert a is rec and b is cnt


def test_plan_cpu_affinity():
    """Ranks get disjoint CPU sets: the cores of their GPU's NUMA node when the topology is known, else contiguous
    slices; more ranks than cores on a node falls back to the slices."""
    from paddle3d_amd import dist as pdist

    flat = [pdist.plan_cpu_affinity(r, 8, range(128)) for r in range(8)]
    assert all(len(c) == 16 for c in flat) and sorted(sum(flat, [])) == list(range(128))
    numa = [0, 0, 0, 0, 1, 1, 1, 1]
    nodes = {0: list(range(0, 64)), 1: list(range(64, 128))}
    by_node = [pdist.plan_cpu_affinity(r, 8, range(128), numa, nodes) for r in range(8)]
    assert sorted(sum(by_node, [])) == list(range(128))
    assert all(set(by_node[r]) <= set(nodes[numa[r]]) for r in range(8))
    # interleaved GPU -> node map (how some boards enumerate): still disjoint, still on the right node
    numa2 = [0, 1, 0, 1, 0, 1, 0, 1]
    by2 = [pdist.plan_cpu_affinity(r, 8, range(128), numa2, nodes) for r in range(8)]
    assert len(set(sum(by2, []))) == 128 and all(set(by2[r]) <= set(nodes[numa2[r]]) for r in range(8))
    # a restricted cgroup (8 cpus allowed, all on node 0, 8 ranks): slices of the allowed set, one cpu each
    tiny = [pdist.plan_cpu_affinity(r, 8, range(8), numa, nodes) for r in range(8)]
    assert all(len(c) >= 1 for c in tiny)
    assert pdist.plan_cpu_affinity(0, 1, range(4)) == [0, 1, 2, 3]
    assert pdist._cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]


def test_gather_pipeline_single_rank_is_one_batch_late():
    from paddle3d_amd import dist as pdist

    pipe = pdist.GatherPipeline()
    a, b = (torch.ones(2, 3, 11), torch.tensor([1, 2], dtype=torch.int32)), (torch.zeros(2, 3, 11), torch.tensor([0, 3], dtype=torch.int32))
    assert pipe.submit(*a) is None
    got = pipe.submit(*b)
    assert torch.equal(got[0], a[0]) and torch.equal(got[1], a[1])
    got = pipe.flush()
    assert torch.equal(got[0], b[0]) and torch.equal(got[1], b[1])
    assert pipe.flush() is None


def test_gather_pipeline_inputs_may_be_reused_after_submit():
    """A captured HIP graph (or a preallocated record buffer) writes every batch into the same tensors: what submit
    hands back one batch later must be the values at submit time, not the buffer's later contents."""
    from paddle3d_amd import dist as pdist

    pipe = pdist.GatherPipeline()
    rec, cnt = torch.zeros(2, 3, 11), torch.zeros(2, dtype=torch.int32)
    rec.fill_(1.0)
    cnt.fill_(1)
    assert pipe.submit(rec, cnt) is None
    rec.fill_(2.0)  # batch k + 1 overwrites the static buffers ...
    cnt.fill_(2)
    got = pipe.submit(rec, cnt)
    assert (got[0] == 1.0).all() and (got[1] == 1).all()  # ... batch k's result is untouched
    rec.fill_(3.0)
    got = pipe.flush()
    assert (got[0] == 2.0).all() and (got[1] == 2).all()


def test_h2d_stage_order_and_capacity():
    """dist.H2DStage on a CPU device (plain copies): batches come out in submission order, the buffers alternate, a
    third submit without a release is refused."""
    from paddle3d_amd import dist as pdist

    stage = pdist.H2DStage((2, 3), torch.float32, "cpu")
    host = [torch.full((2, 3), float(k)) for k in range(5)]
    stage.submit(host[0])
    seen = []
    for k in range(5):
        if k + 1 < 5:
            stage.submit(host[k + 1])
        x = stage.acquire()
        seen.append((float(x[0, 0]), x.data_ptr()))
        stage.release()
    assert [v for v, _ in seen] == [0.0, 1.0, 2.0, 3.0, 4.0]
    assert seen[0][1] == seen[2][1] != seen[1][1]
    stage.submit(host[0])
    stage.submit(host[1])
    with pytest.raises(RuntimeError):
        stage.submit(host[2])
    with pytest.raises(RuntimeError):
        pdist.H2DStage((1,), torch.float32, "cpu").acquire()


def test_choose_launch_policy():
    from paddle3d_amd import dist as pdist

    assert pdist.choose_launch(1.3, 8.4) == "eager"        # the measured single-rank case
    assert pdist.choose_launch(1.3, 4.3) == "eager"        # AMP step
    assert pdist.choose_launch(5.0, 8.4) == "graph"        # a loaded host: enqueue above half a step
    assert pdist.choose_launch(5.0, 8.4, "eager") == "eager" and pdist.choose_launch(0.1, 8.4, "graph") == "graph"
    assert pdist.choose_launch(1.0, 0.0) == "eager"
