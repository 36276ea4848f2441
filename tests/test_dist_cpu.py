"""World-size-2 gloo test of the frame sharding + result all-gather (the N>1 path of bench.py)."""
import os
import socket
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
