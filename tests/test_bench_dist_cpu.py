"""bench.py's N > 1 path on CPU: `python bench.py --gpus 2` has to launch two ranks by itself, run the bench's own
timed loop (barriers, MAX over ranks, the all-gather inside the step) over gloo and print n_gpus = 2; a launch that
cannot give the asked-for number of ranks has to fail loudly.  The device ops are replaced at the run() boundary by
--stub-ops (no GPU in this container); everything around them is the code the driver runs."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_self_launch_two_ranks_over_gloo():
    r = _run(["--gpus", "2", "--stub-ops", "--steps", "3", "--warmup", "1", "--batch", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["collective_backend"] == "gloo"
    assert line["steps"] == 3 and line["warmup"] == 1 and line["stub"] is True
    assert line["frames_gathered"] == 4  # every rank's frames arrived through the all-gather
    assert abs(line["value"] - 2 * 2 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]


def test_eight_ranks_over_gloo_with_extras():
    """What the driver's 8-GPU run exercises, minus the device: eight ranks, CPU affinity per local rank, the
    overlapped all-gather (every rank's records arrive intact and in rank order -- asserted inside the stub), the
    strong-scaling pass over a fixed 32-frame set sharded by dist.shard_frames, the per-rank H2D-inclusive figure."""
    r = _run(["--gpus", "8", "--stub-ops", "--steps", "3", "--warmup", "1", "--batch", "2", "--strong-frames", "32"],
             timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["frames_gathered"] == 16
    assert line["result_hand_off"] == "overlap" and line["scaling"] == "weak"
    ss = line["extras"]["strong_scaling"]
    assert ss["scaling"] == "strong" and ss["frames"] == 32 and ss["frames_per_rank"] == 4
    assert ss["batches_per_rank_per_pass"] == 2 and ss["value"] > 0
    assert line["extras"]["h2d_inclusive"]["value"] > 0
    assert line["extras"]["h2d_overlapped"]["value"] > 0  # the double-buffered stage (dist.H2DStage) on every rank
    assert "cpu_affinity_rank0" in line["config"]


def test_sync_gather_mode_still_works():
    r = _run(["--gpus", "2", "--stub-ops", "--steps", "2", "--warmup", "1", "--batch", "2", "--gather", "sync",
              "--strong-frames", "0", "--no-affinity"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["result_hand_off"] == "sync" and line["frames_gathered"] == 4 and line["extras"] == {}
    assert line["config"]["cpu_affinity_rank0"] == "not pinned"


def test_single_rank_line_has_launch_fields():
    r = _run(["--gpus", "1", "--stub-ops", "--steps", "2", "--warmup", "0", "--batch", "2", "--repeats", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["collective_backend"] is None
    rb = line["extras"]["repeat_blocks"]
    assert rb["blocks"] == 2 and rb["min"] <= rb["median"] <= rb["max"]


def test_refuses_more_gpus_than_visible():
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return  # a 2+-GPU box may run it for real
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stderr + r.stdout)


def test_refuses_world_size_mismatch():
    # a launcher that started one rank although --gpus 2 was asked for must not produce an n_gpus = 1 line
    r = _run(["--gpus", "2", "--stub-ops", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
