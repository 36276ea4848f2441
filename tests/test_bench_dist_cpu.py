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
    assert line["collective_backend"] == "gloo" and line["config"]["launch"] and line["config"]["launch_reason"]
    assert all(0.0 <= v <= 1.0 for _, v in _fracs(line))
    assert line["result_hand_off"] == "overlap" and line["scaling"] == "weak"
    ss = line["extras"]["strong_scaling"]
    assert ss["scaling"] == "strong" and ss["frames"] == 32 and ss["frames_per_rank"] == 4
    assert ss["batches_per_rank_per_pass"] == 2 and ss["value"] > 0
    assert line["extras"]["h2d_inclusive"]["value"] > 0
    assert line["extras"]["h2d_overlapped"]["value"] > 0  # the double-buffered stage (dist.H2DStage) on every rank
    assert "cpu_affinity_rank0" in line["config"]


def _fracs(node, path=""):
    """Every `frac` anywhere in a bench line (recursively), with its path."""
    found = []
    if isinstance(node, dict):
        for k, v in node.items():
            if k == "frac" and v is not None:
                found.append((path + "/frac", v))
            else:
                found += _fracs(v, path + "/" + str(k))
    elif isinstance(node, list):
        for i, v in enumerate(node):
            found += _fracs(v, f"{path}[{i}]")
    return found


def test_roofline_objects_are_priced_per_pipe():
    """Round 5's AMP / bf16x3 lines divided fp16 work by the fp32 peak (frac 1.32 and 1.20).  The helpers every
    workload now builds its roofline objects with price each part against the pipe it runs on: the mix's peak is the
    rate with every part at ITS peak, so a correct flop count cannot give frac > 1."""
    sys.path.insert(0, ROOT)
    from benchlib.common import MFMA_PEAK_TFLOPS, hbm_roofline, mfma_roofline

    assert MFMA_PEAK_TFLOPS == {"f32": 157.3, "f16": 2500.0, "bf16": 2500.0, "bf16x3": 2500.0 / 6.0}
    # round 5's AMP dense graph: 127.2 GFLOP x 16 frames in 3.234 ms on the fp16 pipe = 629 TFLOP/s = 0.25
    r = mfma_roofline({"f16": 127.2e9 * 16}, 3.234, 16)
    assert r["peak"] == 2500.0 and abs(r["achieved"] - 629.3) < 1.0 and abs(r["frac"] - 0.2517) < 1e-3
    # a mix: each part exactly at its pipe's peak -> frac 1 (never above)
    t_ms = (1e12 / 157.3e12 + 6e12 / (2500e12 / 6)) * 1e3
    r = mfma_roofline({"f32": 1e12, "bf16x3": 6e12}, t_ms)
    assert abs(r["frac"] - 1.0) < 1e-9 and 157.3 < r["peak"] < 2500.0 / 6 and set(r["pipes"]) == {"f32", "bf16x3"}
    # the same flops booked on the wrong (fp32) pipe would have read 3.5: the denominators differ by that much
    assert abs(mfma_roofline({"f32": 7e12}, t_ms)["frac"] - 7e12 / (t_ms * 1e-3) / 157.3e12) < 1e-9
    h = hbm_roofline(295.68e6, 0.1329, 16, traffic=498.8e6, traffic_source="profiles/r05_b16_traffic.json")
    assert abs(h["frac"] - 0.278) < 1e-3 and h["traffic_source"].endswith("r05_b16_traffic.json")
    assert h["algorithmic_bytes_per_unit"] == 295.68e6 / 16


def test_dense_graph_flops_follow_the_kernels_that_run():
    """The dense graph's roofline books every layer on the pipe its kernel issues on (round 6: the stride-2 openers and the
    FPN levels moved to bf16x3, the first layer became a sparse convolution whose products are counted on the batch)."""
    import types

    import torch

    sys.path.insert(0, ROOT)
    from benchlib.pillars import dense_flops, dense_flops_by_pipe, first_layer_pairs
    from paddle3d_amd.ops import conv

    direct, executed = dense_flops()
    model = types.SimpleNamespace(backbone=types.SimpleNamespace(sparse_first=True))
    old = conv.S2_BF16X3, conv.PATCH_BF16X3
    try:
        conv.S2_BF16X3 = conv.PATCH_BF16X3 = False
        plain = dense_flops_by_pipe(None)
        assert set(plain) == {"f32"} and abs(plain["f32"] - executed) < 1e-3 * executed  # everything on the fp32 pipe
        conv.S2_BF16X3 = conv.PATCH_BF16X3 = True
        now = dense_flops_by_pipe(model, first_layer_pairs=67000.0)
        first_dense = 2 * 64 * 64 * 9 * 256 * 256
        moved = 2 * 9 * 128 * 128 * 64 * 128 + 2 * 9 * 64 * 64 * 128 * 256 + 2 * 4 * 64 * 128 * 128 * 128 + \
            2 * 128 * 128 * 128 * 128 + 2 * 256 * 128 * 128 * 128
        assert abs(now["bf16x3"] - (moved + 67000.0 * 2 * 64 * 64)) < 1.0
        assert abs(now["f32"] - (executed - moved - first_dense)) < 1e-3 * executed
    finally:
        conv.S2_BF16X3, conv.PATCH_BF16X3 = old
    # pairs of the stride-2 3x3 / pad 1 layer: even coordinate -> one output, odd -> two (one at the far border)
    coors = torch.tensor([[[0, 0, 0, 0], [0, 0, 1, 1], [0, 0, 2, 3], [0, 0, 511, 511], [0, 0, 511, 10], [-1, 0, 0, 0]]])
    assert first_layer_pairs(coors) == 1 + 4 + 2 + 1 + 1


def test_stub_line_fracs_and_launch_fields():
    r = _run(["--gpus", "2", "--stub-ops", "--steps", "2", "--warmup", "1", "--batch", "2", "--strong-frames", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    fr = _fracs(line)
    assert len(fr) >= 2 and all(0.0 <= v <= 1.0 for _, v in fr), fr
    assert line["config"]["launch"].startswith("eager") and line["config"]["launch_reason"]
    assert line["rooflines"]["stub_mfma"]["pipes"]["bf16x3"]["peak"] == 2500.0 / 6.0


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
