"""the multi-sweep merge against the reference's own `LoadPointCloud.__call__`
(paddle3d/transforms/reader.py:91-170) EXECUTED on synthetic `.bin` sweeps (tests/golden/make_reader_golden.py ->
python_reader.npz).  CPU: the oracle's restatement, bit for bit; GPU: `pd3_merge_sweeps`, bit for bit."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ["nusc10", "padded", "allpad", "raw5", "single"]


@pytest.fixture(scope="module")
def rg():
    return np.load(os.path.join(HERE, "golden", "python_reader.npz"))


def replay(rg, name):
    """The call's inputs in the order the reference visited the sweeps (its np.random.choice permutation)."""
    dim, use_dim, use_time_lag, _ = (int(v) for v in rg[f"{name}.cfg"])
    src, perm = rg[f"{name}.src"], rg[f"{name}.perm"]
    sweeps = [rg[f"{name}.frame{int(src[i])}"] for i in perm]
    mats = [rg[f"{name}.mats"][i] if rg[f"{name}.has_mat"][i] else None for i in perm]
    lags = [float(rg[f"{name}.lags"][i]) for i in perm]
    return dict(key=rg[f"{name}.frame0"], sweeps=sweeps, mats=mats, lags=lags, use_dim=None if use_dim < 0 else use_dim,
                use_time_lag=bool(use_time_lag), radius=float(rg[f"{name}.radius"][0]), out=rg[f"{name}.out"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_merge_sweeps_is_the_reference(rg, name):
    from oracle import pyoracle as O

    c = replay(rg, name)
    got = O.merge_sweeps_numpy(c["key"], c["sweeps"], c["mats"], c["lags"], use_dim=c["use_dim"],
                               use_time_lag=c["use_time_lag"], radius=c["radius"])
    assert got.dtype == np.float32 and got.shape == c["out"].shape
    np.testing.assert_array_equal(got.view(np.uint32), c["out"].view(np.uint32))


def test_golden_covers_the_edge_cases(rg):
    """The fixture really holds what it claims: ego returns were dropped from the sweeps but not from the key frame,
    |x| == radius survives, the padded key frame went through untransformed, lags are float32(lag)."""
    c = replay(rg, "allpad")
    k = c["key"]
    n0 = len(k)
    out = c["out"]
    np.testing.assert_array_equal(out[:n0, :4], k[:, :4])
    assert (out[:n0, 4] == 0).all()
    inside = (np.abs(k[:, 0]) < 1) & (np.abs(k[:, 1]) < 1)
    assert inside.sum() > 0 and len(out) == n0 + 9 * (n0 - inside.sum())
    np.testing.assert_array_equal(out[n0:n0 + (n0 - inside.sum()), :3].view(np.uint32), k[~inside, :3].view(np.uint32))
    c = replay(rg, "nusc10")
    lag_col = c["out"][len(c["key"]):, 4]
    assert set(np.unique(lag_col).tolist()) == {float(np.float32(v)) for v in c["lags"]}
    edge = np.concatenate([f[(np.abs(f[:, 0]) == 1) | (np.abs(f[:, 1]) == 1)] for f in c["sweeps"]])
    assert len(edge) >= 8


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_merge_sweeps_is_the_reference(rg, name):
    import torch

    from paddle3d_amd.ops import sweeps as sw

    c = replay(rg, name)
    dev = torch.device("cuda", 0)
    got = sw.merge_sweeps(torch.from_numpy(c["key"]).to(dev), [torch.from_numpy(s).to(dev) for s in c["sweeps"]],
                          c["mats"] if c["sweeps"] else None, c["lags"] if c["sweeps"] else None, use_dim=c["use_dim"],
                          use_time_lag=c["use_time_lag"], sweep_remove_radius=c["radius"]).cpu().numpy()
    assert got.shape == c["out"].shape
    np.testing.assert_array_equal(got.view(np.uint32), c["out"].view(np.uint32))
