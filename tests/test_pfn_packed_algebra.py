"""The algebra the packed PillarFeatureNet kernel rests on (csrc/pfn.hip), restated in NumPy fp32 and
held against the oracle's statement of PFNLayer (pillar_encoder.py:81-105):

  * a pillar's maximum over the rows of layer 2 is taken BEFORE base, BatchNorm and ReLU are applied:
        max_r relu(bn2(t_r + base)) == relu(bn2(ext_r(t_r) + base)),   ext = max for scale >= 0, min for scale < 0
    with the sign folded into the columns of W2[0:C1] so that only maxima are taken;
  * the padded rows of a pillar that is not full are one constant row (y1 = relu(shift1), t = y1 W2[0:C1]) that the
    running maxima start from.
Runs on the CPU: it pins the reformulation, the GPU tests pin the kernel.
"""
import numpy as np
import pytest

from paddle3d_amd import synth


def _fold(p):
    scale = (p["gamma"] / np.sqrt(p["var"] + np.float32(1e-3))).astype(np.float32)
    return scale, (p["beta"] - p["mean"] * scale).astype(np.float32)


def _packed_pfn(vox, npv, c4, params, pillar, rng_):
    f32 = np.float32
    w1, (s1, b1) = params[0]["weight"].astype(f32), _fold(params[0])
    w2, (s2, b2) = params[1]["weight"].astype(f32), _fold(params[1])
    c1 = w1.shape[1]
    sg = np.where(s2 < 0, f32(-1), f32(1)).astype(f32)
    w2a = (w2[:c1] * sg[None, :]).astype(f32)   # sign folded into the columns
    w2b = w2[c1:].astype(f32)
    y1pad = np.maximum(b1, 0).astype(f32)       # relu(bn1(0 * W1))
    tpad = (y1pad @ w2a).astype(f32)
    m, p, d = vox.shape
    out = np.zeros((m, w2.shape[1]), f32)
    vx, vy = f32(pillar[0]), f32(pillar[1])
    xo, yo = f32(vx / 2 + rng_[0]), f32(vy / 2 + rng_[1])
    for i in range(m):
        n = int(min(npv[i], p))
        if npv[i] <= 0:
            continue
        pts = vox[i, :n].astype(f32)
        mean = (pts[:, :3].sum(0, dtype=f32) / f32(npv[i])).astype(f32)
        centre = np.array([f32(c4[i, 3]) * vx + xo, f32(c4[i, 2]) * vy + yo], f32)
        feats = np.concatenate([pts, pts[:, :3] - mean, pts[:, :2] - centre], 1).astype(f32)
        y1 = np.maximum((feats @ w1) * s1 + b1, 0).astype(f32)   # stored points only
        t = (y1 @ w2a).astype(f32)
        m1 = y1.max(0)
        m2 = t.max(0)
        if n < p:  # the padded rows: one constant row
            m1 = np.maximum(m1, y1pad)
            m2 = np.maximum(m2, tpad)
        base = (m1 @ w2b).astype(f32)
        out[i] = np.maximum((sg * m2 + base) * s2 + b2, 0)
    return out


@pytest.mark.parametrize("seed", [0, 1])
def test_packed_pfn_algebra_matches_reference_statement(seed):
    from oracle import pyoracle as O

    rng = np.random.default_rng(seed)
    m, p, d = 300, 20, 5
    npv = rng.integers(0, p + 1, m).astype(np.int32)
    npv[:4] = [p, 1, 0, p - 1]
    vox = rng.uniform(-3, 3, (m, p, d)).astype(np.float32)
    vox *= (np.arange(p)[None, :, None] < npv[:, None, None])
    c4 = np.concatenate([np.zeros((m, 2), np.int32), rng.integers(0, 400, (m, 2)).astype(np.int32)], 1)

    def layer(i, o):
        gamma = rng.uniform(0.5, 1.5, o).astype(np.float32) * rng.choice([-1.0, 1.0], o).astype(np.float32)
        gamma[:2] = 0.0  # a zero scale: the output no longer depends on the rows at all
        return dict(weight=(rng.uniform(-1, 1, (i, o)) / np.sqrt(i)).astype(np.float32), gamma=gamma,
                    beta=rng.normal(0, 0.2, o).astype(np.float32), mean=rng.normal(0, 0.2, o).astype(np.float32),
                    var=rng.uniform(0.5, 1.5, o).astype(np.float32))

    params = [layer(d + 5, 32), layer(64, 64)]
    keep = npv > 0
    ref = O.pfn_forward_torch(vox[keep], npv[keep], c4[keep], params, synth.NUSC_PILLAR, synth.NUSC_RANGE)
    got = _packed_pfn(vox, npv, c4, params, synth.NUSC_PILLAR, synth.NUSC_RANGE)
    assert np.all(got[~keep] == 0)
    # fp32 summation orders differ (NumPy matmul vs torch, base added after the GEMM): 1e-4 is three orders of
    # magnitude below the 1e-3 contract and one above the observed 1e-5
    assert np.abs(got[keep] - ref).max() < 1e-4, np.abs(got[keep] - ref).max()


def test_max_commutes_with_monotone_epilogue():
    """fp32 add, fma and max are monotone: max over rows of relu(fma(t + base, s, b)) equals the epilogue of the
    extreme row, bit for bit, for either sign of s (what lets the kernel keep running maxima only)."""
    rng = np.random.default_rng(3)
    t = rng.normal(0, 2, (1000, 37)).astype(np.float32)
    for s in (np.float32(0.8), np.float32(-1.3), np.float32(0.0)):
        base, b = np.float32(0.37), np.float32(-0.21)
        full = np.maximum((t + base) * s + b, np.float32(0)).max(1)
        ext = t.max(1) if s >= 0 else t.min(1)
        short = np.maximum((ext + base) * s + b, np.float32(0))
        np.testing.assert_array_equal(full.view(np.uint32), short.view(np.uint32))
