"""The arithmetic scheme of csrc/sparse_conv_x3.hip restated on the CPU (no GPU, no kernel): an fp32 value cut into three
bf16 pieces, six of the nine piece products kept.  What the kernel's header claims is checked here on millions of random
operand pairs over the whole exponent range a network's activations and weights live in:
  * hi + mid + lo IS the fp32 value (3 x 8 significand bits = 24; |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|),
  * every piece product is exactly representable in fp32 (so the MFMA's fp32 accumulator adds exact terms),
  * the six kept products sum to the exact product within 2^-24 of it on these samples (worst case 2^-23: the dropped
    mid lo, lo mid, lo lo) -- the rounding of ONE fp32 multiplication -- where two pieces (hi, mid: 16 bits) with three
    products leave 2^-16 and a single bf16 piece 2^-8."""
import numpy as np
import torch


def _cut(x: torch.Tensor):
    hi = x.bfloat16()
    r1 = x - hi.float()          # exact in fp32
    mid = r1.bfloat16()
    r2 = r1 - mid.float()        # exact in fp32
    lo = r2.bfloat16()
    return hi, mid, lo, r1, r2


def _operands(n, seed):
    g = torch.Generator().manual_seed(seed)
    mant = torch.rand(n, generator=g) + 1.0
    expo = torch.randint(-20, 21, (n,), generator=g).float()
    sign = torch.randint(0, 2, (n,), generator=g).float() * 2 - 1
    return (sign * mant * torch.exp2(expo)).float()


def test_three_pieces_reproduce_the_fp32_value():
    x = _operands(2_000_000, 1)
    hi, mid, lo, r1, r2 = _cut(x)
    xd = x.double()
    assert torch.equal((xd - hi.double()).float().double(), xd - hi.double())      # x - hi is an fp32 value
    assert torch.equal(r1.double(), xd - hi.double())
    assert torch.equal(r2.double(), xd - hi.double() - mid.double())
    rest = (xd - hi.double() - mid.double() - lo.double()).abs() / xd.abs()
    assert float(rest.max()) == 0.0                                    # three pieces = 24 bits: exact
    assert float((mid.double().abs() / xd.abs()).max()) <= 2.0 ** -8   # half an ulp of an 8-bit significand
    assert float((lo.double().abs() / xd.abs()).max()) <= 2.0 ** -16


def test_piece_products_are_exact_in_fp32_and_six_of_them_are_an_fp32_product():
    x, w = _operands(2_000_000, 2), _operands(2_000_000, 3)
    xp, wp = _cut(x)[:3], _cut(w)[:3]
    exact = x.double() * w.double()
    kept = torch.zeros_like(exact)
    two = torch.zeros_like(exact)
    for i in range(3):
        for j in range(3):
            prod = xp[i].double() * wp[j].double()
            assert torch.equal(prod.float().double(), prod)      # 8 x 8 significand bits: exact in fp32
            if i + j <= 2:
                kept += prod
            if i + j <= 1:
                two += prod
    rel = ((kept - exact).abs() / exact.abs())
    assert float(rel.max()) < 2.0 ** -24, float(rel.max())
    # the comparison figures quoted in the kernel's header
    assert 2.0 ** -19 < float(((two - exact).abs() / exact.abs()).max()) < 2.0 ** -15
    one = xp[0].double() * wp[0].double()
    assert 2.0 ** -10 < float(((one - exact).abs() / exact.abs()).max()) < 2.0 ** -7


def test_dot_products_match_fp32_accuracy():
    """A whole gather-GEMM row (27 offsets x 128 channels = 3456 terms) by the scheme, accumulated in fp32 like the MFMA
    does, against fp64: no worse than an fp32 dot product of the same operands."""
    rng = np.random.default_rng(4)
    x = torch.from_numpy((rng.normal(size=(256, 3456)) * np.exp(rng.normal(size=(256, 3456)))).astype(np.float32))
    w = torch.from_numpy((rng.normal(size=(3456, 64)) * np.exp(rng.normal(size=(3456, 64))) / 60).astype(np.float32))
    want = x.double() @ w.double()
    xp, wp = _cut(x)[:3], _cut(w)[:3]
    acc = torch.zeros(256, 64, dtype=torch.float32)
    for i, j in ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)):   # the kernel's order: small products first
        acc = acc + (xp[i].float() @ wp[j].float())
    e_x3 = float((acc.double() - want).abs().max())
    e_32 = float(((x @ w).double() - want).abs().max())
    mag = float(want.abs().max())
    assert e_x3 <= max(2.0 * e_32, 2e-7 * mag), (e_x3, e_32, mag)
