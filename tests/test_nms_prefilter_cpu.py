"""The exact early-out of the rotated-box kernels (csrc/iou3d_geom.hpp box_overlap, nms_kernels.hpp nms_cand_kernel):
a pair whose circumscribed circles, widened by 0.25 m, are disjoint is never clipped.  That is only allowed if the
reference finds NO overlap for such a pair -- not a small one: the reference code (iou3d_cpu.cpp:134-229, here its
compiled / restated form in oracle/) must return exactly 0.  Checked on the CPU for dense random boxes, evaluated in
fp32 exactly as the kernels evaluate the test."""
import numpy as np
import pytest


def _circle_test(boxes):
    f = np.float32
    b = boxes.astype(f)
    hx, hy = b[:, 3] / f(2), b[:, 4] / f(2)
    rad = np.sqrt(hx * hx + hy * hy).astype(f)
    dx = (b[:, None, 0] - b[None, :, 0]).astype(f)
    dy = (b[:, None, 1] - b[None, :, 1]).astype(f)
    r = ((rad[:, None] + rad[None, :]).astype(f) + f(0.25)).astype(f)
    return (dx * dx).astype(f) + (dy * dy).astype(f) > (r * r).astype(f)  # True: skipped by the kernels


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_pairs_outside_the_circle_test_have_exactly_zero_overlap(seed):
    from oracle import pyoracle as O

    rng = np.random.default_rng(seed)
    n = 400
    boxes = np.zeros((n, 7), np.float32)
    boxes[:, 0:2] = rng.uniform(-12, 12, (n, 2))          # crowded: many touching and nearly touching pairs
    boxes[:, 2] = rng.uniform(-1, 1, n)
    boxes[:, 3:5] = np.exp(rng.normal(0.6, 0.6, (n, 2)))  # 0.5 .. 8 m
    boxes[:, 5] = 1.5
    boxes[:, 6] = rng.uniform(-4, 4, n)
    kinds = ["port"] + (["ref"] if O.have_ref() else [])
    skipped = _circle_test(boxes)
    assert 0.2 < skipped.mean() < 0.98  # both kinds of pairs are well represented
    for kind in kinds:
        ov = O.boxes_overlap_bev(boxes, boxes, kind)
        assert ov.shape == (n, n)
        assert np.all(ov[skipped] == 0.0), (kind, float(ov[skipped].max()))
        assert (ov[~skipped] > 0).any()


def test_fixed_bubble_network_equals_the_reference_sort():
    """box_overlap sorts up to eight vertices with a fixed eight-element bubble network over +inf-padded angles
    (csrc/iou3d_geom.hpp); the reference bubble-sorts the first `cnt` entries (iou3d_cpu.cpp:201-210).  Same compare
    (a[i] > a[i+1]), same order of passes: the permutation must be identical, ties and all."""
    rng = np.random.default_rng(7)
    for _ in range(3000):
        cnt = int(rng.integers(1, 9))
        ang = rng.choice(np.round(rng.uniform(-3.2, 3.2, 5), 1), cnt).astype(np.float32)  # few distinct values: ties
        # reference: bubble sort of the first cnt entries, carrying the original index
        ra, ri = list(ang), list(range(cnt))
        for j in range(cnt - 1):
            for i in range(cnt - j - 1):
                if ra[i] > ra[i + 1]:
                    ra[i], ra[i + 1] = ra[i + 1], ra[i]
                    ri[i], ri[i + 1] = ri[i + 1], ri[i]
        # kernel: eight slots, pads = +inf
        ka = list(ang) + [np.float32(np.inf)] * (8 - cnt)
        ki = list(range(cnt)) + [-1] * (8 - cnt)
        for j in range(7):
            for i in range(7 - j):
                if ka[i] > ka[i + 1]:
                    ka[i], ka[i + 1] = ka[i + 1], ka[i]
                    ki[i], ki[i + 1] = ki[i + 1], ki[i]
        assert ki[:cnt] == ri and all(k == -1 for k in ki[cnt:])
