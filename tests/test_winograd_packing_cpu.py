"""Host-side packings of the Winograd F(4x4, 3x3) kernels (paddle3d_amd/ops/conv.py) against their written layouts:
the lane order of the ping-pong kernel (include/paddle3d_amd.h: conv3x3_winograd43_pp_bias_relu) holds the same U values
as the block order of the packed kernel, and U itself is G g G^T of the reference layer's folded weight
(second_backbone.py:72-120 / center_head.py:43-220 after BatchNorm folding)."""
import numpy as np
import torch

from paddle3d_amd.ops import conv

G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
              [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)


def test_lane_packing_matches_written_layout():
    g = torch.Generator().manual_seed(5)
    cout, cin = 128, 24
    w = torch.randn(cout, cin, 3, 3, generator=g)
    ul = conv.pack_winograd43_lane_weight(w)
    assert tuple(ul.shape) == (cout // 64, cin // 8, 2, 4, 9, 4, 16, 4) and ul.is_contiguous()
    flat = ul.reshape(cout // 64, cin // 8, 2, 4, 9, 64, 4).numpy()  # [ct][slot][trip][cb][q][lane][j]
    u = np.einsum("ij,ocjk,lk->ocil", G, w.double().numpy(), G).astype(np.float32).reshape(cout, cin, 36)
    rng = np.random.default_rng(1)
    for _ in range(500):
        ct, slot, trip, cb, q, lane, j = (int(rng.integers(n)) for n in (cout // 64, cin // 8, 2, 4, 9, 64, 4))
        co = 64 * ct + 16 * cb + (lane & 15)
        ci = 8 * slot + 4 * trip + (lane >> 4)
        # (one rounding of the fp64 product apart: torch and NumPy contract the two 6x3 factors in different orders)
        assert abs(flat[ct, slot, trip, cb, q, lane, j] - u[co, ci, 4 * q + j]) <= 2e-7 * max(1.0, abs(u[co, ci, 4 * q + j]))


def test_lane_and_block_packings_hold_the_same_values():
    g = torch.Generator().manual_seed(6)
    cout, cin = 64, 16
    w = torch.randn(cout, cin, 3, 3, generator=g)
    ub = conv.pack_winograd43_weight(w, 64).numpy()   # [cout/64][cin/4][4 blocks][4 ci][16 co][36]
    ul = conv.pack_winograd43_lane_weight(w).reshape(1, cin // 8, 2, 4, 9, 4, 16, 4).numpy()
    for slot in range(cin // 8):
        for trip in range(2):
            for cb in range(4):
                blk = ub[0, 2 * slot + trip, cb]                       # [ci4][co16][36]
                lane = ul[0, slot, trip, cb]                           # [q][ci4][co16][j]
                assert np.array_equal(lane.transpose(1, 2, 0, 3).reshape(4, 16, 36), blk)


def test_channel_tiles_are_slices_of_the_first_dimension():
    # the head runs its first stage in slices of channel tiles (centerpoint.py, CenterHead.forward)
    g = torch.Generator().manual_seed(7)
    w = torch.randn(192, 8, 3, 3, generator=g)
    ul = conv.pack_winograd43_lane_weight(w)
    for t in range(3):
        assert torch.equal(ul[t:t + 1], conv.pack_winograd43_lane_weight(w[64 * t:64 * t + 64]))
