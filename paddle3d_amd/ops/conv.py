"""conv3x3_bias_relu: hand-written fp32-MFMA 3x3/stride-1 convolution for the dense BEV graph
(SecondBackbone / CenterHead convolutions, BatchNorm folded)."""
from __future__ import annotations

import torch

from ._common import check, lib, ptr, require_gpu, stream_ptr

__all__ = ["pack_conv3x3_weight", "conv3x3_bias_relu", "supported"]


def supported(cin: int, cout: int, h: int, w: int) -> bool:
    return cin % 8 == 0 and cout % 64 == 0 and (w % 128 == 0 or (w % 64 == 0 and h % 2 == 0) or
                                                 (w % 32 == 0 and h % 4 == 0))


def pack_conv3x3_weight(weight: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> [Cout/64][Cin/8][4 channel pairs][9 taps][2 channels of the pair][64]
    (the order the kernel stages into LDS: one MFMA K step = one tap of one channel pair)."""
    cout, cin = weight.shape[:2]
    assert weight.shape[2:] == (3, 3) and cout % 64 == 0 and cin % 8 == 0
    w = weight.reshape(cout // 64, 64, cin // 8, 4, 2, 9).permute(0, 2, 3, 5, 4, 1)
    return w.reshape(cout // 64, cin // 8, 72, 64).contiguous()


def conv3x3_bias_relu(x: torch.Tensor, w_packed: torch.Tensor, bias, cout: int, relu: bool = True,
                      out: torch.Tensor | None = None) -> torch.Tensor:
    xx = require_gpu(x, "conv3x3_bias_relu")
    n, cin, h, w = xx.shape
    if out is None:
        out = torch.empty((n, cout, h, w), dtype=torch.float32, device=xx.device)
    check(lib().pd3_conv3x3_bias_relu(ptr(xx), ptr(w_packed), ptr(bias), n, cin, cout, h, w, int(bool(relu)),
                                      ptr(out), stream_ptr(xx.device)), "conv3x3_bias_relu")
    return out
