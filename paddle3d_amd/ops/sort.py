"""stable_argsort: the library's radix sort behind the two argsorts of the reference's host glue
(rotate_nms_pcdet, layer_libs.py:230-236; the ranks_feat re-sort of QuickCumsumCuda.backward,
bevdet_transformer.py:60-68).  Equal keys keep their input order."""
from __future__ import annotations

import torch

from ._common import check, lib, ptr, require_gpu, stream_ptr, workspace

__all__ = ["stable_argsort"]


def stable_argsort(keys: torch.Tensor, descending: bool = False, max_key: int | None = None) -> torch.Tensor:
    """keys [n] on the GPU: float32 with descending=True (scores), or int32 / int64 >= 0 ascending (ranks; int64 is
    narrowed; values outside [0, max_key] raise).  Returns int64 indices like torch.argsort(stable=True)."""
    if keys.dim() != 1:
        raise RuntimeError("stable_argsort: keys must be one-dimensional")
    n = int(keys.shape[0])
    dev = keys.device
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64, device=dev)
    if descending:
        k = require_gpu(keys, "stable_argsort", torch.float32)
        mode, mk = 1, 0xFFFFFFFF
    else:
        if keys.dtype not in (torch.int32, torch.int64):
            raise RuntimeError("stable_argsort: ascending keys must be int32 / int64")
        mode, mk = 0, (0x7FFFFFFF if max_key is None else int(max_key))
        if not 0 <= mk <= 0x7FFFFFFF:
            raise RuntimeError("stable_argsort: max_key must lie in [0, 2^31)")
        if not keys.is_cuda:
            raise RuntimeError("Unsupported device type for stable_argsort operator.")
        # a key outside [0, max_key] would be narrowed / sorted on too few digits without any error: one read-back of
        # (min, max) -- this mode serves BevPoolV2.backward (training glue), not an inference step
        lo, hi = (int(v) for v in torch.stack([keys.amin(), keys.amax()]).tolist())
        if lo < 0 or hi > mk:
            raise RuntimeError(f"stable_argsort: ascending keys must lie in [0, {mk}] (found [{lo}, {hi}])")
        k = require_gpu(keys.to(torch.int32), "stable_argsort", torch.int32)
    order = torch.empty((n,), dtype=torch.int32, device=dev)
    L = lib()
    ws = workspace(L.pd3_stable_argsort_workspace(n, mk), dev)
    check(L.pd3_stable_argsort(ptr(k), n, mode, mk, ptr(order), ptr(ws), ws.numel(), stream_ptr(dev)), "stable_argsort")
    return order.long()
