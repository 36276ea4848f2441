"""merge_sweeps: LoadPointCloud's multi-sweep merge (paddle3d/transforms/reader.py:118-164) on the device,
feeding hard_voxelize directly.  merge_sweeps(key_frame, sweeps, ref_from_curr, time_lags, ...) -> [N, D(+1)].

Bit-equal to the reference's own `LoadPointCloud.__call__` executed on the same sweeps in the same order
(tests/golden/make_reader_golden.py -> python_reader.npz; the reference draws the order with
`np.random.choice`, the caller passes the sweeps in the order it wants).  `use_dim` is the reference's int form
(the first `use_dim` columns, what every config on the path uses); its list form is not taken."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._common import check, lib, ptr, require_gpu, stream_ptr, workspace

__all__ = ["merge_sweeps"]


def merge_sweeps(key_frame: torch.Tensor, sweeps, ref_from_curr=None, time_lags=None, use_dim=None,
                 use_time_lag: bool = True, sweep_remove_radius: float = 1.0, return_count: bool = False):
    """key_frame [n0, dim], sweeps: list of [n_i, dim] GPU tensors; ref_from_curr: list of 4x4 float64 arrays, an
    entry may be None like `sweep.meta.ref_from_curr` (or None for no transform at all); time_lags: list of floats.
    Returns the merged cloud sliced to its row count (one host sync), or (padded cloud, device count) with
    return_count=True."""
    frames = [require_gpu(key_frame, "merge_sweeps")] + [require_gpu(s, "merge_sweeps") for s in sweeps]
    dim_in = int(frames[0].shape[1])
    pts = torch.cat(frames, 0).contiguous()
    offs = np.zeros(len(frames) + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    if isinstance(use_dim, (list, tuple, range)):
        raise TypeError("merge_sweeps: use_dim is the number of leading columns (int), not a list")
    use_dim = dim_in if use_dim is None else int(use_dim)
    mats = has = None
    if ref_from_curr is not None:
        mats = np.zeros((len(frames), 16), np.float64)
        mats[0] = np.eye(4).reshape(-1)
        has = np.zeros(len(frames), np.int32)
        for i, m in enumerate(ref_from_curr):
            if m is not None:
                mats[i + 1] = np.asarray(m, np.float64).reshape(-1)
                has[i + 1] = 1
    lags = None
    if time_lags is not None:
        lags = np.zeros(len(frames), np.float32)
        lags[1:] = np.asarray(time_lags, np.float32)
    od = use_dim + (1 if use_time_lag else 0)
    out = torch.empty((pts.shape[0], od), dtype=torch.float32, device=pts.device)
    n_out = torch.empty((1,), dtype=torch.int32, device=pts.device)
    L = lib()
    ws = workspace(L.pd3_merge_sweeps_workspace(pts.shape[0]), pts.device)
    check(L.pd3_merge_sweeps(ptr(pts), ptr(offs), len(frames), dim_in, use_dim, ptr(mats), ptr(has), ptr(lags),
                             int(bool(use_time_lag)), C.c_float(sweep_remove_radius), ptr(out), ptr(n_out),
                             ptr(ws), ws.numel(), stream_ptr(pts.device)), "merge_sweeps")
    if return_count:
        return out, n_out
    return out[: int(n_out.item())]
