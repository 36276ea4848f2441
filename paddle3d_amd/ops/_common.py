"""Shared helpers of the ctypes op shims (torch is only the owner of device memory and streams)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .._lib import Paddle3DAmdError, check, lib

__all__ = ["lib", "check", "Paddle3DAmdError", "ptr", "stream_ptr", "require_gpu", "host_f32", "host_i32",
           "workspace"]


def ptr(t):
    """Device (or host) pointer of a tensor / ndarray as c_void_p; None stays NULL."""
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, np.ndarray):
        return C.c_void_p(t.ctypes.data)
    return C.c_void_p(t.data_ptr())


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(t: torch.Tensor, op: str, dtype=torch.float32) -> torch.Tensor:
    """The reference ops throw PD_THROW("Unsupported device type ...") off-GPU; there is no CPU path here."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"Unsupported device type for {op} operator.")
    if t.dtype != dtype:
        raise RuntimeError(f"{op}: expected {dtype}, got {t.dtype}")
    return t.contiguous()


def host_f32(values, n=None):
    a = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(-1))
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} floats, got {a.size}")
    return a


def host_i32(values):
    return np.ascontiguousarray(np.asarray(values, dtype=np.int32).reshape(-1))


def workspace(nbytes: int, device) -> torch.Tensor:
    # torch's caching allocator hands back 512-byte aligned blocks; the library wants 256
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
