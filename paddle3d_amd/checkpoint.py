"""Paddle checkpoints into the torch-side parameter containers (reference paddle3d/apis/checkpoint.py:148-212,
paddle3d/utils/checkpoint.py:65-122: `paddle.load(path)` of a `.pdparams` file = a pickled dict name -> ndarray).

Layout rules: nn.Linear weights are [in, out] in Paddle (transposed here), BatchNorm
statistics are `_mean` / `_variance`, Conv2D [out, in, kh, kw] and Conv2DTranspose [in, out, kh, kw] are torch's
layouts, paddle.sparse Conv3D / SubmConv3D weights [kd, kh, kw, in, out] are kept as they are (sparse.py uses the
Paddle layout).  Parameter names are the reference's (the mirrors are built with the same attribute names and
Sequential indices), so no name table is needed.
"""
from __future__ import annotations

import pickle
import warnings

import numpy as np
import torch
import torch.nn as nn

__all__ = ["load_pdparams", "save_pdparams", "load_paddle_state_dict"]

_STRUCT_KEY = "StructuredToParameterName@@"


class _NumpyOnlyUnpickler(pickle.Unpickler):
    """A `.pdparams` file is a pickle of {name: ndarray}; unpickling it needs numpy's array reconstruction helpers
    and nothing else.  Everything else is refused, so a downloaded checkpoint cannot run code on load (which
    `pickle.load` -- and `paddle.load` -- would let it)."""

    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                ("numpy", "ndarray"), ("numpy", "dtype"), ("collections", "OrderedDict"),
                ("_codecs", "encode")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to load {module}.{name} from a checkpoint: a .pdparams file holds "
                                     "numpy arrays only")


def load_pdparams(path: str) -> dict:
    """What `paddle.load(path)` returns for a `.pdparams` written by `paddle.save(layer.state_dict(), path)`.
    Only numpy arrays (and the containers around them) are unpickled, see _NumpyOnlyUnpickler."""
    with open(path, "rb") as f:
        state = _NumpyOnlyUnpickler(f, encoding="latin1").load()
    if not isinstance(state, dict):
        raise RuntimeError(f"{path}: not a Paddle state dict (got {type(state).__name__})")
    state.pop(_STRUCT_KEY, None)
    return {k: np.asarray(v) for k, v in state.items()}


def save_pdparams(state: dict, path: str) -> None:
    """Write a dict name -> ndarray in the `.pdparams` wire format (protocol-2 pickle; used by the tests to make
    synthetic checkpoints with the reference's parameter names)."""
    blob = {k: np.asarray(v) for k, v in state.items()}
    blob[_STRUCT_KEY] = {k: k for k in state}
    with open(path, "wb") as f:
        pickle.dump(blob, f, protocol=2)


def _linear_weights(model: nn.Module) -> set:
    return {name + ".weight" for name, m in model.named_modules() if isinstance(m, nn.Linear)}


def load_paddle_state_dict(model: nn.Module, state, strict: bool = True) -> list:
    """Copy a Paddle state dict (or the `.pdparams` file at `state`) into the torch mirror.

    Returns the keys of `state` that found no place.  strict=True raises if any key is left over or any parameter
    / buffer of the model (besides BatchNorm's num_batches_tracked, which Paddle does not have) got no value;
    strict=False only warns."""
    if isinstance(state, (str, bytes)):
        state = load_pdparams(state)
    own = dict(model.state_dict())
    linear = _linear_weights(model)
    unplaced, seen = [], set()
    with torch.no_grad():
        for k, v in state.items():
            t = torch.as_tensor(np.asarray(v))
            name = k.replace("._mean", ".running_mean").replace("._variance", ".running_var")
            if name not in own:
                unplaced.append(k)
                continue
            if name in linear and t.dim() == 2:
                t = t.t()
            if own[name].shape != t.shape:
                unplaced.append(k)
                continue
            own[name].copy_(t)
            seen.add(name)
    unfilled = [k for k in own if k not in seen and not k.endswith("num_batches_tracked")]
    for m in model.modules():  # derived (folded / packed) weights must be rebuilt
        if hasattr(m, "_drop_cache"):
            m._drop_cache()
        if hasattr(m, "_folded"):
            m._folded = None
    if unplaced or unfilled:
        msg = (f"load_paddle_state_dict: {len(unplaced)} checkpoint keys without a place (e.g. {unplaced[:3]}), "
               f"{len(unfilled)} model entries without a value (e.g. {unfilled[:3]})")
        if strict:
            raise RuntimeError(msg)
        warnings.warn(msg)
    return unplaced
