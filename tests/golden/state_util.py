"""Rebuild the synthetic Paddle state dicts of tests/golden/python_layers.npz (values are drawn from a seed by
paddle_shim.fill_state; the fixture stores only key -> shape)."""
import math

import numpy as np


def synth_param(key, shape, rng):  # must stay identical to paddle_shim.synth_param
    if key.endswith("_variance"):
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if key.endswith("_mean"):
        return rng.normal(0, 0.1, shape).astype(np.float32)
    if key.endswith("bias"):
        return rng.normal(0, 0.1, shape).astype(np.float32)
    if len(shape) == 1:
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    fan = float(np.prod(shape[1:])) if len(shape) > 2 else float(shape[0])
    return (rng.normal(0, 1, shape) / math.sqrt(max(fan, 1.0))).astype(np.float32)


def rebuild_state(keys_blob, shapes_blob, seed):
    keys = str(keys_blob).split("\n")
    flat = [int(v) for v in np.asarray(shapes_blob)]
    shapes, cur = [], []
    for v in flat:
        if v == -1:
            shapes.append(tuple(cur))
            cur = []
        else:
            cur.append(v)
    assert len(shapes) == len(keys)
    rng = np.random.default_rng(seed)
    return {k: synth_param(k, s, rng) for k, s in zip(keys, shapes)}  # keys are stored sorted = generation order
