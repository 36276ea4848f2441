"""Golden vectors for the multi-sweep merge, from the reference's own (NumPy) code: `LoadPointCloud.__init__` /
`__call__` (paddle3d/transforms/reader.py:91-170) EXECUTED here on synthetic `.bin` sweeps, with the reference's own
`Sample` / `SampleMeta` (paddle3d/sample.py) carrying `time_lag` / `ref_from_curr` the way
`NuscenesPCDataset.get_sweeps` / `__getitem__` fill them (datasets/nuscenes/nuscenes_pointcloud_det.py:80-155:
`ref_from_curr` = a product of four float64 4x4 transforms, `time_lag` a Python float; a key frame without enough
predecessors is padded with copies of the last sweep, the first pad being the key frame itself with `time_lag = 0`
and `ref_from_curr = None`).

    python tests/golden/make_reader_golden.py        # needs /root/reference; writes python_reader.npz

reader.py's import list (cv2, paddle, PIL, the dataset packages) is not importable here, so the class is exec'd by
line range into a namespace that holds what its body uses: `np`, `Sample`, `PointCloud` (identity: the reference's
PointCloud is an ndarray view of the same bytes), `manager.TRANSFORMS.add_component` and `TransformABC` as no-ops.
The sweep order is the reference's `np.random.choice(len, len, replace=False)`; the generator seeds NumPy's global
state before every call and stores the permutation it produced.
"""
import functools
import importlib.util
import os
import sys
import tempfile
import types
from typing import List, Union

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_pointcloud_class():
    sample = _load("paddle3d_sample_for_reader_golden", "paddle3d/sample.py")
    with open(os.path.join(REF, "paddle3d/transforms/reader.py")) as f:
        lines = f.readlines()
    first = next(i for i, l in enumerate(lines) if l.startswith("class LoadPointCloud"))
    last = next(i for i in range(first + 1, len(lines)) if lines[i].startswith("@manager") or lines[i].startswith("class "))
    assert (first + 1, last) == (91, 170), (first + 1, last)  # reader.py:91-170, the class the docstrings cite
    ns = dict(np=np, Union=Union, List=List, Sample=sample.Sample, PointCloud=lambda d: d, TransformABC=object)
    exec(compile("".join(lines[first:last]), "reader.py:LoadPointCloud", "exec"), ns)
    return ns["LoadPointCloud"], sample


def rigid(rng, max_angle, max_shift):
    a, b, c = rng.uniform(-max_angle, max_angle, 3)
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    rz = np.array([[ca, -sa, 0], [sa, ca, 0], [0, 0, 1.0]])
    ry = np.array([[cb, 0, sb], [0, 1.0, 0], [-sb, 0, cb]])
    rx = np.array([[1.0, 0, 0], [0, cc, -sc], [0, sc, cc]])
    m = np.eye(4)
    m[:3, :3] = rz @ ry @ rx
    m[:3, 3] = rng.uniform(-max_shift, max_shift, 3)
    return m


def ref_from_curr(rng):
    """nuscenes_pointcloud_det.py:106-129: ref_from_car . car_from_global . global_from_car . car_from_current."""
    return functools.reduce(np.dot, [rigid(rng, 0.05, 1.5), rigid(rng, 3.1, 900.0), rigid(rng, 3.1, 900.0),
                                     rigid(rng, 0.05, 1.5)])


def frame(rng, n, dim):
    f = rng.uniform(-60, 60, (n, dim)).astype(np.float32)
    f[:, 2] = rng.uniform(-5, 3, n)
    k = max(4, n // 40)
    f[rng.choice(n, k, replace=False), :2] = rng.uniform(-0.999, 0.999, (k, 2)).astype(np.float32)  # ego returns
    f[rng.choice(n, 4, replace=False), 0] = np.float32(1.0)     # |x| == radius: not "< radius", kept
    f[rng.choice(n, 4, replace=False), 1] = np.float32(-1.0)
    return f


def case(cls, sample_mod, tmp, name, seed, dim, use_dim, use_time_lag, radius, n_sweeps, pads):
    rng = np.random.default_rng(seed)
    frames = [frame(rng, int(rng.integers(500, 900)), dim) for _ in range(1 + n_sweeps - pads)]
    paths = []
    for i, f in enumerate(frames):
        p = os.path.join(tmp, f"{name}_{i}.bin")
        f.tofile(p)
        paths.append(p)
    s = sample_mod.Sample(path=paths[0], modality="lidar")
    mats, lags, src = [], [], []
    t0 = 1533151603.547590
    for i in range(1, len(frames)):
        m = ref_from_curr(rng)
        lag = t0 - 1e-6 * int((t0 - 0.05 * i + rng.uniform(-0.002, 0.002)) * 1e6)  # a Python float, like the dataset's
        mats.append(m)
        lags.append(lag)
        src.append(i)
    for _ in range(pads):  # get_sweeps' padding: the key frame itself first (lag 0, no transform), then repeats
        if not src:
            mats.append(None)
            lags.append(0)
            src.append(0)
        else:
            mats.append(mats[-1])
            lags.append(lags[-1])
            src.append(src[-1])
    for m, lag, i in zip(mats, lags, src):
        sw = sample_mod.Sample(path=paths[i], modality="lidar")
        sw.meta.time_lag = lag
        sw.meta.ref_from_curr = m
        s.sweeps.append(sw)
    np.random.seed(seed)
    perm = np.random.choice(len(s.sweeps), len(s.sweeps), replace=False) if s.sweeps else np.zeros(0, np.int64)
    np.random.seed(seed)
    out = cls(dim=dim, use_dim=use_dim, use_time_lag=use_time_lag, sweep_remove_radius=radius)(s).data
    assert out.dtype == np.float32
    d = {f"{name}.frame{i}": f for i, f in enumerate(frames)}
    d[f"{name}.src"] = np.asarray(src, np.int64)
    d[f"{name}.has_mat"] = np.asarray([m is not None for m in mats], np.bool_)
    d[f"{name}.mats"] = np.stack([m if m is not None else np.zeros((4, 4)) for m in mats]) if mats else np.zeros((0, 4, 4))
    d[f"{name}.lags"] = np.asarray(lags, np.float64)
    d[f"{name}.perm"] = perm.astype(np.int64)
    d[f"{name}.cfg"] = np.asarray([dim, -1 if use_dim is None else use_dim, int(use_time_lag), n_sweeps], np.int64)
    d[f"{name}.radius"] = np.asarray([radius], np.float64)
    d[f"{name}.out"] = out
    return d


def main():
    cls, sample_mod = load_pointcloud_class()
    vec = {}
    with tempfile.TemporaryDirectory() as tmp:
        # the CenterPoint nuScenes pipeline: dim 5, use_dim 4, time lag appended, 9 sweeps
        # (configs/centerpoint/centerpoint_pillars_02voxel_nuscenes_10sweep.yml: LoadPointCloud dim 5 use_dim 4
        #  use_time_lag True sweep_remove_radius 1)
        vec.update(case(cls, sample_mod, tmp, "nusc10", 11, 5, 4, True, 1, 9, 0))
        # a scene start: two real predecessors, seven pads
        vec.update(case(cls, sample_mod, tmp, "padded", 12, 5, 4, True, 1, 9, 7))
        # the very first frame of a scene: nine pads = the key frame itself with lag 0 and no transform
        vec.update(case(cls, sample_mod, tmp, "allpad", 13, 5, 4, True, 1, 9, 9))
        # all five columns, no time lag, a float radius
        vec.update(case(cls, sample_mod, tmp, "raw5", 14, 5, None, False, 1.5, 4, 0))
        # no sweeps at all (KITTI style, dim 4)
        vec.update(case(cls, sample_mod, tmp, "single", 15, 4, None, False, 1, 0, 0))
    path = os.path.join(HERE, "python_reader.npz")
    np.savez_compressed(path, **vec)
    print(path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in vec.items() if k.endswith(".out")})


if __name__ == "__main__":
    main()
