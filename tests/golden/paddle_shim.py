"""A minimal `paddle` over torch (CPU) -- TEST INFRASTRUCTURE ONLY, used by make_python_golden.py to EXECUTE the
reference's own Python layers (Paddle itself is not installable here) and record golden vectors from them.

`install(reference_root)` puts fake `paddle`, `paddle.nn`, `paddle.nn.functional`, ... modules into sys.modules
(only the API surface the hot-path files touch, with Paddle's semantics: [in, out] Linear weights, `axis=` /
`perm=` keywords, BatchNorm buffers `_mean` / `_variance`, `paddle.max` returning values, `paddle.where(cond)`
returning [n, 1] index columns ...) and a skeleton `paddle3d` package whose sub-packages resolve to the
reference's source directories without running their `__init__.py` (those import the whole framework).  The
product (paddle3d_amd/) never imports this file.
"""
from __future__ import annotations

import importlib
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as TF

_DT = {"float32": torch.float32, "float64": torch.float64, "float16": torch.float16, "int32": torch.int32,
       "int64": torch.int64, "int": torch.int64, "bool": torch.bool, "uint8": torch.uint8, "int8": torch.int8}


def _dt(d):
    if d is None or isinstance(d, torch.dtype):
        return d
    if isinstance(d, str):
        return _DT[d]
    return _DT[np.dtype(d).name]


def _ax(kw):
    """axis= -> dim=, keepdim stays."""
    if "axis" in kw:
        kw["dim"] = kw.pop("axis")
    return kw


class Tensor(torch.Tensor):
    """torch.Tensor with Paddle's method spellings (results of torch functions stay of this class)."""

    def transpose(self, perm=None, *rest, **kw):
        if "perm" in kw:
            perm = kw["perm"]
        if isinstance(perm, (list, tuple)):
            return self.permute(*perm)
        return torch.Tensor.transpose(self, perm, *rest)

    def reshape(self, shape, *rest):
        if rest or not isinstance(shape, (list, tuple, torch.Size)):
            shape = (shape,) + rest
        return torch.Tensor.reshape(self, tuple(int(s) for s in shape))

    def astype(self, dtype):
        return self.to(_dt(dtype))

    cast = astype

    def tile(self, reps=None, repeat_times=None):
        return torch.Tensor.repeat(self, *(reps if reps is not None else repeat_times))

    def expand(self, shape, *rest):
        if rest or not isinstance(shape, (list, tuple, torch.Size)):
            shape = (shape,) + rest
        return torch.Tensor.expand(self, *[int(s) for s in shape])

    def sum(self, *a, **kw):
        return torch.Tensor.sum(self, *a, **_ax(kw))

    def mean(self, *a, **kw):
        return torch.Tensor.mean(self, *a, **_ax(kw))

    def max(self, *a, **kw):
        kw = _ax(kw)
        if a or "dim" in kw:
            return torch.Tensor.max(self, *a, **kw).values
        return torch.Tensor.max(self)

    def min(self, *a, **kw):
        kw = _ax(kw)
        if a or "dim" in kw:
            return torch.Tensor.min(self, *a, **kw).values
        return torch.Tensor.min(self)

    def argmax(self, *a, **kw):
        return torch.Tensor.argmax(self, *a, **_ax(kw))

    def argsort(self, axis=-1, descending=False, stable=True):
        return torch.Tensor.argsort(self, dim=axis, descending=descending, stable=True)

    def cumsum(self, *a, **kw):
        return torch.Tensor.cumsum(self, *a, **_ax(kw))

    def unsqueeze(self, axis):
        return torch.Tensor.unsqueeze(self, axis)

    def squeeze(self, axis=None):
        return torch.Tensor.squeeze(self) if axis is None else torch.Tensor.squeeze(self, axis)

    def matmul(self, y):
        return torch.matmul(self, y)

    def numpy(self):
        return self.detach().as_subclass(torch.Tensor).cpu().numpy()

    def clone(self):
        return torch.Tensor.clone(self)


def _wrap(t):
    return t.as_subclass(Tensor) if isinstance(t, torch.Tensor) and not isinstance(t, Tensor) else t


def _mk_paddle():
    p = types.ModuleType("paddle")
    p.Tensor = Tensor
    for name, d in _DT.items():
        if name != "int":
            setattr(p, name, d)

    def to_tensor(data, dtype=None, place=None, stop_gradient=True):
        if isinstance(data, torch.Tensor):
            t = data.clone()
            return _wrap(t.to(_dt(dtype)) if dtype is not None else t)
        a = np.asarray(data)
        if dtype is None and a.dtype == np.float64:
            a = a.astype(np.float32)  # paddle's default dtype
        t = torch.from_numpy(np.ascontiguousarray(a))
        return _wrap(t.to(_dt(dtype)) if dtype is not None else t)

    p.to_tensor = to_tensor
    p.zeros = lambda shape, dtype="float32": _wrap(torch.zeros(tuple(int(s) for s in shape), dtype=_dt(dtype)))
    p.ones = lambda shape, dtype="float32": _wrap(torch.ones(tuple(int(s) for s in shape), dtype=_dt(dtype)))
    p.full = lambda shape, v, dtype="float32": _wrap(torch.full(tuple(shape), v, dtype=_dt(dtype)))
    p.zeros_like = lambda x, dtype=None: _wrap(torch.zeros_like(x, dtype=_dt(dtype)))
    p.ones_like = lambda x, dtype=None: _wrap(torch.ones_like(x, dtype=_dt(dtype)))

    def arange(start=0, end=None, step=1, dtype=None):
        if end is None:
            start, end = 0, start
        if dtype is None:
            dtype = "float32" if any(isinstance(v, float) for v in (start, end, step)) else "int64"
        return _wrap(torch.arange(start, end, step, dtype=_dt(dtype)))

    p.arange = arange
    p.linspace = lambda a, b, n, dtype="float32": _wrap(torch.linspace(a, b, int(n), dtype=_dt(dtype)))
    p.concat = lambda xs, axis=0: _wrap(torch.cat(list(xs), dim=int(axis)))
    p.stack = lambda xs, axis=0: _wrap(torch.stack(list(xs), dim=axis))
    p.reshape = lambda x, shape: _wrap(x).reshape(shape)
    p.transpose = lambda x, perm: _wrap(x).permute(*perm)
    p.cast = lambda x, dtype: _wrap(x).to(_dt(dtype))
    p.sum = lambda x, axis=None, keepdim=False, dtype=None: (_wrap(torch.sum(x)) if axis is None
                                                               else _wrap(torch.sum(x, dim=axis, keepdim=keepdim)))
    p.mean = lambda x, axis=None, keepdim=False: (_wrap(torch.mean(x)) if axis is None
                                                  else _wrap(torch.mean(x, dim=axis, keepdim=keepdim)))
    p.max = lambda x, axis=None, keepdim=False: (_wrap(torch.max(x)) if axis is None
                                                 else _wrap(torch.max(x, dim=axis, keepdim=keepdim).values))
    p.min = lambda x, axis=None, keepdim=False: (_wrap(torch.min(x)) if axis is None
                                                 else _wrap(torch.min(x, dim=axis, keepdim=keepdim).values))
    # paddle.argmax returns the FIRST maximal index (pillar_encoder.py:93-95 comment); so does torch
    p.argmax = lambda x, axis=None, keepdim=False: _wrap(torch.argmax(x, dim=axis, keepdim=keepdim))
    p.argsort = lambda x, axis=-1, descending=False: _wrap(torch.argsort(x, dim=axis, descending=descending,
                                                                          stable=True))
    p.cumsum = lambda x, axis=None: _wrap(torch.cumsum(x.reshape(-1) if axis is None else x,
                                                       dim=0 if axis is None else axis))
    p.index_sample = lambda x, index: _wrap(torch.gather(x, 1, index.long()))
    p.index_select = lambda x, index, axis=0: _wrap(torch.index_select(x, axis if axis >= 0 else x.dim() + axis,
                                                                       index.long()))
    p.gather = lambda x, index, axis=0: _wrap(torch.index_select(x, axis, index.long().reshape(-1)))

    def scatter(x, index, updates, overwrite=True):
        out = x.clone()
        idx = index.long()
        if overwrite:
            out[idx] = updates  # duplicate indices: last write wins, as paddle documents
        else:
            out[idx] = 0
            out.index_add_(0, idx, updates)
        return _wrap(out)

    p.scatter = scatter

    def scatter_(x, index, updates, overwrite=True):
        x.copy_(scatter(x, index, updates, overwrite))
        return x

    p.scatter_ = scatter_

    def where(cond, x=None, y=None):
        if x is None:
            return tuple(_wrap(t.unsqueeze(-1)) for t in torch.where(cond))
        return _wrap(torch.where(cond, x, y))

    p.where = where
    p.inverse = lambda x: _wrap(torch.linalg.inv(x))
    p.matmul = lambda x, y, transpose_x=False, transpose_y=False: _wrap(torch.matmul(
        x.transpose(-1, -2) if transpose_x else x, y.transpose(-1, -2) if transpose_y else y))
    p.norm = lambda x, p=2, axis=None, keepdim=False: _wrap(torch.linalg.vector_norm(x, ord=p, dim=axis,
                                                                                      keepdim=keepdim))
    p.sqrt, p.exp, p.atan2, p.abs = (lambda x: _wrap(torch.sqrt(x))), (lambda x: _wrap(torch.exp(x))), \
        (lambda a, b: _wrap(torch.atan2(a, b))), (lambda x: _wrap(torch.abs(x)))
    p.clip = lambda x, min=None, max=None: _wrap(torch.clamp(x, min=min, max=max))
    p.floor = lambda x: _wrap(torch.floor(x))
    p.round = lambda x: _wrap(torch.round(x))
    p.meshgrid = lambda *xs: [_wrap(t) for t in torch.meshgrid(*(xs[0] if len(xs) == 1 and isinstance(xs[0], (list, tuple))
                                                                    else xs), indexing="ij")]
    p.logical_not = lambda x: _wrap(torch.logical_not(x))
    p.tile = lambda x, reps: _wrap(x.as_subclass(torch.Tensor).repeat(*[int(r) for r in reps]))
    p.split = lambda x, n, axis=0: [_wrap(t) for t in torch.split(x, x.shape[axis] // n if isinstance(n, int) else n,
                                                                  dim=axis)]

    def scatter_nd_add(x, index, updates):
        out = x.clone()
        out.index_put_(tuple(index.long().unbind(-1)), updates, accumulate=True)
        return _wrap(out)

    p.scatter_nd_add = scatter_nd_add
    p.gather_nd = lambda x, index: _wrap(x[tuple(index.long().unbind(-1))])
    p.flatten = lambda x, start_axis=0, stop_axis=-1: _wrap(torch.flatten(x, start_axis, stop_axis))
    p.squeeze = lambda x, axis=None: _wrap(x).squeeze(axis)
    p.unsqueeze = lambda x, axis: _wrap(torch.unsqueeze(x, axis))
    p.is_compiled_with_xpu = lambda: False
    p.is_compiled_with_cuda = lambda: False
    p.no_grad = torch.no_grad
    p.set_device = lambda *_a, **_k: None
    p.ParamAttr = lambda *a, **k: None
    p.in_dynamic_mode = lambda: True

    linalg = types.ModuleType("paddle.linalg")
    linalg.norm = lambda x, p=2, axis=None, keepdim=False: _wrap(torch.linalg.vector_norm(x, ord=p, dim=axis,
                                                                                           keepdim=keepdim))
    p.linalg = linalg

    autograd = types.ModuleType("paddle.autograd")

    class PyLayer:
        """Forward-only stand-in: PyLayer.apply(...) runs forward with a throw-away context."""

        @classmethod
        def apply(cls, *a, **k):
            ctx = types.SimpleNamespace(save_for_backward=lambda *t: None)
            return cls.forward(ctx, *a, **k)

    autograd.PyLayer = PyLayer
    p.autograd = autograd
    return p, linalg, autograd


def _mk_nn(p):
    nn = types.ModuleType("paddle.nn")
    F = types.ModuleType("paddle.nn.functional")
    init = types.ModuleType("paddle.nn.initializer")
    for n in ("Constant", "Uniform", "Normal", "KaimingNormal", "KaimingUniform", "XavierUniform", "XavierNormal",
              "Assign", "TruncatedNormal"):
        setattr(init, n, lambda *a, **k: None)

    F.relu = lambda x: _wrap(TF.relu(x))
    F.sigmoid = lambda x: _wrap(torch.sigmoid(x))
    F.softmax = lambda x, axis=-1: _wrap(TF.softmax(x, dim=axis))
    F.pad = lambda x, pad, mode="constant", value=0.0, data_format="NCHW": _wrap(TF.pad(x, list(pad), mode=mode,
                                                                                         value=value))
    F.max_pool2d = lambda x, k, stride=None, padding=0: _wrap(TF.max_pool2d(x, k, stride, padding))

    class Layer(torch.nn.Module):
        def __call__(self, *a, **k):
            return _wrap(super().__call__(*a, **k)) if True else None

        def sublayers(self, include_self=False):
            mods = list(self.modules())
            return mods if include_self else mods[1:]

        def named_sublayers(self, prefix="", include_self=False):
            return [(n, m) for n, m in self.named_modules(prefix=prefix) if include_self or m is not self]

        def add_sublayer(self, name, layer):
            self.add_module(str(name), layer)
            return layer

        def create_parameter(self, shape, attr=None, dtype="float32", is_bias=False, default_initializer=None):
            return torch.nn.Parameter(torch.zeros(tuple(shape), dtype=_dt(dtype)))

        def full_name(self):
            return type(self).__name__

    def _param(shape):
        return torch.nn.Parameter(torch.zeros(tuple(shape), dtype=torch.float32))

    class Linear(Layer):
        """paddle.nn.Linear: weight [in_features, out_features], y = x W + b."""

        def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
            super().__init__()
            self.weight = _param((in_features, out_features))
            self.bias = None if bias_attr is False else _param((out_features,))

        def forward(self, x):
            y = torch.matmul(x, self.weight)
            return y if self.bias is None else y + self.bias

    class _BatchNorm(Layer):
        """Inference form of paddle.nn.BatchNorm{,1D,2D}: (x - _mean) / sqrt(_variance + epsilon) * weight + bias over
        axis 1 (NC, NCL, NCHW)."""

        def __init__(self, num_features, momentum=0.9, epsilon=1e-5, weight_attr=None, bias_attr=None,
                     data_format=None, use_global_stats=None, name=None):
            super().__init__()
            self._epsilon = epsilon
            if weight_attr is False:
                self.register_buffer("weight", torch.ones(num_features))
            else:
                self.weight = torch.nn.Parameter(torch.ones(num_features))
            if bias_attr is False:
                self.register_buffer("bias", torch.zeros(num_features))
            else:
                self.bias = torch.nn.Parameter(torch.zeros(num_features))
            self.register_buffer("_mean", torch.zeros(num_features))
            self.register_buffer("_variance", torch.ones(num_features))

        def forward(self, x):
            if self.training:
                raise RuntimeError("paddle shim: BatchNorm runs in eval mode only")
            return TF.batch_norm(x, self._mean, self._variance, self.weight, self.bias, False, 0.0, self._epsilon)

    class BatchNorm1D(_BatchNorm):
        pass

    class BatchNorm2D(_BatchNorm):
        pass

    class BatchNorm(_BatchNorm):
        pass

    def _pair(v):
        return (v, v) if isinstance(v, int) else tuple(v)

    class Conv2D(Layer):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                     padding_mode="zeros", weight_attr=None, bias_attr=None, data_format="NCHW"):
            super().__init__()
            k = _pair(kernel_size)
            self.weight = _param((out_channels, in_channels // groups, *k))
            self.bias = None if bias_attr is False else _param((out_channels,))
            self._s, self._p, self._d, self._g = _pair(stride), _pair(padding), _pair(dilation), groups

        def forward(self, x):
            return TF.conv2d(x, self.weight, self.bias, self._s, self._p, self._d, self._g)

    class Conv2DTranspose(Layer):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, groups=1,
                     dilation=1, weight_attr=None, bias_attr=None, data_format="NCHW"):
            super().__init__()
            k = _pair(kernel_size)
            self.weight = _param((in_channels, out_channels // groups, *k))
            self.bias = None if bias_attr is False else _param((out_channels,))
            self._s, self._p = _pair(stride), _pair(padding)

        def forward(self, x):
            return TF.conv_transpose2d(x, self.weight, self.bias, self._s, self._p)

    class Conv1D(Layer):
        pass

    class Conv3D(Layer):
        pass

    class ReLU(Layer):
        def forward(self, x):
            return TF.relu(x)

    class Sequential(torch.nn.Sequential):
        def __call__(self, *a, **k):
            return _wrap(super().__call__(*a, **k))

        def add_sublayer(self, name, layer):
            self.add_module(str(name), layer)
            return layer

    class LayerList(torch.nn.ModuleList):
        pass

    for k, v in dict(Layer=Layer, Linear=Linear, BatchNorm1D=BatchNorm1D, BatchNorm2D=BatchNorm2D, BatchNorm=BatchNorm,
                     Conv2D=Conv2D, Conv2DTranspose=Conv2DTranspose, Conv1D=Conv1D, Conv3D=Conv3D, ReLU=ReLU,
                     Sequential=Sequential, LayerList=LayerList, Identity=torch.nn.Identity).items():
        setattr(nn, k, v)
    nn.functional = F
    nn.initializer = init

    def _other_layer(name):  # PEP 562: layer classes the hot-path files mention but never run (nn.Sigmoid, ...)
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (Layer,), {})
        setattr(nn, name, cls)
        return cls

    nn.__getattr__ = _other_layer
    F.__getattr__ = lambda name: (_ for _ in ()).throw(AttributeError(name)) if name.startswith("__") else _Anything(name)
    return nn, F, init


class _AnyAttr(types.ModuleType):
    """Stub module: every attribute is a harmless callable / decorator namespace."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        v = _Anything(name)
        setattr(self, name, v)
        return v


class _Anything:
    def __init__(self, name="stub"):
        self._name = name

    def __call__(self, *a, **k):
        if len(a) == 1 and not k and (isinstance(a[0], type) or callable(a[0])):
            return a[0]  # decorator use: @manager.X.add_component
        return _Anything(self._name)

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything(name)


def _pkg(name, path=None, permissive=False):
    m = (_AnyAttr if permissive else types.ModuleType)(name)
    m.__path__ = [path] if path else []
    m.__package__ = name
    sys.modules[name] = m
    return m


def install(reference_root="/root/reference"):
    """Fake paddle + skeleton paddle3d into sys.modules.  Returns the fake `paddle` module."""
    p, linalg, autograd = _mk_paddle()
    nn, F, init = _mk_nn(p)
    p.nn = nn
    sys.modules.update({"paddle": p, "paddle.nn": nn, "paddle.nn.functional": F, "paddle.nn.initializer": init,
                        "paddle.linalg": linalg, "paddle.autograd": autograd})
    for extra in ("paddle.distributed", "paddle.distributed.fleet", "paddle.distributed.fleet.utils", "paddle.vision",
                  "paddle.vision.models", "paddle.vision.models.resnet", "paddle.static", "paddle.jit", "paddle.utils", "paddle.utils.cpp_extension"):
        sys.modules[extra] = _AnyAttr(extra)
    # paddle.static.nn.cond in dynamic mode: evaluate the predicate, run one branch (pointpillars_head.py:101-131)
    static_nn = types.ModuleType("paddle.static.nn")
    static_nn.cond = lambda pred, true_fn=None, false_fn=None: true_fn() if bool(pred) else false_fn()
    sys.modules["paddle.static"].nn = static_nn
    sys.modules["paddle.static.nn"] = static_nn
    p.static = sys.modules["paddle.static"]
    root = os.path.join(reference_root, "paddle3d")
    _pkg("paddle3d", root)
    # sub-packages that resolve to the reference's source directories; their __init__.py never runs
    for sub in ("models", "models/voxel_encoders", "models/middle_encoders", "models/backbones", "models/necks",
                "models/layers", "models/transformers", "models/detection", "models/detection/centerpoint",
                "models/detection/bevfusion", "models/detection/pointpillars", "geometries", "utils"):
        _pkg("paddle3d." + sub.replace("/", "."), os.path.join(root, sub), permissive=sub == "models/layers")
    # framework services the layer files import but the forward paths do not need
    for stub in ("paddle3d.apis", "paddle3d.apis.manager", "paddle3d.ops", "paddle3d.models.losses",
                 "paddle3d.utils.logger", "paddle3d.geometries.bbox", "paddle3d.models.layers.param_init",
                 "paddle3d.sample", "paddle3d.utils.checkpoint"):
        sys.modules[stub] = _AnyAttr(stub)
    sys.modules["paddle3d.apis"].manager = _Anything("manager")
    sys.modules["paddle3d.utils.logger"].logger = _Anything("logger")
    sys.modules["paddle3d.models"].layers = sys.modules["paddle3d.models.layers"]
    for name in ("BBoxes3D", "CoordMode"):  # result containers the SSD head imports and only _parse_result_to_sample uses
        setattr(sys.modules["paddle3d.geometries"], name, _Anything(name))
    return p


def load(modname):
    """Import a reference module by its dotted name (after install())."""
    return importlib.import_module(modname)


def exec_lines(path, ranges, namespace):
    """Execute line ranges [(first, last), ...] (1-based, inclusive) of a reference file in `namespace` -- for files
    whose import list drags in the whole framework.  The text is read at run time and never copied into the repo."""
    lines = open(path).read().split("\n")
    import textwrap

    for first, last in ranges:
        src = textwrap.dedent("\n".join(lines[first - 1:last]))
        exec(compile(src, f"{path}:{first}-{last}", "exec"), namespace)
    return namespace


def tensor(a, dtype=None):
    return sys.modules["paddle"].to_tensor(np.asarray(a), dtype=dtype)


def fill_state(module, seed):
    """Deterministic parameters for a shim module: every state-dict entry (sorted by key) drawn from one seeded RNG
    in shape-dependent ranges (variances positive).  The same function on the key -> shape list reproduces them."""
    rng = np.random.default_rng(seed)
    state = {}
    for k in sorted(module.state_dict()):
        t = module.state_dict()[k]
        state[k] = synth_param(k, tuple(t.shape), rng)
    with torch.no_grad():
        for k, v in module.state_dict().items():
            v.copy_(torch.from_numpy(state[k]))
    return {k: list(v.shape) for k, v in state.items()}


def synth_param(key, shape, rng):
    if key.endswith("_variance"):
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if key.endswith("_mean"):
        return rng.normal(0, 0.1, shape).astype(np.float32)
    if key.endswith("bias"):
        return rng.normal(0, 0.1, shape).astype(np.float32)
    if len(shape) == 1:  # BatchNorm scale
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    fan = float(np.prod(shape[1:])) if len(shape) > 2 else float(shape[0])
    return (rng.normal(0, 1, shape) / math.sqrt(max(fan, 1.0))).astype(np.float32)
