"""`sparse_conv3d`: the ops behind paddle.sparse.nn.SubmConv3D / Conv3D as the reference's CenterPoint-Voxel
middle encoder uses them (paddle3d/models/middle_encoders/sparse_resnet.py:31-59, :115-206).  The reference
has no `paddle3d.ops.sparse_conv3d` module -- the arithmetic is Paddle core -- so the signatures here are ours:

  indices(coords, batch, spatial_shape, kernel_size, stride, padding, subm) -> SparseIndices
  plan(coords, batch, spatial_shape, specs) -> SparsePlan   (all index sets of an encoder, ONE host sync)
  plan(..., caps=plan_caps(earlier plan)) -> SparsePlan     (the same WITHOUT a host sync: remembered capacities,
                                                             device row counts, one overflow word to check later)
  features(in_feats, idx, weight, bias=None, scale=None, shift=None, residual=None, relu=False) -> out_feats
  to_dense(feats, coords, batch, spatial_shape) -> [B, C*D, H, W]
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from ._common import check, host_i32, lib, ptr, require_gpu, stream_ptr, workspace

__all__ = ["SparseIndices", "ConvSpec", "SparsePlan", "indices", "plan", "plan_caps", "features", "features_f16",
           "pack_weight_f16", "f16_supported", "SPLIT_BF16", "bf16x3_supported", "bf16x3_pays", "pack_weight_bf16x3",
           "features_bf16x3", "gather_gemm_f16", "tile_order", "to_dense", "out_spatial_shape"]


@dataclass
class SparseIndices:
    out_coords: torch.Tensor   # [n_out, 4] int32 (b, z, y, x)
    nbr: torch.Tensor          # [n_out, K] int32, input row or -1
    n_out: int                 # rows of the arrays (a capacity when n_out_dev is set)
    out_shape: tuple           # (D, H, W)
    kernel_volume: int
    order: torch.Tensor = None      # tile order of the rows (pd3_sparse_tile_order): a scheduling hint for `features`
    n_out_dev: torch.Tensor = None  # [1] int32 on the device: the real row count of a plan made without a host sync


# tile order on / off (the GPU tests run both: the results are the same bytes)
TILE_ORDER = True


def out_spatial_shape(spatial_shape, kernel_size, stride, padding):
    return tuple((s + 2 * p - k) // st + 1 for s, k, st, p in zip(spatial_shape, kernel_size, stride, padding))


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(int(x) for x in v)


def indices(coords: torch.Tensor, batch: int, spatial_shape, kernel_size, stride=1, padding=0,
            subm: bool = False) -> SparseIndices:
    c = require_gpu(coords, "sparse_conv3d", torch.int32)
    if c.dim() != 2 or c.shape[1] != 4:
        raise RuntimeError("sparse_conv3d: coords must be [N, 4] int32 (batch, z, y, x)")
    ks, st, pd = _triple(kernel_size), _triple(stride), _triple(padding)
    n_in = c.shape[0]
    kvol = ks[0] * ks[1] * ks[2]
    dev = c.device
    if n_in == 0:  # an empty voxel set gives an empty sparse tensor (the reference's layers accept nnz == 0)
        shape = tuple(spatial_shape) if subm else out_spatial_shape(spatial_shape, ks, st, pd)
        return SparseIndices(torch.empty((0, 4), dtype=torch.int32, device=dev),
                             torch.empty((0, kvol), dtype=torch.int32, device=dev), 0, shape, kvol)
    if subm:
        out_shape = tuple(spatial_shape)
        cap = n_in
    else:
        out_shape = out_spatial_shape(spatial_shape, ks, st, pd)
        per_in = math.prod(-(-k // s) for k, s in zip(ks, st))
        cap = int(min(n_in * per_in, batch * math.prod(out_shape)))
    out_coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    nbr = torch.empty((cap, kvol), dtype=torch.int32, device=dev)
    n_out = torch.empty((1,), dtype=torch.int32, device=dev)
    L = lib()
    hk, hs, hp, hsh = host_i32(ks), host_i32(st), host_i32(pd), host_i32(spatial_shape)
    ws_bytes = L.pd3_sparse_conv3d_workspace(n_in, ptr(hk), int(subm), cap)
    if ws_bytes == 0:
        raise RuntimeError("sparse_conv3d: invalid sizes")
    ws = workspace(ws_bytes, dev)
    check(L.pd3_sparse_conv3d_indices(ptr(c), n_in, batch, ptr(hsh), ptr(hk), ptr(hs), ptr(hp), int(subm),
                                      ptr(out_coords), ptr(nbr), ptr(n_out), cap, ptr(ws), ws.numel(),
                                      stream_ptr(dev)), "sparse_conv3d_indices")
    n = n_in if subm else int(n_out.item())  # one host sync per strided convolution (v1)
    return SparseIndices(out_coords[:n], nbr[:n], n, out_shape, kvol)


@dataclass(frozen=True)
class ConvSpec:
    """One convolution of a chain: kernel / stride / padding triples, submanifold or regular; convolutions with
    the same `key` (and the same input set) share one rulebook, like the reference's `key=` hint."""
    kernel_size: tuple
    stride: tuple = (1, 1, 1)
    padding: tuple = (0, 0, 0)
    subm: bool = False
    key: object = None
    order: bool = True   # build the tile order for this convolution's rulebook (it pays from 32 output channels on)


@dataclass
class SparsePlan:
    order: torch.Tensor       # [n_in] int64: input row of the i-th row of the (sorted) first index set
    coords: torch.Tensor      # [n_in, 4] int32 coordinates of the first index set (raster order)
    n_in: int
    indices: list             # one SparseIndices per ConvSpec (shared objects where rulebooks are shared)
    counts: list = None       # rows of every index set (host ints; None for a plan made without a host sync)
    n_in_dev: torch.Tensor = None   # [1] int32: rows of the first set (a plan made without a host sync)
    overflow: torch.Tensor = None   # [] bool on the device: some set reached its capacity (results truncated)


def plan_caps(pl: SparsePlan, margin: float = 1.25, quantum: int = 8192):
    """Capacities for later plans of the same encoder and batch size, from a plan made WITH the host sync: every
    set's row count times `margin`, rounded up to `quantum` rows."""
    return [int(-(-int(n * margin + 1) // quantum) * quantum) for n in pl.counts]


def plan(coords: torch.Tensor, batch: int, spatial_shape, specs, caps=None) -> SparsePlan:
    """Index sets and rulebooks of a chain of sparse convolutions (an encoder).  Every index set is a sorted key
    array whose length stays on the device while the chain is enqueued; the lengths are read back ONCE, then the
    rulebooks are built at their exact sizes (LDS-staged hash lookups).  `coords` may contain padding rows
    (batch < 0), e.g. the voxelizer's fixed-shape output; row i of the first set is input row order[i].

    caps (plan_caps of an earlier plan): NO host sync.  Rulebooks and feature rows are allocated at the remembered
    capacities, every kernel reads its row count from device memory, and `overflow` says (on the device) whether a
    set filled its capacity -- the caller reads it wherever it synchronises anyway (where detections are read) and
    replans with the sync if it is set."""
    c = require_gpu(coords, "sparse_conv3d", torch.int32)
    if c.dim() != 2 or c.shape[1] != 4:
        raise RuntimeError("sparse_conv3d: coords must be [N, 4] int32 (batch, z, y, x)")
    specs = [sp if isinstance(sp, ConvSpec) else ConvSpec(*sp) for sp in specs]
    dev, n0 = c.device, int(c.shape[0])
    L = lib()
    n_sets = 1 + sum(1 for sp in specs if not sp.subm)
    counts = torch.zeros((n_sets,), dtype=torch.int32, device=dev)
    order = torch.empty((max(n0, 1),), dtype=torch.int32, device=dev)
    sets = [dict(keys=torch.empty((max(n0, 1),), dtype=torch.int32, device=dev), cap=n0,
                 shape=tuple(int(v) for v in spatial_shape))]
    if n0 > 0:
        ws = workspace(L.pd3_sparse_plan_workspace(n0), dev)
        hsh = host_i32(sets[0]["shape"])  # (host arrays must outlive the call: keep them in locals)
        check(L.pd3_sparse_sort_coords(ptr(c), n0, batch, ptr(hsh), ptr(sets[0]["keys"]),
                                       ptr(order), ptr(counts[0:1]), ptr(ws), ws.numel(), stream_ptr(dev)),
              "sparse_sort_coords")
    pairs, cur = [], 0
    for sp in specs:
        ks, st, pd = _triple(sp.kernel_size), _triple(sp.stride), _triple(sp.padding)
        if sp.subm:
            pairs.append((cur, cur))
            continue
        shape = out_spatial_shape(sets[cur]["shape"], ks, st, pd)
        if min(shape) <= 0:
            raise RuntimeError("sparse_conv3d: empty output shape")
        per_in = math.prod(-(-k // s) for k, s in zip(ks, st))
        cap = int(min(sets[cur]["cap"] * per_in, batch * math.prod(shape)))
        if caps is not None:
            cap = min(cap, int(caps[len(sets)]))
        new = dict(keys=torch.empty((max(cap, 1),), dtype=torch.int32, device=dev), cap=cap, shape=shape)
        if cap > 0 and sets[cur]["cap"] > 0:
            j = len(sets)
            hsh, hk, hs, hp = host_i32(sets[cur]["shape"]), host_i32(ks), host_i32(st), host_i32(pd)
            ws = workspace(L.pd3_sparse_conv_outputs_workspace(batch, ptr(hsh), ptr(hk), ptr(hs), ptr(hp)), dev)
            check(L.pd3_sparse_conv_outputs(ptr(sets[cur]["keys"]), ptr(counts[cur:cur + 1]), sets[cur]["cap"],
                                            batch, ptr(hsh), ptr(hk), ptr(hs), ptr(hp), ptr(new["keys"]),
                                            ptr(counts[j:j + 1]), cap, ptr(ws), ws.numel(), stream_ptr(dev)),
                  "sparse_conv_outputs")
        sets.append(new)
        pairs.append((cur, len(sets) - 1))
        cur = len(sets) - 1
    if caps is None:
        n = [int(v) for v in counts.cpu().tolist()]  # the one host sync of the encoder
    else:
        if len(caps) != n_sets:
            raise RuntimeError("sparse_conv3d.plan: caps must hold one capacity per index set")
        n = [st_["cap"] for st_ in sets]  # arrays at capacity, the real counts stay on the device
    for j, (st_, nn_) in enumerate(zip(sets, n)):
        st_["n"] = min(nn_, st_["cap"])
        st_["coords"] = None
        st_["n_dev"] = counts[j:j + 1] if caps is not None else None
    books, out = {}, []
    for sp, (i, o) in zip(specs, pairs):
        ks, st, pd = _triple(sp.kernel_size), _triple(sp.stride), _triple(sp.padding)
        if sp.subm:
            pd = tuple(k // 2 for k in ks)
        kvol = ks[0] * ks[1] * ks[2]
        tag = (i, o, sp.key, ks, st, pd, sp.subm)  # a key shared by convolutions of different geometry shares nothing
        if tag not in books:
            n_in, n_out = sets[i]["n"], sets[o]["n"]
            nbr = torch.empty((n_out, kvol), dtype=torch.int32, device=dev)
            want_coords = sets[o]["coords"] is None
            oc = torch.empty((n_out, 4), dtype=torch.int32, device=dev) if want_coords else sets[o]["coords"]
            if n_out > 0 and n_in > 0:
                hsh, hk, hs, hp = host_i32(sets[i]["shape"]), host_i32(ks), host_i32(st), host_i32(pd)
                ws = workspace(L.pd3_sparse_rulebook_workspace(batch, ptr(hsh)), dev)
                check(L.pd3_sparse_rulebook(ptr(sets[i]["keys"]), ptr(sets[i]["n_dev"]), n_in, ptr(sets[o]["keys"]),
                                            ptr(sets[o]["n_dev"]), n_out, batch, ptr(hsh), ptr(hk), ptr(hs), ptr(hp),
                                            int(sp.subm), ptr(nbr), ptr(oc) if want_coords else None, ptr(ws),
                                            ws.numel(), stream_ptr(dev)),
                      "sparse_rulebook")
            elif n_out > 0:
                nbr.fill_(-1)
            sets[o]["coords"] = oc
            tile_order = None
            if TILE_ORDER and sp.order and n_out > 0 and 1 < kvol <= 31:
                tile_order = torch.empty((int(L.pd3_sparse_tile_order_entries(n_out)),), dtype=torch.int32, device=dev)
                check(L.pd3_sparse_tile_order(ptr(nbr), ptr(sets[o]["n_dev"]), n_out, kvol, ptr(tile_order),
                                              stream_ptr(dev)), "sparse_tile_order")
            books[tag] = SparseIndices(oc, nbr, n_out, sets[o]["shape"], kvol, tile_order, sets[o]["n_dev"])
        out.append(books[tag])
    if sets[0]["coords"] is None:  # a chain that starts with a regular convolution: decode the first set here
        k = sets[0]["keys"][: sets[0]["n"]].long() & 0xFFFFFFFF
        d, h, w = sets[0]["shape"]
        sets[0]["coords"] = torch.stack([k // (d * h * w), (k // (h * w)) % d, (k // w) % h, k % w], 1).int()
    if caps is None:
        return SparsePlan(order[: sets[0]["n"]].long(), sets[0]["coords"], sets[0]["n"], out, counts=n)
    # a set that filled its capacity may have lost rows (the first set's capacity is the input's own row count)
    limit = torch.tensor([st_["cap"] for st_ in sets[1:]], dtype=torch.int32, device=dev)
    over = (counts[1:] >= limit).any() if n_sets > 1 else torch.zeros((), dtype=torch.bool, device=dev)
    return SparsePlan(order[: sets[0]["n"]].long(), sets[0]["coords"], sets[0]["n"], out, n_in_dev=counts[0:1],
                      overflow=over)


def features(in_feats: torch.Tensor, idx: SparseIndices, weight: torch.Tensor, bias=None, scale=None,
             shift=None, residual=None, relu: bool = False) -> torch.Tensor:
    """weight [kd, kh, kw, Cin, Cout] (Paddle layout)."""
    f = require_gpu(in_feats, "sparse_conv3d")
    w = require_gpu(weight, "sparse_conv3d")
    cin, cout = int(w.shape[-2]), int(w.shape[-1])
    if f.shape[1] != cin or w.numel() != idx.kernel_volume * cin * cout:
        raise RuntimeError("sparse_conv3d: weight / feature shapes do not match")
    out = torch.empty((idx.n_out, cout), dtype=torch.float32, device=f.device)
    if idx.n_out == 0:
        return out
    opt = [None if t is None else require_gpu(t, "sparse_conv3d") for t in (bias, scale, shift, residual)]
    check(lib().pd3_sparse_conv3d_features_ordered(
        ptr(f), ptr(idx.nbr), ptr(idx.n_out_dev), idx.n_out, idx.kernel_volume, cin, cout, ptr(w), ptr(opt[0]),
        ptr(opt[1]), ptr(opt[2]), ptr(opt[3]), int(bool(relu)), ptr(idx.order) if TILE_ORDER else None, ptr(out),
        stream_ptr(f.device)), "sparse_conv3d_features")
    return out


# The fp32 layers from 16 -> 32 channels on run on the bf16 matrix cores with every operand cut into three bf16 pieces
# (csrc/sparse_conv_x3.hip: fp32 arithmetic -- six piece products accumulated in fp32 -- at 2.7 x the fp32 pipe's rate).
# False = the fp32 matrix-core kernel everywhere (pd3_sparse_conv3d_features_ordered), kept as the comparison.
SPLIT_BF16 = True


def bf16x3_supported(cin: int, cout: int, kernel_volume: int) -> bool:
    """Shapes pd3_sparse_conv3d_features_bf16x3 serves."""
    return cin % 16 == 0 and cout in (32, 64, 128) and kernel_volume <= 27


def bf16x3_pays(cin: int, cout: int, kernel_volume: int) -> bool:
    """... and where it is faster than the fp32 matrix-core kernel: from 64 output channels on (measured, 8 scenes:
    32 -> 32 1.00 ms against 0.89-0.94, 32 -> 64 0.79 against 0.94, 64 -> 64 1.6 against 2.4-2.5, 128 -> 128 2.3 against
    3.9-4.0; profiles/r05_sparse_layers.txt, r05_sparse_layers_fp32.txt)."""
    return bf16x3_supported(cin, cout, kernel_volume) and cout >= 64


def pack_weight_bf16x3(weight: torch.Tensor) -> torch.Tensor:
    """weight [kd, kh, kw, Cin, Cout] fp32 (Paddle layout) -> its three bf16 pieces in the operand order of
    features_bf16x3."""
    w = require_gpu(weight, "sparse_conv3d")
    cin, cout = int(w.shape[-2]), int(w.shape[-1])
    kvol = w.numel() // (cin * cout)
    out = torch.empty((3 * w.numel(),), dtype=torch.bfloat16, device=w.device)
    check(lib().pd3_sparse_pack_weight_bf16x3(ptr(w), kvol, cin, cout, ptr(out), stream_ptr(w.device)),
          "sparse_pack_weight_bf16x3")
    return out


def features_bf16x3(in_feats: torch.Tensor, idx: SparseIndices, packed_weight: torch.Tensor, cin: int, cout: int,
                    bias=None, scale=None, shift=None, residual=None, relu: bool = False) -> torch.Tensor:
    """features() on the bf16 matrix cores, fp32 rows in and out (pd3_sparse_conv3d_features_bf16x3)."""
    f = require_gpu(in_feats, "sparse_conv3d")
    if f.shape[1] != cin or packed_weight.numel() != 3 * idx.kernel_volume * cin * cout:
        raise RuntimeError("sparse_conv3d: weight / feature shapes do not match")
    out = torch.empty((idx.n_out, cout), dtype=torch.float32, device=f.device)
    if idx.n_out == 0:
        return out
    opt = [None if t is None else require_gpu(t, "sparse_conv3d") for t in (bias, scale, shift, residual)]
    check(lib().pd3_sparse_conv3d_features_bf16x3(
        ptr(f), ptr(idx.nbr), ptr(idx.n_out_dev), idx.n_out, idx.kernel_volume, cin, cout, ptr(packed_weight),
        ptr(opt[0]), ptr(opt[1]), ptr(opt[2]), ptr(opt[3]), int(bool(relu)), ptr(idx.order) if TILE_ORDER else None,
        ptr(out), stream_ptr(f.device)), "sparse_conv3d_features_bf16x3")
    return out


def f16_supported(cin: int, cout: int, kernel_volume: int) -> bool:
    """Shapes the fp16 gather-GEMM serves (the encoder's layers from 16 -> 32 on)."""
    return cin % 16 == 0 and cout in (32, 64, 128) and kernel_volume <= 27


def pack_weight_f16(weight: torch.Tensor) -> torch.Tensor:
    """weight [kd, kh, kw, Cin, Cout] fp32 (Paddle layout) -> the fp16 operand order of features_f16."""
    w = require_gpu(weight, "sparse_conv3d")
    cin, cout = int(w.shape[-2]), int(w.shape[-1])
    kvol = w.numel() // (cin * cout)
    out = torch.empty((w.numel(),), dtype=torch.float16, device=w.device)
    check(lib().pd3_sparse_pack_weight_f16(ptr(w), kvol, cin, cout, ptr(out), stream_ptr(w.device)),
          "sparse_pack_weight_f16")
    return out


def features_f16(in_feats: torch.Tensor, idx: SparseIndices, packed_weight: torch.Tensor, cin: int, cout: int,
                 bias=None, scale=None, shift=None, residual=None, relu: bool = False,
                 out_f32: bool = False) -> torch.Tensor:
    """features() on the fp16 matrix cores: in_feats / residual fp16 [n, C], packed_weight from pack_weight_f16, fp32
    accumulation, bias / scale / shift fp32; fp16 rows out (fp32 with out_f32)."""
    f = require_gpu(in_feats, "sparse_conv3d", torch.float16)
    if f.shape[1] != cin or packed_weight.numel() != idx.kernel_volume * cin * cout:
        raise RuntimeError("sparse_conv3d: weight / feature shapes do not match")
    out = torch.empty((idx.n_out, cout), dtype=torch.float32 if out_f32 else torch.float16, device=f.device)
    if idx.n_out == 0:
        return out
    opt = [None if t is None else require_gpu(t, "sparse_conv3d") for t in (bias, scale, shift)]
    res = None if residual is None else require_gpu(residual, "sparse_conv3d", torch.float16)
    check(lib().pd3_sparse_conv3d_features_f16(
        ptr(f), ptr(idx.nbr), ptr(idx.n_out_dev), idx.n_out, idx.kernel_volume, cin, cout, ptr(packed_weight),
        ptr(opt[0]), ptr(opt[1]), ptr(opt[2]), ptr(res), int(bool(relu)), ptr(idx.order) if TILE_ORDER else None,
        ptr(out), int(bool(out_f32)), stream_ptr(f.device)), "sparse_conv3d_features_f16")
    return out


def gather_gemm_f16(in_feats: torch.Tensor, nbr: torch.Tensor, packed_weight: torch.Tensor, cin: int, cout: int,
                    out: torch.Tensor, out_off: int = 0, bias=None, relu: bool = False, order=None) -> torch.Tensor:
    """The fp16 gather-GEMM as a general operator (pd3_gather_gemm_f16): out[row, out_off : out_off + cout] =
    relu?(sum_k W[k] . in[nbr[row, k]] + bias) for a static neighbour table `nbr` [rows, K] (-1: no contribution); `out`
    is a row-major fp16 matrix [rows, ld] of which this call writes one channel slice.  The FPN levels under AMP."""
    f = require_gpu(in_feats, "gather_gemm_f16", torch.float16)
    rows, k = int(nbr.shape[0]), int(nbr.shape[1])
    if out.dtype != torch.float16 or not out.is_contiguous() or out.shape[0] != rows:
        raise RuntimeError("gather_gemm_f16: out must be a contiguous fp16 [rows, ld] matrix")
    check(lib().pd3_gather_gemm_f16(ptr(f), ptr(nbr), None, rows, k, cin, cout, ptr(packed_weight), ptr(bias), None, None,
                                    None, int(bool(relu)), ptr(order), ptr(out), 0, int(out.shape[1]), int(out_off),
                                    stream_ptr(f.device)), "gather_gemm_f16")
    return out


def tile_order(nbr: torch.Tensor) -> torch.Tensor:
    """pd3_sparse_tile_order of a neighbour table with all of its rows real."""
    rows, k = int(nbr.shape[0]), int(nbr.shape[1])
    order = torch.empty((int(lib().pd3_sparse_tile_order_entries(rows)),), dtype=torch.int32, device=nbr.device)
    check(lib().pd3_sparse_tile_order(ptr(nbr), None, rows, k, ptr(order), stream_ptr(nbr.device)), "sparse_tile_order")
    return order


def to_dense(feats: torch.Tensor, coords: torch.Tensor, batch: int, spatial_shape, n_dev=None) -> torch.Tensor:
    f = require_gpu(feats, "sparse_to_dense")
    c = require_gpu(coords, "sparse_to_dense", torch.int32)
    d, h, w = (int(x) for x in spatial_shape)
    ch = f.shape[1]
    out = torch.empty((batch, ch * d, h, w), dtype=torch.float32, device=f.device)
    n = f.shape[0]
    if n == 0:
        return out.zero_()
    hsh = host_i32(spatial_shape)
    check(lib().pd3_sparse_to_dense(ptr(f), ptr(c), ptr(n_dev), n, ch, batch, ptr(hsh), ptr(out),
                                    stream_ptr(f.device)), "sparse_to_dense")
    return out
