"""`paddle3d.ops.voxelize` mirror: hard_voxelize(points, voxel_size, point_cloud_range,
max_num_points_in_voxel, max_voxels) -> (voxels, coords, num_points_per_voxel, num_voxels).

Reference operator: paddle3d/ops/voxel/voxelize_op.cc:149-191 (PD_BUILD_OP(hard_voxelize)); caller
paddle3d/models/voxelizers/voxelize.py:39-58.  Same positional arguments, same four outputs with the
same fixed shapes / dtypes (fp32 [V,P,D], int32 [V,3] (z,y,x), int32 [V], int32 [1]).
"""
from __future__ import annotations


import torch

from ._common import check, host_f32, lib, ptr, require_gpu, stream_ptr, workspace

__all__ = ["dynamic_voxelize", "hard_voxelize", "hard_voxelize_batch", "hard_voxelize_index_batch",
           "lds_atomic_order_ok"]

_LDS_ORDER = {}  # device index -> bool, probed once per process


def lds_atomic_order_ok(dev: torch.device) -> bool:
    """The hardware property the wave / tiled forms of hard_voxelize rest on (csrc/selfcheck.hip: lanes of one
    returning LDS add that hit one word are served in ascending lane order, a wave's LDS instructions in program
    order), probed ONCE per process and device through the library itself: 65 k adds in four address patterns, each
    must return the sequential count.  The automatic path choice falls back to the sort form (which does not depend
    on it) when the probe fails -- a new part or driver must never silently reorder points inside a voxel."""
    import numpy as np

    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key in _LDS_ORDER:
        return _LDS_ORDER[key]
    blocks, waves, rounds, table = 32, 4, 8, 256
    rng = np.random.default_rng(1)
    n_w = blocks * waves
    mode = (np.arange(n_w) % 4)[:, None, None]
    a = np.where(mode == 0, rng.integers(0, table, (n_w, rounds, 64)),
                 np.where(mode == 1, rng.integers(0, 4, (n_w, rounds, 64)),
                          np.where(mode == 2, 7, (rng.integers(0, 37, (n_w, rounds, 64)) * 27) % table))).astype(np.uint32)
    a[rng.random(a.shape) < 1 / 11] = 0xFFFFFFFF
    da = torch.from_numpy(a.view(np.int32)).to(dev)
    dold = torch.empty_like(da)
    check(lib().pd3_selfcheck_lds_atomic_order(ptr(da), ptr(dold), blocks, waves, rounds, table, stream_ptr(dev)),
          "selfcheck_lds_atomic_order")
    old = dold.cpu().numpy().view(np.uint32).reshape(n_w, rounds * 64)
    flat = a.reshape(n_w, rounds * 64)
    want = np.full_like(flat, 0xFFFFFFFF)
    cnt = np.zeros((n_w, table), np.uint32)
    rows = np.arange(n_w)
    for i in range(flat.shape[1]):  # sequential over the 512 adds of a wave, vectorised over the waves
        ai = flat[:, i]
        live = ai != 0xFFFFFFFF
        idx = np.where(live, ai, 0)
        want[:, i] = np.where(live, cnt[rows, idx], 0xFFFFFFFF)
        cnt[rows[live], idx[live]] += 1
    ok = bool((old == want).all())
    if not ok:
        import warnings

        warnings.warn("paddle3d_amd: returning LDS atomics are NOT served in lane order on this device; "
                      "hard_voxelize uses its sort form (slower, same results)", RuntimeWarning)
    _LDS_ORDER[key] = ok
    return ok


def hard_voxelize_batch(points: torch.Tensor, voxel_size, point_cloud_range, max_num_points_in_voxel: int,
                        max_voxels: int, num_points: torch.Tensor | None = None, with_batch_coors: bool = False,
                        path: int = 0):
    """points [B, N, D] fp32 on the GPU; num_points optional int32 [B] (valid rows per frame).

    Returns voxels [B,V,P,D], coords [B,V,3], num_points_per_voxel [B,V], num_voxels [B] -- the
    reference's per-sample op results stacked (HardVoxelizer's python loop, voxelize.py:60-82).
    with_batch_coors=True additionally returns coors [B,V,4] = (batch, z, y, x), batch -1 on padding rows.
    path: 0 automatic, 1 generic sort path, 2 tiled path (compact payload array), 3 tiled path (gathered rows),
    5 wave form of the tiled path (6 .. 10: route tile shape forced; 11: heavy group waves at raised priority,
    12: two half batches on two streams, 13: both; 14 .. 16 the 3-D wave form; 17 a measurement form: the points' payload
    carried through the route kernel) (pd3_hard_voxelize_path; the tests run them).
    """
    f64 = isinstance(points, torch.Tensor) and points.dtype == torch.float64
    pts = require_gpu(points, "hard_voxelize", torch.float64 if f64 else torch.float32)
    if pts.dim() != 3:
        raise RuntimeError("hard_voxelize_batch expects points of shape [B, N, D]")
    if f64 and path not in (0, 1):
        raise RuntimeError("hard_voxelize: float64 points run the generic sort path only (path 0 or 1)")
    b, n, d = pts.shape
    dev = pts.device
    vs, pr = host_f32(voxel_size, 3), host_f32(point_cloud_range, 6)
    p, v = int(max_num_points_in_voxel), int(max_voxels)
    voxels = torch.empty((b, v, p, d), dtype=pts.dtype, device=dev)  # HardInferDtype: voxels take the points' dtype
    coords = torch.empty((b, v, 3), dtype=torch.int32, device=dev)
    npv = torch.empty((b, v), dtype=torch.int32, device=dev)
    nv = torch.empty((b,), dtype=torch.int32, device=dev)
    coors4 = torch.empty((b, v, 4), dtype=torch.int32, device=dev) if with_batch_coors else None
    if num_points is not None:
        num_points = require_gpu(num_points, "hard_voxelize", torch.int32)
    L = lib()
    ws_bytes = L.pd3_hard_voxelize_workspace(b, n, d, ptr(vs), ptr(pr), p, v)
    if ws_bytes == 0:
        raise RuntimeError("hard_voxelize: invalid voxel_size / point_cloud_range / sizes")
    ws = workspace(ws_bytes, dev)
    if f64:  # the reference's CPU kernel instantiated for double (PD_DISPATCH_FLOATING_TYPES, voxelize_op.cc:128)
        check(L.pd3_hard_voxelize_f64(ptr(pts), ptr(num_points), b, n, d, ptr(vs), ptr(pr), p, v, ptr(voxels),
                                      ptr(coords), ptr(npv), ptr(nv), ptr(coors4), ptr(ws), ws.numel(),
                                      stream_ptr(dev)), "hard_voxelize")
    else:
        if path == 0 and not lds_atomic_order_ok(dev):
            path = 1  # the sort form does not depend on the order in which an LDS add serves its lanes
        check(L.pd3_hard_voxelize_path(ptr(pts), ptr(num_points), b, n, d, ptr(vs), ptr(pr), p, v, ptr(voxels),
                                       ptr(coords), ptr(npv), ptr(nv), ptr(coors4), ptr(ws), ws.numel(),
                                       stream_ptr(dev), int(path)),
              "hard_voxelize")
    if with_batch_coors:
        return voxels, coords, npv, nv, coors4
    return voxels, coords, npv, nv


def hard_voxelize_index_batch(points: torch.Tensor, voxel_size, point_cloud_range, max_num_points_in_voxel: int,
                              max_voxels: int, num_points: torch.Tensor | None = None):
    """hard_voxelize_batch without the padded [B, V, P, D] tensor (pd3_hard_voxelize_index): returns
    (vox_span [B, V, 2] int32 (start, count), point_list [B * N + 4] int32, coords [B, V, 3], num_points_per_voxel
    [B, V], num_voxels [B], coors [B, V, 4]) -- voxel v of frame b holds the points
    points[b, point_list[b * N + start + k]], k < count -- or None where the grid is not one the wave forms of the
    voxelizer serve (the caller then runs hard_voxelize_batch)."""
    pts = require_gpu(points, "hard_voxelize_index", torch.float32)
    if pts.dim() != 3:
        raise RuntimeError("hard_voxelize_index_batch expects points of shape [B, N, D]")
    b, n, d = pts.shape
    dev = pts.device
    if not lds_atomic_order_ok(dev):
        return None
    vs, pr = host_f32(voxel_size, 3), host_f32(point_cloud_range, 6)
    p, v = int(max_num_points_in_voxel), int(max_voxels)
    L = lib()
    ws_bytes = L.pd3_hard_voxelize_workspace(b, n, d, ptr(vs), ptr(pr), p, v)
    if ws_bytes == 0:
        raise RuntimeError("hard_voxelize: invalid voxel_size / point_cloud_range / sizes")
    span = torch.empty((b, v, 2), dtype=torch.int32, device=dev)
    plist = torch.empty((int(L.pd3_hard_voxelize_index_list_entries(b, n)),), dtype=torch.int32, device=dev)
    coords = torch.empty((b, v, 3), dtype=torch.int32, device=dev)
    npv = torch.empty((b, v), dtype=torch.int32, device=dev)
    nv = torch.empty((b,), dtype=torch.int32, device=dev)
    coors4 = torch.empty((b, v, 4), dtype=torch.int32, device=dev)
    if num_points is not None:
        num_points = require_gpu(num_points, "hard_voxelize_index", torch.int32)
    ws = workspace(ws_bytes, dev)
    rc = L.pd3_hard_voxelize_index(ptr(pts), ptr(num_points), b, n, d, ptr(vs), ptr(pr), p, v, ptr(span), ptr(plist),
                                   ptr(coords), ptr(npv), ptr(nv), ptr(coors4), ptr(ws), ws.numel(), stream_ptr(dev))
    if rc == -3:
        return None
    check(rc, "hard_voxelize_index")
    return span, plist, coords, npv, nv, coors4


def _stage_points(points, op):
    """The reference dispatches on the place of `points` (voxelize_op.cc:149-166): CPU tensors run
    hard_voxelize_cpu and get CPU results, GPU and GPU-pinned tensors run the device kernel.  This library has the
    device kernel only -- bit-identical to hard_voxelize_cpu -- so host tensors are staged through the current
    GPU: (device tensor, whether the results go back to the host)."""
    if not isinstance(points, torch.Tensor):
        raise RuntimeError(f"Unsupported device type for {op} operator.")
    if points.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"{op}: points must be float32 or float64 (PD_DISPATCH_FLOATING_TYPES, "
                           f"voxelize_op.cc:128), got {points.dtype}")
    if points.is_cuda:
        return points, False
    if points.device.type != "cpu":
        raise RuntimeError(f"Unsupported device type for {op} operator.")
    if not torch.cuda.is_available():
        raise RuntimeError(f"{op}: host tensors are staged through the GPU and no GPU is visible "
                           "(this library has no CPU kernel)")
    dev = torch.device("cuda", torch.cuda.current_device())
    return points.to(dev, non_blocking=points.is_pinned()), not points.is_pinned()


def hard_voxelize(points: torch.Tensor, voxel_size, point_cloud_range, max_num_points_in_voxel: int,
                  max_voxels: int, path: int = 0):
    """points [N, D] fp32 or fp64 on the GPU, in pinned host memory (results on the GPU, like the reference's
    `is_gpu_pinned()` branch) or on the CPU (results on the CPU, like hard_voxelize_cpu; computed on the GPU).
    float64 points give float64 voxels (the reference's kernel instantiated for double)."""
    pts, to_host = _stage_points(points, "hard_voxelize")
    pts = require_gpu(pts, "hard_voxelize", pts.dtype)
    if pts.dim() != 2:
        raise RuntimeError("hard_voxelize expects points of shape [N, D]")
    voxels, coords, npv, nv = hard_voxelize_batch(pts.unsqueeze(0), voxel_size, point_cloud_range,
                                                  max_num_points_in_voxel, max_voxels, path=path)
    out = (voxels[0], coords[0], npv[0], nv)
    return tuple(o.cpu() for o in out) if to_host else out


def dynamic_voxelize(points: torch.Tensor, voxel_size, point_cloud_range) -> torch.Tensor:
    """Per-point voxel coordinates [N, 3] int32 = (z, y, x), -1 outside the range; hard_voxelize's cell rule."""
    pts = require_gpu(points, "dynamic_voxelize")
    if pts.dim() != 2:
        raise RuntimeError("dynamic_voxelize expects points of shape [N, D]")
    n, d = pts.shape
    vs, pr = host_f32(voxel_size, 3), host_f32(point_cloud_range, 6)
    coors = torch.empty((n, 3), dtype=torch.int32, device=pts.device)
    check(lib().pd3_dynamic_voxelize(ptr(pts), n, d, ptr(vs), ptr(pr), ptr(coors), stream_ptr(pts.device)),
          "dynamic_voxelize")
    return coors
