"""The C-ABI library loads without a GPU and exports exactly what include/paddle3d_amd.h declares."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from paddle3d_amd import build

    return build.build()


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, "include", "paddle3d_amd.h")).read()
    declared = set(re.findall(r"\b(pd3_\w+)\s*\(", hdr))
    assert len(declared) >= 17
    out = subprocess.check_output(["nm", "-D", "--defined-only", built], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert declared <= exported, declared - exported
    # nothing torch/paddle specific leaks into the ABI
    assert not [s for s in exported if "torch" in s.lower() or "paddle" in s.lower()]


def test_loader_binds_every_symbol(built):
    from paddle3d_amd import _lib

    L = _lib.lib()
    assert L.pd3_version() >= 100
    assert L.pd3_target_arch() == b"gfx950"
    for name in _lib.SYMBOLS:
        assert getattr(L, name).argtypes is not None


def test_library_is_gfx950_code_object(built):
    """The fat binary holds gfx950 code objects and nothing for another GPU family."""
    import re

    data = open(built, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", data))
    assert targets == {b"gfx950"}, targets
    assert not re.search(rb"\bsm_[0-9]{2}\b", data) and b".nv.info" not in data and b"nvptx" not in data


def test_workspace_queries_need_no_gpu(built):
    import numpy as np

    from paddle3d_amd import _lib

    L = _lib.lib()
    vs = np.array([0.2, 0.2, 8.0], np.float32)
    pr = np.array([-51.2, -51.2, -5, 51.2, 51.2, 3], np.float32)
    ws = L.pd3_hard_voxelize_workspace(1, 300000, 5, vs.ctypes.data, pr.ctypes.data, 20, 30000)
    assert 5 * 300000 * 4 <= ws < 64 * 2**20
    assert L.pd3_hard_voxelize_workspace(1, 300000, 5, vs.ctypes.data, (pr * 0).ctypes.data, 20, 30000) == 0
    assert L.pd3_pointpillars_scatter_workspace(2, 512, 512) == 2 * 512 * 512 * 4
    assert L.pd3_nms_workspace(1000) == 1000 * 16 * 8
    assert L.pd3_centerpoint_postprocess_workspace(1, 6, 128, 128, 1000, 83) > 0


def test_ops_device_dispatch_without_a_gpu(built):
    """The reference's hard_voxelize takes CPU, GPU and GPU-pinned points (voxelize_op.cc:149-166); here host tensors
    are staged through the GPU (tests/test_voxelize_gpu.py), so without one the op says so instead of computing on
    the CPU.  float32 and float64 points are taken (PD_DISPATCH_FLOATING_TYPES), other dtypes refused by name.  The other ops are GPU-only in the reference too."""
    import torch

    from paddle3d_amd.ops import iou3d_nms, voxelize

    args = ([0.2, 0.2, 8], [-51.2, -51.2, -5, 51.2, 51.2, 3], 20, 100)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="staged through the GPU and no GPU is visible"):
            voxelize.hard_voxelize(torch.zeros(10, 4), *args)
    with pytest.raises(RuntimeError, match="PD_DISPATCH_FLOATING_TYPES"):
        voxelize.hard_voxelize(torch.zeros(10, 4, dtype=torch.float16), *args)
    with pytest.raises(RuntimeError, match="PD_DISPATCH_FLOATING_TYPES"):
        voxelize.hard_voxelize(torch.zeros(10, 4, dtype=torch.int32), *args)
    with pytest.raises(RuntimeError, match="Unsupported device type for hard_voxelize operator"):
        voxelize.hard_voxelize([[0.0, 0.0, 0.0, 0.0]], *args)
    with pytest.raises(RuntimeError, match="Unsupported device type"):
        iou3d_nms.nms_gpu(torch.zeros(4, 7), 0.5)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from paddle3d_amd import _lib

    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    _lib.lib.cache_clear()
    with pytest.raises(_lib.Paddle3DAmdError, match="no CPU / PyTorch fallback"):
        _lib.lib()
    _lib.lib.cache_clear()


def test_patch_weight_packers_describe_the_layers():
    """Host logic of the patch-GEMM layers (no GPU): un-packing pack_patch_weight's [M/64][K/16][16][64] blocks gives
    the GEMM A matrix whose product with the pixels is the layer -- Conv2D k2 s2 (mode 0), 1x1 with a row count that is
    not a multiple of 64 (mode 1, zero-padded rows), Conv2DTranspose k2 s2 / k4 s4 (modes 2 / 3: rows (co, dy, dx))."""
    import numpy as np
    import torch
    import torch.nn.functional as F

    from paddle3d_amd.ops import conv

    def unpack(p):
        mb, kb = p.shape[0], p.shape[1]
        return p.permute(0, 3, 1, 2).reshape(mb * 64, kb * 16)

    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 16, 4, 8, generator=g)
    # mode 1, 20 rows -> 64 padded
    w = torch.randn(20, 16, 1, 1, generator=g)
    a = unpack(conv.pack_patch_weight(w, 1, False))
    assert a.shape == (64, 16) and not a[20:].any()
    ref = F.conv2d(x, w)[0].reshape(20, -1)
    np.testing.assert_allclose((a @ x[0].reshape(16, -1))[:20].numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    # modes 2 / 3: transposed convolutions with kernel = stride
    for k, mode in ((2, 2), (4, 3)):
        wt = torch.randn(16, 16, k, k, generator=g)
        assert conv.patch_mode(wt, k, True) == mode
        a = unpack(conv.pack_patch_weight(wt, mode, True))            # [co * k * k + dy * k + dx][ci]
        d = (a @ x[0].reshape(16, -1)).reshape(16, k, k, 4, 8)         # [co, dy, dx, y, x]
        out = d.permute(0, 3, 1, 4, 2).reshape(16, 4 * k, 8 * k)       # out[co][k y + dy][k x + dx]
        np.testing.assert_allclose(out.numpy(), F.conv_transpose2d(x, wt, stride=k)[0].numpy(), rtol=1e-5, atol=1e-5)
    # mode 0: Conv2D k2 s2, K = (ci, py, px)
    wc = torch.randn(64, 16, 2, 2, generator=g)
    a = unpack(conv.pack_patch_weight(wc, 0, False))
    cols = F.unfold(x, 2, stride=2)[0]                                  # [ci * 4 + py * 2 + px][pixels]
    np.testing.assert_allclose((a @ cols).numpy(), F.conv2d(x, wc, stride=2)[0].reshape(64, -1).numpy(), rtol=1e-5,
                               atol=1e-5)
    assert conv.patch_supported(0, 64, 128, 496, 432) and not conv.patch_supported(0, 64, 128, 496, 430)
    assert conv.patch_supported(3, 256, 128, 62, 56) and conv.patch_supported(1, 384, 20, 248, 216)
