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


def test_ops_refuse_cpu_tensors(built):
    import torch

    from paddle3d_amd.ops import iou3d_nms, voxelize

    with pytest.raises(RuntimeError, match="Unsupported device type for hard_voxelize operator"):
        voxelize.hard_voxelize(torch.zeros(10, 4), [0.2, 0.2, 8], [-51.2, -51.2, -5, 51.2, 51.2, 3], 20, 100)
    with pytest.raises(RuntimeError, match="Unsupported device type"):
        iou3d_nms.nms_gpu(torch.zeros(4, 7), 0.5)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from paddle3d_amd import _lib

    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    _lib.lib.cache_clear()
    with pytest.raises(_lib.Paddle3DAmdError, match="no CPU / PyTorch fallback"):
        _lib.lib()
    _lib.lib.cache_clear()
