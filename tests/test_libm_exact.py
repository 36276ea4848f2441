"""csrc/libm_exact.hpp (the device's sinf / cosf / expf / atanf / atan2f) against the libm of this machine, on the
host: the header is plain C++ when it is not compiled by hipcc, every operation in it is exact IEEE arithmetic, so
equality here is equality on the device (tests/test_nms_post_bev_gpu.py::test_device_libm_is_glibc checks that too).
tools/libm_exact_check.cpp walks every 256th float per unary function here (all 2^32 with stride 1: 0 mismatches,
profiles/r03_libm_exact.txt) and ~5 M atan2f pairs."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _glibc_version():
    import ctypes

    try:
        f = ctypes.CDLL(None).gnu_get_libc_version
        f.restype = ctypes.c_char_p
        a, b = f().decode().split(".")[:2]
        return int(a), int(b)
    except Exception:  # noqa: BLE001 -- not glibc
        return None


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_libm_exact_host(tmp_path):
    cpu = open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else ""
    if " fma" not in cpu:
        pytest.skip("host without FMA: glibc runs its unfused sinf / cosf / expf build here, not the one restated")
    ver = _glibc_version()
    if ver is None or not ((2, 28) <= ver <= (2, 40)):
        pytest.skip(f"libm_exact.hpp restates glibc 2.28 .. 2.40 (found {ver}): 2.41+ ships the CORE-MATH correctly "
                    "rounded atanf / atan2f, which fdlibm's float routines do not match")
    exe = str(tmp_path / "libm_exact_check")
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-pthread",
                    os.path.join(ROOT, "tools", "libm_exact_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe, "256"], capture_output=True, text=True, check=True, timeout=600).stdout
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 5, out
    for l in lines:
        m = re.match(r"(\w+): (\d+) (arguments|pairs), (\d+) mismatches", l)
        assert m, l
        assert int(m.group(2)) > 1_000_000 and int(m.group(4)) == 0, l
