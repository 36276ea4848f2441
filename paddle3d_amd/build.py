"""Build libpaddle3d_amd.so (gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m paddle3d_amd.build            # incremental
    python -m paddle3d_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  The shared object lands in paddle3d_amd/lib/ (git-ignored,
but shipped to the GPU box by gpurun).  Flags that matter for parity with the reference CPU path:
-ffp-contract=off (no implicit FMA contraction; kernels that want FMA call fmaf explicitly) and no
fast-math (correctly rounded fp32 divide / sqrt are hipcc defaults).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libpaddle3d_amd.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs.append(os.path.join(HERE, "..", "include", "paddle3d_amd.h"))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# per-file additions: the ping-pong Winograd kernel computes U = G g G^T next to its MFMAs, where the SLP vectoriser's
# v_pk_* forms (two issue passes each, plus the v_mov shuffles that feed them) cost more than scalar VALU
EXTRA_FLAGS = {}


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    if force or _stale(obj, [src] + _headers()):
        cmd = [HIPCC, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
