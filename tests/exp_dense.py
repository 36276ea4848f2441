"""Experiment driver (not a test): dense graph variants -- NCHW vs channels_last, eager vs hipGraph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddle3d_amd import centerpoint as cpm  # noqa: E402

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = cpm.centerpoint_pillars_nuscenes().cuda().eval()
x = torch.randn(B, 64, 512, 512, device="cuda")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


def fwd(inp):
    return model.bbox_head(model.dense_forward(inp))[0]


with torch.no_grad():
    print("nchw eager ms/step", round(timeit(lambda: fwd(x)), 3))
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fwd(x)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            out = fwd(x)
        print("nchw graph ms/step", round(timeit(lambda: g.replay()), 3))
    except Exception as e:  # noqa: BLE001
        print("graph capture failed:", repr(e)[:200])
    # channels_last: fold weights into channels_last too
    m2 = cpm.centerpoint_pillars_nuscenes().cuda().eval()
    m2._dense_fold()
    m2._dense = tuple([[ (tr, w.contiguous(memory_format=torch.channels_last), b, st, pd) for (tr, w, b, st, pd) in blk] for blk in part] for part in m2._dense)
    m2.bbox_head._build_fused()
    for k in ("w0", "w1", "wf"):
        m2.bbox_head._fused[k] = m2.bbox_head._fused[k].contiguous(memory_format=torch.channels_last)
    xcl = x.contiguous(memory_format=torch.channels_last)
    print("channels_last eager ms/step", round(timeit(lambda: m2.bbox_head(m2.dense_forward(xcl))[0]), 3))
