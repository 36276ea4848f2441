#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(timeout 600 python -m pytest tests/test_voxelize_gpu.py tests/test_model_gpu.py tests/test_pointpillars_gpu.py -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r2d_tests.log
python tools/prof/prof_voxelize.py 16 30000 20 2,3,4 2>&1 | grep -v amdgpu > gpurun_out/r2d_voxpaths.txt
PROF_FILTER=vt_ tools/gpu_prof.sh r2d_vox_path4 $R/tools/prof/prof_voxelize.py 16 30000 20 4 > gpurun_out/r2d_prof4.log 2>&1
for p in 3 4; do
  python bench.py --vox-path $p --no-cpu-baseline --no-extras > gpurun_out/r2d_bench_path$p.json 2> gpurun_out/r2d_bench_path$p.err
done
PROF_FILTER=vt_ tools/gpu_prof.sh r2d_bench_path4 $R/bench.py --vox-path 4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r2d_prof_b4.log 2>&1
for w in pointpillars_kitti bevfusion_lidar; do
  python - <<EOF > gpurun_out/r2d_vox_$w.txt 2>&1
import sys, numpy as np, torch
sys.path.insert(0, "$R")
from paddle3d_amd import synth
from paddle3d_amd.ops import voxelize
if "$w" == "pointpillars_kitti":
    pts = torch.from_numpy(np.stack([synth.kitti_frame(100 + i, 16384) for i in range(16)])).cuda(); a = (list(synth.KITTI_PILLAR), list(synth.KITTI_RANGE), 32, 40000)
else:
    pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(100 + i, dims=4) for i in range(16)])).cuda(); a = ([0.25, 0.25, 8.0], [-50.0, -50.0, -5.0, 50.0, 50.0, 3.0], 64, 40000)
for path in (2, 3, 4):
    for _ in range(3): out = voxelize.hard_voxelize_batch(pts, *a, with_batch_coors=True, path=path)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): out = voxelize.hard_voxelize_batch(pts, *a, with_batch_coors=True, path=path)
    e1.record(); torch.cuda.synchronize(); print("$w path", path, e0.elapsed_time(e1) / 20 * 1e3, "us per 16 frames")
EOF
done
cat gpurun_out/r2d_tests.log gpurun_out/r2d_voxpaths.txt gpurun_out/r2d_vox_path4_kernels.txt gpurun_out/r2d_bench_path4_kernels.txt gpurun_out/r2d_vox_pointpillars_kitti.txt gpurun_out/r2d_vox_bevfusion_lidar.txt | grep -v amdgpu
python - <<EOF
import json
for p in (3, 4):
    d = json.loads(open("gpurun_out/r2d_bench_path%d.json" % p).read().strip().splitlines()[-1]); print(p, d["value"], d["per_op_ms"])
EOF
