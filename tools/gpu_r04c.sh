#!/bin/bash
# round 4, third GPU call: 3-D wave form of hard_voxelize, sort path with the slot-per-lane row writer, HardVFE packed v2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rf -k "voxelize or hard_vfe or python_golden_gpu" 2>&1 | tail -30 > gpurun_out/r04c_tests.log
cat gpurun_out/r04c_tests.log
echo "== c4 paths (8 frames)" | tee gpurun_out/r04c_vox.txt
timeout 300 python tools/prof/prof_voxelize.py 8 160000 20 1,14,15,16,1,14 c4 2>&1 | grep -v "^$" | tee -a gpurun_out/r04c_vox.txt
for p in 1 14; do
  PROF_FILTER=pd3 PROF_TOP=12 timeout 300 tools/gpu_prof.sh r04c_c4p$p tools/prof/prof_voxelize.py 8 160000 10 $p c4 > /dev/null 2>&1
  echo "# path $p" | tee -a gpurun_out/r04c_vox.txt; cat gpurun_out/r04c_c4p${p}_kernels.txt | tee -a gpurun_out/r04c_vox.txt
done
echo "== c3 check" | tee -a gpurun_out/r04c_vox.txt
timeout 300 python tools/prof/prof_voxelize.py 16 30000 30 5,11,5,11 2>&1 | grep -v "^$" | tee -a gpurun_out/r04c_vox.txt
echo "== bevfusion"
python bench.py --workload bevfusion_lidar --no-cpu-baseline > gpurun_out/r04c_bevf.json 2> gpurun_out/r04c_bevf.err
python -c "
import json;d=json.load(open('gpurun_out/r04c_bevf.json'));print(d['value'], d['per_op_ms'])"
echo "== voxel model"
python bench.py --workload centerpoint_voxel --no-cpu-baseline --no-extras > gpurun_out/r04c_voxel.json 2> gpurun_out/r04c_voxel.err
python -c "
import json;d=json.load(open('gpurun_out/r04c_voxel.json'));print(d['value'], d['per_op_ms'], d['roofline']['frac'])"
