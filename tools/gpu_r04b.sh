#!/bin/bash
# round 4, second GPU call: HardVFE packed form, wave-form variants of hard_voxelize (A/B in one process), config-4 kernel lists
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rf -k "hard_vfe or voxelize or map_proxy or python_golden_gpu" 2>&1 | tail -25 > gpurun_out/r04b_tests.log
cat gpurun_out/r04b_tests.log
echo "== vox variants" | tee gpurun_out/r04b_vox.txt
timeout 300 python tools/prof/prof_voxelize.py 16 30000 50 5,11,12,13,5,11,12,13 2>&1 | grep -v "^$" | tee -a gpurun_out/r04b_vox.txt
echo "== c4 sort path" | tee -a gpurun_out/r04b_vox.txt
timeout 300 python tools/prof/prof_voxelize.py 8 160000 20 1 c4 2>&1 | grep -v "^$" | tee -a gpurun_out/r04b_vox.txt
PROF_FILTER=pd3 PROF_TOP=12 timeout 300 tools/gpu_prof.sh r04b_c4vox tools/prof/prof_voxelize.py 8 160000 10 1 c4 > /dev/null 2>&1
cat gpurun_out/r04b_c4vox_kernels.txt | tee -a gpurun_out/r04b_vox.txt
echo "== bevfusion"
python bench.py --workload bevfusion_lidar --no-cpu-baseline > gpurun_out/r04b_bevf.json 2> gpurun_out/r04b_bevf.err
python -c "
import json;d=json.load(open('gpurun_out/r04b_bevf.json'));print(d['value'], d['per_op_ms'])"
echo "== voxel model kernels"
PROF_FILTER=pd3 PROF_TOP=40 timeout 600 tools/gpu_prof.sh r04b_voxel bench.py --workload centerpoint_voxel --no-cpu-baseline --no-extras --steps 5 --warmup 2 --repeats 0 > gpurun_out/r04b_voxel_prof.log 2>&1
tail -3 gpurun_out/r04b_voxel_prof.log; head -45 gpurun_out/r04b_voxel_kernels.txt
