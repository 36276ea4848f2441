#!/bin/bash
# round 4, fourth GPU call: fp16 conv kernel + AMP graph, 3-D wave form after the register fast path
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rf -s -k "f16 or amp or wave3d or (voxelize and c4)" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r04d_tests.log
cat gpurun_out/r04d_tests.log
echo "== c4 paths (8 frames)" | tee gpurun_out/r04d_vox.txt
timeout 300 python tools/prof/prof_voxelize.py 8 160000 20 14,15,16,14 c4 2>&1 | grep -v "^$" | tee -a gpurun_out/r04d_vox.txt
PROF_FILTER=pd3 PROF_TOP=8 timeout 300 tools/gpu_prof.sh r04d_c4p14 tools/prof/prof_voxelize.py 8 160000 10 14 c4 > /dev/null 2>&1
cat gpurun_out/r04d_c4p14_kernels.txt | tee -a gpurun_out/r04d_vox.txt
echo "== amp bench"
python bench.py --workload centerpoint_pillars_amp --no-cpu-baseline --no-extras > gpurun_out/r04d_amp.json 2> gpurun_out/r04d_amp.err
tail -2 gpurun_out/r04d_amp.err
python -c "
import json;d=json.load(open('gpurun_out/r04d_amp.json'));print(d['value'], d['per_op_ms'], d.get('amp_error'))"
PROF_FILTER=pd3 PROF_TOP=16 timeout 300 tools/gpu_prof.sh r04d_amp bench.py --workload centerpoint_pillars_amp --steps 5 --warmup 2 --no-cpu-baseline --no-extras --repeats 0 > gpurun_out/r04d_amp_prof.log 2>&1
cat gpurun_out/r04d_amp_kernels.txt
