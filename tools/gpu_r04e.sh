#!/bin/bash
# round 4, fifth GPU call: ILP row writer (C3 + C4), full GPU suite, default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
echo "== c3 paths (16 frames)" | tee gpurun_out/r04e_vox.txt
timeout 300 python tools/prof/prof_voxelize.py 16 30000 50 5,0,5,0 2>&1 | grep -v "^$" | tee -a gpurun_out/r04e_vox.txt
PROF_FILTER=pd3 PROF_TOP=6 timeout 300 tools/gpu_prof.sh r04e_c3p0 tools/prof/prof_voxelize.py 16 30000 20 0 > /dev/null 2>&1
cat gpurun_out/r04e_c3p0_kernels.txt | tee -a gpurun_out/r04e_vox.txt
echo "== c4 paths (8 frames)" | tee -a gpurun_out/r04e_vox.txt
timeout 300 python tools/prof/prof_voxelize.py 8 160000 20 1,14,1,14 c4 2>&1 | grep -v "^$" | tee -a gpurun_out/r04e_vox.txt
PROF_FILTER=pd3 PROF_TOP=6 timeout 300 tools/gpu_prof.sh r04e_c4p14 tools/prof/prof_voxelize.py 8 160000 10 14 c4 > /dev/null 2>&1
cat gpurun_out/r04e_c4p14_kernels.txt | tee -a gpurun_out/r04e_vox.txt
tools/gpu_r04.sh r04e
