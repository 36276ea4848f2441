#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python tools/prof/prof_voxelize.py 16 30000 30 3,13,23,3,13,23 2>&1 | grep -v amdgpu > gpurun_out/r2e_voxpaths.txt
cat gpurun_out/r2e_voxpaths.txt
