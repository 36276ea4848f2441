#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 200 python tools/prof/prof_wino_trace.py ${1:-4,0} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r04h_trace.txt
