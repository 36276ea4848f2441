#!/bin/bash
# round 4: the ping-pong Winograd kernel, second form (U in micro-steps behind every MFMA, raw rows by buffer_load ... lds)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -q -rf -s -k "pingpong" 2>&1 | grep -v "^$" | tail -8 | tee gpurun_out/r04i_tests.log
timeout 300 python tools/prof/prof_wino43.py 16 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r04i_wino.txt
timeout 200 python tools/prof/prof_wino_trace.py ${1:-0,1,2} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r04i_trace.txt
