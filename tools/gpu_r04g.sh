#!/bin/bash
# round 4, seventh GPU call: the ping-pong Winograd kernel -- parity tests and per-layer A/B against the packed form
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -rf -s -k "pingpong or stable_argsort or amp_graph" 2>&1 | grep -v "^$" | tail -15 | tee gpurun_out/r04g_tests.log
timeout 300 python tools/prof/prof_wino43.py 16 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r04g_wino.txt
