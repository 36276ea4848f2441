#!/bin/bash
# round 4: ping-pong Winograd wired into the layers with cin >= 128 -- conv / model / graph tests, per-layer A/B, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rf -x -k "conv or model or graph or centerpoint or cache" 2>&1 | grep -v "^$" | tail -8 | tee gpurun_out/r04k_tests.log
timeout 300 python tools/prof/prof_wino43.py 16 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r04k_wino.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep -v "amdgpu.ids" | tail -3 | tee gpurun_out/r04k_bench.json
