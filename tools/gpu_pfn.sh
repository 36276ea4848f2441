#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pfn.sh <tag>  -- PFN parity tests on every kernel form, then their timings
tag=${1:-pfn}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_scatter_pfn_gpu.py tests/test_properties_gpu.py tests/test_python_golden_gpu.py -m gpu -x -q -k "pfn or pillar_feature" 2>&1 | tail -15 > gpurun_out/${tag}_tests.log
cat gpurun_out/${tag}_tests.log
timeout 600 python tools/prof/prof_pfn.py 16 2>&1 | tee gpurun_out/${tag}_prof.txt
