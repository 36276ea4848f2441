cd ${GRAFT_REPO_ROOT:-.}
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/r2z_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2z_smoke.log 2>&1
bash tools/gpu_round_end.sh > gpurun_out/r2z_round_end.log 2>&1
grep -v amdgpu gpurun_out/r2z_tests.log; tail -2 gpurun_out/r2z_smoke.log; tail -12 gpurun_out/r2z_round_end.log | cut -c1-400
