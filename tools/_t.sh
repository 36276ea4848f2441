cd ${GRAFT_REPO_ROOT:-.}
(timeout 600 python -m pytest tests/test_voxelize_gpu.py tests/test_pointpillars_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -6) | grep -v amdgpu
python tools/prof/prof_voxelize.py 16 30000 30 3,2,3 2>&1 | grep path
PROF_FILTER=vt_ tools/gpu_prof.sh t_vox3 $PWD/tools/prof/prof_voxelize.py 16 30000 20 3 > /dev/null 2>&1; cat gpurun_out/t_vox3_kernels.txt
