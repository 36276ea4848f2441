set -x
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r2b_tests.log
python tools/prof/prof_voxelize.py 16 30000 20 2,3 > gpurun_out/r2b_voxpaths.txt 2>&1
PROF_FILTER=vt_ tools/gpu_prof.sh r2b_vox_path2 tools/prof/prof_voxelize.py 16 30000 20 2 > /dev/null 2>&1
PROF_FILTER=vt_ tools/gpu_prof.sh r2b_vox_path3 tools/prof/prof_voxelize.py 16 30000 20 3 > /dev/null 2>&1
python bench.py --workload pointpillars_kitti --no-cpu-baseline > gpurun_out/r2b_bench_kitti.json 2> gpurun_out/r2b_bench_kitti.err
PROF_TOP=30 tools/gpu_prof.sh r2b_kitti bench.py --workload pointpillars_kitti --steps 10 --warmup 3 > /dev/null 2>&1
tail -3 gpurun_out/r2b_tests.log; cat gpurun_out/r2b_voxpaths.txt | grep -v amdgpu; cat gpurun_out/r2b_vox_path2_kernels.txt gpurun_out/r2b_vox_path3_kernels.txt; head -c 1500 gpurun_out/r2b_bench_kitti.json; tail -3 gpurun_out/r2b_bench_kitti.err
