#!/bin/bash
# round-2 measurement run (GPU box): kernel splits, bench lines, PMC traffic.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python tools/prof/prof_voxelize.py 16 30000 20 2,3 2>&1 | grep -v amdgpu > gpurun_out/r2z_voxpaths.txt
PROF_FILTER=vt_ tools/gpu_prof.sh r2z_vox_path2 $R/tools/prof/prof_voxelize.py 16 30000 20 2 > gpurun_out/r2z_prof2.log 2>&1
PROF_FILTER=vt_ tools/gpu_prof.sh r2z_vox_path3 $R/tools/prof/prof_voxelize.py 16 30000 20 3 > gpurun_out/r2z_prof3.log 2>&1
PROF_TOP=40 tools/gpu_prof.sh r2z_kitti $R/bench.py --workload pointpillars_kitti --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2z_prof_kitti.log 2>&1
python bench.py > gpurun_out/r2z_bench_b16.json 2> gpurun_out/r2z_bench_b16.err
PROF_TOP=60 tools/gpu_prof.sh r2z_bench_b16 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r2z_prof_b16.log 2>&1
cp /tmp/prof_r2z_bench_b16/r2z_bench_b16_kernel_stats.csv gpurun_out/ 2>/dev/null
tools/gpu_traffic.sh r2z_b16 16 30000 > gpurun_out/r2z_traffic.log 2>&1
for w in pointpillars_kitti bevfusion_lidar centerpoint_voxel bev_pool_v2; do
  python bench.py --workload $w --no-cpu-baseline > gpurun_out/r2z_bench_$w.json 2> gpurun_out/r2z_bench_$w.err
done
cat gpurun_out/r2z_voxpaths.txt gpurun_out/r2z_vox_path3_kernels.txt
head -c 600 gpurun_out/r2z_bench_b16.json
