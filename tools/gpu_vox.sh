#!/bin/bash
# usage (GPU box, repo root): tools/gpu_vox.sh <paths, e.g. 3,5,6,7,8> [pytest: 0/1]
# hard_voxelize alone: parity tests on every path, HIP-event times per path, rocprof per-kernel split, clocks logged
paths=${1:-3,5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
out=$R/gpurun_out/vox_paths.txt
: > $out
if [ "${2:-1}" = "1" ]; then
  timeout 900 python -m pytest tests/test_voxelize_gpu.py -x -q 2>&1 | tail -5 | tee -a $out
fi
echo "# clocks before" >> $out; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" >> $out
timeout 300 python tools/prof/prof_voxelize.py 16 30000 50 $paths 2>&1 | grep -v "^$" | tee -a $out
echo "# clocks after" >> $out; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" >> $out
echo "# shuffled points" >> $out
timeout 300 python tools/prof/prof_voxelize.py 16 30000 20 $paths shuffle 2>&1 | grep -v "^$" | tee -a $out
for p in $(echo $paths | tr , ' '); do
  PROF_FILTER=pd3 PROF_TOP=8 timeout 300 tools/gpu_prof.sh vox_p$p tools/prof/prof_voxelize.py 16 30000 20 $p > /dev/null 2>&1
  echo "# path $p" >> $out; cat $R/gpurun_out/vox_p${p}_kernels.txt >> $out
done
