#!/bin/bash
# temporary: group-kernel variants with parts switched off (results are wrong on purpose; times only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/vox_dbg.txt; : > $out
for d in 0 1 2 3 4 12 28 60 32; do
  echo "# dbg $d" >> $out
  PD3_VW_DBG=$d PROF_FILTER=pd3::v PROF_TOP=4 timeout 300 tools/gpu_prof.sh vox_d$d tools/prof/prof_voxelize.py 16 30000 20 ${1:-8} > /dev/null 2>&1
  cat $R/gpurun_out/vox_d${d}_kernels.txt >> $out
done
