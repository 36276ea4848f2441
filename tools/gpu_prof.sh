#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_prof.sh <tag> <python script + args...>
# runs the command under `rocprofv3 --kernel-trace --stats` from /tmp and writes gpurun_out/<tag>_kernels.txt
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p /tmp/prof_$tag $R/gpurun_out
cd /tmp
# a script path relative to the repo root still resolves from /tmp
if [ -f "$R/$1" ]; then set -- "$R/$1" "${@:2}"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python "$@" > /tmp/prof_$tag/run.log 2>&1
grep -v "rocprofv3\|amdgpu.ids" /tmp/prof_$tag/run.log | tail -4 | cut -c1-300
python $R/tools/prof_summary.py /tmp/prof_$tag/${tag}_kernel_stats.csv ${PROF_TOP:-24} $R/gpurun_out/${tag}_kernels.txt "${PROF_FILTER:-}"
