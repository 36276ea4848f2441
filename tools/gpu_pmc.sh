#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc.sh <tag> "<counters>" <python script + args...>
# one rocprofv3 --pmc pass (counters only, no trace domains), writes gpurun_out/<tag>_pmc.txt
tag=$1; ctrs=$2; shift; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p /tmp/pmc_$tag $R/gpurun_out
cd /tmp
if [ -f "$R/$1" ]; then set -- "$R/$1" "${@:2}"; fi   # a script path relative to the repo root still resolves from /tmp
rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$tag -o $tag -- python "$@" > /tmp/pmc_$tag/run.log 2>&1
tail -1 /tmp/pmc_$tag/run.log | cut -c1-200
python $R/tools/pmc_summary.py /tmp/pmc_$tag/${tag}_counter_collection.csv $R/gpurun_out/${tag}_pmc.txt
