#!/bin/bash
# round-end measurement run (GPU box): the whole GPU suite, the default bench line, the step's kernel list, PMC
# traffic, hard_voxelize alone on every path.  Outputs under gpurun_out/<tag>_*; copy what is to be judged to profiles/.
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${tag}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
python bench.py > gpurun_out/${tag}_bench_b16.json 2> gpurun_out/${tag}_bench.err
PROF_TOP=90 tools/gpu_prof.sh ${tag}_bench_b16 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --repeats 0 > gpurun_out/${tag}_prof.log 2>&1
cp /tmp/prof_${tag}_bench_b16/${tag}_bench_b16_kernel_stats.csv gpurun_out/ 2>/dev/null
tools/gpu_traffic.sh ${tag}_b16 16 30000 > gpurun_out/${tag}_traffic.log 2>&1
tools/gpu_vox.sh 3,5,6,7,8 0 > /dev/null 2>&1
cp gpurun_out/vox_paths.txt gpurun_out/${tag}_vox_paths.txt
cat gpurun_out/${tag}_tests.log gpurun_out/${tag}_smoke.log
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench_b16.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "vox frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
print(d["per_op_ms"]); print(d["extras"]["repeat_blocks"])
print({k: (v.get("value"), v.get("error")) for k, v in d["extras"]["other_workloads"].items()})
PY
