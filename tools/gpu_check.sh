#!/bin/bash
# usage (GPU box, repo root): tools/gpu_check.sh <tag>  -- the whole GPU suite, the default bench line, the step's kernel list
tag=${1:-chk}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${tag}_tests.log
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
PROF_TOP=70 PROF_FILTER=pd3 tools/gpu_prof.sh ${tag}_step bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/${tag}_prof.log 2>&1
cat gpurun_out/${tag}_tests.log
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "vox frac", d["roofline"]["frac"])
print(d["per_op_ms"])
print(d.get("extras",{}).get("repeat_blocks"))
PY
