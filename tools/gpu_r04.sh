#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r04.sh <tag> [pytest -k expression]  -- GPU suite (all failures listed), default bench line
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
if [ -n "$2" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -s -k "$2" 2>&1 | tail -40 > gpurun_out/${tag}_tests.log
else
  timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 > gpurun_out/${tag}_tests.log
fi
cat gpurun_out/${tag}_tests.log
if [ -z "$NO_BENCH" ]; then
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -3 gpurun_out/${tag}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "vox frac", d["roofline"]["frac"])
print(d["per_op_ms"])
e=d.get("extras",{})
print(e.get("repeat_blocks"))
print("map_proxy", e.get("map_proxy"))
print("strong", e.get("strong_scaling"))
print({k: (v.get("value"), v.get("error")) for k, v in e.get("other_workloads",{}).items()})
PY
fi
