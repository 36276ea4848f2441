#!/bin/bash
# round 4, profile refresh after the ping-pong Winograd kernel went into the dense graph: full GPU suite, smoke, the
# default bench line, the step's kernel list under rocprofv3, PMC traffic
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | grep -v "^$" | tail -8 | tee gpurun_out/r04l_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r04l_smoke.log
python bench.py > gpurun_out/r04_bench_b16.json 2> gpurun_out/r04_bench.err
PROF_TOP=90 tools/gpu_prof.sh r04_bench_b16 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --repeats 0 > gpurun_out/r04_prof.log 2>&1
cp /tmp/prof_r04_bench_b16/r04_bench_b16_kernel_stats.csv gpurun_out/ 2>/dev/null
tools/gpu_traffic.sh r04_b16 16 30000 > gpurun_out/r04_traffic.log 2>&1
tail -5 gpurun_out/r04_traffic.log
python - <<PY
import json
d=json.load(open("gpurun_out/r04_bench_b16.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "vox frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
print(d["per_op_ms"]); e=d["extras"]; print(e["repeat_blocks"])
print({k: (v.get("value"), v.get("error")) for k, v in e["other_workloads"].items()})
PY
head -14 gpurun_out/r04_bench_b16_kernels.txt
