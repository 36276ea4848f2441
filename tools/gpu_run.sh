#!/bin/bash
# The one GPU-box run script (replaces the per-run gpu_r0N*.sh / gpu_check.sh / gpu_round_end.sh of rounds 3-4).
# usage (GPU box, repo root):  tools/gpu_run.sh <tag> <stage> [<stage> ...]
# stages, run in the order given; every output lands under gpurun_out/<tag>_* (copy what is to be judged to profiles/):
#   tests[=<pytest -k expression>]   the GPU suite (or the selected tests), tail in <tag>_tests.log
#   smoke                            __graft_entry__.smoke()
#   bench[=<workload>]               the bench line (default workload, or --workload <w>) -> <tag>_bench[_<w>].json
#   prof[=<workload>]                rocprofv3 --kernel-trace --stats of 10 steps -> <tag>_step[_<w>]_kernels.txt
#   traffic                          two --pmc passes (FETCH_SIZE / WRITE_SIZE) over the default step
#   vox[=<paths>]                    hard_voxelize alone per path (tools/gpu_vox.sh, no pytest) -> <tag>_vox_paths.txt
#   py=<script and args, '+' for spaces>   any tools/prof script, output in <tag>_<script>.txt
#   sh=<command, '+' for spaces>     anything else, output in <tag>_sh.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for stage in "$@"; do
  name=${stage%%=*}; arg=""; [ "$stage" != "$name" ] && arg=${stage#*=}
  echo "== $stage"
  case $name in
    tests)
      if [ -n "$arg" ]; then timeout 1800 python -m pytest tests -m gpu -q -rf --tb=short -k "$arg" > gpurun_out/${tag}_tests_full.log 2>&1
      else timeout 1800 python -m pytest tests -m gpu -q -rf --tb=short > gpurun_out/${tag}_tests_full.log 2>&1; fi
      grep -v "^$" gpurun_out/${tag}_tests_full.log | grep -E "^E |^FAILED|^ERROR|passed|failed|Error" | tail -40 | tee gpurun_out/${tag}_tests.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${tag}_smoke.log ;;
    bench)
      sfx=""; wl=""; [ -n "$arg" ] && sfx=_$arg && wl="--workload $arg"
      timeout 900 python bench.py $wl > gpurun_out/${tag}_bench${sfx}.json 2> gpurun_out/${tag}_bench${sfx}.err
      python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${tag}_bench${sfx}.json"))
    print("value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "roofline", d["roofline"].get("frac"), "traffic", d["roofline"].get("traffic"))
    print(d.get("per_op_ms"))
    e = d.get("extras", {})
    print({k: e[k] for k in ("repeat_blocks", "h2d_inclusive", "h2d_overlapped") if k in e})
    print({k: (v.get("value"), v.get("error")) for k, v in e.get("other_workloads", {}).items()})
    mp = e.get("map_proxy", {})
    print("map_proxy", {k: mp.get(k) for k in ("value", "reverse", "classes_scored")})
except Exception as ex:
    print("bench line unreadable:", ex); print(open("gpurun_out/${tag}_bench${sfx}.err").read()[-2000:])
PY
      ;;
    prof)
      sfx=""; wl=""; [ -n "$arg" ] && sfx=_$arg && wl="--workload $arg"
      PROF_TOP=90 tools/gpu_prof.sh ${tag}_step${sfx} bench.py $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras --repeats 0 > gpurun_out/${tag}_prof${sfx}.log 2>&1
      cp /tmp/prof_${tag}_step${sfx}/${tag}_step${sfx}_kernel_stats.csv gpurun_out/ 2>/dev/null
      head -16 gpurun_out/${tag}_step${sfx}_kernels.txt ;;
    traffic)
      tools/gpu_traffic.sh ${tag}_b16 16 30000 pair > gpurun_out/${tag}_traffic.log 2>&1; tail -5 gpurun_out/${tag}_traffic.log
      tools/gpu_traffic.sh ${tag}_b16_fused 16 30000 fused >> gpurun_out/${tag}_traffic.log 2>&1 ;;
    vox)
      tools/gpu_vox.sh ${arg:-5,11} 0 > /dev/null 2>&1; cp gpurun_out/vox_paths.txt gpurun_out/${tag}_vox_paths.txt
      grep -v "^#" gpurun_out/${tag}_vox_paths.txt | head -30 ;;
    py)
      cmd=${arg//+/ }; base=$(basename ${cmd%% *} .py)
      timeout 900 python $cmd 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/${tag}_${base}.txt | tail -40 ;;
    sh)
      cmd=${arg//+/ }
      timeout 900 bash -c "$cmd" 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/${tag}_sh.txt | tail -40 ;;
    *) echo "unknown stage $stage" ;;
  esac
done
