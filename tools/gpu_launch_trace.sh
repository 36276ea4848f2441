#!/bin/bash
# usage (GPU box, repo root): tools/gpu_launch_trace.sh <tag> <bench.py arguments...>
# Per-LAUNCH durations of one timed step of bench.py (rocprofv3 --kernel-trace): the kernels of the last step in launch
# order with their durations -> gpurun_out/<tag>_launches.txt (the per-kernel averages hide which layer a launch was).
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p /tmp/lt_$tag $R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/lt_$tag -o $tag -- python $R/bench.py "$@" --no-cpu-baseline --no-extras --repeats 0 > /tmp/lt_$tag/run.log 2>&1
python - /tmp/lt_$tag/${tag}_kernel_trace.csv $R/gpurun_out/${tag}_launches.txt "$*" <<'PY'
import csv, os, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last step = from the last launch of the step's first kernel
first = None
names = [r["Kernel_Name"] for r in rows]
anchor = os.environ.get("LT_ANCHOR", "vw_route")
starts = [i for i, n in enumerate(names) if anchor in n] + [len(rows)]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
import os
whole = [s for s in steps if any(os.environ.get("LT_END", "cp_output") in r["Kernel_Name"] for r in s)]
for i, st in enumerate(whole):
    print("step %d: %d launches, span %.1f us" % (i, len(st), (int(st[-1]["End_Timestamp"]) - int(st[0]["Start_Timestamp"])) / 1e3))
step = whole[int(os.environ.get("LT_STEP", "-1"))]  # LT_STEP picks one (default: the last whole step)
t0 = int(step[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("# bench.py %s: the launches of the last step, in order (us since the step's first launch, duration us)\n" % sys.argv[3])
    tot = 0.0
    for r in step:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        f.write("%9.1f %8.1f  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, r["Kernel_Name"][:110]))
    f.write("# sum of durations %.1f us; span %.1f us\n" % (tot, (int(step[-1]["End_Timestamp"]) - t0) / 1e3))
PY
