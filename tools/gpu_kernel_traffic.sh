#!/bin/bash
# usage (GPU box, repo root): tools/gpu_kernel_traffic.sh <tag> <kernel substrings, comma separated> <bench.py arguments...>
# HBM-side traffic of named kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: counters only, no trace
# domains) over `bench.py <arguments>`, per-dispatch averages per matching kernel -> gpurun_out/<tag>_kernel_traffic.json
# (fetch x 2 per the gfx950 note of MI355X_MICROARCH.md, write uncorrected).
tag=$1; pats=$2; shift; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p /tmp/kt_$tag $R/gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/kt_$tag -o ${tag}_$c -- python $R/bench.py "$@" --no-cpu-baseline --no-extras --repeats 0 > /tmp/kt_$tag/run_$c.log 2>&1
done
python - "$pats" /tmp/kt_$tag/${tag}_FETCH_SIZE_counter_collection.csv /tmp/kt_$tag/${tag}_WRITE_SIZE_counter_collection.csv $R/gpurun_out/${tag}_kernel_traffic.json "$*" <<'PY'
import csv, json, sys
from collections import defaultdict
pats = sys.argv[1].split(",")
per = {}
def load(path, which):
    acc, n = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if any(p in k for p in pats):
            acc[k] += float(r["Counter_Value"]); n[k].add(r.get("Dispatch_Id"))
            d = per.setdefault(k, {}).setdefault(which, defaultdict(float))
            d[r.get("Dispatch_Id")] += float(r["Counter_Value"])
    return acc, {k: len(v) for k, v in n.items()}
(f, nf), (w, nw) = load(sys.argv[2], "f"), load(sys.argv[3], "w")
out = {"command": "bench.py " + sys.argv[5], "note": "bytes per dispatch, averaged over the run's dispatches; fetch x 2 (gfx950 FETCH_SIZE), write uncorrected", "kernels": {}}
for k in sorted(set(f) | set(w)):
    fb = 2.0 * f.get(k, 0.0) * 1024.0 / max(1, nf.get(k, 1)); wb = w.get(k, 0.0) * 1024.0 / max(1, nw.get(k, 1))
    out["kernels"][k[:90]] = {"dispatches": nf.get(k, nw.get(k, 0)), "fetch_bytes_x2": fb, "write_bytes": wb, "bytes_per_dispatch": fb + wb,
                              # the dispatches in launch order (a bench run may launch one kernel in several forms)
                              "fetch_bytes_x2_each": [round(2048.0 * v) for _, v in sorted(per.get(k, {}).get("f", {}).items(), key=lambda t: int(t[0]))][:40],
                              "write_bytes_each": [round(1024.0 * v) for _, v in sorted(per.get(k, {}).get("w", {}).items(), key=lambda t: int(t[0]))][:40]}
json.dump(out, open(sys.argv[4], "w"), indent=1); print(json.dumps(out, indent=1)[:3000])
PY
