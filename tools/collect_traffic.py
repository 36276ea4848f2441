"""Aggregate the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a bench.py run into per-op HBM
traffic per launch -> profiles/<tag>_traffic.json (read back by bench.py to fill `roofline.traffic`).

    usage: collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps> <out.json> <batch> <max_voxels>

Units / corrections (MI355X_MICROARCH.md section HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half of a streamed read -- calibrated on vt_route_kernel, whose only HBM reads are
the points (batch * 6.0 MB): the x2 factor reproduces that byte count to 2 %; WRITE_SIZE of the output writer
(vt_rows_kernel) matches its known 12.5 MB per scene to 3 % without correction.  Per-kernel rows are kept next to
the per-op sums (`kernels`).
"""
import csv
import json
import sys
from collections import defaultdict

OPS = {
    "hard_voxelize": ("vw_route", "vw_group", "vw_assign", "vw_rows", "vw_finish_index",
                      "vt_route", "vt_group", "vt_assign", "vt_rows", "cell_key", "seg_head",
                      "gather_voxels", "voxel_meta", "EpiVoxelStart", "LoadNonNegative"),
    "pillar_feature_net": ("pfn_",),
    "pointpillars_scatter": ("fill_i32", "inverse_map", "canvas_write"),
    "centerpoint_postprocess": ("cp_decode", "cp_score", "cp_topk", "cp_nms_boxes", "cp_output", "nms_mask", "nms_cand",
                                "nms_pairs", "nms_sweep"),
}
# the radix sort / scan kernels are shared: with the tiled voxelizer active they belong to the postprocess
SHARED = ("rs_hist", "rs_scatter", "scan_reduce", "scan_partials", "scan_apply")


def load(path):
    """kernel -> (sum of the counter over its dispatches, dispatches).  A dispatch appears once per counter row;
    with a derived counter rocprofv3 may write several rows per dispatch (one per dimension), so dispatches are
    told apart by Dispatch_Id."""
    acc, ids = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]] += float(r["Counter_Value"])
        ids[r["Kernel_Name"]].add(r.get("Dispatch_Id", len(ids[r["Kernel_Name"]])))
    return acc, {k: len(v) for k, v in ids.items()}


def main():
    (fetch, nf), (write, _nw) = load(sys.argv[1]), load(sys.argv[2])
    # steps of the profiled run = dispatches of a kernel that runs once per step (the route kernel of hard_voxelize);
    # the command-line count (steps + warmup) is only the fallback: bench.py also runs a few untimed set-up steps
    once = [n for k, n in nf.items() if "route_kernel" in k]
    launches = max(once) if once else int(sys.argv[3])  # (a one-frame probe may launch another tile shape once)
    out = {"batch": int(sys.argv[5]), "max_voxels": int(sys.argv[6]), "launches_profiled": launches,
           "front": sys.argv[7] if len(sys.argv) > 7 else "pair",
           "note": "bytes per launch (= per bench step) ; fetch corrected x2 (gfx950 FETCH_SIZE), write uncorrected"}
    tiled = any("vt_route" in k or "vw_route" in k for k in fetch)
    ops = dict(OPS)
    owner = "centerpoint_postprocess" if tiled else "hard_voxelize"
    shared = SHARED
    if any("pc_flags" in k for k in fetch):
        # round 6: the first backbone layer runs as a sparse convolution over the occupied pillars and the step's scans are
        # its rulebook's (pillar_conv.hip); neither the wave-form voxelizer nor the post-processing launches one
        shared = tuple(p for p in SHARED if not p.startswith("scan_"))
    ops[owner] = ops[owner] + shared
    for op, pats in ops.items():
        f = sum(v for k, v in fetch.items() if any(p in k for p in pats))
        w = sum(v for k, v in write.items() if any(p in k for p in pats))
        out[op] = {"fetch_size_kib_raw": f / launches, "write_size_kib": w / launches,
                   "bytes_per_launch": (2.0 * f + w) * 1024.0 / launches}
    out["kernels"] = {k[:60]: {"fetch_bytes_x2": 2.0 * fetch.get(k, 0.0) * 1024.0 / launches,
                               "write_bytes": write.get(k, 0.0) * 1024.0 / launches}
                      for k in sorted(set(fetch) | set(write)) if "pd3" in k or "vt_" in k or "vw_" in k}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))


if __name__ == "__main__":
    main()
