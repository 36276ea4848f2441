#!/bin/bash
# usage (GPU box, repo root): tools/gpu_traffic.sh <tag> <batch> <max_voxels>
# two separate --pmc passes (counters only) over bench.py, aggregated into gpurun_out/<tag>_traffic.json
# 4th argument: the front half of the graph, "pair" (default: pd3_hard_voxelize + pd3_pillar_feature_net, the operators the
# bench line's roofline.traffic is about) or "fused" (the model path: index voxelizer + indexed PFN)
tag=$1; batch=${2:-8}; mv=${3:-30000}; front=${4:-pair}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p /tmp/tr_$tag $R/gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/tr_$tag -o ${tag}_$c -- python $R/bench.py --steps 3 --warmup 2 --batch $batch --max-voxels $mv --front $front --no-cpu-baseline --no-extras --repeats 0 > /tmp/tr_$tag/run_$c.log 2>&1
done
python $R/tools/collect_traffic.py /tmp/tr_$tag/${tag}_FETCH_SIZE_counter_collection.csv /tmp/tr_$tag/${tag}_WRITE_SIZE_counter_collection.csv 5 $R/gpurun_out/${tag}_traffic.json $batch $mv $front
