#!/bin/bash
# round 4, sixth GPU call: assign with published prefixes (tests + times), then the round's profile set:
# default bench line, the step's kernel list under rocprofv3, PMC traffic, hard_voxelize alone per path
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_voxelize_gpu.py tests/test_properties_gpu.py -m gpu -q -rf 2>&1 | tail -6 | tee gpurun_out/r04f_tests.log
out=gpurun_out/r04_vox_paths.txt
echo "# clocks" > $out; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4 >> $out
echo "# C3: 16 frames, 0.2 m pillars (512 x 512), P = 20, V = 30000; path 0 = the library's choice (wave form + priorities), 5 = wave form without priorities, 3 = round 2's gather form, 12 = two half batches on two streams" >> $out
timeout 300 python tools/prof/prof_voxelize.py 16 30000 50 3,5,0,12,5,0 2>&1 | grep -v "^$\|amdgpu.ids" | tee -a $out
echo "# C3 shuffled points" >> $out
timeout 300 python tools/prof/prof_voxelize.py 16 30000 20 5,0 shuffle 2>&1 | grep -v "^$\|amdgpu.ids" | tee -a $out
PROF_FILTER=pd3 PROF_TOP=6 timeout 300 tools/gpu_prof.sh r04f_c3p0 tools/prof/prof_voxelize.py 16 30000 20 0 > /dev/null 2>&1
echo "# C3 path 0 per kernel" >> $out; cat gpurun_out/r04f_c3p0_kernels.txt >> $out
echo "# C4: 8 frames, 0.075 m voxels (1440 x 1440 x 40), P = 10, V = 160000; path 1 = sort path (row writer of the wave form since round 4), 14 = 3-D wave form (the library's choice), 15 / 16 = its route tile forced to 8192 / 10240 points" >> $out
timeout 300 python tools/prof/prof_voxelize.py 8 160000 20 1,14,15,16,14 c4 2>&1 | grep -v "^$\|amdgpu.ids" | tee -a $out
for p in 1 14; do
  PROF_FILTER=pd3 PROF_TOP=12 timeout 300 tools/gpu_prof.sh r04f_c4p$p tools/prof/prof_voxelize.py 8 160000 10 $p c4 > /dev/null 2>&1
  echo "# C4 path $p per kernel" >> $out; cat gpurun_out/r04f_c4p${p}_kernels.txt >> $out
done
python bench.py > gpurun_out/r04_bench_b16.json 2> gpurun_out/r04_bench.err
PROF_TOP=90 tools/gpu_prof.sh r04_bench_b16 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --repeats 0 > gpurun_out/r04_prof.log 2>&1
cp /tmp/prof_r04_bench_b16/r04_bench_b16_kernel_stats.csv gpurun_out/ 2>/dev/null
tools/gpu_traffic.sh r04_b16 16 30000 > gpurun_out/r04_traffic.log 2>&1
tail -12 gpurun_out/r04_traffic.log
python - <<PY
import json
d=json.load(open("gpurun_out/r04_bench_b16.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "vox frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
print(d["per_op_ms"]); e=d["extras"]; print(e["repeat_blocks"]); print("map", e.get("map_proxy",{}).get("value"), e.get("map_proxy",{}).get("reverse"), e.get("map_proxy",{}).get("oracle_detections"))
print({k: (v.get("value"), v.get("error")) for k, v in e["other_workloads"].items()})
print(e["other_workloads"].get("centerpoint_pillars_amp",{}).get("amp_error"))
PY
head -30 gpurun_out/r04_bench_b16_kernels.txt
