#!/bin/bash
# builds the micro-benchmarks next to their sources (gfx950); run on the GPU box or here (cross-compile)
cd "$(dirname "$0")"
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$f" -o "${f%.hip}" || exit 1
done
