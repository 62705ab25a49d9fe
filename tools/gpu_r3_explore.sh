#!/bin/bash
# round 3, call 1: suite at HEAD, config 5 with top-level leaves of 4 / 2 / 1 instances, two-level lane accounting
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3_suite.txt || { cat gpurun_out/r3_suite.txt; exit 1; }
cat gpurun_out/r3_suite.txt
{
for mp in 4 2 1; do
  echo "== config 5 4K x 8 spp, top-level max prims $mp"; TLAS_MAX_PRIMS=$mp W=3840 H=2160 SPP=8 timeout -k 10 300 python tools/bench_c5.py || exit 1
done
for mp in 4 1; do
  echo "== lane stats config 5 (1080p x 8), top-level max prims $mp"; TLAS_MAX_PRIMS=$mp INSTANCED=1 timeout -k 10 300 python tools/lane_stats.py pbrt-rs_amd/pbrt_hip/libstats.so || exit 1
done
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r3_explore.txt
