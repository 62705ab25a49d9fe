#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_golden.py tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -3 || exit 1
{
echo "== wide (default build)"; timeout -k 10 200 python tools/trace_bench.py || exit 1
tools/sweep_prebuilt.sh run || exit 1
echo "== 64 spp in one pass"; SPP=64 timeout -k 10 300 python tools/trace_bench.py || exit 1
python tools/lane_stats.py $PWD/pbrt-rs_amd/pbrt_hip/libstats.so
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c_bench.txt
