#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
echo "== wide (default build)"; timeout -k 10 200 python tools/trace_bench.py || exit 1
echo "== binary records (PBRT_HIP_WIDE=0)"; PBRT_HIP_WIDE=0 timeout -k 10 200 python tools/trace_bench.py || exit 1
tools/sweep_prebuilt.sh run || exit 1
echo "== 64 spp in one pass"; SPP=64 timeout -k 10 300 python tools/trace_bench.py || exit 1
echo "== 64 spp in one pass, binary"; PBRT_HIP_WIDE=0 SPP=64 timeout -k 10 300 python tools/trace_bench.py || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2b_bench.txt
timeout -k 10 300 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3
