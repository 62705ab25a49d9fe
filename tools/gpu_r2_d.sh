#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3 || exit 1
{
echo "== wide (default build)"; SPP=${SPP:-32} timeout -k 10 200 python tools/trace_bench.py || exit 1
SPP=${SPP:-32} tools/sweep_prebuilt.sh run || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2d_bench.txt
