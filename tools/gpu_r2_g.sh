#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
for rep in 1 2; do
echo "== wide, breadth-first record order"; SPP=64 timeout -k 10 200 python tools/trace_bench.py || exit 1
echo "== wide, depth-first record order"; PBRT_HIP_WIDE_ORDER=dfs SPP=64 timeout -k 10 200 python tools/trace_bench.py || exit 1
done
echo "== depth-first + 8 queue segments"; PBRT_HIP_WIDE_ORDER=dfs PBRT_HIP_SEGMENTS_ALL=1 SPP=64 timeout -k 10 200 python tools/trace_bench.py || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2g_bench.txt
PBRT_HIP_WIDE_ORDER=dfs timeout -k 10 300 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3
