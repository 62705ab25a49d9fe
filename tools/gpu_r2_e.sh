#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
echo "== wide"; SPP=64 timeout -k 10 200 python tools/trace_bench.py || exit 1
echo "== wide, 8 queue segments for every wavefront"; PBRT_HIP_SEGMENTS_ALL=1 SPP=64 timeout -k 10 200 python tools/trace_bench.py || exit 1
echo "== wide, unsorted queues"; PBRT_HIP_SORT_RAYS=0 SPP=64 timeout -k 10 200 python tools/trace_bench.py || exit 1
echo "== binary"; PBRT_HIP_WIDE=0 SPP=64 timeout -k 10 200 python tools/trace_bench.py || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2e_bench.txt
bash tools/pmc_trace.sh wide2 || exit 1
