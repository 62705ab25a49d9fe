#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_render.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -5 || exit 1
{
echo "== config 5 geometry 1080p x 16 spp, wide two-level"; W=1920 H=1080 SPP=16 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== binary two-level"; PBRT_HIP_WIDE=0 W=1920 H=1080 SPP=16 timeout -k 10 300 python tools/bench_c5.py || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2i_c5.txt
