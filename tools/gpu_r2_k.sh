#!/bin/bash
# A/B in one call (one box): config 5 geometry through today's wide two-level kernel, today's binary kernel, and the
# binary kernel of commit 5cc51f7 (before the general two-level change; tree extracted into _old/ for this run only).
set -o pipefail
mkdir -p gpurun_out
{
echo "== wide two-level (now)"; W=1920 H=1080 SPP=16 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== binary two-level (now)"; PBRT_HIP_WIDE=0 W=1920 H=1080 SPP=16 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== binary two-level (5cc51f7)"; (cd _old && mkdir -p tools && cp ../tools/bench_c5.py tools/ && W=1920 H=1080 SPP=16 timeout -k 10 300 python tools/bench_c5.py) || exit 1
echo "== wide two-level (now), again"; W=1920 H=1080 SPP=16 timeout -k 10 300 python tools/bench_c5.py || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2k_c5.txt
timeout -k 10 600 python -m pytest tests/test_gpu_wide.py -m gpu -x -q 2>&1 | tail -15
