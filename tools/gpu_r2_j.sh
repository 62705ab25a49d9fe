#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
echo "== config 5 geometry 1080p x 16 spp, wide two-level (4 waves)"; W=1920 H=1080 SPP=16 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== wide two-level, 5 waves"; PBRT_LIB=$PWD/pbrt-rs_amd/pbrt_hip/libvar_1.so W=1920 H=1080 SPP=16 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== binary two-level"; PBRT_HIP_WIDE=0 W=1920 H=1080 SPP=16 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== hlbvh device-built scene, wide vs binary"; timeout -k 10 300 python tools/bench_variants.py 2>&1 | tail -3
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2j_c5.txt
timeout -k 10 600 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
