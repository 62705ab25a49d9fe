#!/bin/bash
# packed-fp32 child tests (default) vs scalar (libvar_0), config 3 and config 5 (wide INST at 5 waves now; libvar_1 = 4 waves)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_wide.py tests/test_gpu_intersect.py -m gpu -x -q 2>&1 | tail -8 || exit 1
{
echo "== config 3, 64 spp: packed"; SPP=64 timeout -k 10 300 python tools/trace_bench.py || exit 1
echo "== config 3, 64 spp: scalar"; SPP=64 timeout -k 10 300 python tools/trace_bench.py pbrt-rs_amd/pbrt_hip/libvar_0.so || exit 1
echo "== config 3, 64 spp: packed again"; SPP=64 timeout -k 10 300 python tools/trace_bench.py || exit 1
echo "== config 5: wide two-level (5 waves, packed)"; W=1920 H=1080 SPP=16 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== config 5: wide two-level, 4 waves"; PBRT_LIB=$PWD/pbrt-rs_amd/pbrt_hip/libvar_1.so W=1920 H=1080 SPP=16 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== config 5: wide, scalar"; PBRT_LIB=$PWD/pbrt-rs_amd/pbrt_hip/libvar_0.so W=1920 H=1080 SPP=16 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== config 5: binary two-level"; PBRT_HIP_WIDE=0 W=1920 H=1080 SPP=16 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2n.txt
