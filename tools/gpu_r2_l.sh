#!/bin/bash
# config 5 geometry: wide two-level vs binary two-level (INST = 1 restored), then PMC of both, then the new wide tests
set -o pipefail
mkdir -p gpurun_out
{
echo "== wide two-level"; W=1920 H=1080 SPP=16 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== wide two-level, 5 waves"; PBRT_LIB=$PWD/pbrt-rs_amd/pbrt_hip/libvar_1.so W=1920 H=1080 SPP=16 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== binary two-level"; PBRT_HIP_WIDE=0 W=1920 H=1080 SPP=16 timeout -k 10 300 python tools/bench_c5.py || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2l_c5.txt
timeout -k 10 600 python -m pytest tests/test_gpu_wide.py -m gpu -x -q 2>&1 | tail -15 || exit 1
SCRIPT=tools/bench_c5.py bash tools/pmc_trace.sh c5_wide W=1920 H=1080 SPP=8 NO_COUNT=1 > gpurun_out/r2l_pmc_wide.txt 2>&1 || { tail -5 gpurun_out/r2l_pmc_wide.txt; exit 1; }
SCRIPT=tools/bench_c5.py bash tools/pmc_trace.sh c5_binary W=1920 H=1080 SPP=8 NO_COUNT=1 PBRT_HIP_WIDE=0 > gpurun_out/r2l_pmc_binary.txt 2>&1 || { tail -5 gpurun_out/r2l_pmc_binary.txt; exit 1; }
echo pmc done
