#!/bin/bash
# round 2, first GPU pass over the 4-wide records: parity, then trace-only throughput with and without them
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r2a_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r2a_pytest.txt
tail -5 gpurun_out/r2a_pytest.txt
grep -q "pytest rc 0" gpurun_out/r2a_pytest.txt || exit 1
{
echo "== wide (default build)"; timeout -k 10 200 python tools/trace_bench.py || exit 1
echo "== binary records (PBRT_HIP_WIDE=0)"; PBRT_HIP_WIDE=0 timeout -k 10 200 python tools/trace_bench.py || exit 1
tools/sweep_prebuilt.sh run || exit 1
echo "== 64 spp in one pass"; SPP=64 timeout -k 10 300 python tools/trace_bench.py || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2a_bench.txt
