#!/bin/bash
# after the control-flow work: full GPU suite, PMC of k_trace_wide on config 3, lane statistics, config 5 (INST threshold 24)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 || exit 1
bash tools/pmc_trace.sh r2_final SPP=64 > gpurun_out/r2o_pmc.txt 2>&1 || { tail -5 gpurun_out/r2o_pmc.txt; exit 1; }
{
echo "== lane stats config 3"; SPP=16 timeout -k 10 300 python tools/lane_stats.py pbrt-rs_amd/pbrt_hip/libstats.so || exit 1
echo "== lane stats config 5"; INSTANCED=1 SPP=8 timeout -k 10 300 python tools/lane_stats.py pbrt-rs_amd/pbrt_hip/libstats.so || exit 1
echo "== config 5: default"; W=1920 H=1080 SPP=16 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== config 5 at its stated size (3840x2160, 4 spp per pass x 2)"; W=3840 H=2160 SPP=8 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r2o.txt
