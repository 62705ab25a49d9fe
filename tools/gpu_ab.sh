#!/bin/bash
# A/B inside one call (one box): the built library against pbrt-rs_amd/pbrt_hip/libvar_*.so, config 3 (64 spp) and
# config 5's geometry (1080p x 16 spp), after the wide / intersect parity tests. usage: tools/gpu_ab.sh [tag]
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_wide.py tests/test_gpu_intersect.py -m gpu -x -q 2>&1 | tail -8 || exit 1
{
echo "== config 3, 64 spp: default"; SPP=64 timeout -k 10 300 python tools/trace_bench.py || exit 1
for v in pbrt-rs_amd/pbrt_hip/libvar_*.so; do
  [ -e "$v" ] || continue
  echo "== config 3, 64 spp: $v"; SPP=64 timeout -k 10 300 python tools/trace_bench.py $v || exit 1
done
echo "== config 3, 64 spp: default again"; SPP=64 timeout -k 10 300 python tools/trace_bench.py || exit 1
echo "== config 5: default"; W=1920 H=1080 SPP=16 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
for v in pbrt-rs_amd/pbrt_hip/libvar_*.so; do
  [ -e "$v" ] || continue
  echo "== config 5: $v"; PBRT_LIB=$PWD/$v W=1920 H=1080 SPP=16 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
done
echo "== config 5: binary two-level"; PBRT_HIP_WIDE=0 W=1920 H=1080 SPP=16 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
} 2>&1 | grep -v "amdgpu.ids\|scene + host\|wide records" | tee gpurun_out/ab_${1:-run}.txt
