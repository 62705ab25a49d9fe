#!/bin/bash
# round 3: parity tests, then the same-box A/B of tools/gpu_ab.sh (config 3 at 64 spp, config 5 at 4K x 8 spp here), then
# the two-level lane accounting. usage: tools/gpu_ab.sh tag [extra pytest files]
set -o pipefail
tag=${1:-run}; shift
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_intersect.py "$@" -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r3_ab_tests_$tag.txt
grep -q " passed" gpurun_out/r3_ab_tests_$tag.txt && ! grep -q "failed\|error" gpurun_out/r3_ab_tests_$tag.txt || exit 1
{
echo "== config 3, 64 spp: default"; SPP=64 timeout -k 10 300 python tools/trace_bench.py || exit 1
for v in pbrt-rs_amd/pbrt_hip/libvar_*.so; do
  [ -e "$v" ] || continue
  echo "== config 3, 64 spp: $v"; SPP=64 timeout -k 10 300 python tools/trace_bench.py $v || exit 1
done
echo "== config 3, 64 spp: default again"; SPP=64 timeout -k 10 300 python tools/trace_bench.py || exit 1
echo "== config 5 (4K x 8 spp): default"; W=3840 H=2160 SPP=8 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
for v in pbrt-rs_amd/pbrt_hip/libvar_*.so; do
  [ -e "$v" ] || continue
  echo "== config 5 (4K x 8 spp): $v"; PBRT_LIB=$PWD/$v W=3840 H=2160 SPP=8 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
done
echo "== config 5 (4K x 8 spp): default again"; W=3840 H=2160 SPP=8 timeout -k 10 300 python tools/bench_c5.py || exit 1
if [ -e pbrt-rs_amd/pbrt_hip/libstats.so ]; then
  echo "== lane stats config 5 (1080p x 8)"; INSTANCED=1 timeout -k 10 300 python tools/lane_stats.py pbrt-rs_amd/pbrt_hip/libstats.so || exit 1
  echo "== lane stats config 3 (1080p x 8)"; timeout -k 10 300 python tools/lane_stats.py pbrt-rs_amd/pbrt_hip/libstats.so || exit 1
fi
} 2>&1 | grep -v "amdgpu.ids\|scene + host\|wide records:" | tee gpurun_out/r3_ab_$tag.txt
