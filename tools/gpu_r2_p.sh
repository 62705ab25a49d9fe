#!/bin/bash
# final round-2 pass: full GPU suite, stamped traffic profile + bench line, HLBVH scene-setup times with / without wide records
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 || exit 1
bash tools/measure_traffic.sh > gpurun_out/traffic_run.log 2>&1 || { tail -5 gpurun_out/traffic_run.log; exit 1; }
timeout -k 10 600 python bench.py > gpurun_out/bench_final.log 2>&1 || { tail -5 gpurun_out/bench_final.log; exit 1; }
grep '^{' gpurun_out/bench_final.log | cut -c1-300
{
echo "== HLBVH scene setup, wide records built (default)"; timeout -k 10 300 python tools/bench_hlbvh.py 1000000 10000000 || exit 1
echo "== HLBVH scene setup, PBRT_HIP_WIDE_DEVICE_TREES=0"; PBRT_HIP_WIDE_DEVICE_TREES=0 timeout -k 10 300 python tools/bench_hlbvh.py 1000000 10000000 || exit 1
echo "== config 5, 1080p x 16 spp"; W=1920 H=1080 SPP=16 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== config 5, 3840x2160 x 8 spp"; W=3840 H=2160 SPP=8 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r2p.txt
