#!/bin/bash
# two-level lane accounting with the leaf phase split by branch (libstats.so = -DPB_LANE_STATS build)
set -o pipefail
mkdir -p gpurun_out
{
echo "== lane stats config 5 (1080p x 8)"; INSTANCED=1 timeout -k 10 300 python tools/lane_stats.py pbrt-rs_amd/pbrt_hip/libstats.so || exit 1
echo "== lane stats config 3 (1080p x 8)"; timeout -k 10 300 python tools/lane_stats.py pbrt-rs_amd/pbrt_hip/libstats.so || exit 1
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r3_stats_${1:-run}.txt
