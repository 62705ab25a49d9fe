#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err || { tail -20 gpurun_out/r2f_bench.err; exit 1; }
cat gpurun_out/r2f_bench.json
PB_COMMIT=$1 bash tools/measure_traffic.sh || exit 1
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r2f_bench2.json 2> gpurun_out/r2f_bench2.err || { tail -20 gpurun_out/r2f_bench2.err; exit 1; }
cat gpurun_out/r2f_bench2.json
