#!/bin/bash
# round 5: the whole GPU suite, then the N = 1 frame under both tile orders, then every rank's share of configs 4 and 5 under
# both dealing orders (tools/probe_rank_of_world.py). usage: tools/gpu_r5_check.sh tag
set -o pipefail
tag=${1:-run}
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r5_suite_$tag.txt
grep -q " passed" gpurun_out/r5_suite_$tag.txt && ! grep -q "failed\|error" gpurun_out/r5_suite_$tag.txt || exit 1
for order in morton row-major; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --tile-order $order > gpurun_out/r5_n1_${order}_$tag.json 2> gpurun_out/r5_n1_${order}_err_$tag.txt || { tail -20 gpurun_out/r5_n1_${order}_err_$tag.txt; exit 1; }
  python - <<PY
import json
l = json.loads([x for x in open("gpurun_out/r5_n1_${order}_$tag.json") if x.startswith("{")][-1])
print("$order:", l["value"], l["unit"], "ms/step", l["ms_per_step"], "trace ms/launch", l["roofline"]["avg_launch_ms"], "frac", l["roofline"]["frac"])
PY
done
timeout -k 10 900 python tools/probe_rank_of_world.py 2>&1 | tee gpurun_out/r5_rank_of_world_$tag.txt
