#!/bin/bash
# round 5: BASELINE configs 4 and 5 at FULL size as two-rank jobs started the way the driver starts bench.py (no launcher), both
# ranks on the one GPU of the box, film reduce over gloo (RCCL refuses two ranks on one device): the N > 1 path end to end with the
# HIP kernels in it. A rehearsal, not a scaling number. usage: [RANKS=4 SPP_PASS4=128 SPP_PASS5=16] tools/gpu_two_rank_rehearsal.sh tag
# (RANKS processes share the ONE GPU's memory: give more than two ranks a pass size, the automatic one assumes a GPU of its own)
set -o pipefail
tag=${1:-run}
ranks=${RANKS:-2}
mkdir -p gpurun_out
for cfg in 4 5; do
  steps=2; [ $cfg = 5 ] && steps=1
  pass=${SPP_PASS4:--1}; [ $cfg = 5 ] && pass=${SPP_PASS5:--1}
  timeout -k 10 500 python bench.py --config $cfg --gpus $ranks --steps $steps --warmup 1 --spp-per-pass $pass --dist-backend gloo --one-gpu > gpurun_out/r5_two_ranks_c${cfg}_$tag.json 2> gpurun_out/r5_two_ranks_c${cfg}_err_$tag.txt || { tail -20 gpurun_out/r5_two_ranks_c${cfg}_err_$tag.txt; exit 1; }
  python - <<PY
import json
l = json.loads([x for x in open("gpurun_out/r5_two_ranks_c${cfg}_$tag.json") if x.startswith("{")][-1])
c = l["config"]
print("config $cfg, $ranks ranks on one GPU:", l["value"], l["unit"], "ms/step", l["ms_per_step"], "|", c["workload"][:60], "| launcher:", c["launcher"], "| tile order:", c["tile_order"],
      "| balance", c["load_balance_max_over_mean"], "| ranks' render ms", [r["render_ms_per_step"]["mean"] for r in c["ranks"]], "| kernel", l["roofline"]["kernel"])
PY
done
