#!/bin/bash
# rounds 3-5: the whole GPU suite, the stamped counter profile (tools/measure_traffic.sh), then the bench line as the driver
# runs it (one process, then one rank under torch.distributed.run with RCCL). usage: tools/gpu_bench_check.sh tag
set -o pipefail
tag=${1:-run}
mkdir -p gpurun_out
if [ -z "$SKIP_SUITE" ]; then   # SKIP_SUITE=1: only the stamped profile and the bench lines (the suite ran on these sources already)
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r5_suite_$tag.txt
grep -q " passed" gpurun_out/r5_suite_$tag.txt && ! grep -q "failed\|error" gpurun_out/r5_suite_$tag.txt || exit 1
fi
PB_COMMIT=$(cat .pb_commit 2>/dev/null) timeout -k 10 1500 bash tools/measure_traffic.sh > gpurun_out/r5_measure_traffic_$tag.txt 2>&1 || { tail -20 gpurun_out/r5_measure_traffic_$tag.txt; exit 1; }
tail -3 gpurun_out/r5_measure_traffic_$tag.txt
timeout -k 10 600 python bench.py > gpurun_out/r5_bench_line_$tag.json 2> gpurun_out/r5_bench_err_$tag.txt || { tail -20 gpurun_out/r5_bench_err_$tag.txt; exit 1; }
python - <<PY
import json
l = json.loads([x for x in open("gpurun_out/r5_bench_line_$tag.json") if x.startswith("{")][-1])
r = l["roofline"]
print("bench:", l["value"], l["unit"], "ms/step", l["ms_per_step"], "| roofline:", r["bound"], "achieved", r["achieved"], "frac", r["frac"], "algorithmic_frac", r["algorithmic_frac"],
      "ta_busy", r["ta_busy"], "valu_issue", r["valu_issue"], "gather_frac", r["gather_frac"], "stale", r["stale"])
print("shade:", r.get("shade")); print("secondary:", l.get("secondary")); print("cpu:", l.get("cpu_baseline"))
PY
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --abi-reduce-check --abi-check-strict > gpurun_out/r5_bench_torchrun_$tag.json 2> gpurun_out/r5_bench_torchrun_err_$tag.txt || { tail -20 gpurun_out/r5_bench_torchrun_err_$tag.txt; exit 1; }
python - <<PY
import json
l = json.loads([x for x in open("gpurun_out/r5_bench_torchrun_$tag.json") if x.startswith("{")][-1])
print("torchrun 1 rank (nccl):", l["value"], l["unit"], l["config"]["dist_backend"], "| abi film reduce:", l["config"].get("abi_film_reduce")); print("libs:", l["config"]["runtime_libs"])
PY
