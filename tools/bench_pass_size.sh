#!/bin/bash
# bench.py at several pass sizes (samples of a pixel traced concurrently); run on the GPU box
for s in "$@"; do
  timeout -k 10 250 python bench.py --no-cpu-baseline --spp-per-pass $s 2>&1 | tail -1 > gpurun_out/_pp.json
  python3 - "$s" <<'PY'
import json, sys
d = json.load(open("gpurun_out/_pp.json"))
r = d["roofline"]
print("spp_per_pass", sys.argv[1], d["value"], "Mrays/s", d["ms_per_step"], "ms/step; k_trace", r["avg_launch_ms"], "ms x", r["launches_per_step"], "trace fraction", r["trace_fraction_of_step"], flush=True)
PY
done
