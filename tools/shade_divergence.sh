#!/bin/bash
# What does k_shade's per-lane branch on mat.type cost on config 5, and what does shading in material order buy?
# Config 5 geometry at 1920x1080 x 16 spp, depth 16 (tools/bench_c5.py) with PbrtRenderParams.shade_order = 0 (queue
# order), 1 (by material inside each block, LDS counting sort) and 2 (whole queue radix-sorted): frame time, per-kernel
# times (rocprofv3 --kernel-trace --stats), VALU lane utilisation and waiting of k_shade (own --pmc pass).
# Films must be bit-identical (also a -m gpu test: test_shade_order_by_material_keeps_the_film).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export W=1920 H=1080 SPP=16
OUT=gpurun_out/shade_div; rm -rf $OUT; mkdir -p $OUT
for MODE in 0 1 2; do
  export SHADE_ORDER=$MODE
  echo "===== shade_order=$MODE"
  timeout -k 10 300 python3 tools/bench_c5.py 2>&1 | grep -v amdgpu.ids | tee $OUT/run$MODE.txt || exit 1
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$MODE -o s -- python3 tools/bench_c5.py > $OUT/stats$MODE.log 2>&1 || exit 1
  python3 - $OUT/stats$MODE <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f, newline="")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    n = r["Name"]
    short = "k_trace" if "k_trace" in n else ("k_shade_sort_keys" if "k_shade_sort_keys" in n else ("k_shade" if "k_shade" in n else ("radix sort" if "radix" in n else None)))
    if short: print(f"   {short:18s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e6:8.3f} ms  {float(r['TotalDurationNs'])/tot*100:5.1f} % of GPU time")
PY
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc$MODE -o p -- python3 tools/bench_c5.py > $OUT/pmc$MODE.log 2>&1 || exit 1
  python3 tools/pmc_summary.py $OUT/pmc$MODE | grep -A8 "k_shade"
done
python3 - <<'PY'
import os, sys
sys.path.insert(0, "pbrt-rs_amd")
import numpy as np, pbrt_hip
from pbrt_hip import scenes
sc = scenes.instanced_scene(10_000, 1000); cam = scenes.instanced_camera(640, 360)
ctx = pbrt_hip.Context(0); g = pbrt_hip.Scene(ctx, sc)
a, _ = g.render(cam, 640, 360, 8, max_depth=16, seed=1, shade_order=0)
b, _ = g.render(cam, 640, 360, 8, max_depth=16, seed=1, shade_order=1)
c, _ = g.render(cam, 640, 360, 8, max_depth=16, seed=1, shade_order=2)
print("films bit-identical in queue order / by material inside blocks / sorted queue:", a.tobytes() == b.tobytes() == c.tobytes())
PY
find $OUT -name "*.csv" -size +1M -delete
