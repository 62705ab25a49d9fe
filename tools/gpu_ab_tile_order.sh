set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for order in row-major morton row-major morton; do
  rm -rf gpurun_out/ab_$order; 
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$order -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --tile-order $order > gpurun_out/ab_$order.log 2>&1 || { tail -5 gpurun_out/ab_$order.log; exit 1; }
  f=$(find gpurun_out/ab_$order -name "*kernel_stats.csv" | head -1)
  echo "== $order: $(grep -o '"value": [0-9.]*' gpurun_out/ab_$order.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_$order.log | head -1)"
  python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if any(k in n for k in ("k_shade<", "k_trace_wide<false", "k_generate", "k_film_accumulate")):
        print("   ", n.split("(")[0][-40:], row["Calls"], "avg ms", round(float(row["AverageNs"]) / 1e6, 3))
PY
  find gpurun_out/ab_$order -name "*.csv" -size +1M -delete
done
