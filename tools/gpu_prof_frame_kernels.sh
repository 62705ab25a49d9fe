cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/pr
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pr -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pr.log 2>&1 || { tail -5 gpurun_out/pr.log; exit 1; }
f=$(find gpurun_out/pr -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if any(k in n for k in ("k_shade<", "k_trace_wide<false", "k_generate", "k_film_accumulate")):
        print("   ", n.split("(")[0][-44:], row["Calls"], "avg ms", round(float(row["AverageNs"]) / 1e6, 3))
PY
find gpurun_out/pr -name "*.csv" -size +1M -delete
