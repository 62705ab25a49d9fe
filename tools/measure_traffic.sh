#!/bin/bash
# Writes profiles/r02_traffic.json (fabric-side bytes per launch of the dominant traversal kernel, for bench.py's
# roofline.traffic) and profiles/r02_bench_kernel_stats.csv from rocprofv3 runs of the default bench.py command.
# Three separate runs (--kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE), as gpurun requires.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/traffic
rm -rf $OUT; mkdir -p $OUT profiles
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
grep '^{' $OUT/stats.log | tail -1 > $OUT/bench_line_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o $C -- $CMD > $OUT/$C.log 2>&1 || { tail -5 $OUT/$C.log; exit 1; }
done
# the issue side of the same launches: vector instructions, the lane-cycles they kept busy, wave residency
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/SQ -o SQ -- $CMD > $OUT/SQ.log 2>&1 || { tail -5 $OUT/SQ.log; exit 1; }
# ... and how busy the texture addressers were (the unit that turns lane requests into cache accesses)
timeout -k 10 500 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/TA -o TA -- $CMD > $OUT/TA.log 2>&1 || { tail -5 $OUT/TA.log; exit 1; }
python3 - <<'PY'
import csv, glob, json, os, re, subprocess
out = "gpurun_out/traffic"
def per_kernel(counter):
    tot, n = {}, {}
    for f in glob.glob(f"{out}/{counter}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if row["Counter_Name"] != counter: continue
            k = row["Kernel_Name"]
            tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
    return tot, n
def per_kernel_in(sub, counter):
    tot, n = {}, {}
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if row["Counter_Name"] != counter: continue
            k = row["Kernel_Name"]
            tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
    return tot, n
fetch, nf = per_kernel("FETCH_SIZE"); write, nw = per_kernel("WRITE_SIZE")
# the timed launches: k_trace_wide<false> (or k_trace<false,false,false> on a scene without wide records)
def pick(d):
    ks = [k for k in d if re.search(r"k_trace_wide<false(, 0)?>|k_trace_wide<\(bool\)0", k)] or [k for k in d if re.search(r"k_trace<false, (false|0), false>", k)]
    return ks[0]
kf, kw = pick(fetch), pick(write)
fb, wb = fetch[kf] * 1024 / nf[kf], write[kw] * 1024 / nw[kw]
stats = glob.glob(f"{out}/stats/**/*kernel_stats.csv", recursive=True)
avg_ns = None
if stats:
    for row in csv.DictReader(open(stats[0], newline="")):
        if row["Name"] == kf: avg_ns = float(row["AverageNs"]); calls = int(row["Calls"])
    os.system(f"cp {stats[0]} profiles/r02_bench_kernel_stats.csv")
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("PB_COMMIT", "unknown")
sq = {}
for c in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE"):
    t, n = per_kernel_in("SQ", c)
    if kf in t: sq[c] = t[kf] / n[kf]
ta = {}
for c in ("TA_TA_BUSY_sum", "GRBM_GUI_ACTIVE"):
    t, n = per_kernel_in("TA", c)
    if kf in t: ta[c] = t[kf] / n[kf]
rec = {
    "ta_busy_fraction": (ta["TA_TA_BUSY_sum"] / 256 / (ta["GRBM_GUI_ACTIVE"] / 8)) if ta.get("GRBM_GUI_ACTIVE") else None,
    "valu_insts_per_launch": sq.get("SQ_INSTS_VALU"), "salu_insts_per_launch": sq.get("SQ_INSTS_SALU"),
    "valu_lane_utilisation": (sq["SQ_THREAD_CYCLES_VALU"] / (sq["SQ_INSTS_VALU"] * 64)) if sq.get("SQ_INSTS_VALU") else None,
    "wave_cycles_waiting_fraction": (sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_WAVE_CYCLES") else None,
    "gpu_cycles_per_launch": (sq["GRBM_GUI_ACTIVE"] / 8) if sq.get("GRBM_GUI_ACTIVE") else None,
    "bytes_per_launch": round(fb + wb), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
    "kernel_name": kf, "launches_profiled": nf[kf], "avg_launch_ns_under_kernel_trace": avg_ns,
    "config": {"n_gpus": 1, "tris": 1000000, "width": 1920, "height": 1080, "spp": 64, "max_depth": 5,
               "kernel": "k_trace_wide" if "wide" in kf else "k_trace"},
    "commit": os.environ.get("PB_COMMIT", commit),
    "profile": "profiles/r02_bench_kernel_stats.csv + rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on "
               "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`; FETCH_SIZE = TCC_EA0_RDREQ x 64 B (Infinity-Cache hits "
               "included), used without the x2 of wide coalesced streams: calibrated on this gather pattern in "
               "profiles/r01_fetch_size_calibration.txt (0.986)",
}
json.dump(rec, open("profiles/r02_traffic.json", "w"), indent=1)
json.dump(rec, open(f"{out}/r02_traffic.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
cp profiles/r02_bench_kernel_stats.csv $OUT/ 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete
