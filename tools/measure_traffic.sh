#!/bin/bash
# Writes profiles/r05_traffic.json — the stamped counter profile behind bench.py's roofline block: per launch of the dominant
# traversal kernel and of k_shade, fabric-side bytes (FETCH_SIZE + WRITE_SIZE), vector instructions and lane utilisation,
# texture-addresser busy cycles — and profiles/r05_bench_kernel_stats.csv, from rocprofv3 runs of the default bench.py
# command. Separate runs (--kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; SQ counters; TA), as gpurun requires.
# Stamped with the commit and with bench.py's hash of the kernel sources: bench.py flags the figures stale when they differ.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/traffic
rm -rf $OUT; mkdir -p $OUT profiles
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
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
ks = [k for k in fetch if "k_shade<" in k or re.search(r"k_shade\(", k)][0]   # the path integrator's shading kernel
stats = glob.glob(f"{out}/stats/**/*kernel_stats.csv", recursive=True)
avg = {}
if stats:
    for row in csv.DictReader(open(stats[0], newline="")):
        avg[row["Name"]] = (float(row["AverageNs"]), int(row["Calls"]))
    os.system(f"cp {stats[0]} profiles/r05_bench_kernel_stats.csv")
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("PB_COMMIT", "unknown")
import sys
sys.path.insert(0, ".")
import bench
def kernel_block(k):
    sq, ta = {}, {}
    for c in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE"):
        t, n = per_kernel_in("SQ", c)
        if k in t: sq[c] = t[k] / n[k]
    for c in ("TA_TA_BUSY_sum", "GRBM_GUI_ACTIVE"):
        t, n = per_kernel_in("TA", c)
        if k in t: ta[c] = t[k] / n[k]
    fb, wb = fetch[k] * 1024 / nf[k], write[k] * 1024 / nw[k]
    return {
        "kernel_name": k, "launches_profiled": nf[k],
        "avg_launch_ns_under_kernel_trace": avg.get(k, (None, 0))[0], "launches_in_stats_run": avg.get(k, (None, 0))[1],
        "bytes_per_launch": round(fb + wb), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
        "ta_busy_cycles_per_launch": (ta["TA_TA_BUSY_sum"] / 256) if ta.get("TA_TA_BUSY_sum") else None,
        "gpu_cycles_per_launch": (ta["GRBM_GUI_ACTIVE"] / 8) if ta.get("GRBM_GUI_ACTIVE") else None,
        "ta_busy_fraction": (ta["TA_TA_BUSY_sum"] / 256 / (ta["GRBM_GUI_ACTIVE"] / 8)) if ta.get("GRBM_GUI_ACTIVE") else None,
        "valu_insts_per_launch": sq.get("SQ_INSTS_VALU"), "salu_insts_per_launch": sq.get("SQ_INSTS_SALU"),
        "valu_lane_utilisation": (sq["SQ_THREAD_CYCLES_VALU"] / (sq["SQ_INSTS_VALU"] * 64)) if sq.get("SQ_INSTS_VALU") else None,
        "wave_cycles_waiting_fraction": (sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_WAVE_CYCLES") else None,
    }
trace, shade = kernel_block(kf), kernel_block(ks)
shade["launches_per_step"] = 6
# k_shade streams SoA path state through the shade queue: FETCH_SIZE reads 0.62 - 0.63 of the bytes that pattern MOVES
# (profiles/shade_fetch_calibration.json, from r04_fetch_size_calibration_shade.txt: coalesced streams are counted at 1/2 per line
# at every lane width, the triangle gathers at face value per line touched; density 0.7 -> 0.62, density 1.0 -> 0.633; a step's
# launches run from density 1.0 on the first bounce downwards: bench.py reports the range); WRITE_SIZE at face value
cal = json.load(open("profiles/shade_fetch_calibration.json"))
shade["fetch_calibration"] = {"factor": cal["factor"], "density": cal["density"], "factor_dense": cal["factor_dense"],
                              "density_dense": cal["density_dense"], "source": cal["source"]}
shade["bytes_per_launch_calibrated"] = round(shade["fetch_bytes_per_launch"] / cal["factor"] + shade["write_bytes_per_launch"])
rec = {
    "trace": trace, "shade": shade,
    "config": {"n_gpus": 1, "tris": 1000000, "width": 1920, "height": 1080, "spp": 64, "max_depth": 5,
               "kernel": "k_trace_wide" if "wide" in kf else "k_trace"},
    "commit": os.environ.get("PB_COMMIT", commit), "source_hash": bench.kernel_source_hash(),
    "profile": "profiles/r05_bench_kernel_stats.csv + rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU "
               "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE; TA_TA_BUSY_sum GRBM_GUI_ACTIVE — one run each) on "
               "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary`; FETCH_SIZE = TCC_EA0_RDREQ x 64 B (Infinity-Cache hits "
               "included), used without the x2 of wide coalesced streams: calibrated on this gather pattern in "
               "profiles/r01_fetch_size_calibration.txt (0.986)",
}
json.dump(rec, open("profiles/r05_traffic.json", "w"), indent=1)
json.dump(rec, open(f"{out}/r05_traffic.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
cp profiles/r05_bench_kernel_stats.csv $OUT/ 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete
