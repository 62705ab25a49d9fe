"""One run of pbrt_hip_probe_state_stream (k_shade's access pattern with a known byte count), for tools/gpu_calibrate_shade.sh:
PARTS (bit mask, see include/pbrt_hip.h), DENSITY (share of the paths in the shade queue), N (paths). Prints one JSON line.
With `report DIR` instead: reads the rocprofv3 --pmc CSVs under DIR and prints counter / known-bytes ratios."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))

NAMES = {1: "nine 16-B SoA records", 2: "two 32-B ray records (32-B stride)", 4: "8-B + 2 x 4-B scalars", 8: "random 48-B triangle (48 MB table)",
         15: "all reads", 16: "all stores (+ the 4-B queue read)", 31: "reads + stores (k_shade's mix)"}

if len(sys.argv) > 2 and sys.argv[1] == "report":
    rows = []
    for d in sorted(glob.glob(os.path.join(sys.argv[2], "p*_d*"))):
        known = json.load(open(os.path.join(d, "known.json")))
        got = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            vals = []
            for f in glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f, newline="")):
                    if row["Counter_Name"] == counter and "k_probe_state_stream" in row["Kernel_Name"]:
                        vals.append(float(row["Counter_Value"]) * 1024.0)
            got[counter] = sum(vals) / len(vals) if vals else None
        rows.append((known, got))
    print(f"{'parts':>5} {'density':>7}  {'what':<44} {'asked read GB':>13} {'FETCH_SIZE GB':>13} {'ratio':>6}   {'asked write GB':>14} {'WRITE_SIZE GB':>13} {'ratio':>6}   {'ms':>7} {'GB/s asked':>10}")
    for known, got in rows:
        fr = got["FETCH_SIZE"] / known["bytes_read"] if got["FETCH_SIZE"] and known["bytes_read"] else float("nan")
        wr = got["WRITE_SIZE"] / known["bytes_written"] if got["WRITE_SIZE"] and known["bytes_written"] else float("nan")
        print(f"{known['parts']:>5} {known['density']:>7.2f}  {NAMES.get(known['parts'], ''):<44} {known['bytes_read'] / 1e9:>13.3f} {(got['FETCH_SIZE'] or 0) / 1e9:>13.3f} {fr:>6.3f}   "
              f"{known['bytes_written'] / 1e9:>14.3f} {(got['WRITE_SIZE'] or 0) / 1e9:>13.3f} {wr:>6.3f}   {known['ms']:>7.3f} {(known['bytes_read'] + known['bytes_written']) / known['ms'] / 1e6:>10.0f}")
    sys.exit(0)

import pbrt_hip
parts, density, n = int(os.environ.get("PARTS", "31")), float(os.environ.get("DENSITY", "0.7")), int(os.environ.get("N", str(64 << 20)))
ctx = pbrt_hip.Context(0)
rd, wr, ms = ctx.probe_state_stream(n, density, 48 << 20, parts)
print(json.dumps(dict(parts=parts, density=density, n_paths=n, bytes_read=rd, bytes_written=wr, ms=ms)))
ctx.close()
