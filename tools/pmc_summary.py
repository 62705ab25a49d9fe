"""Sums rocprofv3 --pmc counter CSVs per kernel (tools/pmc_trace.sh). usage: pmc_summary.py <dir>"""
import csv, glob, os, re, sys
from collections import defaultdict
root = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            m = re.search(r"(k_\w+|trampoline_kernel)", k)
            name = m.group(1) if m else k[:40]
            if "k_trace" in k or "k_intersect" in k or "k_shade" in k:
                t = re.search(r"(k_trace\w*|k_intersect\w*|k_shade\w*)(<[^>]*>)?", k)
                name = t.group(0) if t else name
            c = row["Counter_Name"]
            tot[name][c] += float(row["Counter_Value"])
            calls[name][c] += 1
for name in sorted(tot, key=lambda n: -tot[n].get("SQ_WAVE_CYCLES", tot[n].get("GRBM_GUI_ACTIVE", 0))):
    if not any(s in name for s in ("k_trace", "k_shade", "k_intersect")):
        continue
    c = tot[name]
    n = max(calls[name].values())
    print(f"== {name}: {n} dispatches")
    for k in sorted(c):
        print(f"   {k:32s} {c[k]:.4e}")
    if "SQ_INSTS_VALU" in c and c["SQ_INSTS_VALU"]:
        print(f"   -> VALU lane utilisation {c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_INSTS_VALU'] * 64) * 100:.1f} %")
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        print(f"   -> wait_any {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'] * 100:.0f} % of wave cycles, wait_inst_any {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'] * 100:.0f} %")
    if "TA_TA_BUSY_sum" in c and c.get("GRBM_GUI_ACTIVE"):
        print(f"   -> TA busy {c['TA_TA_BUSY_sum'] / 256 / (c['GRBM_GUI_ACTIVE'] / 8) * 100:.0f} % (GRBM_GUI_ACTIVE / 8 XCDs = {c['GRBM_GUI_ACTIVE'] / 8:.3e} cycles)")
    if "TCC_HIT_sum" in c:
        print(f"   -> L2 hit rate {c['TCC_HIT_sum'] / max(c['TCC_HIT_sum'] + c['TCC_MISS_sum'], 1) * 100:.0f} %")
    if "FETCH_SIZE" in c:
        print(f"   -> FETCH_SIZE {c['FETCH_SIZE'] * 1024 / 1e9:.2f} GB over {calls[name]['FETCH_SIZE']} dispatches")
    if "WRITE_SIZE" in c:
        print(f"   -> WRITE_SIZE {c['WRITE_SIZE'] * 1024 / 1e9:.2f} GB over {calls[name]['WRITE_SIZE']} dispatches")
