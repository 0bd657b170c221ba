#!/usr/bin/env python
"""List the kernels of the traced V-cycle (tools/vcycle_trace.py) in order: start (us from the first), duration, gap."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "vtrace")
summary = len(sys.argv) > 2 and sys.argv[2] == "summary"      # totals, the kernels by total time and the largest gaps only
rows = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
last = max(i for i, r in enumerate(rows) if "k_fill_random" in r[2])
cyc = rows[last + 1:]
t0 = cyc[0][0]
prev_end = t0
busy = 0
by_name, gaps = {}, []
for s, e, name in cyc:
    short = name.split("(")[0].replace("void examg::", "").replace("examg::", "")[:60]
    if not summary:
        print("%9.1f us  dur %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short))
    n, t = by_name.get(short, (0, 0))
    by_name[short] = (n + 1, t + e - s)
    gaps.append(((s - prev_end) / 1e3, short))
    busy += e - s
    prev_end = e
if summary:
    for short, (n, t) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:14]:
        print("%8.1f us  %5d x  %s" % (t / 1e3, n, short))
    gaps.sort(reverse=True)
    print("largest gaps (us, before kernel):", ["%.0f %s" % g for g in gaps[:12]])
    print("gaps > 20 us: %d, sum %.0f us; gaps 5..20 us: %d, sum %.0f us" % (
        sum(1 for g in gaps if g[0] > 20), sum(g[0] for g in gaps if g[0] > 20),
        sum(1 for g in gaps if 5 < g[0] <= 20), sum(g[0] for g in gaps if 5 < g[0] <= 20)))
print("kernels %d  span %.1f us  busy %.1f us  gaps %.1f us" % (len(cyc), (prev_end - t0) / 1e3, busy / 1e3, (prev_end - t0 - busy) / 1e3))
