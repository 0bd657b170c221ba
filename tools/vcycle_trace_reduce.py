#!/usr/bin/env python
"""List the kernels of the traced V-cycle (tools/vcycle_trace.py) in order: start (us from the first), duration, gap."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "vtrace")
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
for s, e, name in cyc:
    short = name.split("(")[0].replace("void examg::", "").replace("examg::", "")[:60]
    print("%9.1f us  dur %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short))
    busy += e - s
    prev_end = e
print("kernels %d  span %.1f us  busy %.1f us  gaps %.1f us" % (len(cyc), (prev_end - t0) / 1e3, busy / 1e3, (prev_end - t0 - busy) / 1e3))
