#!/usr/bin/env python
"""Where does the 27-entry stencil-field kernel's fabric traffic beyond its compulsory bytes come from?  Two launches of
k_stencilfield_unrolled<2, 27> at 512^3: (a) the 27-point neighbourhood, (b) all 27 offsets zero -- the same coefficient, rhs and
output streams, but `u` read at the point itself only.  Under `rocprofv3 --pmc FETCH_SIZE` the difference is what the neighbourhood
reads of `u` cost at the fabric; `--reduce <dir>` prints the per-case averages (FETCH_SIZE doubled, tools/pmc_reduce.py)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--reduce":
    rows = []
    for path in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == "FETCH_SIZE":
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0 * 2.0))
    rows.sort()
    seg, segs = [], []
    for _, name, v in rows:
        if "k_fill_random" in name:
            if seg:
                segs.append(seg)
            seg = []
        elif "k_stencilfield_unrolled" in name:
            seg.append(v)
    if seg:
        segs.append(seg)
    for name, s in zip(("neighbourhood", "centre only"), segs[-2:]):
        print("%-14s fetch %.2f GB per launch (%d launches)" % (name, sum(s) / len(s) / 1e9, len(s)))
    sys.exit(0)

import torch
from exastencils_amd.field import Stencil, helmholtz27_offsets
from exastencils_amd.layout import FieldLayout
from exastencils_amd.lib import GeomC
from exastencils_amd.ops import HipOps

ops = HipOps(0)
n = 512
lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
mark = ops.new_array(64)
ops.fill_random(u, 1)
ops.fill_random(f, 2)
cf = ops.new_array(27 * lf.size)
g = GeomC()
for d in range(3):
    g.h[d] = 1.0 / n
ops.init_helmholtz27(lf.c_struct(), cf, g, 7, (10.0, 2.0), [0, 0, 0], [n + 1] * 3)
b, e = [1, 1, 1], [n, n, n]
for name, offs in (("neighbourhood", helmholtz27_offsets()), ("centre only", [(0, 0, 0)] * 27)):
    st = Stencil(offs, [], cf, lf)
    ops.fill_random(mark, 3)
    fn = lambda: ops.stencil_op(2, lu.c_struct(), u, lf.c_struct(), f, lu.c_struct(), un, st, 0.8, -1, b, e)
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-14s %.3f ms" % (name, e0.elapsed_time(e1) / 5), flush=True)
