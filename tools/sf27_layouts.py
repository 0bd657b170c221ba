#!/usr/bin/env python
"""27-entry stencil field at 512^3 (config 4's operator): coefficient planes (reference layout) against the entry-fastest layout
transformation `[x, y, z, i] => [i, x, y, z]`; Jacobi step and residual, bit comparison and timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.field import Stencil, helmholtz27_offsets
from exastencils_amd.layout import FieldLayout
from exastencils_amd.lib import GeomC
from exastencils_amd.ops import HipOps

from exastencils_amd import lib
ops = HipOps(0, lib.DBG_LIB_PATH)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
u, un, un2, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
ops.fill_random(u, 1); ops.fill_random(f, 2)
cf = ops.new_array(27 * lf.size)
g = GeomC()
for d in range(3):
    g.h[d] = 1.0 / n
ops.init_helmholtz27(lf.c_struct(), cf, g, 7, (10.0, 2.0), [0, 0, 0], [n + 1] * 3)
b, e = [1, 1, 1], [n, n, n]
planes = Stencil(helmholtz27_offsets(), [], cf, lf)
rec = planes.entry_fastest(ops)
Ls, Fs = lu.c_struct(), lf.c_struct()
pts = float(n - 1) ** 3


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for mode in (2, 1, 0):
    ops.stencil_op(mode, Ls, u, Fs, f, Ls, un, planes, 0.8, -1, b, e)
    ops.stencil_op(mode, Ls, u, Fs, f, Ls, un2, rec, 0.8, -1, b, e)
    torch.cuda.synchronize()
    print("mode %d: transformed == planes bitwise: %s" % (mode, bool(torch.equal(un, un2))), flush=True)
for rnd in range(2):
    for name, st, run in [("planes (reference layout)", planes, -1)] + [("records, %2d tiles per wave" % r, rec, r) for r in (1, 2, 3, 4, 8)]:
        ops.L.examg_debug_sf27_run(run)
        tj = timed(lambda: ops.stencil_op(2, Ls, u, Fs, f, Ls, un, st, 0.8, -1, b, e))
        tr = timed(lambda: ops.stencil_op(1, Ls, u, Fs, f, Ls, un, st, 0.8, -1, b, e))
        print("n=%d %-28s jacobi %.3f ms (frac %.3f)  residual %.3f ms (frac %.3f)"
              % (n, name, tj, 240 * pts / tj / 1e6 / 8000, tr, 240 * pts / tr / 1e6 / 8000), flush=True)
