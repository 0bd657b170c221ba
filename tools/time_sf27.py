#!/usr/bin/env python
"""27-entry stencil field at n^3 (default 512): one Jacobi step on coefficient planes / on records, and the one-pass forms on records
(two steps; one step + residual) against their two launches.  Debug build: chunk lengths of the pair kernel."""
import ctypes as C, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd import lib
from exastencils_amd.field import Stencil, helmholtz27_offsets
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps

dbg = "--dbg" in sys.argv
ops = HipOps(0, lib.DBG_LIB_PATH if dbg else None)
L = ops.L
n = ([int(a) for a in sys.argv[1:] if a.isdigit()] or [512])[0]
lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
u, un, f, res = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size)
ops.fill_random(u, 1); ops.fill_random(f, 2)
cf = ops.new_array(27 * lf.size)
ops.fill_random(cf, 3)
cf[:lf.size] += 8.0
planes = Stencil(helmholtz27_offsets(), [], cf, lf)
rec = planes.entry_fastest(ops)
Ls, Fs = lu.c_struct(), lf.c_struct()
b, e = [1, 1, 1], [n, n, n]
pts = (n - 1) ** 3


def timed(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cases = [("one step, planes", lambda: ops.stencil_op(2, Ls, u, Fs, f, Ls, un, planes, 0.8, -1, b, e), 240),
         ("one step, records", lambda: ops.stencil_op(2, Ls, u, Fs, f, Ls, un, rec, 0.8, -1, b, e), 240),
         ("residual, records", lambda: ops.stencil_op(1, Ls, u, Fs, f, Ls, res, rec, 0.0, -1, b, e), 240),
         ("two steps, one pass", lambda: ops.jacobi2(Ls, u, un, None, Fs, f, rec, 0.8, b, e), 240),
         ("step + residual, one pass", lambda: ops.jacobi_residual(Ls, u, un, Fs, f, Ls, res, rec, 0.8, b, e), 248)]
if dbg:
    L.examg_debug_sf27_pair.argtypes = [C.c_int, C.c_int]
    for rows in (1, 2):
        for zc in ((32, 64, 128) if rows == 1 else ((32, 43, 64, 86, 128, 171, 256) if n > 256 else (12, 16, 22, 26, 32, 43))):
            cases.append(("two steps, one pass, %d rows per wave, %d planes" % (rows, zc),
                          (lambda zc=zc, rows=rows: (L.examg_debug_sf27_pair(10 + rows, zc), ops.jacobi2(Ls, u, un, None, Fs, f, rec, 0.8, b, e))), 240))
    cases.append(("step + residual, one pass, 1 row per wave", lambda: (L.examg_debug_sf27_pair(11, 0), ops.jacobi_residual(Ls, u, un, Fs, f, Ls, res, rec, 0.8, b, e)), 248))
for _ in range(5):
    for _, fn, _ in cases:
        fn()
if dbg:
    L.examg_debug_sf27_pair(1, 0)
out = {k: [] for k, _, _ in cases}
for _ in range(3):
    for k, fn, _ in cases:
        if dbg and "planes" not in k:
            L.examg_debug_sf27_pair(1, 0)
        fn(); out[k].append(timed(fn))
for k, _, nb in cases:
    ms = statistics.median(out[k])
    print("n=%d %-36s %.3f ms   %.0f GB/s of %d B per point" % (n, k, ms, nb * pts / ms / 1e6, nb), flush=True)
