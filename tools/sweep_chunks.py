#!/usr/bin/env python
"""Two-stage kernel at 512^3 (debug build): planes per z chunk against time, Jacobi pair and red-black sweep, one process."""
import ctypes as C, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd import lib
from exastencils_amd.field import laplace_fd
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps

ops = HipOps(0, lib.DBG_LIB_PATH)
L = ops.L
n = ([int(a) for a in sys.argv[1:] if a.isdigit()] or [512])[0]
lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
ops.fill_random(u, 1); ops.fill_random(f, 2)
A = laplace_fd(3, (1.0 / n,) * 3)
w = 0.8 / A.diag
b, e = [1, 1, 1], [n, n, n]
Ls, Fs = lu.c_struct(), lf.c_struct()
cases = {"red-black sweep": lambda: ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e), "two Jacobi steps": lambda: ops.jacobi2(Ls, u, un, None, Fs, f, A, w, b, e)}


def timed(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ntzs = [32, 29, 26, 22, 19, 18, 17, 16, 13] if n == 512 else [3, 4, 5, 6, 7, 8, 9, 11, 13, 16]
xy = 120 if n == 512 else 57
res = {(k, t): [] for k in cases for t in ntzs}
for rep in range(4):
    for t in ntzs:
        L.examg_debug_two_stage(0, xy * t, -1, 8)
        for k, fn in cases.items():
            fn(); v = timed(fn)
            if rep:
                res[(k, t)].append(v)
for k in cases:
    print(k + ": " + "  ".join("%d chunks %.4f" % (t, statistics.median(res[(k, t)])) for t in ntzs), flush=True)
