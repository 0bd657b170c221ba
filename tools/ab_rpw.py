#!/usr/bin/env python
"""Two-stage kernel (debug build): two against three rows per wave by block size (chunk count by the launcher's rule), ping-pong passes."""
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


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * reps)


for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [192, 224, 256, 288, 320, 352, 384, 448]:
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
    u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Ls, Fs = lu.c_struct(), lf.c_struct()
    cases = {"sweep": lambda: (ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e), ops.rbgs_sweep_fused(Ls, un, u, Fs, f, A, w, 0, b, e)),
             "pair": lambda: (ops.jacobi2(Ls, u, un, None, Fs, f, A, w, b, e), ops.jacobi2(Ls, un, u, None, Fs, f, A, w, b, e))}
    reps = max(5, min(100, int(2e9 / n ** 3)))
    res = {(k, v): [] for k in cases for v in (8, 83)}
    for rep in range(4):
        for v in (8, 83):
            L.examg_debug_two_stage_lds(v)
            for k, fn in cases.items():
                fn(); t = timed(fn, reps)
                if rep:
                    res[(k, v)].append(t)
    L.examg_debug_two_stage_lds(-1)
    print("n=%d  " % n + "   ".join("%s: 2 rows %.4f, 3 rows %.4f ms" % (k, statistics.median(res[(k, 8)]), statistics.median(res[(k, 83)])) for k in cases), flush=True)
    del u, un, f
