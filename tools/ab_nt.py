#!/usr/bin/env python
"""Two-stage kernel (debug build): non-temporal against plain stores when the output of a pass is the input of the next (ping-pong
passes, as the sweeps of a level run in a cycle), by block size."""
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


for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [128, 192, 256, 320, 384, 512]:
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
    u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Ls, Fs = lu.c_struct(), lf.c_struct()
    cases = {"red-black sweep": lambda: (ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e), ops.rbgs_sweep_fused(Ls, un, u, Fs, f, A, w, 0, b, e)),
             "two Jacobi steps": lambda: (ops.jacobi2(Ls, u, un, None, Fs, f, A, w, b, e), ops.jacobi2(Ls, un, u, None, Fs, f, A, w, b, e))}
    cases["one Jacobi step"] = lambda: (ops.stencil_op(2, Ls, u, Fs, f, Ls, un, A, w, -1, b, e), ops.stencil_op(2, Ls, un, Fs, f, Ls, u, A, w, -1, b, e))
    cases["two half sweeps in place"] = lambda: (ops.stencil_op(2, Ls, u, Fs, f, Ls, u, A, w, 0, b, e), ops.stencil_op(2, Ls, u, Fs, f, Ls, u, A, w, 1, b, e))
    reps = max(5, min(100, int(2e9 / n ** 3)))
    res = {(k, nt): [] for k in cases for nt in (1, 0)}
    for rep in range(4):
        for nt in (1, 0):
            L.examg_debug_two_stage_nt(nt)
            L.examg_debug_zmarch_store(0 if nt else 2)
            for k, fn in cases.items():
                fn(); v = timed(fn, reps)
                if rep:
                    res[(k, nt)].append(v)
    L.examg_debug_two_stage_nt(-1)
    L.examg_debug_zmarch_store(-1)
    print("n=%d  " % n + "\n       ".join("%s: non-temporal %.4f, plain %.4f ms" % (k, statistics.median(res[(k, 1)]), statistics.median(res[(k, 0)])) for k in cases), flush=True)
    del u, un, f
