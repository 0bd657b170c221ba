#!/usr/bin/env python
"""One, two and three Jacobi steps per pass at n^3 (default 512 and 256), ping-pong between two arrays, one process, medians of interleaved
rounds after a run-in; with --dbg (debug build) also the three-step pass with forced z chunks.  Prints ms per launch and per STEP."""
import ctypes as C, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd import lib
from exastencils_amd.field import laplace_fd
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps

dbg = "--dbg" in sys.argv
ops = HipOps(0, lib.DBG_LIB_PATH if dbg else None)
L = ops.L
if dbg:
    L.examg_debug_three_stage.argtypes = [C.c_int] * 2
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [512, 256]


def setk(k):
    """None: nothing to set; -1: the launcher's rule; otherwise planes per chunk."""
    if k is not None:
        L.examg_debug_three_stage(0, k)


def timed(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in sizes:
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
    u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    un.copy_(u)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Ls, Fs = lu.c_struct(), lf.c_struct()
    state = {"a": u, "b": un}

    def pp(call):
        def run():
            call(state["a"], state["b"])
            state["a"], state["b"] = state["b"], state["a"]
        return run

    cases = [("one step", pp(lambda x, y: ops.stencil_op(2, Ls, x, Fs, f, Ls, y, A, w, -1, b, e)), 1, None),
             ("two steps", pp(lambda x, y: ops.jacobi2(Ls, x, y, None, Fs, f, A, w, b, e)), 2, None),
             ("three steps", pp(lambda x, y: ops.jacobi3(Ls, x, y, None, Fs, f, A, w, b, e)), 3, -1 if dbg else None)]
    if dbg:
        for zc in (16, 20, 24, 28, 36, 40, 44, 48, 52, 56, 64, 86, 128):
            cases.append(("three steps, %d planes per chunk" % zc, cases[2][1], 3, zc))
    for _ in range(30):
        for _, fn, _, zc in cases:
            setk(zc)
            fn()
    res = {k: [] for k, _, _, _ in cases}
    for _ in range(5):
        for k, fn, _, zc in cases:
            setk(zc)
            fn()
            res[k].append(timed(fn))
    if dbg:
        setk(-1)
    for k, _, steps, _ in cases:
        m = statistics.median(res[k])
        print("%d^3 %-50s %.4f ms per launch, %.4f ms per step, %.3e LU/s" % (n, k, m, m / steps, (n - 1) ** 3 * steps / (m * 1e-3)), flush=True)
