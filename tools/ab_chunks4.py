#!/usr/bin/env python
"""Two- and three-step passes at 512^3 and 256^3 with chunk lengths whose step count is / is not a multiple of four (debug build, one process):
the plane loops run whole groups of four steps, so a chunk of zc planes costs roundup4(zc + 2) steps (two stages) / roundup4(zc + 4) (three)."""
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
L.examg_debug_two_stage.argtypes = [C.c_int] * 4
L.examg_debug_three_stage.argtypes = [C.c_int] * 2


def timed(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [512, 256]:
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
    u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2); un.copy_(u)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Ls, Fs = lu.c_struct(), lf.c_struct()
    pair = lambda: ops.jacobi2(Ls, u, un, None, Fs, f, A, w, b, e)
    sweep = lambda: ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e)
    triple = lambda: ops.jacobi3(Ls, u, un, None, Fs, f, A, w, b, e)
    cases = [("pair, rule", pair, ("two", -1)), ("sweep, rule", sweep, ("two", -1)), ("triple, rule", triple, ("three", -1))]
    for zc in (14, 16, 18, 20, 22, 26, 30):
        cases.append(("pair, %d planes" % zc, pair, ("two", zc)))
        cases.append(("sweep, %d planes" % zc, sweep, ("two", zc)))
    for zc in (40, 44, 48, 52, 56, 57, 60, 64):
        cases.append(("triple, %d planes" % zc, triple, ("three", zc)))

    def setk(k):
        kind, zc = k
        if kind == "two":
            # blocks target huge -> the minimum chunk length decides (4th argument); -1: default rule
            L.examg_debug_two_stage(0, (1 << 24) if zc > 0 else -1, -1, zc)
        else:
            L.examg_debug_three_stage(0, zc)

    for _ in range(20):
        for _, fn, k in cases:
            setk(k); fn()
    res = {c[0]: [] for c in cases}
    for _ in range(5):
        for name, fn, k in cases:
            setk(k); fn(); res[name].append(timed(fn))
    L.examg_debug_two_stage(0, -1, -1, -1)
    L.examg_debug_three_stage(0, -1)
    for name, _, _ in cases:
        print("%d^3 %-24s %.4f ms" % (n, name, statistics.median(res[name])), flush=True)
