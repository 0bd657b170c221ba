#!/usr/bin/env python
"""Row-marching kernel against the 128-point-window kernel (debug build): bit comparison and timing, planes per chunk / order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd import lib
from exastencils_amd.field import laplace_fd
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps

ops = HipOps(0, os.path.abspath(os.environ.get('LIB', lib.DBG_LIB_PATH)))
L = ops.L


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in [int(a) for a in sys.argv[1:]] or [512]:
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
    u, un, un2, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Ls, Fs = lu.c_struct(), lf.c_struct()
    pts = float(n - 1) ** 3
    for mode in (2, 1, 0):
        L.examg_debug_rowmarch(0, -1, -1)
        ops.stencil_op(mode, Ls, u, Fs, f, Ls, un, A, w, -1, b, e)
        L.examg_debug_rowmarch(1, -1, -1)
        ops.stencil_op(mode, Ls, u, Fs, f, Ls, un2, A, w, -1, b, e)
        torch.cuda.synchronize()
        print("n=%d mode %d: row-marching == window kernel bitwise: %s" % (n, mode, bool(torch.equal(un, un2))), flush=True)
    for _ in range(150):
        ops.stencil_op(2, Ls, u, Fs, f, Ls, un, A, w, -1, b, e)
    res = {}
    if n > 300:
        cfgs = [(0, -1, -1)] + [(1, zc, rm) for zc in (32, 48, 64, 86, 128, 171) for rm in (2, 0)]
    else:      # two-segment form (rows of 144 .. 256 points): on = 12 / 14 -> 2 / 4 rows per wave
        cfgs = [(0, -1, -1)] + [(on, zc, rm) for on in (12, 14) for zc in (8, 16, 32, 64, 128) for rm in (2, 0)]
    for rnd in range(3):
        for cfg in cfgs:
            L.examg_debug_rowmarch(*cfg)
            res.setdefault(cfg, []).append((timed(lambda: ops.stencil_op(2, Ls, u, Fs, f, Ls, un, A, w, -1, b, e)),
                                            timed(lambda: ops.stencil_op(1, Ls, u, Fs, f, Ls, un, A, 0.0, -1, b, e))))
    for cfg, v in res.items():
        med = [sorted(x[i] for x in v)[len(v) // 2] for i in range(2)]
        print("n=%d rowmarch on=%d zc=%3d remap=%2d  jacobi %.4f ms (frac %.3f)  residual %.4f (%.3f)"
              % (n, cfg[0], cfg[1], cfg[2], med[0], 24 * pts / med[0] / 1e6 / 8000, med[1], 24 * pts / med[1] / 1e6 / 8000), flush=True)
    L.examg_debug_rowmarch(-1, -1, -1)
    del u, un, un2, f
    torch.cuda.empty_cache()
