#!/usr/bin/env python
"""One-step z-march kernel, XCD-band order: planes per chunk 2 .. 16 (debug build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd import lib
from exastencils_amd.field import laplace_fd
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps

ops = HipOps(0, lib.DBG_LIB_PATH)
L = ops.L


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in [int(a) for a in sys.argv[1:]] or [512, 1024]:
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
    u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Ls, Fs = lu.c_struct(), lf.c_struct()
    pts = float(n - 1) ** 3
    for _ in range(60):
        ops.stencil_op(2, Ls, u, Fs, f, Ls, un, A, w, -1, b, e)
    res = {}
    for rnd in range(3):
        for chunk in (2, 4, 6, 8, 12, 16):
            L.examg_debug_zmarch(1 << 22, chunk, 2)
            res.setdefault(chunk, []).append((timed(lambda: ops.stencil_op(2, Ls, u, Fs, f, Ls, un, A, w, -1, b, e)),
                                              timed(lambda: ops.stencil_op(2, Ls, u, Fs, f, Ls, u, A, w, 0, b, e))))
    for chunk, v in res.items():
        med = [sorted(x[i] for x in v)[1] for i in range(2)]
        print("n=%d planes per chunk %2d  jacobi %.4f ms (frac %.3f)  half sweep %.4f ms (frac %.3f)"
              % (n, chunk, med[0], 24 * pts / med[0] / 1e6 / 8000, med[1], 24 * pts / med[1] / 1e6 / 8000), flush=True)
    L.examg_debug_zmarch(-1, -1, -1)
    del u, un, f
    torch.cuda.empty_cache()
