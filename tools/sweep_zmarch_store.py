#!/usr/bin/env python
"""One-step z-march kernel: store modes (0 non-temporal, 1 cached on the partial lines at window edges, 2 all cached); debug build."""
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


modes = [int(a) for a in os.environ.get("MODES", "0,1,2").split(",")]
for n in [int(a) for a in sys.argv[1:]] or [512]:
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
    u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Ls, Fs = lu.c_struct(), lf.c_struct()
    pts = float(n - 1) ** 3
    for _ in range(150):
        ops.stencil_op(2, Ls, u, Fs, f, Ls, un, A, w, -1, b, e)
    res = {}
    ref = None
    for rnd in range(5):
        for mode in modes:
            L.examg_debug_zmarch_store(mode)
            res.setdefault(mode, []).append((timed(lambda: ops.stencil_op(2, Ls, u, Fs, f, Ls, un, A, w, -1, b, e)),
                                             timed(lambda: ops.stencil_op(1, Ls, u, Fs, f, Ls, un, A, 0.0, -1, b, e)),
                                             timed(lambda: ops.stencil_op(2, Ls, u, Fs, f, Ls, u, A, w, 0, b, e))))
    for mode, v in res.items():
        med = [sorted(x[i] for x in v)[len(v) // 2] for i in range(3)]
        print("n=%d store mode %d  jacobi %.4f ms (frac %.3f)  residual %.4f (%.3f)  half sweep %.4f ms (frac %.3f)"
              % (n, mode, med[0], 24 * pts / med[0] / 1e6 / 8000, med[1], 24 * pts / med[1] / 1e6 / 8000, med[2], 24 * pts / med[2] / 1e6 / 8000), flush=True)
    L.examg_debug_zmarch_store(-1)
    del u, un, f
    torch.cuda.empty_cache()
