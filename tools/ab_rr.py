#!/usr/bin/env python
"""Residual + restriction in one pass (k_residual_restrict3), debug build: the two workgroup orders in one process at levels 9 and 8."""
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
for lvl in (9, 8):
    n = 1 << lvl
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 1)
    lc = FieldLayout.node(3, (n // 2,) * 3, 1)
    u, f, fc = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lc.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    st = laplace_fd(3, [1.0 / n] * 3)
    A, F, Cc = lu.c_struct(), lf.c_struct(), lc.c_struct()
    fb, fe, cb, ce = [1, 1, 1], [n, n, n], [1, 1, 1], [n // 2] * 3
    run = lambda: ops.residual_restrict(A, u, F, f, None, None, st, Cc, fc, 1.0, fb, fe, cb, ce)

    def timed(reps=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    res = {0: [], 1: []}
    keep = {}
    for rep in range(4):
        for order in (0, 1):
            L.examg_debug_rr_order(order)
            run(); res[order].append(timed())
            if rep == 0:
                keep[order] = fc.clone()
    print("level %d: row-group bands per x tile %.4f ms, x tiles fastest in the bands %.4f ms, same bits: %s"
          % (lvl, statistics.median(res[0]), statistics.median(res[1]), bool(torch.equal(keep[0], keep[1]))), flush=True)
