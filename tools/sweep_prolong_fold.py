#!/usr/bin/env python
"""Correction loop + first post-smoothing sweep: separate launches against the folded pass (debug build: variants)."""
import ctypes as C, os, sys
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
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in (int(a) for a in (sys.argv[1:] or ["256", "512"])):
    for align in (0, 16):
        lu, lf = FieldLayout.node(3, (n, n, n), 1, True, True, align), FieldLayout.node(3, (n, n, n), 0, True, False, align)
        lc = FieldLayout.node(3, (n // 2,) * 3, 1, True, True, align)
        u, un, f, uc = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lc.size)
        ops.fill_random(u, 1); ops.fill_random(f, 2); ops.fill_random(uc, 3)
        A = laplace_fd(3, (1.0 / n,) * 3)
        w = 0.8 / A.diag
        b, e = [1, 1, 1], [n, n, n]
        Ls, Fs, Lc = lu.c_struct(), lf.c_struct(), lc.c_struct()
        tp = timed(lambda: ops.prolong_add(Lc, uc, Ls, u, b, e))
        ops.fill_random(u, 1)
        for nw in (5, 6, 8):
            L.examg_debug_two_stage_lds(nw)
            ts = timed(lambda: ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e))
            tj = timed(lambda: ops.jacobi2(Ls, u, un, None, Fs, f, A, w, b, e))
            line = "n=%d align=%2d nw=%d  prolong %.4f  sweep %.4f  jac2 %.4f |" % (n, align, nw, tp, ts, tj)
            for wpe in (1, 4):
                L.examg_debug_two_stage_prol(wpe)
                tf = timed(lambda: ops.rbgs_sweep_fused_prolong(Ls, u, un, Fs, f, A, w, 0, b, e, Lc, uc))
                tg = timed(lambda: ops.jacobi2_prolong(Ls, u, un, None, Fs, f, A, w, b, e, Lc, uc))
                line += "  wpe%d: sweep+P %.4f jac2+P %.4f" % (wpe, tf, tg)
            print(line, flush=True)
        L.examg_debug_two_stage_lds(-1)
        del u, un, f, uc
