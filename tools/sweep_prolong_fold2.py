#!/usr/bin/env python
"""Folded correction + sweep at 512^3 / 256^3: workgroup count target and minimum z chunk (debug build)."""
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
L.examg_debug_two_stage.argtypes = [C.c_int] * 4


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in (int(a) for a in (sys.argv[1:] or ["512", "256"])):
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
    lc = FieldLayout.node(3, (n // 2,) * 3, 1)
    u, un, f, uc = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lc.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2); ops.fill_random(uc, 3)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Ls, Fs, Lc = lu.c_struct(), lf.c_struct(), lc.c_struct()
    for blocks, minzc in ((2048, 16), (3072, 16), (4096, 16), (6144, 16), (8192, 16), (12288, 8), (16384, 8)):
        L.examg_debug_two_stage(0, blocks, -1, minzc)
        ts = timed(lambda: ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e))
        tf = timed(lambda: ops.rbgs_sweep_fused_prolong(Ls, u, un, Fs, f, A, w, 0, b, e, Lc, uc))
        tg = timed(lambda: ops.jacobi2_prolong(Ls, u, un, None, Fs, f, A, w, b, e, Lc, uc))
        print("n=%d blocks=%5d minzc=%2d  sweep %.4f  sweep+P %.4f  jac2+P %.4f" % (n, blocks, minzc, ts, tf, tg), flush=True)
    del u, un, f, uc
