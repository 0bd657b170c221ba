#!/usr/bin/env python
"""Fused red-black sweep on the mid levels (256^3, 128^3): workgroup shape, count and chunk length (debug build)."""
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


def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in (256, 192, 128, 96):
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
    u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Ls, Fs = lu.c_struct(), lf.c_struct()
    for nw in (5, 6, 8):
        L.examg_debug_two_stage_lds(nw)
        for blocks, minzc in ((3072, 0), (3072, 16), (456, 16), (3072, 8), (3072, 4), (3072, 2)):
            L.examg_debug_two_stage(0, blocks, -1, minzc)
            ts = timed(lambda: ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e))
            print("n=%d nw=%d blocks=%5d minzc=%2d  sweep %.4f" % (n, nw, blocks, minzc, ts), flush=True)
    L.examg_debug_two_stage_lds(-1)
    del u, un, f
