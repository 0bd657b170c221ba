#!/usr/bin/env python
"""residual + restriction in one pass: wave count target / minimum z chunk (debug build)."""
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
L.examg_debug_residual_restrict.argtypes = [C.c_int] * 2


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in (512, 256):
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
    lfc = FieldLayout.node(3, (n // 2,) * 3, 0, True, False)
    u, f, fc = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lfc.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    A = laplace_fd(3, (1.0 / n,) * 3)
    b, e = [1, 1, 1], [n, n, n]
    bc, ec = [1, 1, 1], [n // 2] * 3
    Ls, Fs, Fc = lu.c_struct(), lf.c_struct(), lfc.c_struct()
    for waves, minzc in ((4096, 8), (3072, 8), (6144, 8), (9216, 8), (12288, 8), (18432, 8), (24576, 4), (36864, 4)):
        L.examg_debug_residual_restrict(waves, minzc)
        t = timed(lambda: ops.residual_restrict(Ls, u, Fs, f, Ls, None, A, Fc, fc, 1.0, b, e, bc, ec))
        print("n=%d waves=%5d minzc=%d  %.4f ms" % (n, waves, minzc, t), flush=True)
    del u, f, fc
