#!/usr/bin/env python
"""Launch rules of the one-step z-march kernel, residual + restriction and prolongation on blocks larger than 512^3 (debug build).
Usage: sweep_big_others.py [n ...]"""
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


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in [int(a) for a in sys.argv[1:]] or [1024, 768, 512]:
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
    lc = FieldLayout.node(3, (n // 2,) * 3, 0)
    lcu = FieldLayout.node(3, (n // 2,) * 3, 1)
    u, un, r, f, fc, uc = (ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size),
                           ops.new_array(lc.size), ops.new_array(lcu.size))
    ops.fill_random(u, 1); ops.fill_random(f, 2); ops.fill_random(uc, 3)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    bc, ec = [1, 1, 1], [n // 2] * 3
    Ls, Fs, Cs, CUs = lu.c_struct(), lf.c_struct(), lc.c_struct(), lcu.c_struct()
    pts = float(n - 1) ** 3
    for _ in range(30):
        ops.stencil_op(2, Ls, u, Fs, f, Ls, un, A, w, -1, b, e)
    for blocks, minchunk in ((-1, -1), (1024, 16), (2048, 16), (4096, 16), (8192, 16), (1 << 20, 16), (1 << 20, 32), (1 << 20, 64)):
        L.examg_debug_zmarch(blocks, minchunk, -1)
        tj = timed(lambda: ops.stencil_op(2, Ls, u, Fs, f, Ls, un, A, w, -1, b, e))
        th = timed(lambda: ops.stencil_op(2, Ls, u, Fs, f, Ls, u, A, w, 0, b, e))
        tn = timed(lambda: ops.residual_norm2(Ls, u, Fs, f, A, b, e, Ls, r))
        print("n=%d zmarch blocks=%8d minchunk=%3d  jacobi %.4f ms (frac %.3f)  half sweep %.4f  residual+norm %.4f (frac %.3f)"
              % (n, blocks, minchunk, tj, 24 * pts / tj / 1e6 / 8000, th, tn, 16 * pts / tn / 1e6 / 8000), flush=True)
    L.examg_debug_zmarch(-1, -1, -1)
    for waves, minzc in ((24576, 8), (49152, 8), (98304, 8), (196608, 8), (1 << 22, 8), (1 << 22, 13), (1 << 22, 24)):
        L.examg_debug_residual_restrict(waves, minzc)
        t = timed(lambda: ops.residual_restrict(Ls, u, Fs, f, Ls, r, A, Cs, fc, 1.0, b, e, bc, ec))
        print("n=%d residual+restriction waves=%8d minzc=%2d  %.4f ms (frac %.3f)" % (n, waves, minzc, t, 17 * pts / t / 1e6 / 8000), flush=True)
    L.examg_debug_residual_restrict(0, 8)
    for zb in (1, 2, 4, 8):
        L.examg_debug_prolong(zb)
        t = timed(lambda: ops.prolong_add(CUs, uc, Ls, u, b, e))
        print("n=%d prolongation zb=%d  %.4f ms (frac %.3f)" % (n, zb, t, 17 * pts / t / 1e6 / 8000), flush=True)
    L.examg_debug_prolong(-1)
    del u, un, r, f, fc, uc
    torch.cuda.empty_cache()
