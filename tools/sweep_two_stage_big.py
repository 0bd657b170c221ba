#!/usr/bin/env python
"""Two-step Jacobi pass and fused red-black sweep on blocks larger than 512^3: workgroup count (= chunk length in z), debug build.
Usage: sweep_two_stage_big.py [n ...]"""
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


def timed(fn, reps=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in [int(a) for a in sys.argv[1:]] or [1024, 768]:
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
    u, un, tmp, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Ls, Fs = lu.c_struct(), lf.c_struct()
    comp = 24.0 * (n - 1) ** 3
    for _ in range(40):
        ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e)
    for rnd in range(1):
        for blocks in (8192, 65536):
            L.examg_debug_two_stage(0, blocks, -1, 0)
            tj = timed(lambda: ops.jacobi2(Ls, u, un, tmp, Fs, f, A, w, b, e))
            ts = timed(lambda: ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e))
            print("n=%d blocks=%6d  two Jacobi steps %.4f ms (frac %.3f)  red-black sweep %.4f ms (frac %.3f)"
                  % (n, blocks, tj, comp / tj / 1e6 / 8000, ts, comp / ts / 1e6 / 8000), flush=True)
    # chunk length at the unlimited workgroup count, and the variant with the correction folded in
    lc = FieldLayout.node(3, (n // 2,) * 3, 1)
    uc = ops.new_array(lc.size)
    ops.fill_random(uc, 3)
    for minzc in (8, 12, 16, 24, 32, 48):
        L.examg_debug_two_stage(0, 1 << 20, -1, minzc)
        ts = timed(lambda: ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e))
        tp = timed(lambda: ops.rbgs_sweep_fused_prolong(Ls, u, un, Fs, f, A, w, 0, b, e, lc.c_struct(), uc))
        print("n=%d planes per chunk >= %2d  red-black sweep %.4f ms (frac %.3f)  with correction %.4f ms" % (n, minzc, ts, comp / ts / 1e6 / 8000, tp), flush=True)
    for blocks in (3072, 8192, 16384, 65536):
        L.examg_debug_two_stage(0, blocks, -1, 0)
        tp = timed(lambda: ops.rbgs_sweep_fused_prolong(Ls, u, un, Fs, f, A, w, 0, b, e, lc.c_struct(), uc))
        print("n=%d blocks=%6d  with correction %.4f ms" % (n, blocks, tp), flush=True)
    del u, un, tmp, f, uc
    torch.cuda.empty_cache()
