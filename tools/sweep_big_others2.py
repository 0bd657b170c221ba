#!/usr/bin/env python
"""Launch rules of the wide restriction kernel and of the 7-entry stencil-field z-march kernel on blocks larger than 512^3 (debug build)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd import lib
from exastencils_amd.field import Stencil, stencil_field_offsets
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps

ops = HipOps(0, lib.DBG_LIB_PATH)
L = ops.L
L.examg_debug_stencilfield.argtypes = [C.c_int, C.c_int]


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
    u, un, f, fc = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lc.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    b, e = [1, 1, 1], [n, n, n]
    bc, ec = [1, 1, 1], [n // 2] * 3
    Ls, Fs, Cs = lu.c_struct(), lf.c_struct(), lc.c_struct()
    pts, cpts = float(n - 1) ** 3, float(n // 2 - 1) ** 3
    for waves in (1, 16384, 65536, 262144, 1 << 22):
        L.examg_debug_restrict(waves)
        t = timed(lambda: ops.restrict(Ls, u, Cs, fc, 1.0, bc, ec))
        print("n=%d restriction waves=%8d  %.4f ms (frac %.3f)" % (n, 4096 if waves == 1 else waves, t, 72 * cpts / t / 1e6 / 8000), flush=True)
    L.examg_debug_restrict(1)
    cf = ops.new_array(7 * lf.size)
    ops.fill_random(cf, 3)
    cf += 3.0
    st = Stencil(stencil_field_offsets(3), [], cf, lf)
    for blocks in (2048, 8192, 32768, 131072, 1 << 22):
        L.examg_debug_stencilfield(1, blocks)
        t = timed(lambda: ops.stencil_op(2, Ls, u, Fs, f, Ls, un, st, 0.8, -1, b, e))
        print("n=%d 7-entry stencil field blocks=%8d  jacobi %.4f ms (frac %.3f)" % (n, blocks, t, 80 * pts / t / 1e6 / 8000), flush=True)
    L.examg_debug_stencilfield(1, 0)
    del u, un, f, fc, cf, st
    torch.cuda.empty_cache()
