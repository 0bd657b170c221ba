#!/usr/bin/env python
"""Time the stencil-field (variable-coefficient) Jacobi / residual loops: 7 entries (z-march fast path, all launch
variants, and the generic kernel) and 27 entries (config 4, generic kernel)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from exastencils_amd.field import Stencil, helmholtz27_offsets, stencil_field_offsets
from exastencils_amd.layout import FieldLayout
from exastencils_amd import lib
from exastencils_amd.ops import HipOps

level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ops = HipOps(0, lib.DBG_LIB_PATH)      # debug build: examg_debug_stencilfield / examg_debug_sf27 select the kernel variant
ops.L.examg_debug_stencilfield.argtypes = [C.c_int, C.c_int]
n = 1 << level
lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
ops.fill_random(u, 1)
ops.fill_random(f, 2)
b, e = [1, 1, 1], [n, n, n]
pts = (n - 1) ** 3


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


for name, offs in (("7-entry", stencil_field_offsets(3)), ("27-entry", helmholtz27_offsets())):
    K = len(offs)
    cf = ops.new_array(K * lf.size)
    ops.fill_random(cf, 3)
    cf += 3.0
    st = Stencil(offs, [], cf, lf)
    variants = [(-1, 2048)] + ([(v, bl) for v in (0, 1, 2, 3) for bl in (1024, 2048, 4096)] if K == 7 else [(-2, 0)])
    for mode, mname in ((2, "jacobi"), (1, "residual")):
        for variant, blocks in variants:
            if K == 7:
                ops.L.examg_debug_stencilfield(variant, blocks)
            else:
                ops.L.examg_debug_sf27(1 if variant == -2 else 0)      # -1: generic kernel, -2: unrolled kernel
            ms = timeit(lambda: ops.stencil_op(mode, lu.c_struct(), u, lf.c_struct(), f, lu.c_struct(), un, st, 0.8, -1, b, e))
            bpl = 24 + 8 * K
            print("%-9s %-9s variant %2d blocks %4d n=%d  %.4f ms  %.3e LU/s  %.0f GB/s algorithmic (%d B/LU)"
                  % (name, mname, variant, blocks, n, ms, pts / ms * 1e3, pts * bpl / ms / 1e6, bpl), flush=True)
    del cf, st
