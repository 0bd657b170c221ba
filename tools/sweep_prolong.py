#!/usr/bin/env python
"""Planes per workgroup of the prolongation kernel (debug build)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd import lib
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps

ops = HipOps(0, lib.DBG_LIB_PATH)
L = ops.L
for n in (512, 256):
    lu, luc = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n // 2,) * 3, 1)
    u, uc = ops.new_array(lu.size), ops.new_array(luc.size)
    ops.fill_random(u, 1); ops.fill_random(uc, 2)
    b, e = [1, 1, 1], [n, n, n]
    for zb in (1, 2, 4, 8, 16, 32, 64, 128, 511):
        L.examg_debug_prolong(zb)
        fn = lambda: ops.prolong_add(luc.c_struct(), uc, lu.c_struct(), u, b, e)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        print("n=%d zb=%3d %.4f ms" % (n, zb, e0.elapsed_time(e1) / 20), flush=True)
        ops.fill_random(u, 1)
