#!/usr/bin/env python
"""Tile order (plain vs XCD-contiguous) of the two-stage kernel by block size (debug build)."""
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
for n in (256, 512):
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
    u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Ls, Fs = lu.c_struct(), lf.c_struct()
    for blocks, minzc in ((0, 16), (0, -1), (0, 16), (0, -1)):
        for remap in (2,):
            L.examg_debug_two_stage(0, blocks, -1, minzc)
            fn = lambda: ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            print("n=%d blocks=%5d minzc=%d remap=%d %.4f ms" % (n, blocks, minzc, remap, e0.elapsed_time(e1) / 20), flush=True)
    del u, un, f
