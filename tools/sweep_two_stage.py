#!/usr/bin/env python
"""Workgroup-count sweep of the two-stage kernel (debug build: examg_debug_two_stage) at 512^3 and 256^3."""
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
L.examg_debug_two_stage_lds.argtypes = [C.c_int]
for n in (512, 256):
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
    u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Ls, Fs = lu.c_struct(), lf.c_struct()
    for nw in (8, 5):
        L.examg_debug_two_stage_lds(nw)
        for blocks in (256, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192):
            for remap in (0, 1):
                L.examg_debug_two_stage(0, blocks, remap, 0)
                for kind in ("jacobi2", "rbgs"):
                    fn = (lambda: ops.jacobi2(Ls, u, un, None, Fs, f, A, w, b, e)) if kind == "jacobi2" else (lambda: ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e))
                    fn(); torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        fn()
                    e1.record(); torch.cuda.synchronize()
                    print("n=%d nw=%d blocks=%5d remap=%d %-8s %.4f ms" % (n, nw, blocks, remap, kind, e0.elapsed_time(e1) / 20), flush=True)
