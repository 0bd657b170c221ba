#!/usr/bin/env python
"""Times of the one-pass two-stage kernels at n^3 (default 512 and 256): red-black sweep, Jacobi pair, correction + sweep, zero-start
sweep, and residual + restriction; median of interleaved rounds after a run-in.  Debug build: the register caps of the plain and
the folded passes (1 = uncapped, 4 = 128 VGPRs for four waves per SIMD) are timed side by side."""
import ctypes as C, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd import lib
from exastencils_amd.field import laplace_fd
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps

dbg = "--dbg" in sys.argv
ops = HipOps(0, lib.DBG_LIB_PATH if dbg else None)
L = ops.L
if dbg:
    L.examg_debug_two_stage_prol.argtypes = [C.c_int]
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [512, 256]


def timed(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in sizes:
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
    lc, lfc = FieldLayout.node(3, (n // 2,) * 3, 1), FieldLayout.node(3, (n // 2,) * 3, 0, True, False)
    u, un, f, uc, fc = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lc.size), ops.new_array(lfc.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2); ops.fill_random(uc, 3)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e, bc, ec = [1, 1, 1], [n, n, n], [1, 1, 1], [n // 2] * 3
    Ls, Fs, Lc, Fc = lu.c_struct(), lf.c_struct(), lc.c_struct(), lfc.c_struct()
    base = [("red-black sweep", lambda: ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e), 10),
            ("two Jacobi steps", lambda: ops.jacobi2(Ls, u, un, None, Fs, f, A, w, b, e), 10),
            ("correction + sweep", lambda: ops.rbgs_sweep_fused_prolong(Ls, u, un, Fs, f, A, w, 0, b, e, Lc, uc), 0),
            ("correction + Jacobi pair", lambda: ops.jacobi2_prolong(Ls, u, un, None, Fs, f, A, w, b, e, Lc, uc), 0),
            ("zero-start sweep", lambda: ops.rbgs_sweep_fused_zero(Ls, un, Fs, f, A, w, 0, b, e), None),
            ("residual + restriction", lambda: ops.residual_restrict(Ls, u, Fs, f, Ls, None, A, Fc, fc, 1.0, b, e, bc, ec), None)]
    cases = []
    for name, fn, knob in base:
        if dbg and knob is not None:
            for cap in (1, 4):
                cases.append(("%s, cap %d" % (name, cap), fn, knob + cap))
        else:
            cases.append((name, fn, None))
    for _ in range(40):
        for _, fn, k in cases:
            if k is not None:
                L.examg_debug_two_stage_prol(k)
            fn()
    res = {k: [] for k, _, _ in cases}
    for _ in range(4):
        for k, fn, kn in cases:
            if kn is not None:
                L.examg_debug_two_stage_prol(kn)
            fn(); res[k].append(timed(fn))
    for k, _, _ in cases:
        print("n=%d %-32s median %.4f ms  (%s)" % (n, k, statistics.median(res[k]), " ".join("%.4f" % v for v in res[k])), flush=True)
