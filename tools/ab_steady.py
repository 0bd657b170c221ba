#!/usr/bin/env python
"""Launch-geometry choices re-checked in the steady state (after the clock transient of an idle device): every
configuration is timed in several interleaved rounds after a run-in; prints the median per configuration (debug build)."""
import ctypes as C, os, sys, statistics
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
L.examg_debug_residual_restrict.argtypes = [C.c_int] * 2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
lc, lfc = FieldLayout.node(3, (n // 2,) * 3, 1), FieldLayout.node(3, (n // 2,) * 3, 0, True, False)
u, un, f, uc, fc = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lc.size), ops.new_array(lfc.size)
ops.fill_random(u, 1); ops.fill_random(f, 2); ops.fill_random(uc, 3)
A = laplace_fd(3, (1.0 / n,) * 3)
w = 0.8 / A.diag
b, e = [1, 1, 1], [n, n, n]
bc, ec = [1, 1, 1], [n // 2] * 3
Ls, Fs, Lc, Fc = lu.c_struct(), lf.c_struct(), lc.c_struct(), lfc.c_struct()


def timed(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def ab(title, configs, fn):
    for _ in range(150):      # run-in
        fn()
    res = {k: [] for k, _ in configs}
    for _ in range(4):
        for k, setup in configs:
            setup(); fn()
            res[k].append(timed(fn))
    print(title)
    for k, _ in configs:
        print("   %-28s median %.4f  (%s)" % (k, statistics.median(res[k]), " ".join("%.4f" % v for v in res[k])), flush=True)


ts = lambda blocks, minzc=0: (lambda: L.examg_debug_two_stage(0, blocks, -1, minzc))
ab("fused red-black sweep, workgroup target", [("%d" % k, ts(k)) for k in (2048, 3072, 4096, 6144, 8192, 12288)],
   lambda: ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e))
ab("two Jacobi steps, workgroup target", [("%d" % k, ts(k)) for k in (2048, 3072, 4096, 6144, 8192, 12288)],
   lambda: ops.jacobi2(Ls, u, un, None, Fs, f, A, w, b, e))
ab("correction + sweep, workgroup target", [("%d" % k, ts(k)) for k in (2048, 3072, 4096, 6144, 8192)],
   lambda: ops.rbgs_sweep_fused_prolong(Ls, u, un, Fs, f, A, w, 0, b, e, Lc, uc))
ab("zero-field sweep, workgroup target", [("%d" % k, ts(k)) for k in (3072, 4096, 8192)],
   lambda: ops.rbgs_sweep_fused_zero(Ls, un, Fs, f, A, w, 0, b, e))
L.examg_debug_two_stage(0, 8192 if n >= 400 else 3072, -1, 0)
ab("residual + restriction, wave target", [("%d" % k, (lambda k=k: L.examg_debug_residual_restrict(k, 8))) for k in (4096, 12288, 24576, 36864)],
   lambda: ops.residual_restrict(Ls, u, Fs, f, Ls, None, A, Fc, fc, 1.0, b, e, bc, ec))
