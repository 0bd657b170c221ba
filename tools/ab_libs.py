#!/usr/bin/env python
"""A/B of two builds of libexamg in ONE process, steady state, interleaved rounds: python tools/ab_libs.py libA.so libB.so [n]
(tools/build_round_start_lib.sh builds tools/lab/libexamg_r4a.so, the two-stage kernels as they stood before the plane loops ran in groups of four)"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.field import laplace_fd
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps

paths = sys.argv[1:3]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
opss = [HipOps(0, os.path.abspath(p)) for p in paths]
ops = opss[0]
lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
lc = FieldLayout.node(3, (n // 2,) * 3, 1)
u, un, f, uc = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lc.size)
ops.fill_random(u, 1); ops.fill_random(f, 2); ops.fill_random(uc, 3)
A = laplace_fd(3, (1.0 / n,) * 3)
w = 0.8 / A.diag
b, e = [1, 1, 1], [n, n, n]
Ls, Fs, Lc = lu.c_struct(), lf.c_struct(), lc.c_struct()
lfc = FieldLayout.node(3, (n // 2,) * 3, 0, True, False)
fcz = ops.new_array(lfc.size)
Fcs = lfc.c_struct()
if "--sf27" in sys.argv:
    from exastencils_amd.field import Stencil, helmholtz27_offsets
    lf0 = FieldLayout.node(3, (n, n, n), 0)
    f0 = ops.new_array(lf0.size); ops.fill_random(f0, 5)
    cf = ops.new_array(27 * lf0.size); ops.fill_random(cf, 3); cf[:lf0.size] += 8.0
    rec = Stencil(helmholtz27_offsets(), [], cf, lf0).entry_fastest(ops)
    F0 = lf0.c_struct()
    cases27 = {"27-entry step, records": lambda o: o.stencil_op(2, Ls, u, F0, f0, Ls, un, rec, 0.8, -1, b, e)}
cases = {
    "restriction": lambda o: o.restrict(Ls, un, Fcs, fcz, 1.0, [1, 1, 1], [n // 2] * 3),
    "residual + restriction": lambda o: o.residual_restrict(Ls, u, Fs, f, Ls, None, A, Fcs, fcz, 1.0, b, e, [1, 1, 1], [n // 2] * 3),
    "one Jacobi step": lambda o: o.stencil_op(2, Ls, u, Fs, f, Ls, un, A, w, -1, b, e),
    "residual": lambda o: o.stencil_op(1, Ls, u, Fs, f, Ls, un, A, 0.0, -1, b, e),
    "two Jacobi steps": lambda o: o.jacobi2(Ls, u, un, None, Fs, f, A, w, b, e),
    "fused red-black sweep": lambda o: o.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e),
    "correction + sweep": lambda o: o.rbgs_sweep_fused_prolong(Ls, u, un, Fs, f, A, w, 0, b, e, Lc, uc),
}


if "--sf27" in sys.argv:
    cases = cases27

def timed(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, fn in cases.items():
    for _ in range(150):
        fn(opss[0])
    res = [[], []]
    for _ in range(5):
        for i, o in enumerate(opss):
            fn(o)
            res[i].append(timed(lambda: fn(o)))
    print("%-24s %s %.4f   %s %.4f" % (name, os.path.basename(paths[0]), statistics.median(res[0]), os.path.basename(paths[1]), statistics.median(res[1])), flush=True)
