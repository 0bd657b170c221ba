#!/usr/bin/env python
"""A/B in one process (debug build): the V(3,3) cycle at 512^3 with and without the small-level one-pass kernels (csrc/kernels_small.hip),
replayed from hipGraphs, interleaved; the coarse CG's iteration count; and the generic half sweep on the plain and on the colour-split
layout (7-point through examg_debug_force_generic, 27-point constant stencil).  usage: python tools/ab_small.py [level]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from exastencils_amd import lib
from exastencils_amd.field import Stencil, laplace_fd
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps
from exastencils_amd.solver import ConfigL4, SolverFromL4

L = int(sys.argv[1]) if len(sys.argv) > 1 else 9
ops = HipOps(0, lib.DBG_LIB_PATH)
out = {"level": L}


def solver(small_off):
    ops.L.examg_debug_small(1 if small_off else 0)
    P = SolverFromL4(ConfigL4(nd=3, min_level=L - 5, max_level=L, tol=1e-6, fused_rbgs=True, fused_residual_restrict=True,
                              fused_prolong_min_points=10_000_000, fused_zero_start=True, fused_residual_norm=True), ops)
    P.setup()
    P._update_residual(L)
    P.capture_cycle()
    ops.L.examg_debug_small(0)
    return P


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


A, B = solver(True), solver(False)
for _ in range(30):
    A.replay_cycle()
    B.replay_cycle()
res = {"chain": [], "small": []}
for rnd in range(5):
    res["chain"].append(timed(A.replay_cycle, 10))
    res["small"].append(timed(B.replay_cycle, 10))
out["vcycle_ms_launch_chain_on_small_levels"] = sorted(res["chain"])[2]
out["vcycle_ms_small_level_kernels"] = sorted(res["small"])[2]
out["cg_iterations_last_cycle"] = float(ops.to_host(B._cg_info)[0])
del A, B
torch.cuda.empty_cache()

n = 1 << L
lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
u, f = ops.new_array(lu.size), ops.new_array(lf.size)
ops.fill_random(u, 1)
ops.fill_random(f, 2)
lus, lfs = lu.split_x(), lf.split_x()
us, fs = ops.new_array(lus.size), ops.new_array(lfs.size)
ops.transform_field(lu.c_struct(), u, lus.c_struct(), us)
ops.transform_field(lf.c_struct(), f, lfs.c_struct(), fs)
b, e = [1, 1, 1], [n, n, n]
pts = float(n - 1) ** 3
A7 = laplace_fd(3, (1.0 / n,) * 3)
offs = [(0, 0, 0)] + [(a, b_, c) for a in (-1, 0, 1) for b_ in (-1, 0, 1) for c in (-1, 0, 1) if (a, b_, c) != (0, 0, 0)]
A27 = Stencil(offs, [26.0] + [-1.0 / (1 + abs(o[0]) + abs(o[1]) + abs(o[2])) for o in offs[1:]])
for name, st in (("7pt", A7), ("27pt", A27)):
    w = 0.8 / st.diag
    ops.L.examg_debug_force_generic(1)
    t_plain = timed(lambda: ops.stencil_op(2, lu.c_struct(), u, lf.c_struct(), f, lu.c_struct(), u, st, w, 0, b, e), 10)
    ops.L.examg_debug_force_generic(0)
    t_split = timed(lambda: ops.stencil_op(2, lus.c_struct(), us, lfs.c_struct(), fs, lus.c_struct(), us, st, w, 0, b, e), 10)
    out["half_sweep_%s_generic_plain_ms" % name] = t_plain
    out["half_sweep_%s_colour_split_ms" % name] = t_split
    out["half_sweep_%s_colour_split_gbs_at_16B_per_point" % name] = 16.0 * pts / t_split / 1e6
t_fast = timed(lambda: ops.stencil_op(2, lu.c_struct(), u, lf.c_struct(), f, lu.c_struct(), u, A7, 0.8 / A7.diag, 0, b, e), 10)
out["half_sweep_7pt_zmarch_plain_ms"] = t_fast
print(json.dumps(out, indent=1))
