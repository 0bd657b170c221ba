#!/usr/bin/env python
"""Durations of consecutive two-step Jacobi passes at 512^3 after an idle period and after a run-in: the power-management transient
that bench.py's settle phase (--settle-steps) keeps out of the timed region."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.field import laplace_fd
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps
ops = HipOps(0)
n = 512
lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
ops.fill_random(u, 1); ops.fill_random(f, 2)
A = laplace_fd(3, (1.0 / n,) * 3); w = 0.8 / A.diag
b, e = [1, 1, 1], [n, n, n]
L, F = lu.c_struct(), lf.c_struct()
def pas():
    global u, un
    ops.jacobi2(L, u, un, None, F, f, A, w, b, e); u, un = un, u
for _ in range(3): pas()
for idle_ms in (0, 0, 5, 50, 500):
    torch.cuda.synchronize()
    time.sleep(idle_ms / 1e3)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(12)]
    t0 = time.perf_counter()
    for i in range(11):
        evs[i].record(); pas()
    evs[11].record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("idle %3d ms: issue %.3f ms, wall %.3f ms, passes:" % (idle_ms, t_issue * 1e3, t_all * 1e3), " ".join("%.3f" % evs[i].elapsed_time(evs[i + 1]) for i in range(11)), flush=True)

for pre in (0, 50, 150, 400, 800):
    torch.cuda.synchronize(); time.sleep(0.5)
    for _ in range(pre): pas()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(12)]
    t0 = time.perf_counter()
    for i in range(11):
        evs[i].record(); pas()
    evs[11].record()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("pre %3d passes: wall %.3f ms for 11, passes:" % (pre, t_all * 1e3), " ".join("%.3f" % evs[i].elapsed_time(evs[i + 1]) for i in range(11)), flush=True)
