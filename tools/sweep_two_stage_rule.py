#!/usr/bin/env python
"""Old against new launch rule of the two-stage kernel (debug build, explicit knobs), interleaved:
plain passes  old = 8192 workgroups / >= 16 planes per chunk,  new = 12 planes per chunk;
folded pass   old = 3072 workgroups,                           new = 2048 below 2*10^8 points, 8192 from there."""
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


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in [int(a) for a in sys.argv[1:]] or [384, 448, 512, 640]:
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
    u, un, tmp, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 1); ops.fill_random(f, 2)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Ls, Fs = lu.c_struct(), lf.c_struct()
    lc = FieldLayout.node(3, (n // 2,) * 3, 1)
    uc = ops.new_array(lc.size)
    ops.fill_random(uc, 3)
    for _ in range(60):
        ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e)
    pts = (n - 1) ** 3
    res = {}
    for rnd in range(3):
        for name, (blocks, minzc) in (("plain old", (8192, 16)), ("plain new", (1 << 20, 12))):
            L.examg_debug_two_stage(0, blocks, -1, minzc)
            res.setdefault(name + " jacobi2", []).append(timed(lambda: ops.jacobi2(Ls, u, un, tmp, Fs, f, A, w, b, e)))
            res.setdefault(name + " sweep", []).append(timed(lambda: ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e)))
        for name, blocks in (("folded old", 3072), ("folded new", 8192 if pts >= 200000000 else 2048)):
            L.examg_debug_two_stage(0, blocks, -1, 0)
            res.setdefault(name, []).append(timed(lambda: ops.rbgs_sweep_fused_prolong(Ls, u, un, Fs, f, A, w, 0, b, e, lc.c_struct(), uc)))
    for k, v in res.items():
        print("n=%d %-20s median %.4f ms  (%s)" % (n, k, sorted(v)[1], " ".join("%.4f" % x for x in v)), flush=True)
    del u, un, tmp, f, uc
    torch.cuda.empty_cache()
