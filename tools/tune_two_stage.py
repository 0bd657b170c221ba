#!/usr/bin/env python
"""Sweep the launch knobs of the two-stage kernel (fused red-black sweep / two Jacobi steps) on the GPU box."""
import ctypes as C, itertools, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.field import laplace_fd
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps

level = int(sys.argv[1]) if len(sys.argv) > 1 else 9
ops = HipOps(0)
L = ops.L
L.examg_debug_two_stage.argtypes = [C.c_int] * 4
L.examg_debug_two_stage_lds.argtypes = [C.c_int]
n = 1 << level
lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
ops.fill_random(u, 1); ops.fill_random(f, 2)
A = laplace_fd(3, (1.0 / n,) * 3)
w = 0.8 / A.diag
b, e = [1, 1, 1], [n, n, n]
Ls, Fs = lu.c_struct(), lf.c_struct()
res = []
for kind in ("jacobi2", "rbgs"):
    ref = None
    configs = [(0, 8, 0, 4096)]
    configs += [(nw, 0, remap, blocks) for nw in (5, 8) for remap in (0, 1) for blocks in (2048, 3072, 4096)]
    for lds, wy, remap, blocks in configs:
        L.examg_debug_two_stage_lds(lds)
        L.examg_debug_two_stage(0, blocks, remap, wy if wy else 8)
        fn = (lambda: ops.jacobi2(Ls, u, un, None, Fs, f, A, w, b, e)) if kind == "jacobi2" else (lambda: ops.rbgs_sweep_fused(Ls, u, un, Fs, f, A, w, 0, b, e))
        un.zero_(); fn(); torch.cuda.synchronize()
        chk = un.clone()
        ref = chk if ref is None else ref
        same = bool(torch.equal(chk, ref))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        r = dict(kind=kind, lds=lds, wy=wy, remap=remap, blocks=blocks, ms=ms, same=same)
        res.append(r); print(r, flush=True)
for kind in ("jacobi2", "rbgs"):
    print("best", kind, sorted([r for r in res if r["kind"] == kind], key=lambda r: r["ms"])[:4])
