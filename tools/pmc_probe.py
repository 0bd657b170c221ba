#!/usr/bin/env python
"""Launch the two hot kernels under a list of launch configurations, 3 dispatches each, so that a
`rocprofv3 --pmc FETCH_SIZE` (or WRITE_SIZE) run can attribute fabric traffic to each configuration
(dispatch order = configuration order, written to gpurun_out/pmc_probe_order.json)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from exastencils_amd import lib
import tune_jacobi
lib.LIB_PATH = tune_jacobi.build_tune()
from exastencils_amd.field import laplace_fd
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps

ops = HipOps(0)
L = ops.L
L.examg_debug_tune.argtypes = [C.c_char_p, C.c_int]
L.examg_debug_two_stage.argtypes = [C.c_int] * 4
n = 512
lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
ops.fill_random(u, 1); ops.fill_random(f, 2)
A = laplace_fd(3, (1.0 / n,) * 3)
w = 0.8 / A.diag
b, e = [1, 1, 1], [n, n, n]
Ls, Fs = lu.c_struct(), lf.c_struct()
order = []
torch.cuda.synchronize()
for remap in (0, 1):
    for blocks in (1024, 4096):
        for ry, wy in ((2, 4), (4, 4), (2, 8)):
            for k, v in dict(ry=ry, wy=wy, nt=1, pf=1, remap=remap, blocks=blocks).items():
                L.examg_debug_tune(k.encode(), v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.stencil_op(2, Ls, u, Fs, f, Ls, un, A, w, -1, b, e)
            e1.record(); torch.cuda.synchronize()
            order.append(dict(kernel="zmarch", remap=remap, blocks=blocks, ry=ry, wy=wy, ms=e0.elapsed_time(e1) / 3))
for remap in (0, 1):
    for blocks in (1024, 4096):
        for wy in (4, 8):
            L.examg_debug_two_stage(0, blocks, remap, wy)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.jacobi2(Ls, u, un, None, Fs, f, A, w, b, e)
            e1.record(); torch.cuda.synchronize()
            order.append(dict(kernel="two_stage", remap=remap, blocks=blocks, wy=wy, ms=e0.elapsed_time(e1) / 3))
json.dump(order, open(os.path.join(ROOT, "gpurun_out", "pmc_probe_order.json"), "w"), indent=1)
print("ok", len(order))
