#!/usr/bin/env python
"""Parts of the red-black sweep of a 512^3 block WITH neighbours across its z faces, each alone (no overlap): the interior pass on the
shrunk boxes, a ghost exchange with itself, the first- and second-stage shell launches; and the whole overlapped pass."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.comm import Communicator
from exastencils_amd.domain import RectDomain
from exastencils_amd.field import Field, laplace_fd
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps
from exastencils_amd.smoothers import rbgs_sweep

ops = HipOps(0)
n = 512
axes = sys.argv[1] if len(sys.argv) > 1 else "z"
dom = RectDomain(3, (1, 1, 1), 0, periodic=tuple(a in axes for a in "xyz"))
comm = Communicator(dom, ops, concurrent_ghost_axes=True, consistent_duplicates=True, transport="peer")
lay = FieldLayout.node(3, dom.ncells(9), 1, True, True, 0)
S = Field("Solution", 9, lay, ops, 1, None)
T = Field("Tmp", 9, lay, ops, 1, None)
F = Field("RHS", 9, FieldLayout.node(3, dom.ncells(9), 0, True, False, 0), ops, 1, None)
alt = ops.new_array(lay.size)
ops.fill_random(S.data(), 1); ops.fill_random(F.data(), 2)
A = laplace_fd(3, (1.0 / n,) * 3)
w = 0.8 / A.diag
b, e = dom.loop_bounds(lay)
faces = [(d, s) for d in range(3) for s in (-1, 1) if dom.neighbor(d, s) is not None]


def shrunk(k):
    bb, ee = list(b), list(e)
    for d, s in faces:
        if s < 0: bb[d] = b[d] + k
        else: ee[d] = e[d] - k
    return bb, ee


def slab(d, s, k):
    sb, se = list(b), list(e)
    if s < 0: se[d] = b[d] + k
    else: sb[d] = e[d] - k
    return sb, se


b1, e1 = shrunk(1); b2, e2 = shrunk(2)
state = {"alt": alt}


def whole():
    state["alt"] = rbgs_sweep(ops, comm, dom, S, F, A, w, state["alt"], T, 0)


cases = {"lone block sweep (no neighbours)": lambda: ops.rbgs_sweep_fused(S.lc, S.data(), alt, F.lc, F.data(), A, w, 0, b, e),
         "interior pass, shrunk boxes": lambda: ops.rbgs_sweep_fused_boxes(S.lc, S.data(), alt, None, F.lc, F.data(), A, w, 0, b1, e1, b2, e2),
         "ghost exchange": lambda: comm.exchange(S, None, "ghost", True),
         "first-stage shell launches": lambda: [ops.stencil_op(2, S.lc, S.data(), F.lc, F.data(), S.lc, T.data(), A, w, 0, *slab(d, s, 3)) for d, s in faces],
         "second-stage shell launches": lambda: [ops.stencil_op(2, S.lc, T.data(), F.lc, F.data(), S.lc, alt, A, w, 1, *slab(d, s, 2)) for d, s in faces],
         "whole overlapped pass": whole}


def timed(fn, reps=20):
    e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1_.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1_) / reps * 1e3


for k, fn in cases.items():
    for _ in range(3):
        fn()
    print("%-36s %8.1f us" % (k, statistics.median([timed(fn) for _ in range(3)])), flush=True)
comm.check()
