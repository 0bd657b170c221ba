#!/usr/bin/env python
"""How much do the face launches, pack/unpack and host dispatch of a block WITH neighbours cost per smoother step?
One GPU, loop-back neighbours (tests/test_gpu_solver.py:_LoopbackComm) on a 2x2x2-decomposition rank, 512^3 block."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.domain import RectDomain
from exastencils_amd.field import Field, laplace_fd
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps
from exastencils_amd.smoothers import jacobi_pair, rbgs_sweep
from exastencils_amd.comm import Communicator


class _LoopbackComm:
    """Every interior face receives this block's own opposite inner planes (pack -> unpack on the current stream)."""

    def __init__(self, domain, ops):
        self.domain, self.ops = domain, ops

    def exchange(self, f, slot=None, what="all", axis_only=False):
        lay, nd = f.layout, self.domain.nd
        x = f.data(slot)
        for d in range(nd):
            for side in (-1, 1):
                if self.domain.neighbor(d, side) is None:
                    continue
                sbox, rbox = Communicator.ghost_ranges(lay, nd, d, side)
                buf = self.ops.new_array(Communicator._count(sbox))
                self.ops.pack(f.lc, x, buf, sbox[0], sbox[1])
                self.ops.unpack(f.lc, x, buf, rbox[0], rbox[1])


ops = HipOps(0)
L = 9
for blocks, rank in (((1, 1, 1), 0), ((2, 1, 1), 0), ((1, 1, 2), 0), ((2, 2, 1), 0), ((1, 2, 2), 0), ((2, 2, 2), 0), ((1, 2, 4), 2)):
    dom = RectDomain(3, blocks, rank)
    lay, layf = FieldLayout.node(3, dom.ncells(L), 1), FieldLayout.node(3, dom.ncells(L), 0, False, False)
    S, F, T = Field("S", L, lay, ops, 2, None), Field("F", L, layf, ops, 1, None), Field("T", L, lay, ops, 1, None)
    ops.fill_random(S.data(0), 1); S.data(1).copy_(S.data(0)); T.data().copy_(S.data(0)); ops.fill_random(F.data(), 3)
    A = laplace_fd(3, dom.h(L)); w = 0.8 / A.diag
    comm = _LoopbackComm(dom, ops)
    for overlap in (False, True):
        for _ in range(5):
            jacobi_pair(ops, comm, dom, S, F, A, w, T, overlap=overlap)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 50
        for _ in range(n):
            jacobi_pair(ops, comm, dom, S, F, A, w, T, overlap=overlap)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("blocks %s rank %d overlap %d: %.3f ms per pair = %.3f ms per step" % (blocks, rank, overlap, dt * 1e3, dt * 5e2), flush=True)
    # red-black sweep of the V-cycle program: fused interior + shell against the two in-place half sweeps
    alt = S.data().clone()
    b, e = dom.loop_bounds(lay)
    for mode in ("plain", "sequential", "overlap"):
        def sweep():
            global alt
            if mode == "plain":
                for colour in (0, 1):
                    comm.exchange(S, None, "ghost")
                    ops.stencil_op(2, S.lc, S.data(), F.lc, F.data(), S.lc, S.data(), A, w, colour, b, e)
            else:
                alt = rbgs_sweep(ops, comm, dom, S, F, A, w, alt, T, 0, overlap=(mode == "overlap"))
        for _ in range(5):
            sweep()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 50
        for _ in range(n):
            sweep()
        torch.cuda.synchronize()
        print("blocks %s rank %d red-black sweep, %s: %.3f ms" % (blocks, rank, mode, (time.perf_counter() - t0) / n * 1e3), flush=True)
