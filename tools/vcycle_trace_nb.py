#!/usr/bin/env python
"""One V(3,3) cycle at 512^3 WITH block neighbours (the block is its own neighbour across the periodic faces: tools/vcycle_neighbours.py)
replayed from its hipGraph, for `rocprofv3 --kernel-trace` (tools/vcycle_trace_reduce.py lists the kernels after the last k_fill_random)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.comm import Communicator
from exastencils_amd.domain import RectDomain
from exastencils_amd.ops import HipOps
from exastencils_amd.solver import ConfigL4, SolverFromL4

axes = sys.argv[1] if len(sys.argv) > 1 else "z"
L = 9
ops = HipOps(0)
dom = RectDomain(3, (1, 1, 1), 0, periodic=tuple(a in axes for a in "xyz"))
comm = Communicator(dom, ops, concurrent_ghost_axes=True, consistent_duplicates=True, transport="peer")
P = SolverFromL4(ConfigL4(nd=3, min_level=L - 5, max_level=L, tol=1e-6, fused_rbgs=True, fused_residual_restrict=True, agglomerate_level=L - 3,
                          fused_prolong_min_points=10_000_000, fused_zero_start=True, fused_residual_norm=True, deep_halo=True), ops, dom, comm)
P.setup()
P.capture_cycle()
for _ in range(3):
    P.replay_cycle()
torch.cuda.synchronize()
mark = ops.new_array(64)
ops.fill_random(mark, 1)
P.replay_cycle()
torch.cuda.synchronize()
comm.check()
