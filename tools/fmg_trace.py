#!/usr/bin/env python
"""bench.py's FMG solve (configs[4]'s algorithm on one 512^3 block) for `rocprofv3 --kernel-trace`: the kernels after the LAST
k_fill_random dispatch are one Solve from the zero state (tools/vcycle_trace_reduce.py <dir> [summary])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.ops import HipOps
from exastencils_amd.solver import ConfigL3, SolverFromL3

L = int(sys.argv[1]) if len(sys.argv) > 1 else 9
ops = HipOps(0)
cfg = ConfigL3(nd=3, min_level=2, max_level=L, smoother="rbgs", omega=1.0, stencil="scaled", restrict_scale=1.0, tol=1e-6,
               cg_max=512, bc_fn=1, fmg=True, fused_rbgs=True, fused_residual_restrict=True, fused_prolong_min_points=10_000_000,
               fused_zero_start=True, fused_residual_norm=True, fused_coarse=True)
P = SolverFromL3(cfg, ops)
P.setup()
P.capture()
for _ in range(2):
    P.reset()
    P.Solve(use_graph=True)
P.reset()
torch.cuda.synchronize()
mark = ops.new_array(64)
ops.fill_random(mark, 1)
P.Solve(use_graph=True)
torch.cuda.synchronize()
print("cycles", P.iterations)
