#!/usr/bin/env python
"""bench.py's 27-entry Helmholtz V(3,3) cycle (configs[3]'s operator on one block) for `rocprofv3 --kernel-trace`: the kernels after the
LAST k_fill_random dispatch are one cycle (tools/vcycle_trace_reduce.py <dir> [summary])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.ops import HipOps
from exastencils_amd.solver import ConfigL3, SolverFromL3

L = int(sys.argv[1]) if len(sys.argv) > 1 else 9
ops = HipOps(0)
cfg = ConfigL3(nd=3, min_level=1, max_level=L - 1, frag_len=(2, 2, 2), smoother="jacobi", omega=0.8, stencil="helmholtz27",
               restrict_scale=1.0, tol=1e-8, cg_max=512, bc_fn=0, sol_fn=9, coef_fn=7, kappa=10.0, ksq=2.0, rhs_from_solution=True,
               fused_coarse=True, coef_entry_fastest=True, temporal_blocking=True, fused_smooth_residual=True)
P = SolverFromL3(cfg, ops)
P.setup()
P.capture()
run = P._graphs["cycle"].replay
for _ in range(2):
    run()
torch.cuda.synchronize()
mark = ops.new_array(64)
ops.fill_random(mark, 1)
run()
torch.cuda.synchronize()
