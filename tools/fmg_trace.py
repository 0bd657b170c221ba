#!/usr/bin/env python
"""One FMG solve at 512^3 (BASELINE configs[4]'s algorithm, tools/fmg_time.py) for `rocprofv3 --kernel-trace`: everything after the
last k_fill_random dispatch is the timed solve (tools/vcycle_trace_reduce.py <dir> summary lists busy time and gaps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.ops import HipOps
from exastencils_amd.solver import ConfigL3, SolverFromL3

ops = HipOps(0)
kw = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    kw[k] = int(v)
cfg = ConfigL3(nd=3, min_level=2, max_level=9, smoother="rbgs", omega=1.0, stencil="scaled", restrict_scale=1.0, tol=1e-6,
               cg_max=512, bc_fn=1, fmg=True, fused_rbgs=True, fused_residual_restrict=True, **kw)
P = SolverFromL3(cfg, ops)
P.setup()
P.Solve()
P = SolverFromL3(cfg, ops)
P.setup()
torch.cuda.synchronize()
mark = ops.new_array(64)
ops.fill_random(mark, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
P.Solve()
torch.cuda.synchronize()
print("solve_ms %.3f  v_cycles %d  reduction %.3e" % ((time.perf_counter() - t0) * 1e3, P.iterations, P.res_history[-1] / P.res_history[0]))
