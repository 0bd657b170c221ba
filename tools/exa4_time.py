#!/usr/bin/env python
"""Host overhead of the ExaSlang-4 interpreter: one V(3,3) red-black cycle at 512^3 (levels 4..9) interpreted from
examples/exa4/poisson3d_rbgs.exa4 against the hand-written driver (exastencils_amd.solver.SolverFromL4, same kernels)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from exastencils_amd import exa4
from exastencils_amd.ops import HipOps
from exastencils_amd.solver import ConfigL4, SolverFromL4

ops = HipOps(0)
lo, hi = int(os.environ.get("LO", 4)), int(os.environ.get("HI", 9))
P = exa4.Exa4Program(open(os.path.join(ROOT, "examples", "exa4", "poisson3d_rbgs.exa4")).read(),
                     dict(dimensionality=3, minLevel=lo, maxLevel=hi), ops=ops)
P._apply_bc(P.fields[("u", hi)], 0)      # Application: apply bc to u@finest
P.call("Defect", hi)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


l0 = P.launches
for _ in range(3):      # first call interpreted, second recorded (auto_graph), then replays
    P.call("Cycle", hi)
t_int = timed(lambda: P.call("Cycle", hi))
per_cycle = (P.launches - l0) // 9
g = P.capture("Cycle", hi)
t_graph = timed(g.replay)
S = SolverFromL4(ConfigL4(nd=3, min_level=lo, max_level=hi, tol=1e-6, fused_coarse=False), ops)
S.setup()
t_drv = timed(lambda: S.mgCycle(hi))
S2 = SolverFromL4(ConfigL4(nd=3, min_level=lo, max_level=hi, tol=1e-6, fused_coarse=True, fused_rbgs=True, fused_residual_restrict=True,
                           fused_prolong_min_points=10_000_000, fused_zero_start=True, fused_residual_norm=True), ops)
S2.setup()
S2.capture_cycle()
t_fused = timed(S2.replay_cycle)
# the same program, one launch per statement (no peepholes, no cross-statement fusions): printed values must agree bit for bit
Q = exa4.Exa4Program(open(os.path.join(ROOT, "examples", "exa4", "poisson3d_rbgs.exa4")).read(),
                     dict(dimensionality=3, minLevel=lo, maxLevel=hi), ops=ops, fuse=False, fuse_coarse_solver=True)
R = exa4.Exa4Program(open(os.path.join(ROOT, "examples", "exa4", "poisson3d_rbgs.exa4")).read(),
                     dict(dimensionality=3, minLevel=lo, maxLevel=hi), ops=ops)
Q.run()
R.run()
print(json.dumps({"levels": [lo, hi], "interpreted_cycle_ms": t_int, "launches_per_cycle": per_cycle, "interpreted_cycle_graph_replay_ms": t_graph,
                  "driver_cycle_ms": t_drv, "driver_all_fusions_graph_replay_ms": t_fused, "fusions": P.fusions,
                  "printed_values_equal_unfused": Q.printed_values == R.printed_values, "solve_iterations": len(R.printed_values) - 1,
                  "launches_whole_solve_fused": R.launches, "launches_whole_solve_unfused": Q.launches}))
