#!/usr/bin/env python
"""One V(3,3) cycle at 512^3 (levels 4..9) replayed from its hipGraph, for `rocprofv3 --kernel-trace`: the timeline of the
kernels of a cycle (tools/vcycle_trace_reduce.py lists them with the gaps between them).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/vtrace -- python3 $R/tools/vcycle_trace.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.ops import HipOps
from exastencils_amd.solver import ConfigL4, SolverFromL4

L = int(sys.argv[1]) if len(sys.argv) > 1 else 9
ops = HipOps(0)
P = SolverFromL4(ConfigL4(nd=3, min_level=L - 5, max_level=L, tol=1e-6, fused_rbgs=True, fused_residual_restrict=True,
                          fused_prolong_min_points=10_000_000, fused_zero_start=True), ops)
P.setup()
P.capture_cycle()
for _ in range(3):
    P.replay_cycle()
torch.cuda.synchronize()
mark = ops.new_array(64)
ops.fill_random(mark, 1)          # marker: the cycle after the LAST k_fill_random dispatch is the one to read
P.replay_cycle()
torch.cuda.synchronize()
