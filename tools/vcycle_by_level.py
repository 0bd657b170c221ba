#!/usr/bin/env python
"""V(3,3) cycle time (hipGraph replay) by finest level, coarsest level fixed at 4: the differences are the cost of each level."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.ops import HipOps
from exastencils_amd.solver import ConfigL4, SolverFromL4

ops = HipOps(0)
prev = 0.0
for L in (5, 6, 7, 8, 9):
    P = SolverFromL4(ConfigL4(nd=3, min_level=4, max_level=L, tol=1e-6, fused_rbgs=True, fused_residual_restrict=True), ops)
    P.setup()
    P.capture_cycle()
    for _ in range(3):
        P.replay_cycle()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        P.replay_cycle()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print("levels 4..%d: %.3f ms   (level %d alone: %.3f ms)" % (L, ms, L, ms - prev), flush=True)
    prev = ms
    del P
    torch.cuda.empty_cache()
