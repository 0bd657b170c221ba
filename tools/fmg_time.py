#!/usr/bin/env python
"""BASELINE configs[4] on one GPU: FMG start + red-black V(3,3) cycles on the 512^3 Poisson problem (the driver of
Testing/FMG/3D_Trigonometric.exa4 with the smoother of Testing/Smoothers/RBGS.exa4), with and without the FMG start."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.ops import HipOps
from exastencils_amd.solver import ConfigL3, SolverFromL3

ops = HipOps(0)
out = {}
ONE_PASS = dict(fused_prolong_min_points=10_000_000, fused_zero_start=True, fused_residual_norm=True, fused_coarse=True)
for fmg, extra, name in ((False, ONE_PASS, "v_cycles_only"), (True, {}, "fmg_round1_forms"), (True, ONE_PASS, "fmg")):
    cfg = ConfigL3(nd=3, min_level=2, max_level=9, smoother="rbgs", omega=1.0, stencil="scaled", restrict_scale=1.0, tol=1e-6,
                   cg_max=512, bc_fn=1, fmg=fmg, fused_rbgs=True, fused_residual_restrict=True, **extra)
    P = SolverFromL3(cfg, ops)
    P.setup()
    P.Solve()                       # warm-up (lazy allocations)
    P = SolverFromL3(cfg, ops)
    P.setup()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    P.Solve()
    torch.cuda.synchronize()
    out[name] = {"solve_ms": (time.perf_counter() - t0) * 1e3, "v_cycles": P.iterations,
                                              "residual_reduction": P.res_history[-1] / P.res_history[0]}
# FMG start and cycle replayed from hipGraphs: the launch cost of a compiled host instead of this Python driver's
cfg = ConfigL3(nd=3, min_level=2, max_level=9, smoother="rbgs", omega=1.0, stencil="scaled", restrict_scale=1.0, tol=1e-6,
               cg_max=512, bc_fn=1, fmg=True, fused_rbgs=True, fused_residual_restrict=True, **ONE_PASS)
P = SolverFromL3(cfg, ops)
P.setup()
P.capture()
best = None
for _ in range(3):
    P.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    P.Solve(use_graph=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    best = ms if best is None else min(best, ms)
out["fmg_graphs"] = {"solve_ms": best, "v_cycles": P.iterations, "residual_reduction": P.res_history[-1] / P.res_history[0]}
print(json.dumps(out))
