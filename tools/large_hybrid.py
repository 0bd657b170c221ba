#!/usr/bin/env python
"""The reference's `Testing/Large/Hybrid` problem on ONE MI355X: 2560 x 1024 x 2048 cells (5.4e9 unknowns, 43 GB per field
array, ~205 GB resident without the two-step scratch array, ~250 GB with it), levels 0..9, slotted Jacobi V(3,3), CG on
the coarsest level -- the program of Testing/CommBasic/PureMPI.exa4 (the two files differ only in the boundary statements of
the two extra levels), which the reference runs on 4 MPI ranks x 10 OpenMP fragments.  Prints the lines of
Testing/Large/Hybrid.results (tests/golden/Large_Hybrid.results) and the time per cycle.
    python tools/large_hybrid.py [--pairs]      --pairs: two Jacobi steps per pass (needs the scratch array)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.ops import HipOps
from exastencils_amd.solver import ConfigL3, SolverFromL3

pairs = "--pairs" in sys.argv
ops = HipOps(0)
cfg = ConfigL3(nd=3, min_level=0, max_level=9, frag_len=(5, 2, 4), temporal_blocking=pairs, fused_residual_restrict=True)
t0 = time.perf_counter()
P = SolverFromL3(cfg, ops)
P.setup()
torch.cuda.synchronize()
t1 = time.perf_counter()
P.Solve()
torch.cuda.synchronize()
t2 = time.perf_counter()
want = open(os.path.join(ROOT, "tests", "golden", "Large_Hybrid.results")).read().split()
print("\n".join(P.log))
print(json.dumps({"matches_reference_results": P.log == want, "unknowns": 2561 * 1025 * 2049, "setup_s": t1 - t0, "solve_s": t2 - t1,
                  "iterations": P.iterations, "s_per_cycle": (t2 - t1) / max(1, P.iterations), "pairs": pairs,
                  "max_memory_GB": torch.cuda.max_memory_allocated() / 1e9}))
