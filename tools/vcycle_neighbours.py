#!/usr/bin/env python
"""What the V-cycle WITH block neighbours costs on one GPU without a second process in the way: one 512^3 block that is its own
neighbour across the faces of the periodic dimensions (peer-write transport: pack into the own receive slab, flag, unpack -- the kernels,
the side stream, the shells and the flag protocol of real neighbours, minus the link), against the same cycle of the lone block.
bench.py's V-cycle leg for both (agglomerated coarse levels with neighbours), replayed from hipGraphs."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from exastencils_amd.comm import Communicator
from exastencils_amd.domain import RectDomain
from exastencils_amd.ops import HipOps

L = int(sys.argv[1]) if len(sys.argv) > 1 else 9
ops = HipOps(0)
out = {"level": L}
cases = (("lone_block", (False, False, False), 1), ("neighbours_z", (False, False, True), 2),
         ("neighbours_y_z", (False, True, True), 2), ("neighbours_x_y_z", (True, True, True), 2))
if len(sys.argv) > 2:
    cases = [c for c in cases if c[0] in sys.argv[2:]]
for name, per, world in cases:
    dom = RectDomain(3, (1, 1, 1), 0, periodic=per)
    comm = Communicator(dom, ops, concurrent_ghost_axes=True, consistent_duplicates=True, transport="peer" if any(per) else "auto")
    r = bench.vcycle(ops, dom, comm, L, world, check_dups=False)
    out[name] = {k: r[k] for k in ("vcycle_ms", "totalTimeSolve_ms", "solve_iterations", "vcycle_graph", "vcycle_agglomerate_level") if k in r}
    print(name, out[name], file=sys.stderr, flush=True)
    if hasattr(comm, "close"):
        comm.close()
print(json.dumps(out))
