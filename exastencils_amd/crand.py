"""Start values of `loop over F sequentially { F = native ( "((double)std::rand()/RAND_MAX)" ) }` (Testing/Opts/base.exa4:166-170,
Testing/Misc/inlining.exa4:199-203): the values come from libexamg's restatement of glibc's rand() (include/examg.h:
examg_crand_seed / examg_crand_fill_host, host code), are written on the host in the loop order of the generated nest and uploaded.

Which generator a point's value comes from follows the reference's processes: the generated main() of an MPI program calls
std::srand(mpiRank) (Compiler/src/exastencils/parallelization/api/mpi/MPI_IVs.scala:41-45; srand(0) seeds like the default, 1),
and every process fills its own loop box.  With one process per block here that is this process' generator; when the blocks of a
knowledge file are merged into one fragment (one-process runs of a several-process test) the merged grid is filled block by
block from one generator per former process -- highest rank first, because the duplicate planes two processes share end up with
the value of the LOWER one (`communicate`: own upper duplicate plane -> the upper neighbour's lower one, axis by axis)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from .domain import RectDomain
from .layout import FieldLayout


def random_start(ops, field, slot: Optional[int], domain: RectDomain, former_processes: Optional[Sequence[int]] = None):
    from . import lib as _lib

    L = _lib.load()
    lay, nd = field.layout, domain.nd
    host = np.ascontiguousarray(ops.to_host(field.data(slot)), dtype=np.float64).copy()
    lc = lay.c_struct()

    def fill(seed, b, e):
        st = _lib.CrandStateC()
        _lib.check(L.examg_crand_seed(C.byref(st), int(seed)), "examg_crand_seed")
        _lib.check(L.examg_crand_fill_host(C.byref(lc), host.ctypes.data_as(C.c_void_p), _lib.ivec(b), _lib.ivec(e), C.byref(st)),
                   "examg_crand_fill_host")

    procs = tuple(former_processes) if former_processes is not None else (1, 1, 1)
    if domain.world_size > 1:
        b, e = domain.loop_bounds(lay)
        fill(domain.rank, b, e)
    elif procs != (1, 1, 1):
        flen = tuple(domain.frag_len[d] // procs[d] for d in range(3))
        if any(flen[d] * procs[d] != domain.frag_len[d] for d in range(3)):
            raise ValueError("the merged fragment is not a whole number of former blocks")
        for r in reversed(range(procs[0] * procs[1] * procs[2])):
            sub = RectDomain(nd, procs, r, flen)
            nc = sub.ncells(field.level)
            sb, se = sub.loop_bounds(FieldLayout.node(nd, nc, lay.ghost[0]))
            b = [sub.pos[d] * nc[d] + sb[d] if d < nd else 0 for d in range(3)]
            e = [sub.pos[d] * nc[d] + se[d] if d < nd else 1 for d in range(3)]
            fill(r, b, e)
    else:
        b, e = domain.loop_bounds(lay)
        fill(1, b, e)
    field.data(slot).copy_(ops.from_host(host))
