"""Start values of `loop over F sequentially { F = native ( "((double)std::rand()/RAND_MAX)" ) }` (Testing/Opts/base.exa4:166-170,
Testing/Misc/inlining.exa4:199-203, Testing/PolyExpl/Jac3Dcc.exa4:32-41): the values come from libexamg's restatement of glibc's rand()
(include/examg.h: examg_crand_seed / examg_crand_draw_host, host code), are placed on the host in the loop order of the generated
nest -- x fastest, the statements of the loop body drawing one after the other at every point -- and uploaded.

Which generator a point's value comes from follows the reference's processes: the generated main() of an MPI program calls
std::srand(mpiRank) (Compiler/src/exastencils/parallelization/api/mpi/MPI_IVs.scala:41-45; srand(0) seeds like the default, 1),
and every process fills its own loop box.  With one process per block here that is this process' generator; when the blocks of a
knowledge file are merged into one fragment (one-process runs of a several-process test) the merged grid is filled block by
block from one generator per former process -- highest rank first, because the duplicate planes two processes share end up with
the value of the LOWER one (`communicate`: own upper duplicate plane -> the upper neighbour's lower one, axis by axis)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .domain import RectDomain
from .layout import FieldLayout


class CRand:
    """One C-library generator (glibc's rand(), restated in libexamg): seed like std::srand, draw like (double)std::rand()/RAND_MAX."""

    def __init__(self, seed: int = 1):
        from . import lib as _lib

        self._lib, self._L = _lib, _lib.load()
        self._st = _lib.CrandStateC()
        self.seed(seed)

    def seed(self, seed: int):
        self._lib.check(self._L.examg_crand_seed(C.byref(self._st), int(seed) & 0xFFFFFFFF), "examg_crand_seed")

    def draw(self, n: int) -> np.ndarray:
        out = np.empty(int(n), dtype=np.float64)
        self._lib.check(self._L.examg_crand_draw_host(C.byref(self._st), out.ctypes.data_as(C.c_void_p), int(n)), "examg_crand_draw_host")
        return out


def fill_boxes(ops, targets: Sequence[Tuple[object, Optional[int]]], fills: List[Tuple[CRand, Sequence[int], Sequence[int]]]):
    """targets: (field, slot) per statement of the loop body, in statement order (same layout); fills: (generator, begin, end)
    boxes in the order they are to be written.  Every point draws len(targets) values in a row."""
    lay = targets[0][0].layout
    hosts = [np.ascontiguousarray(f.host_array(ops, s), dtype=np.float64).copy() for f, s in targets]
    k = len(targets)
    for gen, b, e in fills:
        n = [max(0, e[d] - b[d]) for d in range(3)]
        if n[0] * n[1] * n[2] == 0:
            continue
        vals = gen.draw(n[0] * n[1] * n[2] * k).reshape(n[2], n[1], n[0], k)
        sl = tuple(slice(lay.ref(d) + b[d], lay.ref(d) + e[d]) for d in (2, 1, 0))
        for j in range(k):
            hosts[j][sl] = vals[..., j]
    for (f, s), h in zip(targets, hosts):
        f.set_host_array(ops, h, s)


def random_start(ops, field, slot: Optional[int], domain: RectDomain, former_processes: Optional[Sequence[int]] = None,
                 generator: Optional[CRand] = None, more_targets: Sequence[Tuple[object, Optional[int]]] = ()):
    """`field<slot>` (and `more_targets`, the further statements of the same loop body) over the loop's box.  `generator`: the
    process' generator when the program seeded it itself (std::srand) or drew from it before; None: the reference's default per
    process (seed = rank, 1 without MPI)."""
    lay, nd = field.layout, domain.nd
    targets = [(field, slot)] + list(more_targets)
    procs = tuple(former_processes) if former_processes is not None else (1, 1, 1)
    fills = []
    if domain.world_size > 1:
        b, e = domain.loop_bounds(lay)
        fills.append((generator or CRand(domain.rank), b, e))
    elif procs != (1, 1, 1):
        if generator is not None or len(targets) != 1:
            raise NotImplementedError("merged blocks: one statement per loop and the default seeding only")
        flen = tuple(domain.frag_len[d] // procs[d] for d in range(3))
        if any(flen[d] * procs[d] != domain.frag_len[d] for d in range(3)):
            raise ValueError("the merged fragment is not a whole number of former blocks")
        for r in reversed(range(procs[0] * procs[1] * procs[2])):
            sub = RectDomain(nd, procs, r, flen)
            nc = sub.ncells(field.level)
            sb, se = sub.loop_bounds(FieldLayout.node(nd, nc, lay.ghost[0]))
            b = [sub.pos[d] * nc[d] + sb[d] if d < nd else 0 for d in range(3)]
            e = [sub.pos[d] * nc[d] + se[d] if d < nd else 1 for d in range(3)]
            fills.append((CRand(r), b, e))
    else:
        b, e = domain.loop_bounds(lay)
        fills.append((generator or CRand(1), b, e))
    fill_boxes(ops, targets, fills)
