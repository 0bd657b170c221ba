"""Rectangular block decomposition, one fragment per process/GPU.

Reference: domain_rect_numBlocks_{x,y,z} equal blocks, rank = bx + nbx*(by + nby*bz)
(Compiler/src/exastencils/domain/ir/IR_ConnectFragments.scala:46-52), fragment position from the
rank (domain/ir/IR_InitGeneratedDomain.scala:40-104), neighbour validity and iteration offsets
(IR_ConnectFragments.scala:60-73,110-151), grid width h_L = width / (nFragsTotal * fragLen * 2^L)
(domain/ir/IR_DomainFromAABB.scala:31-40).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

from .lib import GeomC


@dataclass
class RectDomain:
    nd: int
    num_blocks: Tuple[int, int, int] = (1, 1, 1)
    rank: int = 0
    frag_len: Tuple[int, int, int] = (1, 1, 1)
    lo: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    hi: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    periodic: Tuple[bool, bool, bool] = (False, False, False)    # domain_rect_periodic_{x,y,z}

    def __post_init__(self):
        self.periodic = tuple(bool(self.periodic[d]) if d < self.nd and d < len(self.periodic) else False for d in range(3))
        self.num_blocks = tuple(int(self.num_blocks[d]) if d < self.nd else 1 for d in range(3))
        self.frag_len = tuple(int(self.frag_len[d]) if d < self.nd else 1 for d in range(3))
        n = self.world_size
        if not (0 <= self.rank < n):
            raise ValueError("rank %d outside the %d blocks" % (self.rank, n))
        r = self.rank
        self.pos = (r % self.num_blocks[0], (r // self.num_blocks[0]) % self.num_blocks[1],
                    r // (self.num_blocks[0] * self.num_blocks[1]))

    @property
    def world_size(self) -> int:
        return self.num_blocks[0] * self.num_blocks[1] * self.num_blocks[2]

    @staticmethod
    def blocks_for(world_size: int, nd: int, scheme: str = "zy") -> Tuple[int, int, int]:
        """Blocks per dimension for a power-of-two number of GPUs.  `domain_rect_numBlocks_*` are free knobs of the reference:
          "zy"   (default) the factors of two go to z and y in turn and never to x (8 -> 1x2x4, 4 -> 1x2x2, 2 -> 1x1x2): with the
                 unit-stride dimension undivided every halo is made of whole rows or planes -- an x-face is 512^2 single doubles,
                 one cache line each, for packing and for the thin shell launches.  Measured with loop-back neighbours
                 (tools/pair_overhead.py, 512^3 block, three interior faces): 0.79 ms per Jacobi pair against 0.86 ms for 2x2x2.
          "cube" x, y, z in turn (8 -> 2x2x2, 4 -> 2x2x1, 2 -> 2x1x1): SURVEY.md 8e / the reference's benchmark configurations
                 (smallest surface; every block of a strong-scaled 512^3 is a 256^3 cube).
        An explicit "bx,by,bz" is parsed by `parse_blocks`."""
        nb = [1, 1, 1]
        if scheme == "zy":
            dims = [2, 1] if nd == 3 else [1]
        elif scheme == "cube":
            dims = [0, 1, 2][:nd]
        else:
            raise ValueError("unknown decomposition scheme %r (zy | cube)" % (scheme,))
        d, n = 0, world_size
        while n > 1:
            if n % 2:
                raise ValueError("world size must be a power of two")
            nb[dims[d % len(dims)]] *= 2
            n //= 2
            d += 1
        return tuple(nb)

    @staticmethod
    def parse_blocks(text: Optional[str], world_size: int, nd: int) -> Tuple[int, int, int]:
        """--blocks: None / "" / "zy" / "cube" (see blocks_for) or explicit "bx,by,bz" whose product must be the world size."""
        if not text:
            return RectDomain.blocks_for(world_size, nd)
        if text in ("zy", "cube"):
            return RectDomain.blocks_for(world_size, nd, text)
        try:
            nb = tuple(int(t) for t in text.replace("x", ",").split(","))
        except ValueError:
            raise ValueError("--blocks %r: expected zy, cube or bx,by,bz" % (text,)) from None
        nb = nb + (1,) * (3 - len(nb))
        if len(nb) != 3 or min(nb) < 1 or nb[0] * nb[1] * nb[2] != world_size or any(nb[d] != 1 for d in range(nd, 3)):
            raise ValueError("--blocks %r does not describe %d blocks in %d dimensions" % (text, world_size, nd))
        return nb

    def rank_of(self, pos: Sequence[int]) -> int:
        return pos[0] + self.num_blocks[0] * (pos[1] + self.num_blocks[1] * pos[2])

    def neighbor(self, d: int, side: int) -> Optional[int]:
        """Rank of the axis neighbour, None on a physical boundary (neighbor_isValid false)."""
        q = list(self.pos)
        q[d] += side
        if d >= self.nd:
            return None
        if not (0 <= q[d] < self.num_blocks[d]):
            if not self.periodic[d]:
                return None
            q[d] %= self.num_blocks[d]          # periodic: the block at the other end (this block itself when there is one)
        return self.rank_of(q)

    def ncells(self, level: int) -> Tuple[int, int, int]:
        return tuple(self.frag_len[d] * (1 << level) if d < self.nd else 0 for d in range(3))

    def h(self, level: int) -> Tuple[float, float, float]:
        return tuple(
            (self.hi[d] - self.lo[d]) / (self.num_blocks[d] * self.frag_len[d] * (1 << level)) if d < self.nd else 0.0
            for d in range(3))

    def geom(self, level: int) -> GeomC:
        g = GeomC()
        h = self.h(level)
        for d in range(3):
            w = (self.hi[d] - self.lo[d]) / self.num_blocks[d]
            g.pos_begin[d] = self.lo[d] + self.pos[d] * w if d < self.nd else 0.0
            g.h[d] = h[d]
        return g

    def face_mask(self) -> int:
        """bit (2*d + (side>0)) set <=> no neighbour across that face."""
        m = 0
        for d in range(self.nd):
            if self.neighbor(d, -1) is None:
                m |= 1 << (2 * d)
            if self.neighbor(d, +1) is None:
                m |= 1 << (2 * d + 1)
        return m

    def loop_bounds(self, layout, reduction: bool = False):
        """Iteration space of `loop over <field>` (baseExt/ir/IR_LoopOverPointsInOneFragment.scala:84-101):
        [DLB + iterationOffsetBegin, DRE + iterationOffsetEnd), offsets 1 / -1 on a physical boundary and 0
        at an interior block face; reduction loops skip the lower duplicate plane (:116-125)."""
        b, e = [0, 0, 0], [1, 1, 1]
        for d in range(self.nd):
            b[d] = layout.idx("DLB", d) + (0 if self.neighbor(d, -1) is not None else 1)
            e[d] = layout.idx("DRE", d) + (0 if self.neighbor(d, +1) is not None else -1)
            if reduction:
                b[d] = max(b[d], layout.dup[d])
        return b, e
