"""Field layouts: the reference's per-dimension regions, kept verbatim because they are the
drop-in contract (Compiler/src/exastencils/field/ir/IR_FieldLayout.scala:30-129):

    pad | ghost | dup | inner | dup | ghost | pad        x fastest, referenceOffset = pad_l + ghost_l

Iterator coordinates (what `loop over` bodies and the kernels' begin/end use) put 0 at the lower
duplicate node; array index = iterator + referenceOffset.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence, Tuple

from .lib import LayoutC


@dataclass(frozen=True)
class FieldLayout:
    nd: int
    inner: Tuple[int, int, int]
    ghost: Tuple[int, int, int]
    dup: Tuple[int, int, int]
    pad_l: Tuple[int, int, int] = (0, 0, 0)
    pad_r: Tuple[int, int, int] = (0, 0, 0)
    communicates_dup: bool = True      # `duplicateLayers = [...] with communication`
    communicates_ghost: bool = True    # `ghostLayers = [...] with communication`
    # layout transformation of the program's `LayoutTransformations` block (include/examg.h: EXAMG_LAYOUT_*): 1 = the colour split
    # `[x, y, z] => [x / 2, y, z, x % 2]` (Testing/LayoutTrafo/rbgs.exa4:2).  Regions, iterator coordinates and boxes stay those of the
    # untransformed layout; only where a value lives changes.
    transform: int = 0

    @staticmethod
    def node(nd: int, ncells: Sequence[int], ghost: int, communicates_dup: bool = True,
             communicates_ghost: bool = True, align: int = 0) -> "FieldLayout":
        """`Layout X< Real, Node >`: one duplicate layer per side, inner = cells + 1 - 2*dup
        (fieldlike/l4/L4_FieldLikeLayoutDecl.scala:49-51).  `align` = simd_vectorSize of
        IR_AddPaddingToFieldLayouts (field/ir/IR_AddPaddingToFieldLayouts.scala:36-41): pads x so
        that the first duplicate point and the row length are multiples of `align` doubles."""
        inner = tuple((int(ncells[d]) - 1) if d < nd else 1 for d in range(3))
        if any(i < 0 for i in inner):
            raise ValueError("a Node field needs at least one cell per dimension")
        g = tuple(ghost if d < nd else 0 for d in range(3))
        du = tuple(1 if d < nd else 0 for d in range(3))
        pl, pr = [0, 0, 0], [0, 0, 0]
        if align:
            pl[0] = (align - g[0] % align) % align
            tot = pl[0] + g[0] + du[0] + inner[0] + du[0] + g[0]
            pr[0] = (align - tot % align) % align
        return FieldLayout(nd, inner, g, du, tuple(pl), tuple(pr), communicates_dup, communicates_ghost)

    # -- sizes ---------------------------------------------------------------------------------
    def tot(self, d: int) -> int:
        return self.pad_l[d] + self.ghost[d] + self.dup[d] + self.inner[d] + self.dup[d] + self.ghost[d] + self.pad_r[d]

    def ref(self, d: int) -> int:
        return self.pad_l[d] + self.ghost[d]

    @property
    def size(self) -> int:
        if self.transform == 1:
            return 2 * ((self.tot(0) + 1) // 2) * self.tot(1) * self.tot(2)
        return self.tot(0) * self.tot(1) * self.tot(2)

    def split_x(self) -> "FieldLayout":
        """This layout under the colour split `[x, y, z] => [x / 2, y, z, x % 2]`."""
        from dataclasses import replace

        return replace(self, transform=1)

    def plain(self) -> "FieldLayout":
        from dataclasses import replace

        return replace(self, transform=0)

    def linear(self, i0: int, i1: int = 0, i2: int = 0) -> int:
        """Array index of iterator point (i0, i1, i2) -- the restatement of the library's index map (csrc/examg_common.h: lidx)."""
        a = (i0 + self.ref(0), i1 + self.ref(1), i2 + self.ref(2))
        if self.transform == 1:
            hx = (self.tot(0) + 1) // 2
            return a[0] // 2 + hx * (a[1] + self.tot(1) * (a[2] + self.tot(2) * (a[0] % 2)))
        return a[0] + self.tot(0) * (a[1] + self.tot(1) * a[2])

    @property
    def shape_zyx(self) -> Tuple[int, int, int]:
        return (self.tot(2), self.tot(1), self.tot(0))

    # -- region markers in iterator coordinates (IR_FieldLayout.defIdxById minus referenceOffset) --
    def idx(self, name: str, d: int) -> int:
        g, du, n = self.ghost[d], self.dup[d], self.inner[d]
        table = {
            "GLB": -g, "GLE": 0, "DLB": 0, "DLE": du, "IB": du, "IE": du + n,
            "DRB": du + n, "DRE": 2 * du + n, "GRB": 2 * du + n, "GRE": 2 * du + n + g,
        }
        return table[name]

    def c_struct(self) -> LayoutC:
        s = LayoutC()
        s.nd = self.nd
        for d in range(3):
            s.pad_l[d], s.pad_r[d] = self.pad_l[d], self.pad_r[d]
            s.ghost_l[d] = s.ghost_r[d] = self.ghost[d]
            s.dup_l[d] = s.dup_r[d] = self.dup[d]
            s.inner[d] = self.inner[d]
        s.transform = self.transform
        return s
