"""Fields, slots and stencils of the ExaSlang-4 surface.

Reference: `Field Solution< global, NodeWithComm, <bc> >[slots]@levels`
(Compiler/src/exastencils/field/ir/IR_Field.scala:37-48), one raw `double*` per (field, level, slot)
(field/ir/IR_FieldData.scala:69-98), `<active>` / `<next>` / `advance` (field/ir/IR_Slot.scala:34-65);
`Stencil` entries with offsets and coefficients (operator/ir/IR_Stencil.scala:34-211), `StencilField`
coefficients stored as a vector-valued field, entry index slowest (stencil/ir/IR_StencilConvolution.scala:73-95).
"""
from __future__ import annotations

from dataclasses import dataclass, field as _dc_field
from typing import List, Optional, Sequence, Tuple

from .layout import FieldLayout
from .lib import StencilC

# analytic function ids (include/examg.h)
FN_ZERO, FN_POLY3D, FN_TRIG2D_SOL, FN_TRIG2D_RHS, FN_KAPPA_POLY, FN_KAPPA_RHS = 0, 1, 2, 3, 4, 5
FN_KAPPA_EXPSOL, FN_KAPPA_COEF, FN_TRIG3D_SOL, FN_SIN3 = 6, 7, 8, 9
FN_KAPPA_POLY2D, FN_KAPPA_RHS2D, FN_KAPPA_EXPSOL2D, FN_KAPPA_COEF2D = 10, 11, 12, 13
FN_POLY2D, FN_SINSINH2D, FN_XSQ = 14, 15, 16


class Field:
    """One field on one level of this process' fragment."""

    def __init__(self, name: str, level: int, layout: FieldLayout, ops, num_slots: int = 1,
                 bc_fn: Optional[int] = FN_ZERO, bc_params: Sequence[float] = ()):
        self.name, self.level, self.layout, self.num_slots = name, level, layout, num_slots
        self.bc_fn, self.bc_params = bc_fn, tuple(bc_params)
        self.lc = layout.c_struct()
        self.slots = [ops.new_array(layout.size) for _ in range(num_slots)]   # initFieldsWithZero
        self.current_slot = 0

    @property
    def active(self) -> int:
        return self.current_slot

    @property
    def next(self) -> int:
        return (self.current_slot + 1) % self.num_slots

    def advance(self):
        self.current_slot = (self.current_slot + 1) % self.num_slots

    def data(self, slot: Optional[int] = None):
        return self.slots[self.current_slot if slot is None else slot]


@dataclass
class Stencil:
    """Entry list in declaration order (the order the convolution is summed in)."""
    offsets: List[Tuple[int, int, int]]
    coefs: List[float] = _dc_field(default_factory=list)
    cfield: object = None                 # device array with len(offsets) coefficient planes, or None
    clayout: Optional[FieldLayout] = None

    @property
    def diag_index(self) -> int:
        return self.offsets.index((0, 0, 0))

    @property
    def diag(self) -> float:
        return self.coefs[self.diag_index]

    def c_struct(self, ptr_of=None) -> StencilC:
        s = StencilC()
        s.nent = len(self.offsets)
        s.diag = self.diag_index
        for k, o in enumerate(self.offsets):
            for d in range(3):
                s.off[k][d] = o[d]
            s.coef[k] = self.coefs[k] if self.coefs else 0.0
        if self.cfield is not None:
            s.cfield = ptr_of(self.cfield)
            s.clayout = self.clayout.c_struct()
        else:
            s.cfield = None
        return s


def _axis(d: int, s: int) -> Tuple[int, int, int]:
    o = [0, 0, 0]
    o[d] = s
    return tuple(o)


def laplace_fd(nd: int, h: Sequence[float], order: str = "mp", squared: str = "pow") -> Stencil:
    """Finite-difference Laplacian scaled by 1/h^2.
    order 'mp': entries c,-x,+x,-y,+y,-z,+z (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:39-47, `h ** 2`);
    order 'pm': entries c,+x,-x,+y,-y,+z,-z (Testing/SISC/3D_ConstCoeff.exa4:55-62, `h * h`)."""
    sq = (lambda v: v ** 2) if squared == "pow" else (lambda v: v * v)
    diag = None
    for d in range(nd):
        t = 2.0 / sq(h[d])
        diag = t if diag is None else diag + t
    offs, co = [(0, 0, 0)], [diag]
    for d in range(nd):
        for s in ((-1, 1) if order == "mp" else (1, -1)):
            offs.append(_axis(d, s))
            co.append(-1.0 / sq(h[d]))
    return Stencil(offs, co)


def laplace_unit(nd: int) -> Stencil:
    """[2*nd; -1] (Testing/Smoothers/Jac.exa4:55-63), entry order c,+x,-x,+y,-y,+z,-z."""
    offs, co = [(0, 0, 0)], [2.0 * nd]
    for d in range(nd):
        for s in (1, -1):
            offs.append(_axis(d, s))
            co.append(-1.0)
    return Stencil(offs, co)


def stencil_field_offsets(nd: int) -> List[Tuple[int, int, int]]:
    """Entry order of `Stencil LaplaceStencil` in Testing/SISC/3D_VarCoeff.exa4:66-74."""
    offs = [(0, 0, 0)]
    for d in range(nd):
        for s in (1, -1):
            offs.append(_axis(d, s))
    return offs


def helmholtz27_offsets() -> List[Tuple[int, int, int]]:
    """Entry order of the 27-entry stencil field of examg_init_helmholtz27: centre first, then (dx,dy,dz) lexicographic."""
    return [(0, 0, 0)] + [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) != (0, 0, 0)]
