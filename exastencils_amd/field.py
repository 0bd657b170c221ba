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

# Named point functions of the reference's programs (boundary values, right-hand sides, exact solutions, coefficient profiles).
# They are HOST-side names only: the library knows expression programs (examg_expr_t) and nothing else -- FN_PROGRAMS below
# spells each one as the postfix program of the expression tree written in the reference program (same tree as the oracle's
# orc_eval_fn, hence the same bits up to libm).
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

    # -- host copies in the UNTRANSFORMED layout, [z, y, x] (field I/O, host-side fills): a field under a layout transformation is
    #    brought to the plain layout on the device first (examg_transform_field) -------------------------------------------------------
    def host_array(self, ops, slot: Optional[int] = None):
        x = self.data(slot)
        if self.layout.transform:
            plain = self.layout.plain()
            tmp = ops.new_array(plain.size)
            ops.transform_field(self.lc, x, plain.c_struct(), tmp)
            x = tmp
        return ops.to_host(x).reshape(self.layout.shape_zyx)

    def set_host_array(self, ops, a, slot: Optional[int] = None):
        import numpy as np

        t = ops.from_host(np.ascontiguousarray(a, dtype=np.float64).reshape(-1))
        if self.layout.transform:
            ops.transform_field(self.layout.plain().c_struct(), t, self.lc, self.data(slot))
        else:
            self.data(slot).copy_(t)


@dataclass
class Stencil:
    """Entry list in declaration order (the order the convolution is summed in)."""
    offsets: List[Tuple[int, int, int]]
    coefs: List[float] = _dc_field(default_factory=list)
    cfield: object = None                 # device array with len(offsets) coefficient planes, or None
    clayout: Optional[FieldLayout] = None
    # coefficient field under the layout transformation `[x, y, z, i] => [i, x, y, z]` (EXAMG_CLAYOUT_ENTRY_FASTEST: the entries
    # of a point contiguous -- one stream instead of len(offsets)); 0: the reference layout (entry index slowest)
    ctransform: int = 0
    # smoother weight of a stencil field as the statement writes it: 0 `(1.0 / diag(A)) * omega`, 1 `omega / diag(A)` (EXAMG_WEIGHT_*)
    wform: int = 0

    def entry_fastest(self, ops) -> "Stencil":
        """This stencil field with its coefficients re-laid out by `transform <field> with [x, y, z, i] => [i, x, y, z]`
        (Compiler/src/exastencils/layoutTransformation/; Testing/LayoutTrafo/*.exa4): a new array, the old one is released by the
        caller dropping this object.  Values do not change; kernels read them through the transformed index."""
        if self.cfield is None or self.ctransform == 1:
            return self
        if not hasattr(ops, "transform_stencilfield"):
            return self           # kernel layers without transformed layouts keep the reference layout (same results)
        out = ops.new_array(len(self.offsets) * self.clayout.size)
        ops.transform_stencilfield(self.clayout.c_struct(), len(self.offsets), self.cfield, out, True)
        return Stencil(self.offsets, self.coefs, out, self.clayout, 1, self.wform)

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
            if hasattr(s, "ctransform"):
                s.ctransform = int(self.ctransform)
            if hasattr(s, "wform"):
                s.wform = int(self.wform)
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
    """Entry order of the 27-entry stencil field of examg_init_helmholtz27: centre first, then the offsets with dz = -1, 0, +1 (dz
    slowest, dx fastest) -- the order in which a pass that marches in z can finish the sum of a point one plane at a time
    (csrc/kernels_sf27pair.hip)."""
    return [(0, 0, 0)] + [(a, b, c) for c in (-1, 0, 1) for b in (-1, 0, 1) for a in (-1, 0, 1) if (a, b, c) != (0, 0, 0)]


def _fn_program(fn: int, p0: float):
    """Postfix program [(op, const)] of named point function `fn`; p0 = kappa of the SISC / FMG programs' `Globals { Val kappa }`."""
    import math

    PI = math.pi
    X, Y, Z = ("x", None), ("y", None), ("z", None)

    def c(v):
        return ("const", float(v))

    def mul(a, b):
        return a + b + [("*", None)]

    def add(a, b):
        return a + b + [("+", None)]

    def sub_(a, b):
        return a + b + [("-", None)]

    def call(name, a):
        return a + [(name, None)]

    def bump(v):                       # (v - (v * v))
        return sub_([v], mul([v], [v]))

    xyz = mul(mul(bump(X), bump(Y)), bump(Z))               # ((x - x^2) * (y - y^2)) * (z - z^2)
    xy = mul(bump(X), bump(Y))
    table = {
        FN_ZERO: lambda: [c(0.0)],
        # ( x * x ) - ( ( 0.5 * y ) * y ) - ( ( 0.5 * z ) * z )                 Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:25
        FN_POLY3D: lambda: sub_(sub_(mul([X], [X]), mul(mul([c(0.5)], [Y]), [Y])), mul(mul([c(0.5)], [Z]), [Z])),
        # cos ( PI * x ) - sin ( ( 2.0 * PI ) * y )                              Examples/Poisson/2D_FD_Poisson_fromL4.exa4:26
        FN_TRIG2D_SOL: lambda: sub_(call("cos", mul([c(PI)], [X])), call("sin", mul([c(2.0 * PI)], [Y]))),
        # ( PI * PI ) * cos ( PI * x ) - ( ( 4.0 * ( PI * PI ) ) * sin ( ( 2.0 * PI ) * y ) )      ...exa4:233
        FN_TRIG2D_RHS: lambda: sub_(mul([c(PI * PI)], call("cos", mul([c(PI)], [X]))),
                                    mul([c(4.0 * (PI * PI))], call("sin", mul([c(2.0 * PI)], [Y])))),
        FN_KAPPA_POLY: lambda: mul([c(p0)], xyz),                                                   # Testing/SISC/3D_ConstCoeff.exa4:43
        FN_KAPPA_RHS: lambda: mul([c(2.0 * p0)], add(add(mul(bump(X), bump(Y)), mul(bump(X), bump(Z))), mul(bump(Y), bump(Z)))),
        FN_KAPPA_EXPSOL: lambda: sub_([c(1.0)], call("exp", mul([c(-1.0 * p0)], xyz))),            # Testing/SISC/3D_VarCoeff.exa4:48
        FN_KAPPA_COEF: lambda: call("exp", mul([c(p0)], xyz)),
        # ( sin ( PI * x ) * sin ( PI * y ) ) * sinh ( ( sqrt ( 2.0 ) * PI ) * z )                 Testing/FMG/3D_Trigonometric.exa4:43
        FN_TRIG3D_SOL: lambda: mul(mul(call("sin", mul([c(PI)], [X])), call("sin", mul([c(PI)], [Y]))),
                                   call("sinh", mul([c(math.sqrt(2.0) * PI)], [Z]))),
        FN_SIN3: lambda: mul(mul(call("sin", mul([c(PI)], [X])), call("sin", mul([c(PI)], [Y]))), call("sin", mul([c(PI)], [Z]))),
        FN_KAPPA_POLY2D: lambda: mul([c(p0)], xy),
        FN_KAPPA_RHS2D: lambda: mul([c(2.0 * p0)], add(bump(X), bump(Y))),
        FN_KAPPA_EXPSOL2D: lambda: sub_([c(1.0)], call("exp", mul([c(-1.0 * p0)], xy))),
        FN_KAPPA_COEF2D: lambda: call("exp", mul([c(p0)], xy)),
        FN_POLY2D: lambda: sub_(mul([X], [X]), mul([Y], [Y])),                                      # Testing/BC/2D_Polynomial.exa4:43
        FN_SINSINH2D: lambda: mul(call("sin", mul([c(PI)], [X])), call("sinh", mul([c(PI)], [Y]))),  # Testing/BC/2D_Trigonometric.exa4:43
        FN_XSQ: lambda: mul([X], [X]),                                                               # Testing/BC/2D_Periodic.exa4:43
    }
    if fn not in table:
        raise ValueError("unknown point function id %r" % (fn,))
    return table[fn]()


_FN_EXPR_CACHE = {}


def fn_expr(fn, params=()):
    """examg_expr_t of a named point function (or `fn` itself when it already is an expression program)."""
    from .lib import ExprC

    if not isinstance(fn, int):
        return fn
    p0 = float(params[0]) if len(params) else 0.0
    key = (int(fn), p0)
    e = _FN_EXPR_CACHE.get(key)
    if e is None:
        e = _FN_EXPR_CACHE[key] = ExprC.from_program(_fn_program(int(fn), p0))
    return e
