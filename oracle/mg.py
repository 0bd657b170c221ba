"""CPU oracle: restatement of the reference's generated multigrid programs.

TEST INFRASTRUCTURE ONLY (see examg_oracle.c).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.

Parity pin: the programs below reproduce the reference's checked-in convergence
histories (tests/golden/*.results, copied data files of /root/reference/Testing) --
see tests/test_oracle_golden.py.

The reference is a code generator (Scala, needs a JVM: not buildable here),
so this module restates the *programs it is given* (ExaSlang-4 files) and the
*semantics it gives them* (loop bounds, layouts, BC ranges, halo ranges):

  ProgramA  Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4 and
            Examples/Poisson/2D_FD_Poisson_fromL4.exa4  (RBGS V(3,3), CG coarse solve)
  ProgramB  Testing/Smoothers/{Jac,RBGS}.exa4, Testing/CommBasic/PureMPI.exa4,
            Testing/SISC/3D_{Const,Var}Coeff.exa4, Testing/FMG/3D_*.exa4
            (generated-from-L3 style: slots, UpResidual/Restriction/Correction functions)

All citations are relative to /root/reference/; C/ = Compiler/src/exastencils/.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libexamg_oracle.so")

MAXE = 27


class LayoutC(C.Structure):
    _fields_ = [("nd", C.c_int32)] + [
        (n, C.c_int32 * 3) for n in ("pad_l", "ghost_l", "dup_l", "inner", "dup_r", "ghost_r", "pad_r")
    ]


class StencilC(C.Structure):
    _fields_ = [
        ("nent", C.c_int32),
        ("diag", C.c_int32),
        ("off", (C.c_int32 * 3) * MAXE),
        ("coef", C.c_double * MAXE),
        ("cfield", C.c_void_p),
        ("clayout", LayoutC),
        ("wform", C.c_int32),
    ]


class GeomC(C.Structure):
    _fields_ = [("pos_begin", C.c_double * 3), ("h", C.c_double * 3)]


_GEN_LIB_PATH = os.path.join(_HERE, "libexamg_oracle_gen.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with the committed Makefile: the checker's build and the generator-shaped one (pragma on the outer
    loop only) that bench.py times as the CPU baseline."""
    src = os.path.join(_HERE, "examg_oracle.c")
    for path in (_LIB_PATH, _GEN_LIB_PATH):
        if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", os.path.basename(path)])
    return _LIB_PATH


_lib = None
_shape = "collapse"


def generator_shape(on: bool = True):
    """Switch the process to the build whose loop nests carry the OpenMP pragma on the outer loop only -- what the generator prints
    with its default `omp_useCollapse = false` (parallelization/api/omp/OMP_Loop.scala:47,103-110).  For timing (bench.py's
    cpu_baseline); results are the same bits either way."""
    global _lib, _shape
    want = "outer" if on else "collapse"
    if want != _shape:
        _shape, _lib = want, None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_GEN_LIB_PATH if _shape == "outer" else _LIB_PATH)
        dp, ip, lp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(LayoutC)
        L.orc_stencil_op.argtypes = [C.c_int, lp, dp, lp, dp, lp, dp, C.POINTER(StencilC), C.c_double, C.c_int, ip, ip]
        L.orc_stencil_op.restype = None
        L.orc_jacobi7_const.argtypes = [lp, dp, lp, dp, dp, C.POINTER(StencilC), C.c_double, ip, ip]
        L.orc_jacobi7_const.restype = None
        L.orc_restrict.argtypes = [lp, dp, lp, dp, C.c_double, ip, ip]
        L.orc_restrict.restype = None
        L.orc_prolong_add.argtypes = [lp, dp, lp, dp, ip, ip]
        L.orc_prolong_add.restype = None
        L.orc_crand_fill.argtypes = [lp, dp, ip, ip, C.c_int]
        L.orc_crand_fill.restype = None
        L.orc_set.argtypes = [lp, dp, C.c_double, ip, ip]
        L.orc_set.restype = None
        L.orc_axpby.argtypes = [lp, dp, lp, dp, C.c_double, C.c_double, ip, ip]
        L.orc_axpby.restype = None
        L.orc_dot.argtypes = [lp, dp, lp, dp, ip, ip]
        L.orc_dot.restype = C.c_double
        L.orc_eval_fn.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double]
        L.orc_eval_fn.restype = C.c_double
        L.orc_fill_fn.argtypes = [lp, dp, C.POINTER(GeomC), C.c_int, C.POINTER(C.c_double), ip, ip]
        L.orc_fill_fn.restype = None
        L.orc_max_err_fn.argtypes = [lp, dp, C.POINTER(GeomC), C.c_int, C.POINTER(C.c_double), ip, ip]
        L.orc_max_err_fn.restype = C.c_double
        L.orc_init_varcoeff7.argtypes = [lp, dp, C.POINTER(GeomC), C.c_int, C.POINTER(C.c_double), ip, ip]
        L.orc_init_varcoeff7.restype = None
        L.orc_init_helmholtz27.argtypes = [lp, dp, C.POINTER(GeomC), C.c_int, C.POINTER(C.c_double), ip, ip]
        L.orc_init_helmholtz27.restype = None
        L.orc_pack.argtypes = [lp, dp, dp, ip, ip]
        L.orc_pack.restype = None
        L.orc_unpack.argtypes = [lp, dp, dp, ip, ip]
        L.orc_unpack.restype = None
        L.orc_fill_random.argtypes = [dp, C.c_long, C.c_uint64]
        L.orc_fill_random.restype = None
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads(min(int(L.orc_num_threads()), cpu_budget()))
        _lib = L
    return _lib


def cpu_budget() -> int:
    """Hardware threads this process may really use: the affinity mask capped by the cgroup CPU quota.  OpenMP's default
    (all host threads) under a quota of a few CPUs oversubscribes them and runs an order of magnitude slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for quota_file, period_file in (("/sys/fs/cgroup/cpu.max", None),
                                    ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            with open(quota_file) as fh:
                txt = fh.read().split()
            if period_file is None:
                q, p = txt[0], txt[1]
            else:
                with open(period_file) as fh:
                    q, p = txt[0], fh.read().split()[0]
            if q != "max" and int(q) > 0:
                n = min(n, max(1, int(q) // int(p)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


# function ids (must match examg_oracle.c and include/examg.h)
FN_ZERO, FN_POLY3D, FN_TRIG2D_SOL, FN_TRIG2D_RHS, FN_KAPPA_POLY, FN_KAPPA_RHS = 0, 1, 2, 3, 4, 5
FN_KAPPA_EXPSOL, FN_KAPPA_COEF, FN_TRIG3D_SOL, FN_SIN3 = 6, 7, 8, 9
FN_KAPPA_POLY2D, FN_KAPPA_RHS2D, FN_KAPPA_EXPSOL2D, FN_KAPPA_COEF2D = 10, 11, 12, 13

APPLY, RESIDUAL, SMOOTH = 0, 1, 2


def _ivec(v: Sequence[int]):
    return (C.c_int * 3)(*[int(x) for x in v])


def _ptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


# ---------------------------------------------------------------------------
# layout  (C/field/ir/IR_FieldLayout.scala:30-129)
# ---------------------------------------------------------------------------


@dataclass(frozen=True)
class Layout:
    nd: int
    inner: Tuple[int, int, int]
    ghost: Tuple[int, int, int]
    dup: Tuple[int, int, int]
    pad_l: Tuple[int, int, int] = (0, 0, 0)
    pad_r: Tuple[int, int, int] = (0, 0, 0)
    comm_dup: bool = True
    comm_ghost: bool = True

    @staticmethod
    def node(nd: int, ncells: Sequence[int], ghost: int, comm_dup=True, comm_ghost=True, align: int = 0) -> "Layout":
        """Node localisation: inner = ncells + 1 - 2*dup (C/fieldlike/l4/L4_FieldLikeLayoutDecl.scala:49-51).
        align>0 reproduces IR_AddPaddingToFieldLayouts (C/field/ir/IR_AddPaddingToFieldLayouts.scala:36-41)
        with simd_vectorSize = align."""
        inner = tuple((ncells[d] + 1 - 2) if d < nd else 1 for d in range(3))
        g = tuple(ghost if d < nd else 0 for d in range(3))
        du = tuple(1 if d < nd else 0 for d in range(3))
        pl, pr = [0, 0, 0], [0, 0, 0]
        if align:
            pl[0] = (align - g[0] % align) % align
            tot = pl[0] + g[0] + du[0] + inner[0] + du[0] + g[0]
            pr[0] = (align - tot % align) % align
        return Layout(nd, inner, g, du, tuple(pl), tuple(pr), comm_dup, comm_ghost)

    def tot(self, d: int) -> int:
        return self.pad_l[d] + self.ghost[d] + self.dup[d] + self.inner[d] + self.dup[d] + self.ghost[d] + self.pad_r[d]

    def ref(self, d: int) -> int:
        return self.pad_l[d] + self.ghost[d]

    @property
    def size(self) -> int:
        return self.tot(0) * self.tot(1) * self.tot(2)

    def c(self) -> LayoutC:
        s = LayoutC()
        s.nd = self.nd
        for d in range(3):
            s.pad_l[d], s.pad_r[d] = self.pad_l[d], self.pad_r[d]
            s.ghost_l[d] = s.ghost_r[d] = self.ghost[d]
            s.dup_l[d] = s.dup_r[d] = self.dup[d]
            s.inner[d] = self.inner[d]
        return s

    def alloc(self) -> np.ndarray:
        return np.zeros(self.size, dtype=np.float64)

    def view(self, a: np.ndarray) -> np.ndarray:
        """[z, y, x] view of a flat field array."""
        return a.reshape(self.tot(2), self.tot(1), self.tot(0))

    # iterator-coordinate markers (iterator 0 == lower duplicate node)
    def it(self, name: str, d: int) -> int:
        g, du, n = self.ghost[d], self.dup[d], self.inner[d]
        return {
            "GLB": -g, "GLE": 0, "DLB": 0, "DLE": du, "IB": du, "IE": du + n,
            "DRB": du + n, "DRE": 2 * du + n, "GRB": 2 * du + n, "GRE": 2 * du + n + g,
        }[name]


# ---------------------------------------------------------------------------
# stencils
# ---------------------------------------------------------------------------


@dataclass
class Stencil:
    offsets: List[Tuple[int, int, int]]
    coefs: List[float]
    cfield: Optional[np.ndarray] = None
    clayout: Optional[Layout] = None

    @property
    def diag_index(self) -> int:
        return self.offsets.index((0, 0, 0))

    def c(self) -> StencilC:
        s = StencilC()
        s.nent = len(self.offsets)
        s.diag = self.diag_index
        for k, o in enumerate(self.offsets):
            for d in range(3):
                s.off[k][d] = o[d]
            s.coef[k] = self.coefs[k] if self.coefs else 0.0
        if self.cfield is not None:
            s.cfield = _ptr(self.cfield)
            s.clayout = self.clayout.c()
        else:
            s.cfield = None
        return s


def laplace_examples(nd: int, h: Sequence[float]) -> Stencil:
    """Stencil Laplace of Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:39-47 (entry order c,-x,+x,-y,+y,-z,+z)."""
    offs, co = [(0, 0, 0)], [0.0]
    diag = None
    for d in range(nd):
        term = 2.0 / (h[d] ** 2)
        diag = term if diag is None else diag + term
    co[0] = diag
    for d in range(nd):
        for s in (-1, 1):
            o = [0, 0, 0]
            o[d] = s
            offs.append(tuple(o))
            co.append(-1.0 / (h[d] ** 2))
    return Stencil(offs, co)


def laplace_tests_scaled(nd: int, h: Sequence[float]) -> Stencil:
    """Stencil Laplace of Testing/SISC/3D_ConstCoeff.exa4:55-62 (entry order c,+x,-x,+y,-y,+z,-z; h*h not h**2)."""
    offs, co = [(0, 0, 0)], [0.0]
    diag = None
    for d in range(nd):
        term = 2.0 / (h[d] * h[d])
        diag = term if diag is None else diag + term
    co[0] = diag
    for d in range(nd):
        for s in (1, -1):
            o = [0, 0, 0]
            o[d] = s
            offs.append(tuple(o))
            co.append(-1.0 / (h[d] * h[d]))
    return Stencil(offs, co)


def laplace_tests_unit(nd: int) -> Stencil:
    """Stencil Laplace of Testing/Smoothers/Jac.exa4:55-63: [2*nd; -1], order c,+x,-x,+y,-y,+z,-z."""
    offs, co = [(0, 0, 0)], [2.0 * nd]
    for d in range(nd):
        for s in (1, -1):
            o = [0, 0, 0]
            o[d] = s
            offs.append(tuple(o))
            co.append(-1.0)
    return Stencil(offs, co)


# ---------------------------------------------------------------------------
# domain decomposition  (C/domain/ir/IR_ConnectFragments.scala:46-151,
#                        C/domain/ir/IR_DomainFromAABB.scala:31-40)
# ---------------------------------------------------------------------------


@dataclass
class Domain:
    nd: int
    nfrag: Tuple[int, int, int]           # total fragments per dim (blocks x fragsPerBlock)
    frag_len: Tuple[int, int, int] = (1, 1, 1)
    lo: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    hi: Tuple[float, float, float] = (1.0, 1.0, 1.0)

    def __post_init__(self):
        self.nfrag = tuple(self.nfrag[d] if d < self.nd else 1 for d in range(3))
        self.frag_len = tuple(self.frag_len[d] if d < self.nd else 1 for d in range(3))

    @property
    def frags(self) -> List[Tuple[int, int, int]]:
        return [(x, y, z) for z in range(self.nfrag[2]) for y in range(self.nfrag[1]) for x in range(self.nfrag[0])]

    def ncells(self, level: int) -> Tuple[int, int, int]:
        return tuple(self.frag_len[d] * (1 << level) if d < self.nd else 0 for d in range(3))

    def h(self, level: int) -> Tuple[float, float, float]:
        return tuple(
            (self.hi[d] - self.lo[d]) / (self.nfrag[d] * self.frag_len[d] * (1 << level)) if d < self.nd else 0.0
            for d in range(3)
        )

    def geom(self, level: int, p: Tuple[int, int, int]) -> GeomC:
        g = GeomC()
        h = self.h(level)
        for d in range(3):
            w = (self.hi[d] - self.lo[d]) / self.nfrag[d]
            g.pos_begin[d] = self.lo[d] + p[d] * w if d < self.nd else 0.0
            g.h[d] = h[d]
        return g

    def has_neigh(self, p, d: int, side: int) -> bool:
        q = p[d] + side
        return 0 <= q < self.nfrag[d]

    def neigh(self, p, d: int, side: int):
        q = list(p)
        q[d] += side
        return tuple(q)


class FragField:
    """One field on one level: an array per fragment and slot (C/field/ir/IR_FieldData.scala:69-98)."""

    def __init__(self, dom: Domain, level: int, layout: Layout, nslots: int = 1, bc_fn: Optional[int] = FN_ZERO,
                 bc_params: Sequence[float] = (0.0,)):
        self.dom, self.level, self.layout, self.nslots = dom, level, layout, nslots
        self.bc_fn, self.bc_params = bc_fn, (C.c_double * 4)(*list(bc_params) + [0.0] * (4 - len(bc_params)))
        self.data: Dict[Tuple[int, int, int], List[np.ndarray]] = {p: [layout.alloc() for _ in range(nslots)] for p in dom.frags}
        self.cur = 0
        self.lc = layout.c()

    # slots (C/field/ir/IR_Slot.scala:34-65)
    @property
    def active(self) -> int:
        return self.cur

    @property
    def next(self) -> int:
        return (self.cur + 1) % self.nslots

    def advance(self):
        self.cur = (self.cur + 1) % self.nslots

    def arr(self, p, slot: Optional[int] = None) -> np.ndarray:
        return self.data[p][self.cur if slot is None else slot]


def loop_bounds(dom: Domain, lay: Layout, p, reduction: bool = False):
    """Iteration space of `loop over <field>` on fragment p, iterator coordinates
    (C/baseExt/ir/IR_LoopOverPointsInOneFragment.scala:84-101): [DLB + iterOffBegin, DRE + iterOffEnd),
    iterOffBegin = 1 / iterOffEnd = -1 on a physical boundary, 0 at an interior face
    (C/domain/ir/IR_ConnectFragments.scala:60-73).  Reductions additionally skip the lower
    duplicate plane (:116-125)."""
    b, e = [0, 0, 0], [1, 1, 1]
    for d in range(dom.nd):
        b[d] = lay.it("DLB", d) + (0 if dom.has_neigh(p, d, -1) else 1)
        e[d] = lay.it("DRE", d) + (0 if dom.has_neigh(p, d, +1) else -1)
        if reduction:
            b[d] = max(b[d], lay.dup[d])
    return b, e


def apply_bc(f: FragField, slot: Optional[int] = None):
    """apply bc (Dirichlet, Node): on every face without neighbour set the duplicate plane,
    tangentially GLB..GRE (C/boundary/ir/IR_ApplyBCFunction.scala:53-83,
    C/boundary/ir/IR_HandleBoundaries.scala:92-118, C/boundary/ir/IR_DirichletBC.scala:37-40)."""
    if f.bc_fn is None:
        return
    L, dom, lay = lib(), f.dom, f.layout
    for p in dom.frags:
        g = dom.geom(f.level, p)
        for d in range(dom.nd):
            for side in (-1, 1):
                if dom.has_neigh(p, d, side):
                    continue
                b, e = [0, 0, 0], [1, 1, 1]
                for t in range(dom.nd):
                    if t == d:
                        b[t], e[t] = (lay.it("DLB", t), lay.it("DLE", t)) if side < 0 else (lay.it("DRB", t), lay.it("DRE", t))
                    else:
                        b[t], e[t] = lay.it("GLB", t), lay.it("GRE", t)
                L.orc_fill_fn(C.byref(f.lc), _ptr(f.arr(p, slot)), C.byref(g), f.bc_fn, f.bc_params, _ivec(b), _ivec(e))


def communicate(f: FragField, slot: Optional[int] = None, what: str = "all"):
    """exch<Field>: duplicate layers (upstream: own upper plane -> '+' neighbour's lower plane),
    then ghost layers, one axis at a time, tangential extent including the ghosts of the other
    axes (C/communication/ir/IR_CommunicateFunction.scala:412-471, IR_PackInfoDuplicate.scala:15-39,
    IR_PackInfoGhost.scala:13-60; comm_onlyAxisNeighbors, comm_syncGhostData, comm_batchCommunication)."""
    L, dom, lay = lib(), f.dom, f.layout
    nd = dom.nd
    if what in ("all", "dup") and lay.comm_dup and max(lay.dup) > 0:
        for d in range(nd):
            for p in dom.frags:
                if not dom.has_neigh(p, d, +1):
                    continue
                q = dom.neigh(p, d, +1)
                sb, se, rb, re_ = [0, 0, 0], [1, 1, 1], [0, 0, 0], [1, 1, 1]
                for t in range(nd):
                    if t == d:
                        sb[t], se[t] = lay.it("DRB", t), lay.it("DRE", t)
                        rb[t], re_[t] = lay.it("DLB", t), lay.it("DLE", t)
                    else:
                        sb[t], se[t] = lay.it("DLB", t), lay.it("DRE", t)
                        rb[t], re_[t] = sb[t], se[t]
                n = int(np.prod([se[t] - sb[t] for t in range(3)]))
                buf = np.empty(n)
                L.orc_pack(C.byref(f.lc), _ptr(f.arr(p, slot)), _ptr(buf), _ivec(sb), _ivec(se))
                L.orc_unpack(C.byref(f.lc), _ptr(f.arr(q, slot)), _ptr(buf), _ivec(rb), _ivec(re_))
    if what in ("all", "ghost") and lay.comm_ghost and max(lay.ghost) > 0:
        for d in range(nd):
            msgs = []
            for p in dom.frags:
                for side in (-1, 1):
                    if not dom.has_neigh(p, d, side):
                        continue
                    q = dom.neigh(p, d, side)
                    g = lay.ghost[d]
                    sb, se, rb, re_ = [0, 0, 0], [1, 1, 1], [0, 0, 0], [1, 1, 1]
                    for t in range(nd):
                        if t == d:
                            if side < 0:   # send first inner planes; neighbour receives in its upper ghost
                                sb[t], se[t] = lay.it("IB", t), lay.it("IB", t) + g
                                rb[t], re_[t] = lay.it("GRB", t), lay.it("GRB", t) + g
                            else:
                                sb[t], se[t] = lay.it("IE", t) - g, lay.it("IE", t)
                                rb[t], re_[t] = lay.it("GLE", t) - g, lay.it("GLE", t)
                        else:
                            sb[t], se[t] = lay.it("GLB", t), lay.it("GRE", t)
                            rb[t], re_[t] = sb[t], se[t]
                    n = int(np.prod([se[t] - sb[t] for t in range(3)]))
                    buf = np.empty(n)
                    L.orc_pack(C.byref(f.lc), _ptr(f.arr(p, slot)), _ptr(buf), _ivec(sb), _ivec(se))
                    msgs.append((q, buf, rb, re_))
            for q, buf, rb, re_ in msgs:
                L.orc_unpack(C.byref(f.lc), _ptr(f.arr(q, slot)), _ptr(buf), _ivec(rb), _ivec(re_))


# ---------------------------------------------------------------------------
# loops
# ---------------------------------------------------------------------------


def stencil_op(mode: int, dom: Domain, u: FragField, u_slot, rhs: Optional[FragField], dst: FragField, dst_slot,
               st_for, w: float, colour: int = -1, over: Optional[FragField] = None):
    """loop over <over> { dst = op(A, u, rhs) } on every fragment."""
    L = lib()
    over = over or dst
    for p in dom.frags:
        b, e = loop_bounds(dom, over.layout, p)
        st = st_for(p)
        sc = st.c()
        L.orc_stencil_op(mode, C.byref(u.lc), _ptr(u.arr(p, u_slot)), C.byref(rhs.lc) if rhs else None,
                         _ptr(rhs.arr(p)) if rhs else None, C.byref(dst.lc), _ptr(dst.arr(p, dst_slot)), C.byref(sc), w,
                         colour, _ivec(b), _ivec(e))


def restrict(dom: Domain, rf: FragField, fc: FragField, scale: float):
    L = lib()
    for p in dom.frags:
        b, e = loop_bounds(dom, fc.layout, p)
        L.orc_restrict(C.byref(rf.lc), _ptr(rf.arr(p)), C.byref(fc.lc), _ptr(fc.arr(p)), scale, _ivec(b), _ivec(e))


def prolong_add(dom: Domain, uc: FragField, uf: FragField):
    L = lib()
    for p in dom.frags:
        b, e = loop_bounds(dom, uf.layout, p)
        L.orc_prolong_add(C.byref(uc.lc), _ptr(uc.arr(p)), C.byref(uf.lc), _ptr(uf.arr(p)), _ivec(b), _ivec(e))


def set_value(dom: Domain, f: FragField, v: float, slot=None):
    L = lib()
    for p in dom.frags:
        b, e = loop_bounds(dom, f.layout, p)
        L.orc_set(C.byref(f.lc), _ptr(f.arr(p, slot)), v, _ivec(b), _ivec(e))


def axpby(dom: Domain, x: FragField, y: FragField, a: float, b_: float, over: Optional[FragField] = None, xslot=None, yslot=None):
    L = lib()
    over = over or y
    for p in dom.frags:
        b, e = loop_bounds(dom, over.layout, p)
        L.orc_axpby(C.byref(x.lc), _ptr(x.arr(p, xslot)), C.byref(y.lc), _ptr(y.arr(p, yslot)), a, b_, _ivec(b), _ivec(e))


def dot(dom: Domain, x: FragField, y: FragField, over: Optional[FragField] = None) -> float:
    """loop over ... with reduction(+): per-fragment sums added in fragment order
    (MPI_Allreduce / OMP reduction order is unspecified in the reference)."""
    L = lib()
    over = over or x
    s = 0.0
    for p in dom.frags:
        b, e = loop_bounds(dom, over.layout, p, reduction=True)
        s = s + L.orc_dot(C.byref(x.lc), _ptr(x.arr(p)), C.byref(y.lc), _ptr(y.arr(p)), _ivec(b), _ivec(e))
    return s


def fill_fn(dom: Domain, f: FragField, fn: int, params: Sequence[float] = (0.0,), slot=None):
    L = lib()
    pp = (C.c_double * 4)(*list(params) + [0.0] * (4 - len(params)))
    for p in dom.frags:
        b, e = loop_bounds(dom, f.layout, p)
        g = dom.geom(f.level, p)
        L.orc_fill_fn(C.byref(f.lc), _ptr(f.arr(p, slot)), C.byref(g), fn, pp, _ivec(b), _ivec(e))


def max_err(dom: Domain, f: FragField, fn: int, params: Sequence[float] = (0.0,), reduction_bounds=False) -> float:
    L = lib()
    pp = (C.c_double * 4)(*list(params) + [0.0] * (4 - len(params)))
    m = 0.0
    for p in dom.frags:
        b, e = loop_bounds(dom, f.layout, p, reduction=reduction_bounds)
        g = dom.geom(f.level, p)
        m = max(m, L.orc_max_err_fn(C.byref(f.lc), _ptr(f.arr(p)), C.byref(g), fn, pp, _ivec(b), _ivec(e)))
    return m


def reduced_prec(x: float) -> str:
    """printWithReducedPrec (C/util/ir/IR_ResolvePrintWithReducedPrec.scala:50-71;
    thresholds C/config/Knowledge.scala:293-305): std::cout with precision 4 (fewer near 1e-12)."""
    if x <= 1.0e-12:
        return "EFFECTIVELY ZERO"
    if x <= 1.0e-11:
        prec = 1
    elif x <= 9.999999999999999e-11:
        prec = 2
    elif x <= 9.999999999999999e-10:
        prec = 3
    else:
        prec = 4
    return "%.*g" % (prec, x)


# ---------------------------------------------------------------------------
# Program A: Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4 (+ the 2-D example)
# ---------------------------------------------------------------------------


@dataclass
class ConfigA:
    nd: int = 3
    min_level: int = 1
    max_level: int = 4
    nfrag: Tuple[int, int, int] = (1, 1, 1)
    frag_len: Tuple[int, int, int] = (1, 1, 1)
    omega: float = 0.8
    n_smooth: int = 3
    tol: float = 1.0e-6
    max_it: int = 100
    cg_max: int = 128
    cg_tol: float = 0.001
    bc_fn: int = FN_POLY3D
    rhs_fn: Optional[int] = None
    sol_fn: Optional[int] = None        # PrintError (2-D example)
    align: int = 0


class ProgramA:
    def __init__(self, cfg: ConfigA):
        self.cfg = cfg
        nd = cfg.nd
        self.dom = Domain(nd, cfg.nfrag, cfg.frag_len)
        self.levels = list(range(cfg.min_level, cfg.max_level + 1))
        lo, hi = cfg.min_level, cfg.max_level
        self.Solution: Dict[int, FragField] = {}
        self.RHS: Dict[int, FragField] = {}
        self.Residual: Dict[int, FragField] = {}
        self.Laplace: Dict[int, Stencil] = {}
        for l in self.levels:
            nc = self.dom.ncells(l)
            with_comm = Layout.node(nd, nc, 1, True, True, cfg.align)
            no_ghost = Layout.node(nd, nc, 0, True, False, cfg.align)
            # Field Solution<global, NodeWithComm, bc>@finest / 0.0 elsewhere  (...exa4:24-25)
            self.Solution[l] = FragField(self.dom, l, with_comm, 1, cfg.bc_fn if l == hi else FN_ZERO)
            self.RHS[l] = FragField(self.dom, l, no_ghost, 1, None)                        # ...exa4:27
            self.Residual[l] = FragField(self.dom, l, no_ghost if l == lo else with_comm, 1, FN_ZERO)  # :29-30
            self.Laplace[l] = laplace_examples(nd, self.dom.h(l))
        nc = self.dom.ncells(lo)
        self.cgTmp0 = FragField(self.dom, lo, Layout.node(nd, nc, 1, True, True, cfg.align), 1, FN_ZERO)  # :32
        self.cgTmp1 = FragField(self.dom, lo, Layout.node(nd, nc, 0, True, False, cfg.align), 1, None)     # :33
        self.log: List[str] = []
        self.res_history: List[float] = []
        self.err_history: List[float] = []
        self.cg_iters: List[int] = []

    # Function ResNorm@(coarsest and finest)  (...exa4:113-119)
    def ResNorm(self, l: int) -> float:
        return math.sqrt(dot(self.dom, self.Residual[l], self.Residual[l]))

    def _residual(self, l: int):
        communicate(self.Solution[l])
        stencil_op(RESIDUAL, self.dom, self.Solution[l], None, self.RHS[l], self.Residual[l], None,
                   lambda p: self.Laplace[l], 0.0)
        apply_bc(self.Residual[l])

    # Function Application (...exa4:251-277)
    def setup(self):
        cfg, hi = self.cfg, self.cfg.max_level
        if cfg.rhs_fn is not None:            # InitRHS@finest (2-D example :231-235)
            fill_fn(self.dom, self.RHS[hi], cfg.rhs_fn)
        apply_bc(self.Solution[hi])

    # Function Solve@finest (...exa4:121-150)
    def Solve(self):
        cfg, hi = self.cfg, self.cfg.max_level
        self._residual(hi)
        initRes = self.ResNorm(hi)
        curRes = initRes
        self.res_history.append(initRes)
        self.log.append(reduced_prec(initRes))
        curIt = 0
        while not (curIt >= cfg.max_it or curRes <= cfg.tol * initRes):
            curIt += 1
            self.mgCycle(hi)
            if cfg.sol_fn is not None:        # PrintError@finest (2-D example :83-95): plain loop bounds
                err = max_err(self.dom, self.Solution[hi], cfg.sol_fn)
                self.err_history.append(err)
                self.log.append(reduced_prec(err))
            self._residual(hi)
            curRes = self.ResNorm(hi)
            self.res_history.append(curRes)
            self.log.append(reduced_prec(curRes))
        self.iterations = curIt
        return curIt

    def _smooth(self, l: int):
        # repeat 3 times { color with { (i0+i1+i2) % 2, communicate; loop; apply bc } }  (...exa4:204-213)
        st = self.Laplace[l]
        w = self.cfg.omega / st.coefs[st.diag_index]          # `0.8 / diag(Laplace)` folded by the generator
        for _ in range(self.cfg.n_smooth):
            for colour in (0, 1):
                communicate(self.Solution[l])
                stencil_op(SMOOTH, self.dom, self.Solution[l], None, self.RHS[l], self.Solution[l], None,
                           lambda p: st, w, colour)
                apply_bc(self.Solution[l])

    # Function mgCycle@(all but coarsest) (...exa4:203-249) / mgCycle@coarsest (:152-201)
    def mgCycle(self, l: int):
        if l == self.cfg.min_level:
            return self._cg(l)
        self._smooth(l)
        self._residual(l)
        communicate(self.Residual[l])
        restrict(self.dom, self.Residual[l], self.RHS[l - 1], 1.0)
        set_value(self.dom, self.Solution[l - 1], 0.0)
        apply_bc(self.Solution[l - 1])
        self.mgCycle(l - 1)
        communicate(self.Solution[l - 1])
        prolong_add(self.dom, self.Solution[l - 1], self.Solution[l])
        apply_bc(self.Solution[l])
        self._smooth(l)

    def _cg(self, l: int):
        dom, A = self.dom, self.Laplace[l]
        Sol, Res, p_, Ap = self.Solution[l], self.Residual[l], self.cgTmp0, self.cgTmp1
        self._residual(l)
        curRes = self.ResNorm(l)
        initRes = curRes
        axpby(dom, Res, p_, 1.0, 0.0)                  # cgTmp0 = Residual
        apply_bc(p_)
        for step in range(self.cfg.cg_max):
            communicate(p_)
            stencil_op(APPLY, dom, p_, None, None, Ap, None, lambda p: A, 0.0)   # cgTmp1 = Laplace * cgTmp0
            alphaNom = dot(dom, Res, Res)
            alphaDenom = dot(dom, p_, Ap, over=p_)
            alpha = alphaNom / alphaDenom if alphaDenom != 0.0 else float("nan")
            axpby(dom, p_, Sol, alpha, 1.0)            # Solution += alpha * cgTmp0
            apply_bc(Sol)
            axpby(dom, Ap, Res, -alpha, 1.0)           # Residual -= alpha * cgTmp1
            apply_bc(Res)
            nextRes = self.ResNorm(l)
            if nextRes <= self.cfg.cg_tol * initRes:
                self.cg_iters.append(step + 1)
                return
            beta = (nextRes * nextRes) / (curRes * curRes)
            axpby(dom, Res, p_, 1.0, beta)             # cgTmp0 = Residual + beta * cgTmp0
            apply_bc(p_)
            curRes = nextRes
        self.cg_iters.append(self.cfg.cg_max)
        self.log.append("Maximum number of cgs iterations (%d) was exceeded" % self.cfg.cg_max)


# ---------------------------------------------------------------------------
# Program B: Testing/Smoothers/{Jac,RBGS}.exa4, CommBasic/PureMPI.exa4, SISC/3D_*.exa4, FMG/3D_*.exa4
# ---------------------------------------------------------------------------


@dataclass
class ConfigB:
    nd: int = 3
    min_level: int = 0
    max_level: int = 4
    nfrag: Tuple[int, int, int] = (1, 1, 1)
    frag_len: Tuple[int, int, int] = (1, 1, 1)
    smoother: str = "jacobi"            # 'jacobi' (2 slots) | 'rbgs'
    omega: float = 0.8
    n_smooth: int = 3
    stencil: str = "unit"               # 'unit' [2nd;-1] | 'scaled' 1/h^2 | 'varcoeff' (7/5-entry stencil field)
    restrict_scale: float = 4.0         # `4.0 * RestrictionStencil` with the unit stencil (Jac.exa4:151)
    tol: float = 1.0e-5
    max_it: int = 100
    cg_max: int = 512
    cg_tol: float = 0.001
    bc_fn: int = FN_POLY3D
    rhs_fn: Optional[int] = None
    sol_fn: Optional[int] = None        # NormError_0 (SISC / FMG programs)
    coef_fn: Optional[int] = None
    kappa: float = 10.0
    fmg: bool = False
    align: int = 0
    ksq: float = 0.0
    rhs_from_solution: bool = False
    # InitSolution of Testing/Opts/base.exa4:166-170: Solution@finest = (double)std::rand()/RAND_MAX in every process of this grid
    # (std::srand(mpiRank) first); None: Solution starts at zero
    init_rand_procs: Optional[Tuple[int, int, int]] = None


def rand_start(dom: "Domain", S: "FragField", procs):
    """Every process of the reference fills ITS loop box of Solution<active>@finest from its own generator, seeded with its rank.
    One fragment per process (decomposed oracle) or all processes' blocks merged into one fragment: then block by block, highest rank
    first -- a duplicate plane shared by two processes ends up with the LOWER one's values (`communicate`, duplicate phase)."""
    lay, nd = S.layout, dom.nd
    nprocs = procs[0] * procs[1] * procs[2]
    if tuple(dom.nfrag) == tuple(procs):
        for p in dom.frags:
            rank = p[0] + procs[0] * (p[1] + procs[1] * p[2])
            b, e = loop_bounds(dom, lay, p)
            lib().orc_crand_fill(C.byref(S.lc), _ptr(S.arr(p)), _ivec(b), _ivec(e), rank if nprocs > 1 else 1)
        return
    if tuple(dom.nfrag) != (1, 1, 1):
        raise ValueError("rand_start: one fragment per process, or one merged fragment")
    sub = Domain(nd, tuple(procs), tuple(dom.frag_len[d] // procs[d] for d in range(3)))
    nc = sub.ncells(S.level)
    sublay = Layout.node(nd, nc, lay.ghost[0])
    p0 = dom.frags[0]
    for p in reversed(sub.frags):
        rank = p[0] + procs[0] * (p[1] + procs[1] * p[2])
        sb, se = loop_bounds(sub, sublay, p)
        b = [p[d] * nc[d] + sb[d] if d < nd else 0 for d in range(3)]
        e = [p[d] * nc[d] + se[d] if d < nd else 1 for d in range(3)]
        lib().orc_crand_fill(C.byref(S.lc), _ptr(S.arr(p0)), _ivec(b), _ivec(e), rank if nprocs > 1 else 1)


class ProgramB:
    def __init__(self, cfg: ConfigB):
        self.cfg = cfg
        nd = cfg.nd
        self.dom = Domain(nd, cfg.nfrag, cfg.frag_len)
        lo, hi = cfg.min_level, cfg.max_level
        self.levels = list(range(lo, hi + 1))
        nslots = 2 if cfg.smoother == "jacobi" else 1
        prm = (cfg.kappa,)
        self.Solution: Dict[int, FragField] = {}
        self.RHS: Dict[int, FragField] = {}
        self.Residual: Dict[int, FragField] = {}
        self.Laplace: Dict[int, Dict[Tuple[int, int, int], Stencil]] = {}
        self.LaplaceCoeff: Dict[int, Dict] = {}
        for l in self.levels:
            nc = self.dom.ncells(l)
            basic = Layout.node(nd, nc, 1, True, True, cfg.align)      # BasicComm / CommFullTempBlockable
            nocomm = Layout.node(nd, nc, 0, False, False, cfg.align)   # NoComm / CommPartTempBlockable
            self.Solution[l] = FragField(self.dom, l, basic, nslots, cfg.bc_fn if l == hi else FN_ZERO, prm)
            self.RHS[l] = FragField(self.dom, l, nocomm, 1, None)
            self.Residual[l] = FragField(self.dom, l, basic, 1, FN_ZERO)
            self.Laplace[l] = {}
            for p in self.dom.frags:
                if cfg.stencil == "unit":
                    st = laplace_tests_unit(nd)
                elif cfg.stencil == "scaled":
                    st = laplace_tests_scaled(nd, self.dom.h(l))
                elif cfg.stencil == "helmholtz27":
                    st = self._init_helmholtz27(l, p, nocomm)
                else:
                    st = self._init_laplace(l, p, nocomm)
                self.Laplace[l][p] = st
        nc = self.dom.ncells(lo)
        self.VecP = FragField(self.dom, lo, Layout.node(nd, nc, 1, True, True, cfg.align), 1, FN_ZERO)
        self.VecGradP = FragField(self.dom, lo, Layout.node(nd, nc, 0, False, False, cfg.align), 1, None)
        self.log: List[str] = []
        self.res_history: List[float] = []
        self.err_history: List[float] = []
        self.cg_iters: List[int] = []

    def _init_laplace(self, l: int, p, lay: Layout) -> Stencil:
        """InitLaplace@l (Testing/SISC/3D_VarCoeff.exa4:206-217): `loop over LaplaceCoeff` bounds."""
        nd = self.cfg.nd
        K = 2 * nd + 1
        cf = np.zeros(K * lay.size)
        b, e = loop_bounds(self.dom, lay, p)
        g = self.dom.geom(l, p)
        pp = (C.c_double * 4)(self.cfg.kappa, 0, 0, 0)
        lc = lay.c()
        lib().orc_init_varcoeff7(C.byref(lc), _ptr(cf), C.byref(g), self.cfg.coef_fn, pp, _ivec(b), _ivec(e))
        st = laplace_tests_scaled(nd, self.dom.h(l))
        return Stencil(st.offsets, [], cf, lay)

    def _init_helmholtz27(self, l: int, p, lay: Layout) -> Stencil:
        cf = np.zeros(27 * lay.size)
        b, e = loop_bounds(self.dom, lay, p)
        g = self.dom.geom(l, p)
        pp = (C.c_double * 4)(self.cfg.kappa, self.cfg.ksq, 0, 0)
        lc = lay.c()
        lib().orc_init_helmholtz27(C.byref(lc), _ptr(cf), C.byref(g), self.cfg.coef_fn, pp, _ivec(b), _ivec(e))
        offs = [(0, 0, 0)] + [(a, b_, c) for c in (-1, 0, 1) for b_ in (-1, 0, 1) for a in (-1, 0, 1) if (a, b_, c) != (0, 0, 0)]
        return Stencil(offs, [], cf, lay)

    # -- leveled functions ------------------------------------------------
    def UpResidual(self, l: int):
        S = self.Solution[l]
        communicate(S, S.active)
        stencil_op(RESIDUAL, self.dom, S, S.active, self.RHS[l], self.Residual[l], None, lambda p: self.Laplace[l][p], 0.0)

    def NormResidual(self, l: int) -> float:
        return math.sqrt(dot(self.dom, self.Residual[l], self.Residual[l]))

    def NormError(self, l: int) -> float:
        return max_err(self.dom, self.Solution[l], self.cfg.sol_fn, (self.cfg.kappa,), reduction_bounds=True)

    def Smoother(self, l: int):
        S, cfg = self.Solution[l], self.cfg
        if cfg.smoother == "jacobi":
            communicate(S, S.active, "ghost")
            if cfg.stencil in ("varcoeff", "helmholtz27"):
                w = cfg.omega
            else:
                st = next(iter(self.Laplace[l].values()))
                w = (1.0 / st.coefs[st.diag_index]) * cfg.omega
            stencil_op(SMOOTH, self.dom, S, S.active, self.RHS[l], S, S.next, lambda p: self.Laplace[l][p], w)
            S.advance()
        else:   # Testing/Smoothers/RBGS.exa4:125-133, colour 0 first
            for colour in (0, 1):
                communicate(S, S.active)
                if cfg.stencil in ("varcoeff", "helmholtz27"):
                    w = cfg.omega
                else:
                    st = next(iter(self.Laplace[l].values()))
                    w = (1.0 / st.coefs[st.diag_index]) * cfg.omega
                stencil_op(SMOOTH, self.dom, S, S.active, self.RHS[l], S, S.active, lambda p: self.Laplace[l][p], w, colour)

    def Restriction(self, l: int):
        communicate(self.Residual[l], None, "ghost")
        restrict(self.dom, self.Residual[l], self.RHS[l - 1], self.cfg.restrict_scale)

    def Correction(self, l: int):
        Sc, Sf = self.Solution[l - 1], self.Solution[l]
        communicate(Sc, Sc.active, "ghost")
        L = lib()
        for p in self.dom.frags:
            b, e = loop_bounds(self.dom, Sf.layout, p)
            L.orc_prolong_add(C.byref(Sc.lc), _ptr(Sc.arr(p)), C.byref(Sf.lc), _ptr(Sf.arr(p)), _ivec(b), _ivec(e))

    def SetSolution(self, l: int, v: float):
        set_value(self.dom, self.Solution[l], v, self.Solution[l].active)

    def VCycle(self, l: int):
        if l == self.cfg.min_level:
            return self.VCycle_0(l)
        for _ in range(self.cfg.n_smooth):
            self.Smoother(l)
        self.UpResidual(l)
        self.Restriction(l)
        self.SetSolution(l - 1, 0.0)
        self.VCycle(l - 1)
        self.Correction(l)
        for _ in range(self.cfg.n_smooth):
            self.Smoother(l)

    def VCycle_0(self, l: int):
        """Coarse-grid CG (Testing/Smoothers/Jac.exa4:75-109)."""
        dom, S, R, P, GP = self.dom, self.Solution[l], self.Residual[l], self.VecP, self.VecGradP
        self.UpResidual(l)
        communicate(R)
        res = self.NormResidual(l)
        initialRes = res
        axpby(dom, R, P, 1.0, 0.0)
        for step in range(self.cfg.cg_max):
            communicate(P)
            stencil_op(APPLY, dom, P, None, None, GP, None, lambda p: self.Laplace[l][p], 0.0, over=P)
            alphaDenom = dot(dom, P, GP, over=P)
            alpha = (res * res) / alphaDenom if alphaDenom != 0.0 else float("nan")
            axpby(dom, P, S, alpha, 1.0, over=S, yslot=S.active)
            axpby(dom, GP, R, -alpha, 1.0, over=S)
            nextRes = self.NormResidual(l)
            if nextRes <= self.cfg.cg_tol * initialRes:
                self.cg_iters.append(step + 1)
                return
            beta = (nextRes * nextRes) / (res * res)
            axpby(dom, R, P, 1.0, beta)
            res = nextRes
        self.cg_iters.append(self.cfg.cg_max)
        self.log.append("Maximum number of cgs iterations (%d) was exceeded" % self.cfg.cg_max)

    # Function Application
    def setup(self):
        cfg, hi = self.cfg, self.cfg.max_level
        if cfg.rhs_from_solution:
            S, F = self.Solution[hi], self.RHS[hi]
            lay = S.layout
            pp = (C.c_double * 4)(cfg.kappa, 0, 0, 0)
            for p in self.dom.frags:
                tmp = lay.alloc()
                gb = [lay.it("GLB", d) if d < cfg.nd else 0 for d in range(3)]
                ge = [lay.it("GRE", d) if d < cfg.nd else 1 for d in range(3)]
                g = self.dom.geom(hi, p)
                lib().orc_fill_fn(C.byref(S.lc), _ptr(tmp), C.byref(g), cfg.sol_fn, pp, _ivec(gb), _ivec(ge))
                b, e = loop_bounds(self.dom, F.layout, p)
                sc = self.Laplace[hi][p].c()
                lib().orc_stencil_op(APPLY, C.byref(S.lc), _ptr(tmp), None, None, C.byref(F.lc), _ptr(F.arr(p)), C.byref(sc), 0.0,
                                     -1, _ivec(b), _ivec(e))
        elif cfg.rhs_fn is not None:
            fill_fn(self.dom, self.RHS[hi], cfg.rhs_fn, (cfg.kappa,))
        if cfg.init_rand_procs is not None:
            rand_start(self.dom, self.Solution[hi], cfg.init_rand_procs)
        for l in self.levels:
            for s in range(self.Solution[l].nslots):
                apply_bc(self.Solution[l], s)
        apply_bc(self.VecP)

    # Function Solve
    def Solve(self):
        cfg, hi = self.cfg, self.cfg.max_level
        self.UpResidual(hi)
        resStart = self.NormResidual(hi)
        res = resStart
        self.res_history.append(res)
        self.log.append(reduced_prec(res))
        if cfg.fmg:
            self.FMG(cfg.min_level)
        numIt = 0
        while not (res < cfg.tol * resStart or numIt >= cfg.max_it):
            numIt += 1
            self.VCycle(hi)
            self.UpResidual(hi)
            res = self.NormResidual(hi)
            self.res_history.append(res)
            if cfg.sol_fn is not None:
                err = self.NormError(hi)
                self.err_history.append(err)
                self.log.append(reduced_prec(err))
            else:
                self.log.append(reduced_prec(res))
        self.log.append(str(numIt))
        self.iterations = numIt
        return numIt

    # -- full multigrid (Testing/FMG/3D_Trigonometric.exa4:189-242) -------------
    def SetFuncDir(self, l: int):
        """`loop over Solution<s> only dup [dir] on boundary { Solution<s> = g }`: region D in the face
        direction, DLB..DRE tangentially (C/baseExt/ir/IR_LoopOverPointsInOneFragment.scala:57-70)."""
        S, dom, lay = self.Solution[l], self.dom, self.Solution[l].layout
        pp = (C.c_double * 4)(self.cfg.kappa, 0, 0, 0)
        for p in dom.frags:
            g = dom.geom(l, p)
            for d in range(dom.nd):
                for side in (-1, 1):
                    if dom.has_neigh(p, d, side):
                        continue
                    b, e = [0, 0, 0], [1, 1, 1]
                    for t in range(dom.nd):
                        if t == d:
                            b[t], e[t] = (lay.it("DLB", t), lay.it("DLE", t)) if side < 0 else (lay.it("DRB", t), lay.it("DRE", t))
                        else:
                            b[t], e[t] = lay.it("DLB", t), lay.it("DRE", t)
                    for s_ in range(S.nslots):
                        lib().orc_fill_fn(C.byref(S.lc), _ptr(S.arr(p, s_)), C.byref(g), self.cfg.bc_fn, pp, _ivec(b), _ivec(e))

    def InitRHS(self, l: int):
        if self.cfg.rhs_fn is not None:
            fill_fn(self.dom, self.RHS[l], self.cfg.rhs_fn, (self.cfg.kappa,))
        else:
            set_value(self.dom, self.RHS[l], 0.0)

    def ResetBC(self, l: int):
        for s_ in range(self.Solution[l].nslots):
            apply_bc(self.Solution[l], s_)

    def FMG(self, l: int):
        self.SetFuncDir(l)
        self.InitRHS(l)
        self.VCycle(l)
        self.Correction(l + 1)
        self.ResetBC(l)
        if l != self.cfg.max_level - 1:
            self.FMG(l + 1)


def compare_with_golden(log: Sequence[str], golden_text: str, eps: float = 1.0e-6) -> List[str]:
    """The reference harness' comparison rule (Testing/run_test.py:12-42): line-wise, numeric lines
    equal within 1e-6 absolute, others exact.  Returns the list of mismatches."""
    want = [t for t in golden_text.split("\n") if t.strip() != ""]
    got = [t for t in log if t.strip() != ""]
    bad = []
    if len(want) != len(got):
        bad.append("line count: got %d want %d" % (len(got), len(want)))
    for i, (g, w) in enumerate(zip(got, want)):
        try:
            if abs(float(g) - float(w)) > eps:
                bad.append("line %d: got %s want %s" % (i, g, w))
        except ValueError:
            if g.strip() != w.strip():
                bad.append("line %d: got %r want %r" % (i, g, w))
    return bad
