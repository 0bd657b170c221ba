"""Peepholes of the ExaSlang-4 interpreter (mixin of exa4.Exa4Program): statement groups that run as ONE pass over HBM with the same
bits -- the red-black sweep of a `color with` block (examg_rbgs_sweep_fused), pairs of slotted Jacobi steps and contracting loops
(examg_jacobi2_boxes), and a coarsest-level function that is statement for statement the generated CG solver (examg_cg_coarse*).
The cross-STATEMENT forms (pending loops + liveness) are in exa4_fusion.py."""
from __future__ import annotations

import math
import os
import random
import re
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

from . import knowledge as _knowledge
from .comm import Communicator
from .domain import RectDomain
from .field import Field, Stencil
from .layout import FieldLayout
from .exa4_parser import (Exa4SyntaxError, Exa4Unsupported, FunctionDecl, Parser, _COORD, _GRIDW, _MATH, _arith,  # noqa: F401
                          _colour_cond, _conjuncts, _const_value, _contains, _find_calls, _has_coord, _lower_cond, _parity_expr, _walk)
from .exa4_common import (APPLY, RESIDUAL, SMOOTH, _FN_2D_ONLY, _FN_ANY_DIM, _FN_WITH_PARAM, _N_FN, _Frame, _Return, fn_eval)  # noqa: F401


class Peepholes:
    # -- peepholes: same results bit for bit, fewer passes over HBM ---------------------------------------------------------
    def _match_smoother(self, st, fr: _Frame):
        """(D, dslot, U, uslot, F, fslot, A, w) if `st` is a damped-residual update  D = U + w * (F - A * U)."""
        if st[0] != "assign" or st[2][0] != "fld":
            return None
        op, lhs, rhs = st[1], st[2], st[3]
        src = wexpr = r = None
        if op == "+=" and rhs[0] == "bin" and rhs[1] == "*":
            src, wexpr, r = lhs, rhs[2], self._residual_form(rhs[3], fr)
        elif op == "=" and rhs[0] == "bin" and rhs[1] == "+" and rhs[2][0] == "fld" and rhs[3][0] == "bin" and rhs[3][1] == "*":
            src, wexpr, r = rhs[2], rhs[3][2], self._residual_form(rhs[3][3], fr)
        if r is None or not self._same_access(src, r[2], fr):
            return None
        D, ds = self._field(lhs, fr)
        U, us = self._field(src, fr)
        F, fs = self._field(r[0], fr)
        wv, A = self._smoother_weight(wexpr, r[1], fr)
        if D.layout.transform or U.layout.transform or F.layout.transform:
            return None         # fields under a layout transformation: the plain loops (the kernel layer's one-pass forms take plain layouts)
        return D, ds, U, us, F, fs, A, wv

    @staticmethod
    def _canonical7(A: Stencil, nd: int) -> bool:
        return nd == 3 and A.cfield is None and len(A.offsets) == 7 and all(sum(1 for c in o if c) <= 1 for o in A.offsets)

    def _try_fused_sweep(self, body, first: int, fr: _Frame, only_field=None, zero_input: bool = False, correction_from=None) -> bool:
        """`color with { (i0+i1+i2) % 2, [communicate u] loop over u { u += w (f - A u) } [apply bc to u] }` on one block:
        both half sweeps in one pass (examg_rbgs_sweep_fused), out of place into a second array that carries the same
        boundary shell, then the two arrays change roles.  `apply bc` re-writes position-only Dirichlet values the sweep
        never touches, so it is a no-op here."""
        multi = self.domain.world_size != 1
        if multi and not any(st[0] == "comm" and st[2] in ("all", "ghost") for st in body):
            return False        # blocks with neighbours: the fused form contains the exchanges of the statement list
        loops = [st for st in body if st[0] == "loop"]
        if len(loops) != 1 or any(st[0] not in ("loop", "comm", "applybc") for st in body):
            return False
        lp = loops[0]
        if lp[2] is not None or lp[3] is not None or lp[4] is not None or len(lp[5]) != 1:
            return False
        m = self._match_smoother(lp[5][0], fr)
        if m is None:
            return False
        D, ds, U, us, F, fs, A, w = m
        if D is not U or ds != us or not self._canonical7(A, self.nd) or U.layout.inner[0] < (self.fuse_min_row_blocks if multi else self.fuse_min_row):
            return False
        if only_field is not None and U is not only_field:
            return False        # a pending `u = 0` / `u += P * uc` rides along with the sweep of the same field only
        if (zero_input or correction_from is not None) and multi:
            return False
        if multi and U.num_slots != 1:
            return False
        for st in body:
            if st[0] in ("comm", "applybc") and self._field(st[-1], fr)[0] is not U:
                return False
        if U.bc_fn is not None and (U.name, U.level, us) not in self._bc_valid:
            return False        # boundary planes not known to hold the Dirichlet values yet: the plain path applies them
        b, e = self.domain.loop_bounds(self._field(lp[1], fr)[0].layout)
        key = (U.name, U.level, us)
        alt = self._alt.get(key)
        if alt is None:
            alt = self._alt[key] = self.ops.new_array(U.layout.size)
        if self._alt_shell.get(key) != self._bc_epoch.get((U.name, U.level), 0):
            lay = U.layout      # the shell (everything outside the loop's box) comes from the field itself
            gb = [lay.idx("GLB", d) if d < self.nd else 0 for d in range(3)]
            ge = [lay.idx("GRE", d) if d < self.nd else 1 for d in range(3)]
            self.ops.axpby(U.lc, U.data(us), U.lc, alt, 1.0, 0.0, gb, ge)
            self._alt_shell[key] = self._bc_epoch.get((U.name, U.level), 0)
            self.launches += 1
        self.launches += 1
        if multi:
            # fused deep interior + two-point shell with its exchanges on a side stream (exastencils_amd/smoothers.py): rbgs_sweep
            # exchanges ghost layers only -- a `communicate u` (duplicate + ghost) in the body keeps its duplicate part here
            from .smoothers import rbgs_sweep

            if any(st[0] == "comm" and st[2] == "all" for st in body):
                self.comm.exchange(U, us, "dup")

            tmp = self._pair_tmp.get((U.name, U.level))
            if tmp is None:
                tmp = self._pair_tmp[(U.name, U.level)] = Field(U.name + "Tmp", U.level, U.layout, self.ops, 1, None)
            self._alt[key] = rbgs_sweep(self.ops, self.comm, self.domain, U, F, A, w, alt, tmp, first)
            return True
        if zero_input:
            # `u = 0.0` just before: the sweep takes the zero field as a constant, the zeroing loop never runs (examg_rbgs_sweep_fused_zero)
            self.ops.rbgs_sweep_fused_zero(U.lc, alt, F.lc, F.data(fs), A, w, first, b, e)
        elif correction_from is not None:
            # `u += P@coarser * u@coarser` just before: interpolated while u is loaded (examg_rbgs_sweep_fused_prolong)
            X, xs = correction_from
            self.ops.rbgs_sweep_fused_prolong(U.lc, U.data(us), alt, F.lc, F.data(fs), A, w, first, b, e, X.lc, X.data(xs))
        else:
            self.ops.rbgs_sweep_fused(U.lc, U.data(us), alt, F.lc, F.data(fs), A, w, first, b, e)
        self._alt[key], U.slots[us] = U.slots[us], alt
        return True

    def _try_jacobi_pairs(self, body, n: int, fr: _Frame) -> bool:
        """`repeat n times { Smoother ( ) }` with Smoother = [communicate ghost of u<active>; loop over u { u<next> =
        u<active> + w (f - A u<active>) }; advance u]: consecutive pairs as one pass over HBM (exastencils_amd/smoothers.py)."""
        if len(body) != 1 or body[0][0] != "callstmt":
            return False
        c = body[0][1]
        if c[1] not in self.functions or c[3]:
            return False
        lvl = self._level_of(c[2], fr) if c[2] is not None else fr.level
        fn = self._resolve(c[1], lvl)
        fb = fn.body
        if len(fb) != 3 or fb[0][0] != "comm" or fb[1][0] != "loop" or fb[2][0] != "advance":
            return False
        if fb[0][2] != "ghost":
            return False        # jacobi_pair exchanges ghost layers only: `communicate u` / `communicate dup of u` keep the plain path
        cfr = _Frame(lvl if fn.levels is not None else None, {})
        lp = fb[1]
        if lp[2] is not None or lp[3] is not None or lp[4] is not None or len(lp[5]) != 1:
            return False
        m = self._match_smoother(lp[5][0], cfr)
        if m is None:
            return False
        D, ds, U, us, F, fs, A, w = m
        if D is not U or U.num_slots != 2 or us != U.active or ds != U.next or A.cfield is not None:
            return False
        if self._field(fb[0][3], cfr) != (U, us) or self._field(fb[2][1], cfr)[0] is not U or self._field(lp[1], cfr)[0] is not U:
            return False
        # the pair reads the boundary planes of <active> in both steps; the two plain steps read those of <next> in the
        # second: only equal when both slots are known to hold the same boundary values
        if U.bc_fn is not None:
            if not all((U.name, U.level, sl) in self._bc_valid for sl in range(2)):
                return False
        elif self._bc_epoch.get((U.name, U.level), 0) != 0:
            return False
        from .smoothers import jacobi_pair, jacobi_triple

        key = (U.name, U.level)
        tmp = self._pair_tmp.get(key)
        if tmp is None:
            tmp = self._pair_tmp[key] = Field(U.name + "Tmp", U.level, U.layout, self.ops, 1, None)
        k = n
        while k >= 3 and k != 4 and jacobi_triple(self.ops, self.comm, self.domain, U, F, A, w, tmp):    # three steps per pass on a lone block
            self.launches += 1
            k -= 3
        while k >= 2:
            self.launches += 1
            jacobi_pair(self.ops, self.comm, self.domain, U, F, A, w, tmp)
            k -= 2
        if k:
            self._exec_block(body, fr)
        return True

    # -- `repeat n times with contraction [..] { loop ..; advance .. }` (temporal blocking with deep ghost layers) ------------
    def _exec_contract(self, s, fr: _Frame):
        """IR_ContractingLoop.expandSpecial (baseExt/ir/IR_ContractingLoop.scala:130-196): the loop is unrolled; the k-th
        `loop over` of the unrolled sequence runs on bounds widened by (total - 1 - k) x the contraction at interior faces, so
        that no exchange is needed inside.  Slotted Jacobi bodies run as two-step passes (examg_jacobi2_boxes: first step on
        the box widened by e, second on the box widened by e - 1) -- the reference's own use of the construct
        (Testing/PolyExpl/Jac3Dcc.exa4:27: 5 ghost layers, 5 steps)."""
        _, nexpr, counter, pos, neg, body = s
        n = int(self._eval(nexpr, fr))
        if any(st[0] not in ("loop", "advance") for st in body):
            raise Exa4Unsupported("repeat ... with contraction: body may hold `loop over` and `advance` statements only")
        nloops = sum(1 for st in body if st[0] == "loop")
        expand = n * nloops - 1
        it = 0
        if self.fuse and counter is None and nloops == 1 and len(body) == 2 and body[0][0] == "loop" and body[1][0] == "advance":
            m = self._contract_pair_plan(body, fr)
            while m is not None and n - it >= 2:
                U, F, A, w, tmp = m
                lb, le = self.domain.loop_bounds(U.layout)
                b1, e1 = self._contract_bounds(U.layout, lb, le, expand, pos, neg)
                b2, e2 = self._contract_bounds(U.layout, lb, le, expand - 1, pos, neg)
                if n - it >= 3 and n - it != 4 and hasattr(self.ops, "jacobi3"):
                    # three steps in one pass where they run on the same box (a block without neighbours: nothing is widened) --
                    # Testing/PolyExpl/Jac3Dcc.exa4:27's five steps are then a pass of three and a pass of two
                    b3, e3 = self._contract_bounds(U.layout, lb, le, expand - 2, pos, neg)
                    if list(b1) == list(b2) == list(b3) and list(e1) == list(e2) == list(e3) and \
                            (not hasattr(self.ops, "three_stage_eligible") or self.ops.three_stage_eligible(U.lc, F.lc, A, list(b1), list(e1))):
                        self.launches += 1
                        self.ops.jacobi3(U.lc, U.data(U.active), U.data(U.next), tmp.data(), F.lc, F.data(), A, w, b1, e1)
                        U.advance()
                        expand -= 3
                        it += 3
                        continue
                self.launches += 1
                self.ops.jacobi2_boxes(U.lc, U.data(U.active), U.data(U.next), tmp.data(), F.lc, F.data(), A, w, b1, e1, b2, e2)
                U.advance()
                expand -= 2
                it += 2
        saved = fr.contract
        try:
            for k in range(it, n):
                if counter:
                    fr.vars[counter] = k
                for st in body:
                    if st[0] == "loop":
                        fr.contract = (expand, pos, neg)
                        self._exec_loop(st, fr)
                        expand -= 1
                    else:
                        self._exec(st, fr)
        finally:
            fr.contract = saved
        if counter:
            fr.vars[counter] = n

    def _contract_pair_plan(self, body, fr: _Frame):
        """(U, F, A, w, scratch) if body is `loop over U { U<next> = U<active> + w (F - A U<active>) }; advance U` on a two-slot
        field with constant coefficients whose two slots hold the same boundary values (same condition as _try_jacobi_pairs)."""
        lp = body[0]
        if lp[2] is not None or lp[3] is not None or lp[4] is not None or len(lp[5]) != 1:
            return None
        m = self._match_smoother(lp[5][0], fr)
        if m is None:
            return None
        D, ds, U, us, F, fs, A, w = m
        if D is not U or U.num_slots != 2 or us != U.active or ds != U.next or A.cfield is not None:
            return None
        if self._field(body[1][1], fr)[0] is not U or self._field(lp[1], fr)[0] is not U:
            return None
        # both slots must carry the same values on the physical boundary planes (the pass reads <active>'s in both steps):
        # either `apply bc` put the field's Dirichlet values into both, or nothing has written them since the zero fill
        valid = [(U.name, U.level, sl) in self._bc_valid for sl in range(2)]
        untouched = self._bc_epoch.get((U.name, U.level), 0) == 0 and not any(valid)
        if not (untouched or (U.bc_fn is not None and all(valid))):
            return None
        key = (U.name, U.level)
        tmp = self._pair_tmp.get(key)
        if tmp is None:
            tmp = self._pair_tmp[key] = Field(U.name + "Tmp", U.level, U.layout, self.ops, 1, None)
        return U, F, A, w, tmp

    # -- coarse-grid CG as one kernel ---------------------------------------------------------------------------------------
    def _inline(self, body, lvl: int, depth: int = 0):
        """Statement list with calls to parameterless, value-less functions of the same level replaced by their bodies."""
        out = []
        for st in body:
            if st[0] == "callstmt" and st[1][1] in self.functions and not st[1][3] and depth < 4:
                c = st[1]
                clvl = self._level_of(c[2], _Frame(lvl, {})) if c[2] is not None else lvl
                fn = self._resolve(c[1], clvl)
                if clvl != lvl or fn.params or any(x[0] == "return" for x in fn.body):
                    return None
                sub = self._inline(fn.body, lvl, depth + 1)
                if sub is None:
                    return None
                out += sub
            else:
                out.append(st)
        return out

    def _norm_of(self, e, fr: _Frame):
        """Field R if `e` is a call of a function  { Var s = 0; loop over R with reduction(+ : s) { s += R * R }; return sqrt(s) }."""
        if e[0] != "call" or e[1] not in self.functions or e[3]:
            return None
        lvl = self._level_of(e[2], fr) if e[2] is not None else fr.level
        b = self._resolve(e[1], lvl).body
        if len(b) != 3 or b[0][0] != "decl" or b[1][0] != "loop" or b[2][0] != "return":
            return None
        var, lp = b[0][1], b[1]
        if lp[2] is not None or lp[4] != ("+", var) or len(lp[5]) != 1 or b[2][1] != ("call", "sqrt", None, [("id", var, None)]):
            return None
        if lp[3] is not None and any(_lower_cond(c) is None for c in _conjuncts(lp[3])):
            return None
        st = lp[5][0]
        if st[0] != "assign" or st[1] != "+=" or st[2] != ("id", var, None):
            return None
        r = st[3]
        cfr = _Frame(lvl, {})
        if r[0] == "bin" and r[1] == "*" and self._same_access(r[2], r[3], cfr) and self._same_access(r[2], lp[1], cfr):
            return self._field(r[2], cfr)[0]
        return None

    def _coarse_cg_plan(self, fn: FunctionDecl, lvl: int):
        key = (fn.name, lvl)
        if key not in self._cg_plans:
            try:
                self._cg_plans[key] = self._match_coarse_cg(fn, lvl)
            except (Exa4SyntaxError, Exa4Unsupported, IndexError, KeyError, TypeError):
                self._cg_plans[key] = None
        return self._cg_plans[key]

    def _match_coarse_cg(self, fn: FunctionDecl, lvl: int):
        """The conjugate-gradient solver the reference's generator emits for `mgCycle@coarsest`
        (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:152-201), recognised statement by statement -- and the form its layer-3
        solver generator writes (Function VCycle_0@coarsest, Testing/Smoothers/Jac.exa4:75-109): alpha = res * res / alphaDenom
        with the norm carried over instead of a sum of squares, both vector updates in one loop, no `apply bc` statements, the
        solution possibly slotted (accessed through <active> only)."""
        if self.domain.world_size != 1 or not hasattr(self.ops, "cg_coarse"):
            return None
        if self.domain.face_mask() != (1 << (2 * self.nd)) - 1:
            return None         # a periodic block is its own neighbour: the solver's `communicate` statements do something
        body = self._inline(fn.body, lvl)
        if body is None:
            return None
        fr = _Frame(lvl, {})
        pos = [0]

        def peek():
            return body[pos[0]] if pos[0] < len(body) else ("end",)

        def take():
            pos[0] += 1
            return body[pos[0] - 1]

        def loop1(st):        # plain loop with one assignment
            if st[0] == "loop" and st[2] is None and st[3] is None and st[4] is None and len(st[5]) == 1 and st[5][0][0] == "assign":
                return st[5][0]
            return None

        def fld(e):
            return self._field(e, fr)[0]

        def active_only(e):       # a slotted field may take part if the solver touches its active slot only
            return e[0] == "fld" and e[2] in (None, "active", "activeSlot", "current", "currentSlot")

        def opt_comm(F):
            if peek()[0] == "comm" and fld(peek()[3]) is F:
                take()

        def opt_bc(F):
            if peek()[0] == "applybc" and fld(peek()[1]) is F:
                take()
                return True
            return False

        # communicate u; r = f - A u; apply bc to r
        if peek()[0] == "comm":
            take()
        a = loop1(take())
        rf = self._residual_form(a[3], fr) if a and a[1] == "=" else None
        if rf is None:
            return None
        R, F, A, U = fld(a[2]), fld(rf[0]), rf[1], fld(rf[2])
        if not active_only(rf[2]):
            return None
        bc_r = opt_bc(R)
        opt_comm(R)
        # Var rr = Norm(); Var rr0 = rr
        d1, d2 = take(), take()
        if d1[0] != "decl" or self._norm_of(d1[2], fr) is not R or d2[0] != "decl" or d2[2] != ("id", d1[1], None):
            return None
        rr, rr0 = d1[1], d2[1]
        # p = r; apply bc to p
        a = loop1(take())
        if not a or a[1] != "=" or a[3][0] != "fld" or fld(a[3]) is not R:
            return None
        P = fld(a[2])
        bc_p = opt_bc(P)
        if peek()[0] == "decl" and pos[0] + 1 < len(body) and body[pos[0] + 1][0] == "repeat" and body[pos[0] + 1][2] == peek()[1]:
            take()                                          # Var curStep : Integer = 0 -- the repeat's counter
        rep = take()
        if rep[0] != "repeat" or pos[0] < len(body) and not all(x[0] == "callstmt" and x[1][1] == "print" for x in body[pos[0]:]):
            return None
        max_it = int(self._eval(rep[1], fr))
        tail = body[pos[0]:]                                # print statements after the loop: reached when it runs out of iterations
        body, pos[0] = self._inline(rep[3], lvl), 0
        if body is None:
            return None
        opt_comm(P)
        a = loop1(take())                                   # q = A p
        m = self._sten_times_field(a[3], fr) if a and a[1] == "=" else None
        if m is None or m[1] != "stencil" or m[0] != 1.0 or m[2] is not A or fld(m[4]) is not P:
            return None
        Q = fld(a[2])

        def reduction(x, y):                                # Var v = 0; loop ... reduction(+ : v) { v += x * y }; [Var w = v]
            d = take()
            lp = take()
            if d[0] != "decl" or lp[0] != "loop" or lp[2] is not None or lp[4] != ("+", d[1]) or len(lp[5]) != 1:
                return None
            if lp[3] is not None and any(_lower_cond(c) is None for c in _conjuncts(lp[3])):
                return None
            st = lp[5][0]
            if st[0] != "assign" or st[1] != "+=" or st[2] != ("id", d[1], None) or st[3][0] != "bin" or st[3][1] != "*":
                return None
            if {id(fld(st[3][2])), id(fld(st[3][3]))} != {id(x), id(y)}:
                return None
            name = d[1]
            if peek()[0] == "decl" and peek()[2] == ("id", name, None):
                name = take()[1]
            return name

        sq = lambda v: ("bin", "*", ("id", v, None), ("id", v, None))
        mark = pos[0]
        num = reduction(R, R)
        from_norm = num is None                             # no sum of squares: alpha's numerator is the squared norm
        if from_norm:
            pos[0] = mark
        den = reduction(P, Q)
        d = take()
        if not den or d[0] != "decl" or d[2] != ("bin", "/", sq(rr) if from_norm else ("id", num, None), ("id", den, None)):
            return None
        alpha = d[1]
        st = take()
        if st[0] == "loop" and st[2] is None and st[3] is None and st[4] is None and len(st[5]) == 2 and all(x[0] == "assign" for x in st[5]):
            a, a2 = st[5]                                   # both updates in one loop
            bc_u = False
        else:
            a = loop1(st)
            bc_u = None
        # u += alpha p
        if (not a or a[1] != "+=" or not active_only(a[2]) or fld(a[2]) is not U or a[3] != ("bin", "*", ("id", alpha, None), a[3][3])
                or fld(a[3][3]) is not P):
            return None
        if bc_u is None:
            bc_u = opt_bc(U)
            a2 = loop1(take())
        a = a2                                              # r -= alpha q
        if not a or a[1] != "-=" or fld(a[2]) is not R or a[3] != ("bin", "*", ("id", alpha, None), a[3][3]) or fld(a[3][3]) is not Q:
            return None
        if opt_bc(R) != bc_r:
            return None
        d = take()                                          # Var rrNew = Norm()
        if d[0] != "decl" or self._norm_of(d[2], fr) is not R:
            return None
        new = d[1]
        c = take()                                          # if ( rrNew <= tol * rr0 ) { return }
        if (c[0] != "if" or c[3] or len(c[2]) != 1 or c[2][0] != ("return", None) or c[1][0] != "bin" or c[1][1] != "<="
                or c[1][2] != ("id", new, None) or c[1][3][0] != "bin" or c[1][3][1] != "*" or c[1][3][3] != ("id", rr0, None)):
            return None
        tol = float(self._eval(c[1][3][2], fr))
        d = take()                                          # Var beta = (rrNew * rrNew) / (rr * rr)
        if d[0] != "decl" or d[2] != ("bin", "/", sq(new), sq(rr)):
            return None
        beta = d[1]
        a = loop1(take())                                   # p = r + beta p
        if (not a or a[1] != "=" or fld(a[2]) is not P or a[3][0] != "bin" or a[3][1] != "+" or a[3][2][0] != "fld" or fld(a[3][2]) is not R
                or a[3][3] != ("bin", "*", ("id", beta, None), a[3][3][3]) or fld(a[3][3][3]) is not P):
            return None
        if opt_bc(P) != bc_p:
            return None
        if take() != ("assign", "=", ("id", rr, None), ("id", new, None)) or pos[0] != len(body):
            return None
        # the kernel applies homogeneous Dirichlet values to r, p and u on every face, or leaves every boundary plane alone: the
        # program must do one or the other
        from .lib import CG_ALPHA_FROM_NORM, CG_NO_BC

        flags = CG_ALPHA_FROM_NORM if from_norm else 0
        if not (bc_r or bc_p or bc_u):
            flags |= CG_NO_BC
        else:
            for fld_, has in ((R, bc_r), (P, bc_p), (U, bc_u)):
                if not has or fld_.bc_fn != 0:
                    return None
        if any(x.num_slots != 1 for x in (F, R, P, Q)) or (U.num_slots != 1 and not flags & CG_NO_BC):
            return None
        if any(x.layout.transform for x in (U, F, R, P, Q)):
            return None         # the one-kernel solver takes plain layouts
        return U, F, R, P, Q, A, max_it, tol, tail, flags

    def _run_coarse_cg(self, plan):
        U, F, R, P, Q, A, max_it, tol, tail, flags = plan
        b, e = self.domain.loop_bounds(U.layout)
        if not hasattr(self, "_cg_info"):
            self._cg_info = self.ops.new_array(4)
        self._cg_tail = (tail, U.level)
        self.launches += 1
        self.ops.cg_coarse(U.lc, U.data(), F.lc, F.data(), R.lc, R.data(), P.lc, P.data(), Q.lc, Q.data(), A,
                           self.domain.geom(U.level), self.domain.face_mask(), max_it, tol, b, e, self._cg_info, flags=flags)
        return None
