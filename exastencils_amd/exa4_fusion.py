"""Cross-statement fusions of the ExaSlang-4 interpreter (exastencils_amd/exa4.py): the one-pass forms of the hand-written drivers
(exastencils_amd/solver.py) reached from a PROGRAM, whatever functions its statements are spread over.

A loop that one of the one-pass kernels can absorb is not launched when the interpreter meets it but kept PENDING:

  Residual = RHS - A * Solution            pending until  RHS@coarser = Restriction * Residual   -> examg_residual_restrict
                                                     or   s += Residual * Residual (reduction)   -> examg_residual_norm2  (opt-in:
                                                          `fuse_residual_norm`; the sum's order, hence its last bits, change)
  Solution@coarser = 0.0                   pending until  the red-black sweep that follows        -> examg_rbgs_sweep_fused_zero
  Solution += Prolongation * Solution@coarser  pending until  the red-black sweep that follows   -> examg_rbgs_sweep_fused_prolong

Statements in between pass through if they cannot observe the difference (`apply bc` of the pending field, scalar statements,
function calls -- their bodies are gated statement by statement -- and `communicate` on a block without neighbours); anything else
makes the pending loop run first.  The first two forms never store the residual, so they also need a proof that nothing reads it:
a LIVENESS scan over everything that can still execute (`_dead_after`) -- the rest of every active statement list up the call
stack, loop bodies once more for their next iteration, callees through memoised summaries -- which must find no read of the field
before a loop overwrites all of it.  Both are decided from the program text alone; results are bit-identical to `fuse=False`
(tests/test_exa4.py), the pass count is the hand-written driver's.

Reference: what the generator would reach with loop fusion across inlined functions (Compiler/src/exastencils/optimization/ir/
IR_GeneralSimplify, polyhedron/ -- schedule-level fusion); there is no counterpart of the never-stored residual in the reference."""
from __future__ import annotations

from typing import Optional, Tuple

READ, WRITTEN, NEITHER = "read", "written", "neither"


class LazyFusions:
    """Mixin of Exa4Program: continuation stack, pending loops, liveness."""

    def _lazy_init(self):
        self._pending: Optional[dict] = None
        self._cont: list = []                 # active statement lists, innermost last: [body, index, frame, is_loop, is_function]
        self._summary: dict = {}
        self.fusions = {"residual_restrict": 0, "residual_norm": 0, "zero_start": 0, "folded_correction": 0}
        self.fused_prolong_min_points = 10_000_000     # the fold pays from ~10^7 points: 256^3 0.110 ms against 0.090 + 0.062 for sweep and correction, 128^3 0.028 against 0.023
        # residual + norm sums the squares in the residual kernel's own (fixed) order, not in the order of the dot kernel: the
        # printed norm then differs from `fuse=False` in the last bits (fields stay bit-identical).  Opt-in for that reason.
        self.fuse_residual_norm = False

    # -- who may defer ---------------------------------------------------------------------------------------------------
    def _lazy_enabled(self) -> bool:
        return bool(self.fuse and self.domain.world_size == 1 and not any(self.domain.periodic))

    def _try_defer(self, st, target, box, fr) -> bool:
        """Keep the single-statement loop `st` over the whole box of `target` pending if it is one of the three absorbable kinds."""
        if not self._lazy_enabled() or self._pending is not None or st[0] != "assign" or st[2][0] != "fld":
            return False
        op, lhs, rhs = st[1], st[2], st[3]
        D, ds = self._field(lhs, fr)
        if D is not target or D.num_slots != 1 or D.layout.transform:
            return False
        b, e = box
        fb, fe = self.domain.loop_bounds(D.layout)
        if list(b) != list(fb) or list(e) != list(fe):
            return False
        if op == "=":
            if self._is_scalar(rhs):
                if float(self._eval(rhs, fr)) != 0.0 or D.bc_fn != 0 or (D.name, D.level, ds) not in self._bc_valid or \
                        not hasattr(self.ops, "rbgs_sweep_fused_zero") or D.layout.inner[0] < self.fuse_min_row or self.nd != 3:
                    return False
                self._pending = dict(kind="zero", D=D, ds=ds, b=b, e=e)
                return True
            r = self._residual_form(rhs, fr)
            if r is None or not hasattr(self.ops, "residual_restrict") or self.nd != 3:
                return False
            F, fs = self._field(r[0], fr)
            U, us = self._field(r[2], fr)
            A = r[1]
            if U.layout.transform or F.layout.transform:
                return False
            if U is D or F is D or not self._canonical7(A, self.nd) or D.layout.inner[0] < self.fuse_min_row:
                return False
            self._pending = dict(kind="residual", D=D, ds=ds, U=U, us=us, F=F, fs=fs, A=A, b=b, e=e)
            return True
        if op == "+=" and rhs[0] == "bin" and rhs[1] == "*":
            m = self._sten_times_field(rhs, fr)
            if m is None or m[1] != "prolongation" or m[0] != 1.0 or not hasattr(self.ops, "rbgs_sweep_fused_prolong") or self.nd != 3:
                return False
            X, xs = self._field(m[4], fr)
            if X.layout.transform:
                return False
            pts = 1
            for d in range(3):
                pts *= max(1, e[d] - b[d])
            # the fold pays on large levels (one read-modify-write pass less) and on launch-bound ones (rows shorter than 64 points: one
            # kernel less); in between the separate correction is faster
            if X.level != D.level - 1 or (pts < self.fused_prolong_min_points and D.layout.inner[0] >= 64) or \
                    (D.bc_fn is not None and (D.name, D.level, ds) not in self._bc_valid):
                return False
            self._pending = dict(kind="prolong", D=D, ds=ds, X=X, xs=xs, b=b, e=e)
            return True
        return False

    def _flush_pending(self):
        """Run the pending loop as the statement it is."""
        P, self._pending = self._pending, None
        if P is None:
            return
        ops, D = self.ops, P["D"]
        self.launches += 1
        if P["kind"] == "residual":
            U, F = P["U"], P["F"]
            ops.stencil_op(1, U.lc, U.data(P["us"]), F.lc, F.data(P["fs"]), D.lc, D.data(P["ds"]), P["A"], 0.0, -1, P["b"], P["e"])
        elif P["kind"] == "zero":
            ops.set(D.lc, D.data(P["ds"]), 0.0, P["b"], P["e"])
        else:
            X = P["X"]
            ops.prolong_add(X.lc, X.data(P["xs"]), D.lc, D.data(P["ds"]), P["b"], P["e"])

    # -- the gate: every statement passes here while a loop is pending ------------------------------------------------------
    def _gate(self, s, fr) -> bool:
        """True if `s` was absorbed together with the pending loop (nothing left to execute)."""
        k = s[0]
        if k in ("decl", "assign", "callstmt", "if", "repeat", "until", "levelscope", "return"):
            # no field access of their own (what they contain is gated when it executes) -- unless one of their expressions names a
            # field: outside a loop that is an argument of a host-side builtin (printField / writeField / readField, norms of a
            # field), which reads or writes the array NOW.  The pending loop runs first.
            e = s[2] if k == "decl" else (s[3] if k == "assign" else (None if k == "levelscope" else s[1]))
            if e is not None and self._has_field_ref(e):
                self._flush_pending()
            return False
        P = self._pending
        if k == "comm":
            return False                      # a block without neighbours: the generated exch function is empty
        if k == "applybc":
            f, _ = self._field(s[1], fr)
            if f is P["D"]:
                return False                  # boundary planes only: commutes with the pending loop over inner points
        elif k == "loop" and P["kind"] == "residual":
            if self._consume_residual(s, fr):
                return True
        elif k == "color" and P["kind"] in ("zero", "prolong"):
            from .exa4 import _parity_expr

            shift = _parity_expr(s[1][0], self.nd) if len(s[1]) == 1 else None
            if shift is not None:
                kw = dict(zero_input=True) if P["kind"] == "zero" else dict(correction_from=(P["X"], P["xs"]))
                self._pending = None          # the sweep either absorbs it ...
                if self._try_fused_sweep(s[2], (0 - shift) % 2, fr, only_field=P["D"], **kw):
                    self.fusions["zero_start" if P["kind"] == "zero" else "folded_correction"] += 1
                    return True
                self._pending = P             # ... or it runs first
        self._flush_pending()
        return False

    @classmethod
    def _has_field_ref(cls, e) -> bool:
        """Does the expression tree name a field anywhere (also inside the arguments of a call)?"""
        if isinstance(e, (list, tuple)):
            if len(e) >= 4 and e[0] == "fld" and isinstance(e[1], str):
                return True
            return any(cls._has_field_ref(x) for x in e)
        return False

    def _consume_residual(self, s, fr) -> bool:
        P = self._pending
        _, target, only, where, reduction, body = s
        if only is not None or where is not None or fr.colour is not None or fr.contract is not None or len(body) != 1:
            return False
        st = body[0]
        D, U, F, A = P["D"], P["U"], P["F"], P["A"]
        if reduction is None:
            # RHS@coarser = [scale *] Restriction * Residual
            if st[0] != "assign" or st[1] != "=" or st[2][0] != "fld":
                return False
            m = self._sten_times_field(st[3], fr)
            if m is None or m[1] != "restriction":
                return False
            X, xs = self._field(m[4], fr)
            C, cs = self._field(st[2], fr)
            if X is not D or xs != P["ds"] or C.level != D.level - 1 or self._field(target, fr)[0] is not C:
                return False
            if not self._dead_after((D.name, D.level)):
                return False
            cb, ce = self.domain.loop_bounds(C.layout)
            self._pending = None
            self.launches += 1
            self.ops.residual_restrict(U.lc, U.data(P["us"]), F.lc, F.data(P["fs"]), D.lc, D.data(P["ds"]), A, C.lc, C.data(cs), m[0],
                                       P["b"], P["e"], cb, ce)
            self.fusions["residual_restrict"] += 1
            return True
        # Var s = 0; loop over Residual with reduction ( + : s ) { s += Residual * Residual }
        op, var = reduction
        if not self.fuse_residual_norm or op != "+" or st[0] != "assign" or st[1] != "+=" or st[2] != ("id", var, None) or \
                not hasattr(self.ops, "residual_norm2"):
            return False
        rhs = st[3]
        if not (rhs[0] == "bin" and rhs[1] == "*" and rhs[2][0] == "fld" and rhs[3][0] == "fld"):
            return False
        X, xs = self._field(rhs[2], fr)
        Y, ys = self._field(rhs[3], fr)
        f, _ = self._field(target, fr)
        if X is not D or Y is not D or xs != P["ds"] or ys != P["ds"] or f is not D:
            return False
        boxes, colour = self._loop_boxes(f, only, where, reduction, fr)
        if colour is not None or len(boxes) != 1 or not self._dead_after((D.name, D.level)):
            return False
        b, e = boxes[0]
        self._pending = None
        self.launches += 1
        t = self.ops.residual_norm2(U.lc, U.data(P["us"]), F.lc, F.data(P["fs"]), A, b, e, D.lc, D.data(P["ds"]))
        fr.vars[var] = fr.vars[var] + self.comm.reduce_value(t, "sum")
        self.fusions["residual_norm"] += 1
        return True

    # -- liveness: can anything that may still execute read `key` before it is overwritten? -------------------------------------
    def _dead_after(self, key: Tuple[str, int]) -> bool:
        may_return = False
        for body, idx, fr, is_loop, is_fn in reversed(self._cont):
            st, ret = self._scan(body[idx + 1:], fr.level, key, fr.colour is not None)
            if st == READ:
                return False
            if st == WRITTEN and not (may_return or ret):
                return True
            may_return = may_return or ret
            if is_loop:                       # the body may run again from its start (and the loop may also end here)
                st2, ret2 = self._scan(body[:idx + 1], fr.level, key, fr.colour is not None)
                if st2 == READ:
                    return False
                may_return = may_return or ret2
            if is_fn:
                may_return = False            # a `return` ends at this list: the caller's statements run either way
        return True                           # the program ends without another read

    def _scan(self, stmts, lvl, key, in_colour: bool):
        """(READ | WRITTEN | NEITHER, a `return` may leave the list early) for a statement list executed from its start."""
        ret = False
        for s in stmts:
            st, r = self._scan_stmt(s, lvl, key, in_colour)
            if st == READ:
                return READ, ret or r
            ret = ret or r
            if st == WRITTEN:
                return (NEITHER if ret else WRITTEN), ret
            if s[0] == "return":
                return NEITHER, True
        return NEITHER, ret

    def _matches(self, node, lvl, key) -> bool:
        if node[1] != key[0]:
            return False
        try:
            from .exa4 import _Frame

            return self._level_of(node[3], _Frame(lvl, {})) == key[1]
        except Exception:                     # noqa: BLE001 -- cannot tell: assume it is the field
            return True

    def _expr_reads(self, e, lvl, key) -> bool:
        """Does evaluating `e` read the field: a reference to it, or a call of a function that reads it first?"""
        if isinstance(e, (list, tuple)):
            if len(e) >= 4 and e[0] == "fld" and isinstance(e[1], str):
                return self._matches(e, lvl, key)
            if len(e) >= 4 and e[0] == "call" and isinstance(e[1], str):
                if any(self._expr_reads(a, lvl, key) for a in e[3]):
                    return True
                if e[1] in self.functions:
                    return self._fn_summary(e[1], e[2], lvl, key) == READ
                return False
            return any(self._expr_reads(x, lvl, key) for x in e)
        return False

    def _expr_writes(self, e, lvl, key) -> bool:
        """A call in `e` that definitely overwrites the field before reading it."""
        if isinstance(e, (list, tuple)):
            if len(e) >= 4 and e[0] == "call" and isinstance(e[1], str) and e[1] in self.functions:
                return self._fn_summary(e[1], e[2], lvl, key) == WRITTEN
            if len(e) >= 1 and e[0] in ("fld", "str", "num", "id"):
                return False
            return any(self._expr_writes(x, lvl, key) for x in e if isinstance(x, (list, tuple)))
        return False

    def _fn_summary(self, name, lspec, lvl, key):
        from .exa4 import _Frame

        try:
            flvl = self._level_of(lspec, _Frame(lvl, {})) if lspec is not None else lvl
            fn = self._resolve(name, flvl)
        except Exception:                     # noqa: BLE001
            return READ
        flvl = flvl if fn.levels is not None else None
        ck = (name, flvl, key)
        if ck in self._summary:
            return self._summary[ck]
        self._summary[ck] = READ              # recursion at the same level: assume the worst
        st, _ = self._scan(fn.body, flvl, key, False)
        self._summary[ck] = st
        return st

    def _scan_stmt(self, s, lvl, key, in_colour: bool):
        k = s[0]
        if k in ("decl", "assign", "callstmt", "return"):
            exprs = [s[2]] if k == "decl" else ([s[3]] if k == "assign" else [s[1]])
            for e in exprs:
                if e is None:
                    continue
                if self._expr_reads(e, lvl, key):
                    return READ, False
                if self._expr_writes(e, lvl, key):
                    return WRITTEN, False
            return NEITHER, k == "return"
        if k == "loop":
            _, target, only, where, reduction, body = s
            full = only is None and where is None and reduction is None and not in_colour
            if where is not None and self._expr_reads(where, lvl, key):
                return READ, False
            res = NEITHER
            for st in body:
                if st[0] == "decl":
                    if st[2] is not None and self._expr_reads(st[2], lvl, key):
                        return READ, False
                    continue
                if st[0] != "assign":
                    return READ, False
                if self._expr_reads(st[3], lvl, key):
                    return READ, False
                lhs = st[2]
                if isinstance(lhs, tuple) and lhs[0] == "fld" and self._matches(lhs, lvl, key):
                    if st[1] != "=":
                        return READ, False
                    if full and target[0] == "fld" and target[1] == lhs[1] and self._matches(target, lvl, key):
                        res = WRITTEN
            return res, False
        if k == "comm":
            if s[1] == "finish" or self.domain.world_size == 1 and not any(self.domain.periodic):
                return NEITHER, False
            return (READ if self._matches(s[3], lvl, key) else NEITHER), False
        if k == "applybc":
            return NEITHER, False
        if k == "advance":
            return (READ if self._matches(s[1], lvl, key) else NEITHER), False
        if k in ("repeat", "contract"):
            body = s[3] if k == "repeat" else s[-1]
            st, ret = self._scan(body, lvl, key, in_colour)
            if st == READ:
                return READ, ret
            n = 0
            try:
                from .exa4 import _Frame

                n = int(self._eval(s[1], _Frame(lvl, {})))
            except Exception:                 # noqa: BLE001 -- the count depends on run-time values: it may be zero
                n = 0
            return (WRITTEN if (st == WRITTEN and n >= 1) else NEITHER), ret
        if k == "until":
            if self._expr_reads(s[1], lvl, key):
                return READ, False
            st, ret = self._scan(s[2], lvl, key, in_colour)
            return (READ if st == READ else NEITHER), ret
        if k == "if":
            if self._expr_reads(s[1], lvl, key):
                return READ, False
            a, ra = self._scan(s[2], lvl, key, in_colour)
            b, rb = self._scan(s[3] or [], lvl, key, in_colour)
            if READ in (a, b):
                return READ, ra or rb
            return (WRITTEN if a == WRITTEN and b == WRITTEN else NEITHER), ra or rb
        if k == "color":
            return self._scan(s[2], lvl, key, True)
        if k == "levelscope":
            try:
                inside = lvl in self.levels_of(s[1], lvl)
            except Exception:                 # noqa: BLE001
                return READ, False
            return self._scan(s[2], lvl, key, in_colour) if inside else (NEITHER, False)
        return READ, False                    # a statement kind this scan does not know: assume it reads
