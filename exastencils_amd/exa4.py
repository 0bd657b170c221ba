"""ExaSlang-4 subset reader and interpreter: run the reference's own multigrid programs on libexamg.

The reference compiles an ExaSlang-4 program (`*.exa4`) plus a `.knowledge` file into C++/CUDA whose hot loops are
what libexamg implements.  This module reads the same two inputs and executes the program directly:
declarations become device fields and host stencils, leveled functions are interpreted, and every `loop over` body is
recognised as one of the emitted-loop kinds and issued as ONE library call on the current HIP stream -- there is no
per-point interpretation, no generated code and no CPU arithmetic on field data.

Grammar covered (Compiler/src/exastencils/parsers/l4/L4_Parser.scala): `Domain` (:286-296), `Layout` with
`ghostLayers` / `duplicateLayers [with communication]` (:378-396), `Field name< domain, layout, bc >[slots]@levels`
(:398-408), `Stencil` with offset entries, `from [...] with` mapping entries and `from default restriction|prolongation`
(:430-466), `StencilField` (:468-472), `Globals` (:227-233), leveled `Function`s (:235-260) with `Var`/`Val`,
assignments, `loop over f [only dup [..] on boundary] [where c] [with reduction (op : v)]` (:303-340),
`communicate [ghost|dup of]`, `apply bc to`, `advance`, `repeat n times [count v]`, `repeat until`, `if/else`,
`color with`, `return`, level scopes `@(...) { }` (:653-667), declaration level lists `@all`, `@(a to b)`, `@(a, b)`,
`@(a and b)`, `@(all but x)`, `coarsest + 1`, and access levels `@current|coarser|finer|finest|coarsest|<n>`.

Loop bodies recognised (anything else raises Exa4Unsupported -- nothing is silently approximated):
  F = c | F = G | F = analytic(x,y,z) | F += a*G | F -= a*G | F = G + b*F            examg_set/fill_fn/axpby
  R = RHS - A*U | D = A*U                                                            examg_residual / stencil_op(APPLY)
  U<next> = U<active> + w(diag A) * (RHS - A*U<active>)                              examg_jacobi
  U += w(diag A) * (RHS - A*U)   inside `color with {(i0+i1+i2) % 2, ...}` or `where (c == (i0+i1+i2) % 2)`
                                                                                      examg_rbgs_colour
  RHS@coarser = [s *] Restriction * Residual | U += Prolongation@coarser * U@coarser  examg_restrict / examg_prolong_add
  s += F*G (reduction +) | s = max(s, fabs(F - analytic)) (reduction max)             examg_dot / examg_max_err_fn
  A:[o] = expr (every entry of a stencil field)                                      examg_init_varcoeff7 / examg_fill_expr
Analytic point functions (boundary values, right-hand sides, exact solutions) are matched numerically against the
built-in function ids of include/examg.h; any other expression over the node position -- user functions included -- is
compiled to a postfix program (examg_expr_t) that the device evaluates per point in the order of the expression tree
(examg_fill_expr / examg_apply_dirichlet_expr / examg_max_err_expr).  Stencil-field entries `A:[o] = expr` take
examg_init_varcoeff7 when they are -div(a grad) with a built-in `a`, else one expression program per coefficient plane.

Fewer passes than statements, where the statements allow it (`fuse=True`, bit-identical): a `color with` red-black
sweep is one out-of-place pass (examg_rbgs_sweep_fused), `repeat n times { Smoother ( ) }` with a slotted Jacobi body runs
as n/2 two-step passes (examg_jacobi2_boxes) -- both only while the boundary planes involved are known to hold the
field's Dirichlet values.  A coarsest-level function that is statement for statement the generated CG solver -- in the form of the
layer-4 benchmark program or in the one the layer-3 solver generator writes (_match_coarse_cg) -- becomes one persistent kernel
(examg_cg_coarse / examg_cg_coarse_variant; `fuse_coarse_solver`, agrees to reduction-order rounding); a cycle is then free of host
synchronisation and `capture()` records it into a hipGraph.

Two deliberate readings of printed-L4 files (Testing/Smoothers/Jac.exa4:43 declares the finest `Solution` without level
and slot count): a declaration never overrides an earlier one on the same level (the reference's collection lookup
returns the first match, knowledge/l4/L4_KnowledgeCollection.scala:78-79), and a field name has one slot count, the
maximum over its declarations.
"""
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

APPLY, RESIDUAL, SMOOTH = 0, 1, 2


# =====================================================================================================================
# analytic point functions: python mirror of eval_fn (exastencils_amd/csrc/examg_common.h), used for recognition only
# =====================================================================================================================
def fn_eval(fn: int, p: Sequence[float], x: float, y: float, z: float) -> float:
    PI = math.pi
    k = p[0] if p else 0.0
    if fn == 0:
        return 0.0
    if fn == 1:
        return ((x * x) - ((0.5 * y) * y)) - ((0.5 * z) * z)
    if fn == 2:
        return math.cos(PI * x) - math.sin((2.0 * PI) * y)
    if fn == 3:
        return (PI * PI) * math.cos(PI * x) - ((4.0 * (PI * PI)) * math.sin((2.0 * PI) * y))
    if fn == 4:
        return k * (((x - (x * x)) * (y - (y * y))) * (z - (z * z)))
    if fn == 5:
        return (2.0 * k) * ((((x - (x * x)) * (y - (y * y))) + ((x - (x * x)) * (z - (z * z)))) + ((y - (y * y)) * (z - (z * z))))
    if fn == 6:
        return 1.0 - math.exp((-1.0 * k) * (((x - (x * x)) * (y - (y * y))) * (z - (z * z))))
    if fn == 7:
        return math.exp(k * (((x - (x * x)) * (y - (y * y))) * (z - (z * z))))
    if fn == 8:
        return (math.sin(PI * x) * math.sin(PI * y)) * math.sinh((math.sqrt(2.0) * PI) * z)
    if fn == 9:
        return (math.sin(PI * x) * math.sin(PI * y)) * math.sin(PI * z)
    if fn == 10:
        return k * ((x - (x * x)) * (y - (y * y)))
    if fn == 11:
        return (2.0 * k) * ((x - (x * x)) + (y - (y * y)))
    if fn == 12:
        return 1.0 - math.exp((-1.0 * k) * ((x - (x * x)) * (y - (y * y))))
    if fn == 13:
        return math.exp(k * ((x - (x * x)) * (y - (y * y))))
    if fn == 14:
        return (x * x) - (y * y)
    if fn == 15:
        return math.sin(PI * x) * math.sinh(PI * y)
    if fn == 16:
        return x * x
    raise ValueError("function id %d" % fn)


_FN_WITH_PARAM = {4, 5, 6, 7, 10, 11, 12, 13}
_FN_2D_ONLY = {2, 3, 10, 11, 12, 13, 14, 15}     # ignore z
_FN_ANY_DIM = {0, 16}
_N_FN = 17


from .exa4_fusion import LazyFusions  # noqa: E402


# =====================================================================================================================
# interpreter
# =====================================================================================================================
class _Return(Exception):
    def __init__(self, value):
        self.value = value


@dataclass
class _Frame:
    level: Optional[int]
    vars: Dict[str, object]
    colour: Optional[int] = None
    contract: Optional[tuple] = None     # (extent, posExt, negExt) inside `repeat .. with contraction`: loops widen at interior faces


class Exa4Program(LazyFusions):
    """One ExaSlang-4 program bound to a kernel layer (`ops`: HipOps on the GPU), a block decomposition and a
    communicator.  `run()` executes `Function Application`; printed lines are collected in `self.out`."""

    def __init__(self, text: str, knowledge: Optional[Dict] = None, ops=None, domain: Optional[RectDomain] = None, comm=None,
                 echo: bool = False, fuse: bool = True, fuse_coarse_solver: Optional[bool] = None):
        """fuse: run red-black sweeps and pairs of slotted Jacobi steps as single passes over HBM where the program's
        statements allow it (bit-identical results; `fuse=False` issues exactly one launch per loop statement)."""
        self.ast = Parser(text).parse()
        self.k = dict(knowledge or {})
        d = _knowledge.derive(self.k)
        self.nd = d["dimensionality"]
        self.min_level, self.max_level = d["min_level"], d["max_level"]
        if ops is None:
            from .ops import HipOps

            ops = HipOps()              # raises without libexamg.so / GPU: no fallback
        self.ops = ops
        lo, hi = (0.0,) * 3, (1.0,) * 3
        if self.ast.domain:
            lo = tuple(float(_const_value(e)) for e in self.ast.domain[1]) + (0.0,) * (3 - self.nd)
            hi = tuple(float(_const_value(e)) for e in self.ast.domain[2]) + (1.0,) * (3 - self.nd)
        self._merged_blocks = None
        if domain is None:
            # one process: the reference's blocks x fragments become one fragment of the same global grid
            flen = tuple(d["frags_total"][i] * d["frag_len"][i] for i in range(3))
            domain = RectDomain(self.nd, (1, 1, 1), 0, flen, lo[:3], hi[:3], d["periodic"])
            if d["num_blocks"] != (1, 1, 1) and d["frags_per_block"] == (1, 1, 1):
                self._merged_blocks = (d["num_blocks"], d["frag_len"])      # what a per-process std::rand() needs to know (_exec_rand_fill)
        self.domain = domain
        self.comm = comm or Communicator(domain, ops)
        self.echo = echo
        self.out: List[str] = []
        self.printed_values: List[float] = []     # every Real handed to print / printWithReducedPrec, unrounded
        self.json_results: Dict[str, Dict] = {}
        self.json_dir: Optional[str] = None       # where printJSON writes its file (None: keep it in json_results only)
        self.timers: Dict[str, float] = {}
        self._timer_start: Dict[str, float] = {}
        self.launches = 0
        self._precision = 6
        self.globals: Dict[str, object] = {}
        self.fields: Dict[Tuple[str, int], Field] = {}
        self.stencils: Dict[Tuple[str, int], Stencil] = {}
        self.transfer: Dict[str, str] = {}
        self._fn_cache: Dict[Tuple, Tuple] = {}
        self._rng = random.Random(20240229)
        self.fuse = fuse
        # a coarsest-level function that is exactly the generated conjugate-gradient solver runs as ONE persistent kernel
        # (examg_cg_coarse; single block).  Same algorithm and statement order; the in-kernel reductions sum in another
        # (fixed) order, so iterates agree with the statement-by-statement run to rounding, not bit for bit.
        self.fuse_coarse_solver = fuse if fuse_coarse_solver is None else fuse_coarse_solver
        self._cg_plans: Dict[Tuple[str, int], object] = {}
        self.fuse_min_row = 64      # rows shorter than one wavefront's tile stay on the per-colour kernels
        self._alt: Dict[Tuple[str, int, int], object] = {}
        self._alt_shell: Dict[Tuple[str, int, int], int] = {}
        self._bc_epoch: Dict[Tuple[str, int], int] = {}
        self._pair_tmp: Dict[Tuple[str, int], Field] = {}
        self._bc_valid = set()      # (field, level, slot) whose physical-boundary planes hold the field's Dirichlet values
        self._lazy_init()           # pending loops of the cross-statement fusions (exastencils_amd/exa4_fusion.py)
        self._declare()

    # -- declarations -> objects --------------------------------------------------------------------------------------
    def levels_of(self, spec, cur: Optional[int] = None) -> List[int]:
        lo, hi = self.min_level, self.max_level
        if spec is None or spec[0] == "all":
            return list(range(lo, hi + 1))
        k = spec[0]
        if k == "single":
            base = spec[1]
            if isinstance(base, str):
                if base in ("current", "coarser", "finer") and cur is None:
                    raise Exa4SyntaxError("@%s outside a leveled function" % base)
                base = {"current": cur, "coarser": None if cur is None else cur - 1, "finer": None if cur is None else cur + 1,
                        "finest": hi, "coarsest": lo}[base]
            return [base + spec[2]]
        if k == "range":
            a, b = self.levels_of(spec[1], cur)[0], self.levels_of(spec[2], cur)[0]
            return list(range(min(a, b), max(a, b) + 1))
        if k == "list":
            out = []
            for s in spec[1]:
                out += self.levels_of(s, cur)
            return sorted(set(out))
        if k == "but":
            drop = set(self.levels_of(spec[2], cur))
            return [l for l in self.levels_of(spec[1], cur) if l not in drop]
        raise Exa4SyntaxError("level specification %r" % (spec,))

    def _declare(self):
        a, nd, dom = self.ast, self.nd, self.domain
        self.functions: Dict[str, List[FunctionDecl]] = {}
        for fn in a.functions:
            self.functions.setdefault(fn.name, []).append(fn)
        for name, e in a.globals:
            self.globals[name] = self._eval(e, _Frame(None, {}))
        layouts = {l.name: l for l in a.layouts}
        nslots: Dict[str, int] = {}
        for fd in a.fields:
            nslots[fd.name] = max(nslots.get(fd.name, 1), fd.slots)
        self._sfield_of = {}
        sfield_fields = {s.field for s in a.sfields}
        for fd in a.fields:
            ld = layouts.get(fd.layout)
            if ld is None:
                raise Exa4SyntaxError("field %s uses the undeclared layout %s" % (fd.name, fd.layout))
            if ld.localization != "Node":
                raise Exa4Unsupported("layout %s: localization %s (node fields only)" % (ld.name, ld.localization))
            if ld.vec_len != 1 and fd.name not in sfield_fields:
                raise Exa4Unsupported("vector-valued field %s outside a StencilField" % fd.name)
            for lvl in self.levels_of(fd.levels):
                if (fd.name, lvl) in self.fields or not (self.min_level <= lvl <= self.max_level):
                    continue
                nc = dom.ncells(lvl)
                ghost = tuple(ld.ghost[i] if i < nd and i < len(ld.ghost) else 0 for i in range(3))
                dup = tuple((ld.dup[i] if i < len(ld.dup) else 1) if i < nd else 0 for i in range(3))
                if any(dup[i] != 1 for i in range(nd)):
                    raise Exa4Unsupported("layout %s: node fields need one duplicate layer" % ld.name)
                inner = tuple(nc[i] - 1 if i < nd else 1 for i in range(3))
                if ld.inner and tuple(ld.inner[:nd]) != inner[:nd]:
                    # an explicit size (fieldlike/l4/L4_FieldLikeLayoutDecl.scala:49-51: innerPoints replaces the default 2^level * fragLen
                    # - 1): the benchmark programs of Testing/PolyExpl put a 256^3 array on level 0.  Loops follow the layout; grid
                    # widths and node positions stay those of the level, as in the reference.  Only without inter-grid transfers
                    if self.min_level != self.max_level or dom.world_size != 1:
                        raise Exa4Unsupported("layout %s: innerPoints %s differ from the %s points level %d gives one fragment"
                                              % (ld.name, list(ld.inner[:nd]), list(inner[:nd]), lvl))
                    inner = tuple(int(ld.inner[i]) if i < nd else 1 for i in range(3))
                lay = FieldLayout(nd, inner, ghost, dup, (0, 0, 0), (0, 0, 0), ld.dup_comm, ld.ghost_comm)
                bc_fn, bc_par = None, ()
                if fd.bc is not None:
                    bc_fn, bc_par = self._analytic(fd.bc, lvl)
                if ld.vec_len != 1:
                    f = Field.__new__(Field)     # coefficient planes: allocated by the stencil field below
                    f.name, f.level, f.layout, f.num_slots, f.bc_fn, f.bc_params = fd.name, lvl, lay, 1, None, ()
                    f.lc, f.slots, f.current_slot, f.vec_len = lay.c_struct(), [self.ops.new_array(ld.vec_len * lay.size)], 0, ld.vec_len
                else:
                    f = Field(fd.name, lvl, lay, self.ops, nslots[fd.name], bc_fn, bc_par)
                self.fields[(fd.name, lvl)] = f
        for sd in a.stencils:
            if sd.transfer:
                self.transfer[sd.name] = sd.transfer
        self._stencil_decls = {s.name: s for s in a.stencils if not s.transfer}
        for sf in a.sfields:
            sd = self._stencil_decls.get(sf.stencil)
            if sd is None:
                raise Exa4SyntaxError("stencil field %s: stencil %s is not declared" % (sf.name, sf.stencil))
            for lvl in self.levels_of(sf.levels):
                cf = self.fields.get((sf.field, lvl))
                if cf is None:
                    continue
                offs = [tuple(o) + (0,) * (3 - len(o)) for o, _ in sd.entries]
                if getattr(cf, "vec_len", 1) != len(offs):
                    raise Exa4SyntaxError("stencil field %s: %d entries but %d coefficients per point" % (sf.name, len(offs), getattr(cf, "vec_len", 1)))
                self.stencils[(sf.name, lvl)] = Stencil(offs, [], cf.slots[0], cf.layout)
        self._apply_layout_transformations()

    def _apply_layout_transformations(self):
        """`LayoutTransformations { transform <field>@<levels> with [x, y, z, i] => [i, x, y, z] }` on the coefficient field of a
        stencil field (Compiler/src/exastencils/layoutTransformation/l4/L4_LayoutSection.scala; Testing/LayoutTrafo/*.exa4): the
        entries of a point become contiguous -- APPLIED: loops read the coefficients through the transformed index
        (EXAMG_CLAYOUT_ENTRY_FASTEST), one stream instead of one per entry.  The other directives of the reference's test programs
        (colour splits, axis permutations, concat / rename of scalar fields) change no value either and stay recorded only: scalar
        fields keep the reference layout, the library's contract with its callers."""
        import re

        self._sf_entry_fastest, self._sf_rec, self._sf_dirty = set(), {}, {}
        nd = self.nd
        src = ",".join("xyz"[:nd]) + ",i"
        dst = "i," + ",".join("xyz"[:nd])
        coef_of = {}
        for sf in self.ast.sfields:
            coef_of.setdefault(sf.field, []).append(sf)
        for text in getattr(self.ast, "layout_transformations", []):
            m = re.match(r"^transform (.+?) with \[(.+?)\] => \[(.+?)\]$", text.strip())
            if not m or m.group(2).replace(" ", "") != src or m.group(3).replace(" ", "") != dst:
                continue
            items, depth, cur = [], 0, []
            for tok in m.group(1).split(" "):          # items are separated by commas outside level lists
                depth += tok == "("
                depth -= tok == ")"
                if tok == "," and depth == 0:
                    items.append(cur)
                    cur = []
                else:
                    cur.append(tok)
            items.append(cur)
            for it in items:
                if not it:
                    continue
                fname, spec = it[0], " ".join(it[2:]) if len(it) > 2 and it[1] == "@" else "all"
                for sf in coef_of.get(fname, []):
                    lvls = self.levels_of(self._parse_level_text(spec))
                    for lvl in lvls:
                        if (sf.name, lvl) in self.stencils:
                            self._sf_entry_fastest.add((sf.name, lvl))

    def _parse_level_text(self, spec: str):
        """Level specification of a LayoutTransformations item (`all`, `finest`, `4`, `(4 to finest)`) as the parser's level node."""
        from .exa4_parser import Parser

        return Parser("@ " + spec).decl_levels()

    def stencil(self, name: str, lvl: int) -> Stencil:
        """The stencil as loops use it: a stencil field whose coefficient field is under the entry-fastest layout transformation is
        handed out in that layout (re-laid out from the planes the initialisation statements write, when they have changed)."""
        s = self._stencil_planes(name, lvl)
        key = (name, lvl)
        if key in getattr(self, "_sf_entry_fastest", ()) and s.cfield is not None and hasattr(self.ops, "transform_stencilfield"):
            t = self._sf_rec.get(key)
            if t is None:
                t = self._sf_rec[key] = s.entry_fastest(self.ops)
                self.launches += 1
            elif self._sf_dirty.get(key, False):
                self.ops.transform_stencilfield(s.clayout.c_struct(), len(s.offsets), s.cfield, t.cfield, True)
                self.launches += 1
            self._sf_dirty[key] = False
            return t
        return s

    def _stencil_planes(self, name: str, lvl: int) -> Stencil:
        s = self.stencils.get((name, lvl))
        if s is None:
            sd = self._stencil_decls.get(name)
            if sd is None:
                raise Exa4SyntaxError("stencil %s is not declared" % name)
            if lvl not in self.levels_of(sd.levels):
                raise Exa4SyntaxError("stencil %s is not declared on level %d" % (name, lvl))
            fr = _Frame(lvl, {})
            offs = [tuple(o) + (0,) * (3 - len(o)) for o, _ in sd.entries]
            s = self.stencils[(name, lvl)] = Stencil(offs, [float(self._eval(e, fr)) for _, e in sd.entries])
        return s

    # -- analytic function recognition ----------------------------------------------------------------------------------
    def _point_eval(self, e, lvl: Optional[int], x: float, y: float, z: float):
        fr = _Frame(lvl, {"__x": x, "__y": y, "__z": z})
        return float(self._eval(e, fr))

    def _recognise(self, e, lvl: Optional[int]) -> Tuple[int, Tuple[float, ...]]:
        """Function id and parameters of include/examg.h that reproduce the point expression `e`."""
        key = (repr(e), lvl)
        if key in self._fn_cache:
            return self._fn_cache[key]
        pts = [(self._rng.uniform(0.05, 0.95), self._rng.uniform(0.05, 0.95), self._rng.uniform(0.05, 0.95)) for _ in range(12)]
        vals = [self._point_eval(e, lvl, *p) for p in pts]
        cands = [float(v) for v in self.globals.values() if isinstance(v, (int, float)) and not isinstance(v, bool)]
        for fn in range(_N_FN):
            if (fn in _FN_2D_ONLY) != (self.nd == 2) and fn not in _FN_ANY_DIM:
                continue
            for par in (cands if fn in _FN_WITH_PARAM else [None]):
                p = (par,) if par is not None else ()
                if all(abs(fn_eval(fn, p, *pt) - v) <= 1e-12 * max(1.0, abs(v)) for pt, v in zip(pts, vals)):
                    self._fn_cache[key] = (fn, p)
                    return fn, p
        raise Exa4Unsupported("analytic expression is none of the built-in point functions (include/examg.h EXAMG_FN_*)")

    def _compile_point_expr(self, e, lvl: Optional[int], subst: Optional[Dict[str, list]] = None) -> list:
        """Postfix program (include/examg.h EXAMG_OP_*) of an expression over the node position: operands in the order of
        the expression tree, user functions inlined, everything that does not depend on the position folded to constants
        exactly where the tree has it."""
        k = e[0]
        fr = _Frame(lvl, {})
        if k == "num":
            return [("const", float(e[1]))]
        if k == "neg":
            return self._compile_point_expr(e[1], lvl, subst) + [("neg", None)]
        if k == "id":
            name = e[1]
            if subst is not None and name in subst:
                return list(subst[name])
            m = _COORD.match(name)
            if m:
                return [(m.group(2), None)]
            return [("const", float(self._eval(e, fr)))]        # globals, PI, vf_gridWidth_*
        if k == "bin" and e[1] in ("+", "-", "*", "/"):
            return self._compile_point_expr(e[2], lvl, subst) + self._compile_point_expr(e[3], lvl, subst) + [(e[1], None)]
        if k == "bin" and e[1] == "**":
            base = self._compile_point_expr(e[2], lvl, subst)
            if e[3][0] == "num" and float(e[3][1]) == 2.0:
                return base + base + [("*", None)]                # the generator expands integer powers into products
            return base + self._compile_point_expr(e[3], lvl, subst) + [("pow", None)]
        if k == "call":
            name, args = e[1], e[3]
            if name in self.functions:
                fn = self._resolve(name, self._level_of(e[2], fr) if e[2] is not None else lvl)
                if len(fn.body) != 1 or fn.body[0][0] != "return" or fn.body[0][1] is None or len(fn.params) != len(args):
                    raise Exa4Unsupported("function %s inside a point expression must be a single return statement" % name)
                inner = {p: self._compile_point_expr(a, lvl, subst) for p, a in zip(fn.params, args)}
                return self._compile_point_expr(fn.body[0][1], lvl, inner)
            un = {"sin": "sin", "cos": "cos", "exp": "exp", "sinh": "sinh", "cosh": "cosh", "sqrt": "sqrt", "tan": "tan", "log": "log",
                  "fabs": "fabs", "abs": "fabs", "tanh": "tanh"}
            if name in un and len(args) == 1:
                return self._compile_point_expr(args[0], lvl, subst) + [(un[name], None)]
            if name in ("pow", "max", "min") and len(args) == 2:
                return self._compile_point_expr(args[0], lvl, subst) + self._compile_point_expr(args[1], lvl, subst) + [(name, None)]
        raise Exa4Unsupported("point expression with %s %r" % (k, e[1] if len(e) > 1 else ""))

    def _analytic(self, e, lvl: Optional[int]):
        """(fn id, params) of a built-in point function when the expression is one, else an expression program (ExprC)."""
        try:
            return self._recognise(e, lvl)
        except Exa4Unsupported:
            pass
        key = ("expr", repr(e), lvl)
        if key not in self._fn_cache:
            from .lib import ExprC

            try:
                self._fn_cache[key] = (ExprC.from_program(self._compile_point_expr(e, lvl)), ())
            except ValueError as ex:
                raise Exa4Unsupported(str(ex))
        return self._fn_cache[key]

    # -- expression evaluation (host scalars) ---------------------------------------------------------------------------
    def _level_of(self, spec, fr: _Frame) -> int:
        if spec is None:
            if fr.level is None:
                raise Exa4SyntaxError("leveled access without level outside a leveled function")
            return fr.level
        lv = self.levels_of(spec, fr.level)
        if len(lv) != 1:
            raise Exa4SyntaxError("access needs a single level")
        return lv[0]

    def _eval(self, e, fr: _Frame):
        k = e[0]
        if k == "num" or k == "str":
            return e[1]
        if k == "neg":
            return -self._eval(e[1], fr)
        if k == "not":
            return not self._eval(e[1], fr)
        if k == "bin":
            return _arith(e[1], self._eval(e[2], fr), self._eval(e[3], fr))
        if k == "id":
            name = e[1]
            if name in fr.vars:
                return fr.vars[name]
            if name in self.globals:
                return self.globals[name]
            if name == "PI":
                return math.pi
            m = _GRIDW.match(name)
            if m:
                return self.domain.h(self._level_of(e[2], fr))["xyz".index(m.group(1))]
            m = _COORD.match(name)
            if m:
                v = fr.vars.get("__" + m.group(2))
                if v is None:
                    raise Exa4Unsupported("%s outside a point expression" % name)
                return v
            raise Exa4SyntaxError("unknown name %r" % name)
        if k == "call":
            return self._call(e, fr)
        if k in ("fld", "sten", "sentry"):
            raise Exa4Unsupported("field / stencil access %s outside a recognised loop body" % e[1])
        raise Exa4SyntaxError("expression %r" % (e,))

    def _call(self, e, fr: _Frame):
        name, lspec, args = e[1], e[2], e[3]
        if name == "diag":
            a = args[0]
            if a[0] != "sten":
                raise Exa4Unsupported("diag of a non-stencil")
            st = self.stencil(a[1], self._level_of(a[2], fr))
            if st.cfield is not None:
                raise Exa4Unsupported("diag of a stencil field outside the smoother form ((1.0 / diag(A)) * omega)")
            return st.diag
        if name in _MATH and name not in self.functions:
            return _MATH[name](*[self._eval(a, fr) for a in args])
        if name.split("_")[0] in ("printField", "writeField", "readField") and name not in self.functions:
            return self._field_io(name, args, fr)
        if name in ("startTimer", "stopTimer", "getTotalFromTimer", "getTotalTime", "getMeanFromTimer") and args and \
                args[0][0] == "id" and args[0][1] not in fr.vars and args[0][1] not in self.globals:
            # timers may be named by a bare identifier (Testing/PolyExpl/Jac3Dcc.exa4:49: startTimer(benchTimer))
            args = [("str", args[0][1])] + list(args[1:])
        if name == "getKnowledge":
            key = self._eval(args[0], fr)
            return self.k.get(key, {"testing_enabled": False, "testing_printRes": True, "testing_printErr": True}.get(key, False))
        if name in self.functions:
            lvl = None
            if lspec is not None:
                lvl = self._level_of(lspec, fr)
            return self.call(name, lvl if lvl is not None else fr.level, [self._eval(a, fr) for a in args], fr)
        return self._builtin(name, [self._eval(a, fr) for a in args], fr)

    # -- field I/O ---------------------------------------------------------------------------------------
    def _field_io(self, name: str, args: list, fr: _Frame):
        """printField / writeField / readField [ _lock | _fpp ] ( "file", field [, includeGhost [, binary [, condition [, separator ]]]] )
        (Compiler/src/exastencils/field/ir/IR_PrintField.scala:38-110, IR_ReadField / IR_WriteField; argument order as in
        Testing/IOTest/3D_Scalar_CheckEquality_ReadAfterWrite.exa4:78-100).  "$blockId" in the file name becomes the rank.
        Blocks of a decomposition write one after the other into the same file (the reference's MPI_Sequential, "lock"
        interface); the data leave / enter the device through the kernel layer's to_host / from_host."""
        from . import io as xio

        base, iface = (name.split("_") + ["lock"])[:2]
        pos = [i for i, a in enumerate(args) if a[0] == "fld"]
        if not pos:
            raise Exa4Unsupported("%s without a field argument" % name)
        f, slot = self._field(args[pos[0]], fr)
        fname = str(self._eval(args[0], fr)).replace("$blockId", str(self.domain.rank))
        rest = [self._eval(a, fr) for a in args[pos[0] + 1:]]
        if iface in ("hdf5", "nc", "mpiio", "sion"):
            # write/readField_hdf5 ( file, dataset, field ), _nc ( file, variable, field [, includeGhost] ), _mpiio ( file, field ),
            # _sion ( file, field [, includeGhost [, condition]] ) (IOTest:115-168).  Those libraries are not part of this image: the
            # values go to `file` as the raw doubles of the lock / fpp interfaces -- the same round trip, not those file formats
            if base == "printField":
                raise Exa4Unsupported("%s: visualisation output of the %s interface" % (name, iface))
            include_ghost = bool(rest[0]) if rest and iface in ("nc", "sion") else False
            binary, condition, separator = True, (rest[1] if len(rest) > 1 and iface == "sion" else True), " "
        else:
            include_ghost = bool(rest[0]) if len(rest) > 0 else False
            binary = bool(rest[1]) if len(rest) > 1 else (base != "printField" and iface != "lock")
            condition = rest[2] if len(rest) > 2 else True
            separator = str(rest[3]) if len(rest) > 3 else " "
        if not isinstance(condition, bool):
            raise Exa4Unsupported("%s: only constant conditions" % name)
        d = os.path.dirname(fname)
        if d and self.domain.rank == 0:
            os.makedirs(d, exist_ok=True)
        dist = getattr(self.comm, "dist", None)
        shared = dist is not None and "$blockId" not in str(self._eval(args[0], fr))
        self.ops.synchronize()
        for turn in range(self.domain.world_size if shared else 1):
            if not shared or turn == self.domain.rank:
                if not condition:
                    if base != "readField" and turn == 0:
                        open(fname, "w").close()
                elif base == "readField":
                    if shared and self.domain.world_size > 1:
                        raise Exa4Unsupported("readField from one file shared by several blocks")
                    if binary:
                        xio.read_field(fname, f, self.ops, slot, include_ghost)
                    else:
                        xio.read_field_ascii(fname, f, self.ops, slot, include_ghost, separator)
                elif binary:
                    if shared and self.domain.world_size > 1:
                        raise Exa4Unsupported("binary writeField into one file shared by several blocks")
                    xio.write_field(fname, f, self.ops, slot, include_ghost)
                else:
                    xio.print_field(fname, f, self.ops, self.domain, slot, include_ghost, separator, None, append=shared and turn > 0,
                                    precision=int(self.k.get("field_printFieldPrecision", -1)))
            if shared:
                dist.barrier()
        if base == "readField":
            self._bc_valid.discard((f.name, f.level, slot))       # whatever the boundary planes held, the file's values replace it
            self._bc_epoch[(f.name, f.level)] = self._bc_epoch.get((f.name, f.level), 0) + 1
        return None

    def _range(self, name: str, push: bool):
        torch = getattr(self.ops, "torch", None)
        if torch is None or getattr(getattr(self.ops, "device", None), "type", "cpu") == "cpu":
            return
        try:
            if push:
                torch.cuda.nvtx.range_push(name)
            else:
                torch.cuda.nvtx.range_pop()
        except Exception:       # profiler ranges are an aid, never a reason to stop a program
            pass

    # -- built-in statements ----------------------------------------------------------------------------------------------
    def _emit(self, line: str):
        self.out.append(line)
        if self.echo and self.domain.rank == 0:
            print(line, flush=True)

    def _fmt(self, v) -> str:
        if isinstance(v, bool):
            return "true" if v else "false"
        if isinstance(v, float):
            return "%.*g" % (self._precision, v)
        return str(v)

    def _builtin(self, name: str, args: list, fr: _Frame):
        from .solver import reduced_prec

        if name == "print":
            self.printed_values += [a for a in args if isinstance(a, float)]
            self._emit(" ".join(self._fmt(a) for a in args))
        elif name == "printWithReducedPrec":
            self.printed_values.append(float(args[0]))
            self._emit(reduced_prec(float(args[0])))
        elif name == "native" and re.fullmatch(r"\s*std::srand\s*\(\s*(\d+)\s*\)\s*;?\s*", str(args[0])):
            from .crand import CRand

            seed = int(re.search(r"\d+", str(args[0])).group(0))      # native('std::srand(42)') (Testing/PolyExpl/Jac3Dcc.exa4:33)
            if getattr(self, "_crand", None) is None:
                self._crand = CRand(seed)
            else:
                self._crand.seed(seed)
        elif name == "native":
            m = re.search(r"cout\.precision\((\w+)\)", str(args[0]))
            if m and "oldPrec =" not in str(args[0]):
                self._precision = int(m.group(1)) if m.group(1).isdigit() else 6
        elif name == "startTimer":
            # IR_Stopwatch (Compiler/src/exastencils/timing/ir/IR_Stopwatch.scala:31-84): wall-clock timer; on the GPU also a
            # profiler range of the same name (roctx, through torch.cuda.nvtx), so rocprofv3 --marker-trace shows the program's
            # own timers around the kernels they enclose
            self.ops.synchronize()
            self._range(str(args[0]), True)
            self._timer_start[args[0]] = time.perf_counter()
        elif name == "stopTimer":
            self.ops.synchronize()
            self._range(str(args[0]), False)
            self.timers[args[0]] = self.timers.get(args[0], 0.0) + time.perf_counter() - self._timer_start.pop(args[0])
        elif name == "printAllTimers":
            for key, val in self.timers.items():
                self._emit("Mean mean total time for Timer %s: %g" % (key, val * 1e3))
        elif name == "getTotalTime" or name == "getTotalFromTimer":
            return self.timers.get(args[0], 0.0) * 1e3
        elif name == "exit":
            raise SystemExit(int(args[0]) if args else 0)
        elif name in ("initGlobals", "initDomain", "initGeometry", "destroyGlobals", "initFieldsWithZero"):
            pass        # fields are allocated zeroed at declaration (initFieldsWithZero)
        elif name in ("benchmarkStart", "benchmarkStop"):
            pass        # likwid / time markers of Benchmark/run_benchmark.py: the timers around them carry the numbers
        elif name == "printJSON":
            # printJSON ( "file", 'key', value, ... ) (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:273-276)
            doc = {str(args[i]): args[i + 1] for i in range(1, len(args) - 1, 2)}
            self.json_results[str(args[0])] = doc
            if self.json_dir is not None and self.domain.rank == 0:
                import json

                with open(os.path.join(self.json_dir, str(args[0])), "w") as fh:
                    json.dump(doc, fh)
        else:
            raise Exa4Unsupported("function %s" % name)
        return None

    # -- functions ------------------------------------------------------------------------------------------------------
    def _resolve(self, name: str, lvl: Optional[int]) -> FunctionDecl:
        for fn in self.functions[name]:
            if fn.levels is None:
                return fn
            if lvl is not None and lvl in self.levels_of(fn.levels):
                return fn
        raise Exa4SyntaxError("function %s is not declared on level %r" % (name, lvl))

    def call(self, name: str, lvl: Optional[int] = None, args: Sequence = (), caller: Optional[_Frame] = None):
        fn = self._resolve(name, lvl)
        if self.fuse_coarse_solver and fn.levels is not None and lvl == self.min_level and not fn.params:
            plan = self._coarse_cg_plan(fn, lvl)
            if plan is not None:
                self._flush_pending()
                return self._run_coarse_cg(plan)
        fr = _Frame(lvl if fn.levels is not None else None, dict(zip(fn.params, args)))
        if caller is not None and "__x" in caller.vars:      # point expression: coordinates stay visible in callees
            for c in ("__x", "__y", "__z"):
                fr.vars.setdefault(c, caller.vars[c])
        try:
            self._exec_block(fn.body, fr, fn=True)
        except _Return as r:
            return r.value
        finally:
            if not self._cont:
                self._flush_pending()     # back at the caller of the interpreter: every field holds what the program says
        return None

    # -- hipGraph capture of a function call ----------------------------------------------------------------------------------
    def _roles(self):
        st = [(k, tuple(t.data_ptr() for t in f.slots), f.current_slot) for k, f in sorted(self.fields.items())]
        return st, sorted((k, v.data_ptr()) for k, v in self._alt.items())

    def capture(self, name: str, lvl: Optional[int] = None, args: Sequence = ()):
        """Record one call of a function (typically the cycle function on the finest level) into a hipGraph and return it;
        `graph.replay()` then re-issues all its kernels without host work.  The call must be free of host-visible
        reductions (a generated coarse-grid CG qualifies through examg_cg_coarse) and must leave every array in the role
        it had before (an even number of out-of-place sweeps / slot advances)."""
        torch = self.ops.torch
        dev = self.ops.device
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        # twice outside the capture: lazily created second arrays / scratch fields exist and carry their boundary planes
        for _ in range(2):
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self.call(name, lvl, args)
            cur.wait_stream(side)
        before = self._roles()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                self.call(name, lvl, args)
        except RuntimeError as ex:
            raise Exa4Unsupported("%s cannot be captured (host-visible reduction or allocation inside?): %s" % (name, ex))
        if self._roles() != before:
            raise Exa4Unsupported("%s leaves arrays in other roles than it found them: a replay would read stale data" % name)
        return g

    def run(self, name: str = "Application"):
        self.call(name)
        self._flush_cg_limit()
        return self.out

    def _flush_cg_limit(self):
        """The one-call coarse solve counts on the device how often its loop ran out of iterations (info[3]); the statements the
        program has after that loop -- a print -- run that often when the program ends."""
        if not hasattr(self, "_cg_tail"):
            return
        tail, lvl = self._cg_tail
        n = int(self.ops.to_host(self._cg_info)[3])
        for _ in range(n - getattr(self, "_cg_limit_seen", 0)):
            self._exec_block(tail, _Frame(lvl, {}))
        self._cg_limit_seen = n

    # -- statements -----------------------------------------------------------------------------------------------------
    def _exec_block(self, body: list, fr: _Frame, loop: bool = False, fn: bool = False):
        """loop: the list is a loop body that may run again; fn: a function body (a `return` ends here).  The stack of active lists
        is what the liveness scan of the cross-statement fusions walks (exa4_fusion.py: _dead_after)."""
        entry = [body, 0, fr, loop, fn]
        self._cont.append(entry)
        try:
            i = 0
            while i < len(body):
                entry[1] = i
                self._exec(body[i], fr)
                i += 1
        finally:
            self._cont.pop()

    def _field(self, e, fr: _Frame) -> Tuple[Field, int]:
        if e[0] != "fld":
            raise Exa4SyntaxError("field access expected, found %r" % (e,))
        lvl = self._level_of(e[3], fr)
        f = self.fields.get((e[1], lvl))
        if f is None:
            raise Exa4SyntaxError("field %s is not declared on level %d" % (e[1], lvl))
        s = e[2]
        if s is None or s in ("active", "activeSlot", "current", "currentSlot"):
            slot = f.active
        elif s in ("next", "nextSlot"):
            slot = f.next
        elif s in ("previous", "previousSlot"):
            slot = (f.current_slot - 1) % f.num_slots
        else:
            slot = int(s) % f.num_slots
        return f, slot

    def _exec(self, s, fr: _Frame):
        k = s[0]
        if self._pending is not None and self._gate(s, fr):
            return
        if k == "decl":
            fr.vars[s[1]] = self._eval(s[2], fr) if s[2] is not None else 0
        elif k == "assign":
            op, lhs, rhs = s[1], s[2], s[3]
            if lhs[0] != "id":
                raise Exa4Unsupported("assignment to %s outside a loop" % lhs[0])
            v = self._eval(rhs, fr)
            tgt = fr.vars if lhs[1] in fr.vars or lhs[1] not in self.globals else self.globals
            tgt[lhs[1]] = v if op == "=" else _arith(op[0], tgt[lhs[1]], v)
        elif k == "callstmt":
            self._call(s[1], fr)
        elif k == "loop":
            self._exec_loop(s, fr)
        elif k == "comm":
            if s[1] == "finish":
                return
            f, slot = self._field(s[3], fr)
            self.comm.exchange(f, slot, s[2])
        elif k == "applybc":
            f, slot = self._field(s[1], fr)
            self._apply_bc(f, slot)
        elif k == "advance":
            f, _ = self._field(s[1], fr)
            f.advance()
        elif k == "repeat":
            n = int(self._eval(s[1], fr))
            if self.fuse and s[2] is None and n >= 2 and self._try_jacobi_pairs(s[3], n, fr):
                return
            for it in range(n):
                if s[2]:
                    fr.vars[s[2]] = it
                self._exec_block(s[3], fr, loop=True)
            if s[2]:
                fr.vars[s[2]] = n
        elif k == "contract":
            self._exec_contract(s, fr)
        elif k == "until":
            while not self._eval(s[1], fr):
                self._exec_block(s[2], fr, loop=True)
        elif k == "if":
            self._exec_block(s[2] if self._eval(s[1], fr) else s[3], fr)
        elif k == "color":
            if len(s[1]) != 1:
                raise Exa4Unsupported("color with more than one colour expression")
            shift = _parity_expr(s[1][0], self.nd)
            if shift is None:
                raise Exa4Unsupported("colour expression other than (i0 + i1 [+ i2]) % 2")
            if self.fuse and self._try_fused_sweep(s[2], (0 - shift) % 2, fr):
                return
            saved = fr.colour
            for c in (0, 1):
                fr.colour = (c - shift) % 2
                self._exec_block(s[2], fr)
            fr.colour = saved
        elif k == "levelscope":
            if fr.level in self.levels_of(s[1], fr.level):
                self._exec_block(s[2], fr)
        elif k == "return":
            raise _Return(self._eval(s[1], fr) if s[1] is not None else None)
        else:
            raise Exa4SyntaxError("statement %r" % (k,))

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
        if D is not U or ds != us or not self._canonical7(A, self.nd) or U.layout.inner[0] < self.fuse_min_row:
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
        from .smoothers import jacobi_pair

        key = (U.name, U.level)
        tmp = self._pair_tmp.get(key)
        if tmp is None:
            tmp = self._pair_tmp[key] = Field(U.name + "Tmp", U.level, U.layout, self.ops, 1, None)
        k = n
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

    def _apply_bc(self, f: Field, slot: int):
        if f.bc_fn is None:
            return
        if self.fuse and (f.name, f.level, slot) in self._bc_valid:
            # Dirichlet values are functions of the position: the planes already hold exactly what this statement would write (every
            # writer of boundary planes -- `loop over .. only ..`, readField -- takes the entry out of _bc_valid) -- no launch, no bit changes
            return
        self._bc_valid.add((f.name, f.level, slot))
        mask = self.domain.face_mask()
        if mask:
            self.launches += 1
            if isinstance(f.bc_fn, int):
                self.ops.apply_dirichlet(f.lc, f.data(slot), self.domain.geom(f.level), f.bc_fn, f.bc_params, mask)
            else:
                self.ops.apply_dirichlet_expr(f.lc, f.data(slot), self.domain.geom(f.level), f.bc_fn, mask)

    # -- loops ----------------------------------------------------------------------------------------------------------
    def _loop_boxes(self, f: Field, only, where, reduction, fr: _Frame):
        """Iteration boxes of the loop and the colour its condition selects (None: all points)."""
        dom, lay, nd = self.domain, f.layout, self.nd
        colour = fr.colour
        lower = [False] * 3
        if where is not None:
            for c in _conjuncts(where):
                par = _colour_cond(c, nd)
                d = _lower_cond(c)
                if par is not None:
                    colour = par
                elif d is not None:
                    lower[d] = True
                else:
                    raise Exa4Unsupported("loop condition other than a colour test or (i_d > 0)")
        if only is not None:
            # `only <region> [dir] [on boundary]` (baseExt/ir/IR_LoopOverPointsInOneFragment.scala:57-72, 293-299): per dimension the
            # region's whole extent <R>LB..<R>RE where dir is 0, its lower part <R>LB..<R>LE / upper part <R>RB..<R>RE where dir is
            # -1 / +1, no iteration offsets; `on boundary`: only where the block has no neighbour in that direction
            region, direction, on_boundary = only
            code = {"dup": "D", "ghost": "G", "inner": "I"}.get(region)
            if code is None:
                raise Exa4Unsupported("loop ... only %s" % region)
            direction = tuple(direction) + (0,) * (3 - len(direction))
            if on_boundary:
                nz = [d for d in range(nd) if direction[d] != 0]
                if len(nz) != 1:
                    raise Exa4Unsupported("loop only ... on boundary: axis directions only")
                if dom.neighbor(nz[0], direction[nz[0]]) is not None:
                    return [], colour
            b, e = [0, 0, 0], [1, 1, 1]
            for t in range(nd):
                if code == "I":
                    lo_b, lo_e, hi_b, hi_e = "IB", "IE", "IB", "IE"
                else:
                    lo_b, lo_e, hi_b, hi_e = code + "LB", code + "LE", code + "RB", code + "RE"
                if direction[t] == 0:
                    b[t], e[t] = lay.idx(lo_b, t), lay.idx(hi_e, t)
                elif direction[t] < 0:
                    b[t], e[t] = lay.idx(lo_b, t), lay.idx(lo_e, t)
                else:
                    b[t], e[t] = lay.idx(hi_b, t), lay.idx(hi_e, t)
            return [(b, e)], colour
        b, e = dom.loop_bounds(lay, reduction is not None)
        for d in range(nd):
            if lower[d]:
                b[d] = max(b[d], 1)
        if fr.contract is not None and reduction is None:
            b, e = self._contract_bounds(lay, b, e, *fr.contract)
        return [(b, e)], colour

    def _contract_bounds(self, lay, b, e, extent: int, pos, neg):
        """Loop bounds inside `repeat n times with contraction`: widened by `extent` x the contraction per side at interior faces
        (iteration offset 0), unchanged on physical boundaries (IR_ContractingLoop.extendBoundsBegin / End,
        Compiler/src/exastencils/baseExt/ir/IR_ContractingLoop.scala:45-87); the layers computed redundantly must exist."""
        b, e = list(b), list(e)
        for d in range(self.nd):
            if self.domain.neighbor(d, -1) is not None:
                b[d] -= extent * (neg[d] if d < len(neg) else 0)
            if self.domain.neighbor(d, +1) is not None:
                e[d] += extent * (pos[d] if d < len(pos) else 0)
            if b[d] - 1 < lay.idx("GLB", d) or e[d] + 1 > lay.idx("GRE", d):
                raise Exa4Unsupported("contraction by %d layers needs deeper ghost layers than the layout has" % extent)
        return b, e

    def _exec_loop(self, s, fr: _Frame):
        _, target, only, where, reduction, body = s
        f, _ = self._field(target, fr)
        boxes, colour = self._loop_boxes(f, only, where, reduction, fr)
        if reduction is not None:
            return self._exec_reduction(f, boxes, reduction, body, fr)
        if body and all(st[0] == "assign" and st[2][0] == "sentry" for st in body):
            return self._exec_stencil_field_init(body, boxes, fr)
        if (body and colour is None and all(st[0] == "assign" and st[1] == "=" and st[2][0] == "fld" and self._is_std_rand(st[3])
                                            for st in body)):
            return self._exec_rand_fill([st[2] for st in body], boxes, fr)
        cmp_ = self._match_compare_loop(body, fr)
        if cmp_ is not None and colour is None:
            return self._exec_compare_loop(cmp_, boxes, fr)
        if colour is None and self._is_check_loop(body):
            return self._exec_check_loop(body, boxes, fr)
        if (len(body) == 1 and len(boxes) == 1 and colour is None and only is None and where is None and fr.contract is None
                and self._try_defer(body[0], f, boxes[0], fr)):
            return      # absorbed by a one-pass form later, or run when the next statement could tell the difference
        for st in body:
            if st[0] != "assign":
                raise Exa4Unsupported("statement %r inside a loop body" % st[0])
            if only is not None and st[2][0] == "fld":      # boundary planes rewritten: second arrays of fused sweeps are stale
                tf = self._field(st[2], fr)[0]
                self._bc_epoch[(tf.name, tf.level)] = self._bc_epoch.get((tf.name, tf.level), 0) + 1
                self._bc_valid -= {(tf.name, tf.level, sl) for sl in range(tf.num_slots)}
            for b, e in boxes:
                self._exec_point_assign(st, b, e, colour, fr)

    # `loop over F sequentially { F = native("((double)std::rand()/RAND_MAX)") }` (Testing/Opts/base.exa4:166-170): the generated
    # loop nest calls the C library's rand() once per point, x fastest, in every process after std::srand(mpiRank); the values come
    # from libexamg's restatement of glibc's generator (exastencils_amd/crand.py), written on the host and uploaded.
    @staticmethod
    def _is_std_rand(e) -> bool:
        return (e[0] == "call" and e[1] == "native" and len(e[3]) == 1 and e[3][0][0] == "str"
                and e[3][0][1].replace(" ", "") == "((double)std::rand()/RAND_MAX)")

    def _exec_rand_fill(self, targets, boxes, fr: _Frame):
        """One loop whose statements all draw from std::rand(): every point draws once per statement, in statement order."""
        from .crand import CRand, random_start

        fs = [self._field(t, fr) for t in targets]
        f, slot = fs[0]
        b, e = self.domain.loop_bounds(f.layout)
        if len(boxes) != 1 or list(boxes[0][0]) != list(b) or list(boxes[0][1]) != list(e):
            raise Exa4Unsupported("std::rand() start values on a restricted iteration space")
        if any(g.layout.shape_zyx != f.layout.shape_zyx for g, _ in fs):
            raise Exa4Unsupported("std::rand() start values for fields of different layouts in one loop")
        merged = self._merged_blocks[0] if self._merged_blocks is not None else None
        gen = getattr(self, "_crand", None)
        if merged is not None and (gen is not None or getattr(self, "_rand_drawn", False)):
            raise Exa4Unsupported("merged blocks: one loop drawing from std::rand(), with the default seeding")
        if merged is None and gen is None:      # this process' generator: seeded by the generated main() (rank; 1 without MPI)
            gen = self._crand = CRand(self.domain.rank if self.domain.world_size > 1 else 1)
        self._rand_drawn = True
        random_start(self.ops, f, slot, self.domain, merged, generator=gen, more_targets=fs[1:])
        self.launches += 1

    # `loop over B sequentially { Var d : Real = fabs ( B - A ); if ( d > tol ) { print ( ... ) ... return v } }`
    # (Testing/IOTest/3D_Scalar_CheckEquality_ReadAfterWrite.exa4:25-33): a search for the first point where two fields differ by
    # more than a tolerance.  One difference loop and one max-reduction on the device decide whether such a point exists; only then
    # are the fields brought to the host to find the first one in loop order for the program's messages and its `return`.
    def _match_compare_loop(self, body, fr: _Frame):
        if len(body) != 2 or body[0][0] != "decl" or body[1][0] != "if" or body[1][3]:
            return None
        name, init = body[0][1], body[0][2]
        if init is None or init[0] != "call" or init[1] not in ("fabs", "abs") or len(init[3]) != 1:
            return None
        d = init[3][0]
        if d[0] != "bin" or d[1] != "-" or d[2][0] != "fld" or d[3][0] != "fld":
            return None
        cond, guards = None, []      # `diff > tol`, possibly and-ed with conditions that do not depend on the point
        for c in _conjuncts(body[1][1]):
            if c[0] == "bin" and c[1] in (">", ">=") and c[2] == ("id", name, None) and self._is_scalar(c[3]) and cond is None:
                cond = c
            elif self._is_scalar(c) and ("id", name, None) not in list(_walk(c)) and not any(
                    x[0] == "id" and x[1] in ("i0", "i1", "i2") for x in _walk(c)):
                guards.append(c)
            else:
                return None
        if cond is None:
            return None
        if not all(bool(self._eval(gd, fr)) for gd in guards):
            return ("skip",)
        then = body[1][2]
        if not then or then[-1][0] != "return" or any(st[0] not in ("callstmt", "return") for st in then):
            return None
        return d[2], d[3], cond[1], cond[3], then

    def _exec_compare_loop(self, m, boxes, fr: _Frame):
        import numpy as np

        if m == ("skip",):
            return
        ea, eb, op, tol_e, then = m
        A, sa = self._field(ea, fr)
        B, sb = self._field(eb, fr)
        tol = float(self._eval(tol_e, fr))
        if not hasattr(self, "_cmp_tmp") or self._cmp_tmp.numel() < A.layout.size:
            self._cmp_tmp = self.ops.new_array(A.layout.size)
        worst = 0.0
        for b, e in boxes:
            self.ops.axpby(A.lc, A.data(sa), A.lc, self._cmp_tmp, 1.0, 0.0, b, e)            # tmp = A
            self.ops.axpby(B.lc, B.data(sb), A.lc, self._cmp_tmp, -1.0, 1.0, b, e)           # tmp -= B
            t = self.ops.max_err_fn(A.lc, self._cmp_tmp, self.domain.geom(A.level), 0, (), b, e)
            self.launches += 3
            worst = max(worst, self.ops.scalar_value(self.comm.allreduce(t, "max")))
        if not (worst > tol if op == ">" else worst >= tol):
            return
        # a point beyond the tolerance exists: the first one in loop order (x fastest) on this block, for the program's messages
        ha = self.ops.to_host(A.data(sa)).reshape(A.layout.shape_zyx)
        hb = self.ops.to_host(B.data(sb)).reshape(B.layout.shape_zyx)
        for b, e in boxes:
            sl = tuple(slice(A.layout.ref(d) + b[d], A.layout.ref(d) + e[d]) for d in (2, 1, 0))
            bad = np.argwhere(np.abs(ha[sl] - hb[sl]) > tol if op == ">" else np.abs(ha[sl] - hb[sl]) >= tol)
            if len(bad):
                k2, k1, k0 = (int(v) for v in bad[0])
                vals = {"i0": b[0] + k0, "i1": b[1] + k1, "i2": b[2] + k2}
                self._cmp_point = (vals, float(ha[sl][k2, k1, k0]), float(hb[sl][k2, k1, k0]))
                break
        self._exec_block_at_point(then, fr, A, sa, B, sb)

    # A loop that writes no field -- point-wise `Var`s and `if ( cond ) { print ( ... ) }` -- is a check of the data, not part of the
    # hot path (Testing/PolyExpl/Jac3Dcc.exa4:58-65: `Var s = Solution<active> * Solution<nextSlot>; if (s == 0.0 || s == 1./0. || ...)
    # print`): the fields it reads come to the host once, the expressions are evaluated over the whole box with numpy, and the
    # prints run for the offending points in loop order.
    @staticmethod
    def _is_check_loop(body) -> bool:
        def ok(st):
            if st[0] == "decl":
                return True
            if st[0] == "if":
                return not st[3] and all(x[0] == "callstmt" and x[1][1] == "print" for x in st[2])
            return False
        return bool(body) and all(ok(st) for st in body) and any(st[0] == "if" for st in body)

    def _np_eval(self, e, env, fr: _Frame, box):
        import numpy as np

        k = e[0]
        if k == "num":
            return float(e[1]) if not isinstance(e[1], bool) else e[1]
        if k == "str":
            return e[1]
        if k == "fld":
            f, slot = self._field(e, fr)
            key = (f.name, f.level, slot)
            if key not in env["_fields"]:
                lay = f.layout
                sl = tuple(slice(lay.ref(d) + box[0][d], lay.ref(d) + box[1][d]) for d in (2, 1, 0))
                env["_fields"][key] = self.ops.to_host(f.data(slot)).reshape(lay.shape_zyx)[sl]
            return env["_fields"][key]
        if k == "id":
            if e[1] in env:
                return env[e[1]]
            if e[1] in ("i0", "i1", "i2"):
                d = int(e[1][1])
                n = [box[1][t] - box[0][t] for t in range(3)]
                shape = [1, 1, 1]
                shape[2 - d] = n[d]
                return (np.arange(box[0][d], box[1][d]).reshape(shape) + np.zeros((n[2], n[1], n[0]), dtype=np.int64))
            return self._eval(e, fr)
        if k == "neg":
            return -self._np_eval(e[1], env, fr, box)
        if k == "not":
            return np.logical_not(self._np_eval(e[1], env, fr, box))
        if k == "bin":
            a, b = self._np_eval(e[2], env, fr, box), self._np_eval(e[3], env, fr, box)
            op = e[1]
            with np.errstate(all="ignore"):
                if op in ("&&", "and"):
                    return np.logical_and(a, b)
                if op in ("||", "or"):
                    return np.logical_or(a, b)
                if op == "/":
                    return np.divide(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))
                table = {"+": np.add, "-": np.subtract, "*": np.multiply, "**": np.power, "%": np.mod, "==": np.equal, "!=": np.not_equal,
                         "<": np.less, "<=": np.less_equal, ">": np.greater, ">=": np.greater_equal}
                if op not in table:
                    raise Exa4Unsupported("operator %s in a check loop" % op)
                return table[op](a, b)
        if k == "call" and e[1] in ("fabs", "abs", "sqrt", "exp", "sin", "cos") and len(e[3]) == 1:
            fn = {"fabs": np.abs, "abs": np.abs, "sqrt": np.sqrt, "exp": np.exp, "sin": np.sin, "cos": np.cos}[e[1]]
            with np.errstate(all="ignore"):
                return fn(self._np_eval(e[3][0], env, fr, box))
        raise Exa4Unsupported("expression %s in a check loop" % (k,))

    def _exec_check_loop(self, body, boxes, fr: _Frame):
        import numpy as np

        self.ops.synchronize()
        for box in boxes:
            n = [box[1][d] - box[0][d] for d in range(3)]
            if n[0] * n[1] * n[2] == 0:
                continue
            env = {"_fields": {}}
            for st in body:
                if st[0] == "decl":
                    env[st[1]] = self._np_eval(st[2], env, fr, box) if st[2] is not None else 0.0
                    continue
                mask = np.broadcast_to(np.asarray(self._np_eval(st[1], env, fr, box), dtype=bool), (n[2], n[1], n[0]))
                for k2, k1, k0 in np.argwhere(mask)[:1000]:      # (a check that fires on every point need not print them all)
                    pt = {"i0": box[0][0] + int(k0), "i1": box[0][1] + int(k1), "i2": box[0][2] + int(k2)}
                    for x in st[2]:
                        out = []
                        for a in x[1][3]:
                            v = self._np_eval(a, {**env, **pt}, fr, box) if a[0] != "str" else a[1]
                            if isinstance(v, np.ndarray):
                                v = np.broadcast_to(v, (n[2], n[1], n[0]))[k2, k1, k0].item()
                            out.append(v)
                        self._emit(" ".join(self._fmt(v) for v in out))

    def _exec_block_at_point(self, stmts, fr: _Frame, A, sa, B, sb):
        """The statements of the compare loop's `if` at the offending point: prints see the fields' values and i0 / i1 / i2 there."""
        vals, va, vb = getattr(self, "_cmp_point", ({"i0": -1, "i1": -1, "i2": -1}, float("nan"), float("nan")))
        for st in stmts:
            if st[0] == "return":
                raise _Return(self._eval(st[1], fr) if st[1] is not None else None)
            c = st[1]
            if c[1] != "print":
                self._exec(st, fr)
                continue
            out = []
            for a in c[3]:
                if a[0] == "fld":
                    f, _ = self._field(a, fr)
                    out.append(va if f is A else vb)
                elif a[0] == "id" and a[1] in vals:
                    out.append(vals[a[1]])
                else:
                    out.append(self._eval(a, fr))
            self._emit(" ".join(self._fmt(x) for x in out))

    # pattern helpers ---------------------------------------------------------------------------------------------------
    def _is_scalar(self, e) -> bool:
        return not _contains(e, ("fld", "sten", "sentry")) and not _has_coord(e, self.functions)

    def _same_access(self, a, b, fr: _Frame) -> bool:
        if a[0] != "fld" or b[0] != "fld":
            return False
        fa, sa = self._field(a, fr)
        fb, sb = self._field(b, fr)
        return fa is fb and sa == sb

    def _sten_times_field(self, e, fr: _Frame):
        """(scale, kind, stencil-or-name, stencil level, field expr) for `[s *] S * F`."""
        if e[0] != "bin" or e[1] != "*" or e[3][0] != "fld":
            return None
        s, scale = e[2], 1.0
        if s[0] == "bin" and s[1] == "*" and s[3][0] == "sten" and self._is_scalar(s[2]):
            scale, s = float(self._eval(s[2], fr)), s[3]
        if s[0] != "sten":
            return None
        if s[1] in self.transfer:
            return scale, self.transfer[s[1]], s[1], None, e[3]
        return scale, "stencil", self.stencil(s[1], self._level_of(s[2], fr)), None, e[3]

    def _residual_form(self, e, fr: _Frame):
        """(F, A, U) for `F - A * U`."""
        if e[0] != "bin" or e[1] != "-" or e[2][0] != "fld":
            return None
        m = self._sten_times_field(e[3], fr)
        if m is None or m[1] != "stencil" or m[0] != 1.0:
            return None
        return e[2], m[2], m[4]

    def _smoother_weight(self, w, A: Stencil, fr: _Frame):
        """(omega, stencil as the kernel needs it): a stencil field carries the FORM of the weight -- the kernels evaluate per point
        what the statement says, `(1.0 / diag(A)) * omega` (Testing/SISC/3D_VarCoeff.exa4:145) or `omega / diag(A)`
        (Testing/PolyExpl/RBGS3Dvc.exa4:52); the two round differently."""
        if A.cfield is None:
            return float(self._eval(w, fr)), A
        if (w[0] == "bin" and w[1] == "*" and w[2][0] == "bin" and w[2][1] == "/" and w[2][2] == ("num", 1.0)
                and w[2][3][0] == "call" and w[2][3][1] == "diag" and self._is_scalar(w[3])):
            return float(self._eval(w[3], fr)), A
        if w[0] == "bin" and w[1] == "/" and w[3][0] == "call" and w[3][1] == "diag" and self._is_scalar(w[2]):
            import dataclasses

            return float(self._eval(w[2], fr)), dataclasses.replace(A, wform=1)
        raise Exa4Unsupported("smoother weight on a stencil field must read ((1.0 / diag(A)) * omega) or (omega / diag(A))")

    def _exec_point_assign(self, st, b, e, colour, fr: _Frame):
        op, lhs, rhs = st[1], st[2], st[3]
        ops = self.ops
        if lhs[0] != "fld":
            raise Exa4Unsupported("loop body assigns to %s" % lhs[0])
        D, ds = self._field(lhs, fr)
        self.launches += 1
        if op == "=":
            if self._is_scalar(rhs):
                return ops.set(D.lc, D.data(ds), float(self._eval(rhs, fr)), b, e)
            if not _contains(rhs, ("fld", "sten", "sentry")):
                fn, par = self._analytic(rhs, D.level)
                if isinstance(fn, int):
                    return ops.fill_fn(D.lc, D.data(ds), self.domain.geom(D.level), fn, par, b, e)
                return ops.fill_expr(D.lc, D.data(ds), self.domain.geom(D.level), fn, b, e)
            if rhs[0] == "fld":
                X, xs = self._field(rhs, fr)
                return ops.axpby(X.lc, X.data(xs), D.lc, D.data(ds), 1.0, 0.0, b, e)
            r = self._residual_form(rhs, fr)
            if r is not None:
                F, fs = self._field(r[0], fr)
                U, us = self._field(r[2], fr)
                return ops.stencil_op(RESIDUAL, U.lc, U.data(us), F.lc, F.data(fs), D.lc, D.data(ds), r[1], 0.0, -1, b, e)
            m = self._sten_times_field(rhs, fr)
            if m is not None:
                X, xs = self._field(m[4], fr)
                if m[1] == "restriction":
                    if X.level != D.level + 1:
                        raise Exa4Unsupported("restriction between levels %d and %d" % (X.level, D.level))
                    return ops.restrict(X.lc, X.data(xs), D.lc, D.data(ds), m[0], b, e)
                if m[1] == "stencil" and m[0] == 1.0:
                    return ops.stencil_op(APPLY, X.lc, X.data(xs), None, None, D.lc, D.data(ds), m[2], 0.0, -1, b, e)
            if rhs[0] == "bin" and rhs[1] == "+" and rhs[2][0] == "fld":
                # U_src + w * (F - A * U_src)   |   G + beta * D
                t = rhs[3]
                if t[0] == "bin" and t[1] == "*":
                    r = self._residual_form(t[3], fr)
                    if r is not None and self._same_access(rhs[2], r[2], fr):
                        return self._smooth(D, ds, rhs[2], t[2], r, b, e, colour, fr)
                    if t[3][0] == "fld" and self._same_access(lhs, t[3], fr) and self._is_scalar(t[2]):
                        X, xs = self._field(rhs[2], fr)
                        return ops.axpby(X.lc, X.data(xs), D.lc, D.data(ds), 1.0, float(self._eval(t[2], fr)), b, e)
        elif op in ("+=", "-="):
            sign = 1.0 if op == "+=" else -1.0
            if rhs[0] == "fld":          # x += y | x -= y  (y + (-1.0) * x is exactly y - x)
                X, xs = self._field(rhs, fr)
                return ops.axpby(X.lc, X.data(xs), D.lc, D.data(ds), sign, 1.0, b, e)
            if rhs[0] == "bin" and rhs[1] == "*":
                if rhs[3][0] == "fld" and self._is_scalar(rhs[2]):
                    X, xs = self._field(rhs[3], fr)
                    return ops.axpby(X.lc, X.data(xs), D.lc, D.data(ds), sign * float(self._eval(rhs[2], fr)), 1.0, b, e)
                r = self._residual_form(rhs[3], fr)
                if r is not None and op == "+=" and self._same_access(lhs, r[2], fr):
                    return self._smooth(D, ds, lhs, rhs[2], r, b, e, colour, fr)
                m = self._sten_times_field(rhs, fr)
                if m is not None and m[1] == "prolongation" and op == "+=" and m[0] == 1.0:
                    X, xs = self._field(m[4], fr)
                    if X.level != D.level - 1:
                        raise Exa4Unsupported("prolongation between levels %d and %d" % (X.level, D.level))
                    return ops.prolong_add(X.lc, X.data(xs), D.lc, D.data(ds), b, e)
        self.launches -= 1
        raise Exa4Unsupported("loop body statement is none of the recognised kernels: %s %s ..." % (lhs[1], op))

    def _smooth(self, D: Field, ds: int, src, w, r, b, e, colour, fr: _Frame):
        F, fs = self._field(r[0], fr)
        U, us = self._field(src, fr)
        A = r[1]
        wv, A = self._smoother_weight(w, A, fr)
        in_place = D is U and ds == us
        if in_place and colour is None:
            raise Exa4Unsupported("in-place smoother update without colouring (lexicographic Gauss-Seidel)")
        if not in_place and colour is not None:
            raise Exa4Unsupported("coloured update into another slot")
        return self.ops.stencil_op(SMOOTH, U.lc, U.data(us), F.lc, F.data(fs), D.lc, D.data(ds), A, wv,
                                   -1 if colour is None else colour, b, e)

    def _exec_reduction(self, f: Field, boxes, reduction, body, fr: _Frame):
        op, var = reduction
        locals_: Dict[str, object] = {}
        for st in body:
            if st[0] == "decl":
                locals_[st[1]] = st[2]
                continue
            if st[0] != "assign" or st[2] != ("id", var, None):
                raise Exa4Unsupported("reduction loop body")
            rhs = st[3]
            if op == "+" and st[1] == "+=" and rhs[0] == "bin" and rhs[1] == "*" and rhs[2][0] == "fld" and rhs[3][0] == "fld":
                X, xs = self._field(rhs[2], fr)
                Y, ys = self._field(rhs[3], fr)
                acc = 0.0
                for b, e in boxes:
                    self.launches += 1
                    t = self.ops.dot(X.lc, X.data(xs), Y.lc, Y.data(ys), b, e)
                    acc += self.ops.scalar_value(self.comm.allreduce(t, "sum"))
                fr.vars[var] = fr.vars[var] + acc
                continue
            if op == "max" and st[1] == "=" and rhs[0] == "call" and rhs[1] == "max" and len(rhs[3]) == 2:
                other = [a for a in rhs[3] if a != ("id", var, None)]
                if len(other) == 1:
                    t = other[0]
                    if t[0] == "id" and t[1] in locals_:
                        t = locals_[t[1]]
                    if t[0] == "call" and t[1] in ("fabs", "abs") and t[3][0][0] == "bin" and t[3][0][1] == "-" and t[3][0][2][0] == "fld":
                        X, xs = self._field(t[3][0][2], fr)
                        fn, par = self._analytic(t[3][0][3], X.level)
                        acc = fr.vars[var]
                        for b, e in boxes:
                            self.launches += 1
                            if isinstance(fn, int):
                                r = self.ops.max_err_fn(X.lc, X.data(xs), self.domain.geom(X.level), fn, par, b, e)
                            else:
                                r = self.ops.max_err_expr(X.lc, X.data(xs), self.domain.geom(X.level), fn, b, e)
                            acc = max(acc, self.ops.scalar_value(self.comm.allreduce(r, "max")))
                        fr.vars[var] = acc
                        continue
            raise Exa4Unsupported("reduction %s over this loop body" % op)

    def _exec_stencil_field_init(self, body, boxes, fr: _Frame):
        """`A:[o] = expr` for every entry of a 7/5-entry stencil field: -div(a grad) with a at the half points."""
        name = body[0][2][1]
        lvl = self._level_of(body[0][2][2], fr)
        A = self._stencil_planes(name, lvl)       # initialisation statements write the planes; loops get the transformed copy
        if A.cfield is None:
            raise Exa4Unsupported("%s is not a stencil field" % name)
        if hasattr(self, "_sf_dirty"):
            self._sf_dirty[(name, lvl)] = True
        nd = self.nd
        want = [(0, 0, 0)]
        for d in range(nd):
            for sgn in (1, -1):
                o = [0, 0, 0]
                o[d] = sgn
                want.append(tuple(o))
        got = {tuple(st[2][3]) + (0,) * (3 - len(st[2][3])): st[3] for st in body}
        offs = [tuple(o) for o in A.offsets]
        if set(got) != set(offs) or len(got) != len(body) or any(st[1] != "=" for st in body):
            raise Exa4Unsupported("stencil field initialisation: one assignment per declared entry")
        try:
            if offs != want:
                raise Exa4Unsupported("not the 5/7-entry form of examg_init_varcoeff7")
            fn, par = self._varcoeff_function(got, want, lvl)
        except Exa4Unsupported:
            want = offs
            # any other discretisation / coefficient function: every entry is a point expression of its own, filled into
            # its coefficient plane by the expression kernel (one launch per entry)
            size = A.clayout.size
            for k, o in enumerate(want):
                prog, _ = self._analytic(got[o], lvl)
                plane = A.cfield[k * size:(k + 1) * size]
                for b, e in boxes:
                    self.launches += 1
                    if isinstance(prog, int):
                        self.ops.fill_fn(A.clayout.c_struct(), plane, self.domain.geom(lvl), prog, _, b, e)
                    else:
                        self.ops.fill_expr(A.clayout.c_struct(), plane, self.domain.geom(lvl), prog, b, e)
            return
        for b, e in boxes:
            self.launches += 1
            self.ops.init_varcoeff7(A.clayout.c_struct(), A.cfield, self.domain.geom(lvl), fn, par, b, e)

    def _varcoeff_function(self, got, want, lvl: int):
        """Coefficient function id if the entries are -div(a grad) with a built-in `a` at the half points, in the exact
        form of examg_init_varcoeff7."""
        nd = self.nd
        calls = [c for c in _find_calls(got[want[1]]) if c[1] in self.functions]
        if not calls:
            raise Exa4Unsupported("stencil field initialisation without a coefficient function")
        cfn = self.functions[calls[0][1]][0]
        coef_expr = ("call", cfn.name, None, [("id", "vf_nodePosition_" + "xyz"[i], None) for i in range(len(cfn.params))])
        fn, par = self._recognise(coef_expr, lvl)
        h = self.domain.h(lvl)
        for _ in range(6):      # check the seven expressions against the kernel's formula at random points
            x, y, z = (self._rng.uniform(0.1, 0.9) for _ in range(3))
            a = lambda dx, dy, dz: fn_eval(fn, par, x + dx, y + dy, z + dz)
            ref = {}
            diag = None
            for d in range(nd):
                off = [0.0, 0.0, 0.0]
                off[d] = 0.5 * h[d]
                ap, am = a(*off), a(*[-v for v in off])
                term = (ap + am) / (h[d] * h[d])
                diag = term if diag is None else diag + term
                ref[want[1 + 2 * d]] = (-1.0 * ap) / (h[d] * h[d])
                ref[want[2 + 2 * d]] = (-1.0 * am) / (h[d] * h[d])
            ref[want[0]] = diag
            for o in want:
                v = self._point_eval(got[o], lvl, x, y, z)
                if abs(v - ref[o]) > 1e-11 * max(1.0, abs(ref[o])):
                    raise Exa4Unsupported("stencil field entry %r is not -a(x +- h/2)/h^2" % (o,))
        return fn, par


# =====================================================================================================================
def load(exa4_path: str, knowledge_path: Optional[str] = None, **kw) -> Exa4Program:
    with open(exa4_path) as f:
        text = f.read()
    k = _knowledge.parse_file(knowledge_path) if knowledge_path else {}
    return Exa4Program(text, k, **kw)


def main(argv=None):
    import argparse

    ap = argparse.ArgumentParser(description="run an ExaSlang-4 multigrid program on libexamg (MI355X)")
    ap.add_argument("exa4")
    ap.add_argument("knowledge", nargs="?")
    ap.add_argument("--set", action="append", default=[], metavar="key=value", help="override a knowledge flag")
    args = ap.parse_args(argv)
    k = _knowledge.parse_file(args.knowledge) if args.knowledge else {}
    for kv in args.set:
        _knowledge.parse_text(kv, None, k)
    with open(args.exa4) as f:
        prog = Exa4Program(f.read(), k, echo=True)
    prog.json_dir = os.getcwd()
    prog.run()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
