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

from .exa4_common import (APPLY, RESIDUAL, SMOOTH, _FN_2D_ONLY, _FN_ANY_DIM, _FN_WITH_PARAM, _N_FN, _Frame, _Return, fn_eval)  # noqa: E402,F401
from .exa4_builtins import Builtins  # noqa: E402
from .exa4_fusion import LazyFusions  # noqa: E402
from .exa4_peepholes import Peepholes  # noqa: E402


class Exa4Program(LazyFusions, Peepholes, Builtins):
    """One ExaSlang-4 program bound to a kernel layer (`ops`: HipOps on the GPU), a block decomposition and a
    communicator.  `run()` executes `Function Application`; printed lines are collected in `self.out`."""

    def __init__(self, text: str, knowledge: Optional[Dict] = None, ops=None, domain: Optional[RectDomain] = None, comm=None,
                 echo: bool = False, fuse: bool = True, fuse_coarse_solver: Optional[bool] = None, auto_graph: Optional[bool] = None):
        """fuse: run red-black sweeps and pairs of slotted Jacobi steps as single passes over HBM where the program's
        statements allow it (bit-identical results; `fuse=False` issues exactly one launch per loop statement).
        auto_graph: a leveled function without parameters whose statements never return to the host (no reductions, prints or
        builtins anywhere below it: typically the cycle function) is recorded into a hipGraph at its second call and replayed from
        then on -- the launch sequence of such a call is fixed once the peepholes and fusions have run, so the interpreter's host
        work is paid once (default: on with `fuse` on the HIP kernel layer, one block)."""
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
        # shortest row the one-pass forms are used on: the kernel layer has them for every row length (two-stage kernel from 64 points,
        # csrc/kernels_small.hip below), blocks with neighbours keep 64 (interior + shell split)
        self.fuse_min_row = 3
        self.fuse_min_row_blocks = 64
        on_gpu = hasattr(ops, "torch") and getattr(getattr(ops, "device", None), "type", "cpu") != "cpu"
        self.auto_graph = bool((fuse if auto_graph is None else auto_graph) and on_gpu and domain.world_size == 1 and not any(domain.periodic))
        self._auto_graphs: Dict[Tuple[str, int], object] = {}     # (function, level) -> recorded graph | False (not capturable)
        self._auto_calls: Dict[Tuple[str, int], int] = {}
        self._host_free: Dict[Tuple[str, int], bool] = {}
        self._graph_depth = 0
        self.graph_replays = 0
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
        (EXAMG_CLAYOUT_ENTRY_FASTEST), one stream instead of one per entry.  `transform <field> with [x, y, z] => [x / 2, y, z, x % 2]`
        (the colour split of Testing/LayoutTrafo/rbgs.exa4:2; 2-D: `[x, y] => [x / 2, y, x % 2]`) on a scalar field -- APPLIED on the
        HIP kernel layer: the field's arrays are allocated under EXAMG_LAYOUT_SPLIT_X and every loop reaches it through the
        transformed index (the points of one red-black colour of a row are contiguous: 32 B per update for a half sweep instead
        of 48); one-pass peepholes keep to plain fields.  The other directives of the reference's test programs (axis permutations,
        concat / rename of scalar fields) change no value either and stay recorded only."""
        import re

        self._sf_entry_fastest, self._sf_rec, self._sf_dirty = set(), {}, {}
        nd = self.nd
        src = ",".join("xyz"[:nd]) + ",i"
        dst = "i," + ",".join("xyz"[:nd])
        coef_of = {}
        for sf in self.ast.sfields:
            coef_of.setdefault(sf.field, []).append(sf)
        axes = "xyz"[:nd]
        split_src = ",".join(axes)
        split_dst = ",".join(["x/2"] + list(axes[1:]) + ["x%2"])
        self._split_fields = set()
        for text in getattr(self.ast, "layout_transformations", []):
            m = re.match(r"^transform (.+?) with \[(.+?)\] => \[(.+?)\]$", text.strip())
            if m and m.group(2).replace(" ", "") == split_src and m.group(3).replace(" ", "") == split_dst:
                self._apply_colour_split(m.group(1))
                continue
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

    def _transform_items(self, text: str):
        """(field name, levels) of the items of a `transform a@(..), b with ..` directive (items are separated by commas outside level lists)."""
        items, depth, cur = [], 0, []
        for tok in text.split(" "):
            depth += tok == "("
            depth -= tok == ")"
            if tok == "," and depth == 0:
                items.append(cur)
                cur = []
            else:
                cur.append(tok)
        items.append(cur)
        for it in items:
            if it:
                yield it[0], self.levels_of(self._parse_level_text(" ".join(it[2:]) if len(it) > 2 and it[1] == "@" else "all"))

    def _apply_colour_split(self, items: str):
        """The named scalar fields under `[x, y, z] => [x / 2, y, z, x % 2]`: their (still zero) arrays are re-allocated in the split
        layout.  Only where the kernel layer has transformed layouts (HipOps: examg_transform_field); the CPU stand-in of the tests
        keeps the directive recorded."""
        if not hasattr(self.ops, "transform_field"):
            return
        for fname, lvls in self._transform_items(items):
            for lvl in lvls:
                f = self.fields.get((fname, lvl))
                if f is None or f.layout.transform:
                    continue
                f.layout = f.layout.split_x()
                f.lc = f.layout.c_struct()
                f.slots = [self.ops.new_array(f.layout.size) for _ in range(f.num_slots)]
                self._split_fields.add((fname, lvl))

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
        if self.auto_graph and self._graph_depth == 0 and fn.levels is not None and not fn.params and not args and \
                self._is_host_free(name, lvl) and self._call_through_graph(name, lvl):
            return None
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

    # -- functions that never return to the host, replayed from a hipGraph (auto_graph) -----------------------------------------------
    def _is_host_free(self, name: str, lvl: int) -> bool:
        """Can a call of this leveled function be issued without ever looking at a device value or doing host-visible work?  Decided
        from the program text: loops without reductions, `communicate` / `apply bc` / `advance`, repeats with literal counts, colour
        blocks, conditions and scalar statements without calls or field accesses, calls of functions that qualify themselves -- and
        the coarsest-level function when it is the generated CG solver (one kernel)."""
        key = (name, lvl)
        if key in self._host_free:
            return self._host_free[key]
        self._host_free[key] = False          # recursion at the same level: no
        try:
            fn = self._resolve(name, lvl)
            ok = fn.levels is not None and not fn.params
            if ok and self.fuse_coarse_solver and lvl == self.min_level and self._coarse_cg_plan(fn, lvl) is not None:
                ok = True
            elif ok:
                ok = self._host_free_body(fn.body, lvl)
        except (Exa4SyntaxError, Exa4Unsupported, KeyError, IndexError, TypeError):
            ok = False
        self._host_free[key] = ok
        return ok

    def _host_free_body(self, stmts, lvl: int) -> bool:
        def pure(e) -> bool:      # an expression without calls and field accesses
            if isinstance(e, (list, tuple)):
                if len(e) >= 1 and e[0] in ("call", "fld") and len(e) >= 4:
                    return False
                return all(pure(x) for x in e)
            return True

        for s in stmts:
            k = s[0]
            if k == "loop":
                if s[4] is not None:
                    return False
            elif k in ("comm", "applybc", "advance"):
                continue
            elif k == "repeat":
                if s[1][0] != "num" or s[2] is not None or not self._host_free_body(s[3], lvl):
                    return False
            elif k == "color":
                if not self._host_free_body(s[2], lvl):
                    return False
            elif k == "levelscope":
                if lvl in self.levels_of(s[1], lvl) and not self._host_free_body(s[2], lvl):
                    return False
            elif k == "if":
                if not pure(s[1]) or not self._host_free_body(s[2], lvl) or not self._host_free_body(s[3] or [], lvl):
                    return False
            elif k in ("decl", "assign"):
                e = s[2] if k == "decl" else s[3]
                if e is not None and not pure(e):
                    return False
            elif k == "callstmt":
                c = s[1]
                if c[0] != "call" or c[1] not in self.functions or c[3]:
                    return False
                clvl = self._level_of(c[2], _Frame(lvl, {})) if c[2] is not None else lvl
                if not self._is_host_free(c[1], clvl):
                    return False
            else:
                return False
        return True

    def _host_state(self):
        """What executing statements changes on the HOST side (boundary bookkeeping, second arrays, slots): saved before a function is
        recorded, put back if the recording is discarded (its launches were never issued)."""
        return {"bc_valid": set(self._bc_valid), "bc_epoch": dict(self._bc_epoch), "alt": dict(self._alt), "alt_shell": dict(self._alt_shell),
                "pair_tmp": dict(self._pair_tmp), "sf_dirty": dict(self._sf_dirty), "sf_rec": dict(self._sf_rec),
                "fields": {k: (list(f.slots), f.current_slot) for k, f in self.fields.items()},
                "cg": (getattr(self, "_cg_tail", None), getattr(self, "_cg_info", None))}

    def _restore_host_state(self, st):
        self._bc_valid, self._bc_epoch, self._alt, self._alt_shell = st["bc_valid"], st["bc_epoch"], st["alt"], st["alt_shell"]
        self._pair_tmp, self._sf_dirty, self._sf_rec = st["pair_tmp"], st["sf_dirty"], st["sf_rec"]
        for k, (slots, cur) in st["fields"].items():
            self.fields[k].slots, self.fields[k].current_slot = slots, cur
        for attr, v in zip(("_cg_tail", "_cg_info"), st["cg"]):
            if v is None and hasattr(self, attr):
                delattr(self, attr)
            elif v is not None:
                setattr(self, attr, v)

    def _call_through_graph(self, name: str, lvl: int) -> bool:
        """True if the call was issued as a graph replay.  First call: interpreted (lazily created arrays come into being); second
        call: recorded -- the interpreter runs the function under a stream capture, nothing executes -- and replayed once;
        afterwards: replayed, as long as every array is in the role it had when the graph was recorded."""
        torch = self.ops.torch
        key = (name, lvl)
        rec = self._auto_graphs.get(key)
        if rec is False or torch.cuda.is_current_stream_capturing():
            return False
        if rec is None:
            n = self._auto_calls[key] = self._auto_calls.get(key, 0) + 1
            if n < 2:
                return False
            self._flush_pending()
            before, l0, f0 = self._roles(), self.launches, dict(self.fusions)
            saved = self._host_state()
            g = torch.cuda.CUDAGraph()
            self._graph_depth += 1
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    fn = self._resolve(name, lvl)
                    self._exec_block(fn.body, _Frame(lvl, {}), fn=True)
                    self._flush_pending()     # a loop the function leaves pending for its caller's next statement: issued here
                ok = self._roles() == before
            except (RuntimeError, Exa4Unsupported, _Return):
                ok = False
            finally:
                self._graph_depth -= 1
            recorded, fused = self.launches - l0, {k: v - f0.get(k, 0) for k, v in self.fusions.items()}
            self.launches, self.fusions = l0, f0
            if not ok or recorded == 0:
                # nothing of the recording was issued: the interpreter's bookkeeping goes back to where it was, the real call follows
                self._pending = None
                self._restore_host_state(saved)
                self._auto_graphs[key] = False
                return False
            rec = self._auto_graphs[key] = {"graph": g, "roles": before, "launches": recorded, "fusions": fused}
        if self._roles() != rec["roles"]:
            return False                      # arrays swapped roles since (an odd number of out-of-place sweeps elsewhere): interpret
        self._flush_pending()
        rec["graph"].replay()
        self.launches += rec["launches"]
        for k, v in rec["fusions"].items():
            self.fusions[k] = self.fusions.get(k, 0) + v
        self.graph_replays += 1
        return True

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
        self.comm.check()         # a wait of the peer-write transport that gave up is an error of the run, not silent garbage
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
                    acc += self.comm.reduce_value(t, "sum")
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
                            acc = max(acc, self.comm.reduce_value(r, "max"))
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
