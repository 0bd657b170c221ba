"""Built-ins of the ExaSlang-4 interpreter (mixin of exa4.Exa4Program): print / timers / printJSON and friends, field I/O
(printField, writeField, readField: SURVEY.md 8 f-4), and the host-side loop kinds of the reference's test programs -- std::rand() fills
(Testing/Opts), point-by-point compare loops (Testing/IOTest) and value checks (Testing/PolyExpl)."""
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


class Builtins:
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
            worst = max(worst, self.comm.reduce_value(t, "max"))
        if not (worst > tol if op == ">" else worst >= tol):
            return
        # a point beyond the tolerance exists SOMEWHERE (the verdict is all-reduced, so that every block leaves the function together
        # when the `if` returns): the first one in loop order (x fastest) on THIS block, if it has one, for the program's messages
        self._cmp_point = None
        ha = A.host_array(self.ops, sa)
        hb = B.host_array(self.ops, sb)
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
                env["_fields"][key] = f.host_array(self.ops, slot)[sl]
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
        here = getattr(self, "_cmp_point", None)
        vals, va, vb = here if here is not None else ({"i0": -1, "i1": -1, "i2": -1}, float("nan"), float("nan"))
        for st in stmts:
            if st[0] == "return":
                raise _Return(self._eval(st[1], fr) if st[1] is not None else None)
            c = st[1]
            if c[1] != "print":
                self._exec(st, fr)
                continue
            if here is None:
                continue        # the offending point lies on another block: that process prints it (the reference prints where it finds it)
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
