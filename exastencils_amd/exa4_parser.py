"""ExaSlang-4 subset: lexer, declarations, recursive-descent parser and the small AST helpers of the interpreter
(exastencils_amd/exa4.py).  Grammar reference: Compiler/src/exastencils/parsers/l4/L4_Parser.scala (see exa4.py for the
line ranges of the constructs covered).

AST: tuples.  Expressions: ("num", v) ("str", s) ("id", name, level) ("fld", name, slot, level) ("sten", name, level)
("sentry", name, level, offset) ("call", name, level, args) ("bin", op, a, b) ("neg", a) ("not", a).  Statements: ("decl", ..)
("assign", op, lhs, rhs) ("loop", target, only, where, reduction, body) ("comm", phase, what, target) ("applybc", target)
("advance", target) ("repeat", n, counter, body) ("until", cond, body) ("if", cond, then, else) ("color", exprs, body)
("levelscope", levels, body) ("return", expr) ("callstmt", call).  Level specifications: ("single", base, delta) ("all",)
("range", a, b) ("list", items) ("but", spec, spec)."""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field as _dcf
from typing import Dict, List, Optional, Tuple

class Exa4SyntaxError(ValueError):
    pass


class Exa4Unsupported(NotImplementedError):
    pass


# =====================================================================================================================
# lexer
# =====================================================================================================================
_TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?)
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<str>"[^"\n]*"|'[^'\n]*')
  | (?P<op>=>|\*\*|\+=|-=|\*=|/=|==|!=|<=|>=|&&|\|\||[@()\[\]{}<>,:=+\-*/%!.])
""", re.X | re.S)


@dataclass
class Tok:
    kind: str
    text: str
    line: int


def tokenize(text: str) -> List[Tok]:
    out, pos, line = [], 0, 1
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise Exa4SyntaxError("line %d: cannot read %r" % (line, text[pos:pos + 20]))
        kind = m.lastgroup
        if kind != "ws":
            out.append(Tok(kind, m.group(), line))
        line += m.group().count("\n")
        pos = m.end()
    out.append(Tok("eof", "", line))
    return out


# =====================================================================================================================
# declarations
# =====================================================================================================================
@dataclass
class LayoutDecl:
    name: str
    datatype: str
    vec_len: int
    localization: str
    levels: object
    ghost: Tuple[int, ...] = ()
    dup: Tuple[int, ...] = ()
    ghost_comm: bool = False
    dup_comm: bool = False
    inner: Tuple[int, ...] = ()


@dataclass
class FieldDecl:
    name: str
    domain: str
    layout: str
    bc: object          # None | expression
    slots: int
    levels: object


@dataclass
class StencilDecl:
    name: str
    levels: object
    entries: List[Tuple[Tuple[int, ...], object]] = _dcf(default_factory=list)   # (offset, coefficient expression)
    transfer: Optional[str] = None      # 'restriction' | 'prolongation'


@dataclass
class StencilFieldDecl:
    name: str
    field: str
    stencil: str
    levels: object


@dataclass
class FunctionDecl:
    name: str
    levels: object
    params: List[str]
    body: list


_STMT_WORDS = {"loop", "communicate", "apply", "advance", "repeat", "if", "Var", "Val", "Variable", "Value", "color", "return",
               "begin", "finish", "print"}
_SLOT_WORDS = {"active", "activeSlot", "current", "currentSlot", "next", "nextSlot", "previous", "previousSlot"}
_LEVEL_WORDS = {"current", "coarser", "finer", "finest", "coarsest", "all"}
_MATH = {"sqrt": math.sqrt, "fabs": abs, "abs": abs, "sin": math.sin, "cos": math.cos, "tan": math.tan, "exp": math.exp,
         "sinh": math.sinh, "cosh": math.cosh, "tanh": math.tanh, "log": math.log, "pow": math.pow, "max": max, "min": min,
         "floor": math.floor, "ceil": math.ceil, "ldexp": lambda x, e: math.ldexp(float(x), int(e)), "fmod": math.fmod,
         "asin": math.asin, "acos": math.acos, "atan": math.atan, "atan2": math.atan2, "log10": math.log10}
_COORD = re.compile(r"^vf_(nodePosition|nodePos|boundaryCoord|boundaryPosition|boundaryPos)_([xyz])$")
_GRIDW = re.compile(r"^vf_gridWidth_([xyz])$")


# =====================================================================================================================
# parser
# =====================================================================================================================
class Parser:
    def __init__(self, text: str):
        self.toks = tokenize(text)
        self.p = 0
        self.field_names = set(re.findall(r"^\s*Field\s+([A-Za-z_]\w*)", text, re.M))
        self.stencil_names = set(re.findall(r"^\s*Stencil\s+([A-Za-z_]\w*)", text, re.M))
        self.sfield_names = set(re.findall(r"^\s*StencilField\s+([A-Za-z_]\w*)", text, re.M))
        self.domain = None
        self.layouts: List[LayoutDecl] = []
        self.fields: List[FieldDecl] = []
        self.stencils: List[StencilDecl] = []
        self.sfields: List[StencilFieldDecl] = []
        self.globals: List[Tuple[str, object]] = []
        self.functions: List[FunctionDecl] = []

    # -- token helpers --------------------------------------------------------------------------------------------
    def peek(self, k: int = 0) -> Tok:
        return self.toks[min(self.p + k, len(self.toks) - 1)]

    def next(self) -> Tok:
        t = self.toks[self.p]
        self.p += 1
        return t

    def at(self, text: str, k: int = 0) -> bool:
        return self.peek(k).text == text and self.peek(k).kind != "str"

    def accept(self, text: str) -> bool:
        if self.at(text):
            self.p += 1
            return True
        return False

    def expect(self, text: str) -> Tok:
        if not self.at(text):
            t = self.peek()
            raise Exa4SyntaxError("line %d: expected %r, found %r" % (t.line, text, t.text))
        return self.next()

    def ident(self) -> str:
        t = self.next()
        if t.kind != "id":
            raise Exa4SyntaxError("line %d: expected a name, found %r" % (t.line, t.text))
        return t.text

    # -- program ----------------------------------------------------------------------------------------------------
    def parse(self) -> "Parser":
        while self.peek().kind != "eof":
            t = self.peek()
            if t.text == "Domain":
                self._domain()
            elif t.text == "Layout":
                self._layout()
            elif t.text == "Field":
                self._field()
            elif t.text == "Stencil":
                self._stencil()
            elif t.text == "StencilField":
                self._stencil_field()
            elif t.text == "Globals":
                self._globals()
            elif t.text == "LayoutTransformations":
                self._layout_transformations()
            elif t.text in ("Function", "Func", "Def", "noinline"):
                self._function()
            else:
                raise Exa4SyntaxError("line %d: unexpected %r at top level" % (t.line, t.text))
        return self

    def _bad_comm(self):
        t = self.peek()
        raise Exa4SyntaxError("line %d: communicate: expected all / dup / ghost, found %r" % (t.line, t.text))

    def _layout_transformations(self):
        """`LayoutTransformations { transform F@lvls with [x, y, z] => [...] | concat @lvls A, B into M | rename F@lvl to N }`
        (layoutTransformation/l4/L4_LayoutSection.scala; Testing/LayoutTrafo/rbgs.exa4:1-6): directives about where the values of
        a field live in memory -- they change no value a program computes or prints (the reference checks these programs against
        the .results files of the untransformed ones, .gitlab-ci.yml:760-767).  Recorded, not applied: fields stay in the
        reference layout, which is the library's contract with its callers."""
        self.expect("LayoutTransformations")
        self.expect("{")
        depth, cur, line = 1, [], self.peek().line
        self.layout_transformations = getattr(self, "layout_transformations", [])
        while depth:
            t = self.next()
            if t.kind == "eof":
                raise Exa4SyntaxError("line %d: LayoutTransformations block is not closed" % line)
            if t.text == "{":
                depth += 1
            elif t.text == "}":
                depth -= 1
                if depth == 0:
                    break
            if t.text in ("transform", "concat", "rename") and depth == 1 and cur:
                self.layout_transformations.append(" ".join(cur))
                cur = []
            cur.append(t.text)
        if cur:
            self.layout_transformations.append(" ".join(cur))

    def _const_list(self) -> list:
        self.expect("[")
        out = [self.expr(no_rel=True)]
        while self.accept(","):
            out.append(self.expr(no_rel=True))
        self.expect("]")
        return out

    def _domain(self):
        self.expect("Domain")
        name = self.ident()
        self.expect("<")
        lo = self._const_list()
        self.expect("to")
        hi = self._const_list()
        self.expect(">")
        self.domain = (name, lo, hi)

    def _layout(self):
        self.expect("Layout")
        name = self.ident()
        self.expect("<")
        dt, vec = self.ident(), 1
        if self.accept("<"):        # ColumnVector<Real,7>
            self.ident()
            self.expect(",")
            vec = int(self.next().text)
            self.expect(">")
        self.expect(",")
        loc = self.ident()
        self.expect(">")
        levels = self.decl_levels()
        d = LayoutDecl(name, dt, vec, loc, levels)
        self.expect("{")
        while not self.accept("}"):
            key = self.ident()
            self.expect("=")
            vals = tuple(int(_const_value(e)) for e in self._const_list())
            comm = False
            if self.accept("with"):
                self.expect("communication")
                comm = True
            if key == "ghostLayers":
                d.ghost, d.ghost_comm = vals, comm
            elif key == "duplicateLayers":
                d.dup, d.dup_comm = vals, comm
            elif key == "innerPoints":
                d.inner = vals          # explicit inner extent (Testing/PolyExpl/Jac3Dcc.exa4:2-5); must agree with the level
            else:
                raise Exa4Unsupported("layout option %r" % key)
        self.layouts.append(d)

    def _field(self):
        self.expect("Field")
        name = self.ident()
        self.expect("<")
        dom = self.ident()
        self.expect(",")
        lay = self.ident()
        self.expect(",")
        bc = None
        if self.at("None"):
            self.next()
        else:
            bc = self.expr(no_rel=True)
        self.expect(">")
        slots = 1
        if self.accept("["):
            slots = int(self.next().text)
            self.expect("]")
        self.fields.append(FieldDecl(name, dom, lay, bc, slots, self.decl_levels()))

    def _stencil(self):
        self.expect("Stencil")
        name = self.ident()
        if self.accept("from"):
            self.expect("default")
            kind = self.ident()
            self.expect("on")
            loc = self.ident()
            self.expect("with")
            interp = self.next().text.strip("\"'")
            if kind not in ("restriction", "prolongation") or loc != "Node" or interp != "linear":
                raise Exa4Unsupported("default %s on %s with %r" % (kind, loc, interp))
            self.stencils.append(StencilDecl(name, None, [], kind))
            return
        levels = self.decl_levels()
        d = StencilDecl(name, levels)
        self.expect("{")
        mapped = []
        while not self.accept("}"):
            lhs = self._const_list()
            if self.accept("=>"):
                off = tuple(int(_const_value(e)) for e in lhs)
                d.entries.append((off, self.expr()))
            else:
                self.expect("from")
                src = self._const_list()
                self.expect("with")
                mapped.append((src, self.expr()))
            self.accept(",")
        if mapped:
            d.transfer = _classify_transfer(mapped)
        self.stencils.append(d)

    def _stencil_field(self):
        self.expect("StencilField")
        name = self.ident()
        self.expect("<")
        f = self.ident()
        self.expect("=>")
        s = self.ident()
        self.expect(">")
        self.sfields.append(StencilFieldDecl(name, f, s, self.decl_levels()))

    def _globals(self):
        self.expect("Globals")
        self.expect("{")
        while not self.accept("}"):
            self.next()                    # Var | Val
            name = self.ident()
            self.expect(":")
            self._datatype()
            val = self.expr() if self.accept("=") else ("num", 0.0)
            self.globals.append((name, val))

    def _datatype(self):
        self.ident()
        if self.at("<") and self.peek(1).kind == "id":    # Vector<Real, 3>
            depth = 0
            while True:
                t = self.next()
                depth += t.text == "<"
                depth -= t.text == ">"
                if depth == 0:
                    break

    def _function(self):
        self.accept("noinline")
        self.next()
        name = self.ident()
        levels = self.decl_levels()
        params = []
        if self.accept("("):
            while not self.accept(")"):
                params.append(self.ident())
                self.expect(":")
                self._datatype()
                self.accept(",")
        if self.accept(":"):
            self._datatype()
        self.functions.append(FunctionDecl(name, levels, params, self.block()))

    # -- levels -----------------------------------------------------------------------------------------------------
    def decl_levels(self):
        if not self.at("@"):
            return None
        self.next()
        return self._level_item()

    def _level_item(self):
        if self.accept("("):
            spec = self._level_list()
            self.expect(")")
            return spec
        return self._level_atom()

    def _level_atom(self):
        t = self.next()
        if t.kind == "num":
            return ("single", int(t.text), 0)
        if t.text == "all":
            return ("all",)
        if t.text in _LEVEL_WORDS:
            return ("single", t.text, 0)
        raise Exa4SyntaxError("line %d: %r is not a level" % (t.line, t.text))

    def _level_single(self):
        if self.accept("("):
            spec = self._level_list()
            self.expect(")")
        else:
            spec = self._level_atom()
        while self.at("+") or self.at("-"):
            sign = 1 if self.next().text == "+" else -1
            delta = sign * int(self.next().text)
            if spec[0] != "single":
                raise Exa4SyntaxError("level arithmetic on a level list")
            spec = ("single", spec[1], spec[2] + delta)
        return spec

    def _level_list(self):
        items = [self._level_range()]
        while self.at(",") or self.at("and"):
            self.next()
            items.append(self._level_range())
        spec = items[0] if len(items) == 1 else ("list", items)
        if self.accept("but"):
            spec = ("but", spec, self._level_list())
        return spec

    def _level_range(self):
        a = self._level_single()
        if self.accept("to"):
            return ("range", a, self._level_single())
        return a

    # -- statements -------------------------------------------------------------------------------------------------
    def block(self) -> list:
        self.expect("{")
        out = []
        while not self.accept("}"):
            out.append(self.stmt())
        return out

    def stmt(self):
        t = self.peek()
        w = t.text
        if t.kind == "id":
            if w in ("Var", "Val", "Variable", "Value"):
                self.next()
                name = self.ident()
                self.expect(":")
                self._datatype()
                init = self.expr() if self.accept("=") else None
                return ("decl", name, init)
            if w == "loop":
                return self._loop()
            if w in ("communicate", "begin", "finish"):
                phase = "sync"
                if w != "communicate":
                    phase = self.next().text
                self.expect("communicate")
                what = "all"
                if (self.at("ghost") or self.at("dup") or self.at("all")) and self.at("of", 1):
                    what = self.next().text
                    self.next()
                elif self.at("ghost") or self.at("dup") or self.at("all"):
                    # `communicate dup ghost [0, 0, 0] of F` (Testing/Misc/inlining.exa4:157; communication/l4/L4_Communicate.scala):
                    # a list of layer kinds, each with an optional index range `[b] [to [e]]` of the layers meant.  The kinds are
                    # honoured; an index range is widened to all layers of its kind -- more layers exchanged, the same values in them
                    kinds = set()
                    while not self.at("of"):
                        kinds.add(self.expect(self.peek().text).text if self.peek().text in ("ghost", "dup", "all") else self._bad_comm())
                        if self.at("["):
                            self._const_list()
                            if self.accept("to"):
                                self._const_list()
                    self.next()
                    what = "all" if ("all" in kinds or kinds >= {"dup", "ghost"}) else kinds.pop()
                target = self.postfix()
                if self.accept("where"):      # conditional exchange (Testing/Smoothers/RBGS.exa4:126): same values, full exchange
                    self.expr()
                return ("comm", phase, what, target)
            if w == "apply":
                self.next()
                self.expect("bc")
                self.expect("to")
                return ("applybc", self.postfix())
            if w == "advance":
                self.next()
                return ("advance", self.postfix())
            if w == "repeat":
                self.next()
                if self.accept("until"):
                    cond = self.expr()
                    return ("until", cond, self.block())
                n = self.expr()
                self.expect("times")
                counter = None
                if self.accept("count"):
                    counter = self.ident()
                if self.accept("with"):
                    # `repeat n times with contraction [px, py, pz] [, [nx, ny, nz]] { .. }` (parsers/l4/L4_Parser.scala: contractionLoop;
                    # baseExt/ir/IR_ContractingLoop.scala): positive extents, negative ones default to the same
                    self.expect("contraction")
                    pos = tuple(int(_const_value(e)) for e in self._const_list())
                    neg = pos
                    if self.accept(","):
                        neg = tuple(int(_const_value(e)) for e in self._const_list())
                    return ("contract", n, counter, pos, neg, self.block())
                return ("repeat", n, counter, self.block())
            if w == "if":
                self.next()
                cond = self.expr()
                then = self.block()
                other = []
                if self.accept("else"):
                    other = [self.stmt()] if self.at("if") else self.block()
                return ("if", cond, then, other)
            if w == "color":
                self.next()
                self.expect("with")
                self.expect("{")
                colours = []
                while True:
                    colours.append(self.expr())
                    self.expect(",")
                    if self.peek().text in _STMT_WORDS:
                        break
                body = []
                while not self.accept("}"):
                    body.append(self.stmt())
                return ("color", colours, body)
            if w == "return":
                line = self.next().line
                if self.peek().line == line and not self.at("}"):
                    return ("return", self.expr())
                return ("return", None)
        if w == "@":
            self.next()
            spec = self._level_item()
            return ("levelscope", spec, self.block())
        lhs = self.postfix()
        for op in ("=", "+=", "-=", "*=", "/="):
            if self.at(op):
                self.next()
                return ("assign", op, lhs, self.expr())
        if lhs[0] == "call":
            return ("callstmt", lhs)
        raise Exa4SyntaxError("line %d: statement starting with %r not understood" % (t.line, w))

    def _loop(self):
        self.expect("loop")
        self.expect("over")
        if self.at("fragments"):          # `loop over fragments { ... }`: one fragment per process here
            self.next()
            if self.accept("with"):       # `with reduction ( + : res )`: the loop over the field inside carries the same clause and
                self.expect("reduction")  # does the reduction (and the all-reduce across blocks); one fragment per process
                self.expect("(")
                while not self.accept(")"):
                    self.next()
            return ("if", ("num", True), self.block(), [])
        target = self.postfix()
        only = None
        if self.accept("only"):
            region = self.ident()
            direction = tuple(int(_const_value(e)) for e in self._const_list())
            on_boundary = False
            if self.accept("on"):
                self.expect("boundary")
                on_boundary = True
            only = (region, direction, on_boundary)
        # `sequentially` (baseExt/l4/L4_LoopOverField.scala: no OpenMP/SIMD for this loop) concerns the reference's CPU code
        # generation only; the statements of a `loop over` are independent per point either way
        self.accept("sequentially")
        if self.at("starting") or self.at("ending") or self.at("stepping"):
            raise Exa4Unsupported("line %d: loop modifier %r" % (self.peek().line, self.peek().text))
        where = self.expr() if self.accept("where") else None
        reduction = None
        if self.accept("with"):
            self.expect("reduction")
            self.expect("(")
            op = self.next().text
            self.expect(":")
            reduction = (op, self.ident())
            self.expect(")")
        return ("loop", target, only, where, reduction, self.block())

    # -- expressions ------------------------------------------------------------------------------------------------
    def expr(self, no_rel: bool = False):
        return self._or(no_rel)

    def _or(self, nr):
        a = self._and(nr)
        while self.at("||"):
            self.next()
            a = ("bin", "||", a, self._and(nr))
        return a

    def _and(self, nr):
        a = self._cmp(nr)
        while self.at("&&"):
            self.next()
            a = ("bin", "&&", a, self._cmp(nr))
        return a

    def _cmp(self, nr):
        a = self._add()
        while not nr and self.peek().text in ("==", "!=", "<", "<=", ">", ">=") and self.peek().kind == "op":
            op = self.next().text
            a = ("bin", op, a, self._add())
        return a

    def _add(self):
        a = self._mul()
        while (self.at("+") or self.at("-")):
            op = self.next().text
            a = ("bin", op, a, self._mul())
        return a

    def _mul(self):
        a = self._unary()
        while self.at("*") or self.at("/") or self.at("%"):
            op = self.next().text
            a = ("bin", op, a, self._unary())
        return a

    def _unary(self):
        if self.at("-"):
            self.next()
            if self.peek().kind == "num" and not self.at("**", 1):
                t = self.next()
                return _num(t.text, -1)
            return ("neg", self._unary())
        if self.at("+"):
            self.next()
            return self._unary()
        if self.at("!"):
            self.next()
            return ("not", self._unary())
        return self._pow()

    def _pow(self):
        a = self.postfix()
        if self.at("**"):
            self.next()
            return ("bin", "**", a, self._unary())
        return a

    def postfix(self):
        t = self.next()
        if t.kind == "num":
            return _num(t.text, 1)
        if t.kind == "str":
            return ("str", t.text[1:-1])
        if t.text == "(":
            e = self.expr()
            self.expect(")")
            return e
        if t.kind != "id":
            raise Exa4SyntaxError("line %d: unexpected %r in an expression" % (t.line, t.text))
        name = t.text
        if name in ("true", "false"):
            return ("num", name == "true")
        slot = None
        if name in self.field_names and self.at("<") and self.at(">", 2) and (
                self.peek(1).kind == "num" or self.peek(1).text in _SLOT_WORDS):
            self.next()
            s = self.next()
            slot = int(s.text) if s.kind == "num" else s.text
            self.next()
        level = None
        if self.at("@") and not self.at("[", 1):
            self.next()
            level = self._level_item()
        if self.at("@") and self.at("[", 1):
            raise Exa4Unsupported("line %d: offset access %s@[...]" % (t.line, name))
        if name in self.field_names:
            return ("fld", name, slot, level)
        if name in self.stencil_names or name in self.sfield_names:
            if self.at(":") and self.at("[", 1):
                self.next()
                off = tuple(int(_const_value(e)) for e in self._const_list())
                return ("sentry", name, level, off)
            return ("sten", name, level)
        if self.at("("):
            self.next()
            args = []
            while not self.accept(")"):
                args.append(self.expr())
                self.accept(",")
            return ("call", name, level, args)
        return ("id", name, level)


def _num(text: str, sign: int):
    if re.fullmatch(r"\d+", text):
        return ("num", sign * int(text))
    return ("num", sign * float(text))


def _const_value(e, env: Optional[Dict[str, float]] = None):
    """Value of an expression over literals (and the names in env)."""
    k = e[0]
    if k == "num":
        return e[1]
    if k == "neg":
        return -_const_value(e[1], env)
    if k == "id" and env is not None and e[1] in env:
        return env[e[1]]
    if k == "bin":
        return _arith(e[1], _const_value(e[2], env), _const_value(e[3], env))
    raise Exa4SyntaxError("constant expected, found %r" % (e,))


def _arith(op: str, a, b):
    if op == "+":
        return a + b
    if op == "-":
        return a - b
    if op == "*":
        return a * b
    if op == "/":
        if isinstance(a, int) and isinstance(b, int) and not isinstance(a, bool):
            q = abs(a) // abs(b)          # C integer division truncates towards zero
            return q if (a >= 0) == (b >= 0) else -q
        try:
            return a / b
        except ZeroDivisionError:       # IEEE semantics of the generated C++
            return float("nan") if a == 0 or a != a else math.copysign(float("inf"), a)
    if op == "%":
        return math.fmod(a, b) if isinstance(a, float) or isinstance(b, float) else int(math.fmod(a, b))
    if op == "**":
        return a ** b if not (isinstance(b, float) and b == 2.0) else a ** 2
    if op == "==":
        return a == b
    if op == "!=":
        return a != b
    if op == "<":
        return a < b
    if op == "<=":
        return a <= b
    if op == ">":
        return a > b
    if op == ">=":
        return a >= b
    if op == "&&":
        return bool(a) and bool(b)
    if op == "||":
        return bool(a) or bool(b)
    raise Exa4SyntaxError("operator %r" % op)


def _classify_transfer(mapped) -> str:
    """`[i0, i1] from [2.0 * i0 - 1.0, ...] with w` entries: which inter-grid operator, and is it the linear one?"""
    env = {"i0": 0.0, "i1": 0.0, "i2": 0.0}
    offs = [tuple(float(_const_value(e, env)) for e in src) for src, _ in mapped]
    weights = [float(_const_value(w)) for _, w in mapped]
    prolong = any(abs(o) == 0.5 for off in offs for o in off)
    nd = len(offs[0])
    if len(offs) != 3 ** nd:
        raise Exa4Unsupported("inter-grid stencil with %d entries in %dD" % (len(offs), nd))
    for off, w in zip(offs, weights):
        want = 1.0
        for o in off:
            far = (abs(o) == 0.5) if prolong else (abs(o) == 1.0)
            want *= (0.5 if far else 1.0) if prolong else (0.25 if far else 0.5)
        if abs(w - want) > 1e-15:
            raise Exa4Unsupported("inter-grid stencil is not the linear %s" % ("prolongation" if prolong else "restriction"))
    return "prolongation" if prolong else "restriction"




# -- AST helpers ----------------------------------------------------------------------------------------------------------
def _walk(e):
    if isinstance(e, tuple):
        yield e
        for c in e[1:]:
            if isinstance(c, tuple):
                yield from _walk(c)
            elif isinstance(c, list):
                for x in c:
                    if isinstance(x, tuple):
                        yield from _walk(x)


def _contains(e, kinds) -> bool:
    return any(n[0] in kinds for n in _walk(e) if n and isinstance(n[0], str))


def _find_calls(e):
    return [n for n in _walk(e) if n and n[0] == "call"]


def _has_coord(e, functions) -> bool:
    for n in _walk(e):
        if n and n[0] == "id" and isinstance(n[1], str) and (_COORD.match(n[1]) or re.fullmatch(r"i[012]", n[1])):
            return True
    return False


def _conjuncts(e):
    if e[0] == "bin" and e[1] == "&&":
        return _conjuncts(e[2]) + _conjuncts(e[3])
    return [e]


def _index_sum(e, nd: int):
    """constant c if e == c + i0 + i1 [+ i2] (each index once), else None."""
    seen, const = [], 0

    def rec(x):
        nonlocal const
        if x[0] == "bin" and x[1] == "+":
            return rec(x[2]) and rec(x[3])
        if x[0] == "id" and re.fullmatch(r"i[012]", x[1]):
            seen.append(x[1])
            return True
        if x[0] == "num" and isinstance(x[1], int):
            const += x[1]
            return True
        return False

    if rec(e) and sorted(seen) == ["i%d" % d for d in range(nd)]:
        return const
    return None


def _parity_expr(e, nd: int):
    """shift s if e == (s + i0 + i1 [+ i2]) % 2."""
    if e[0] == "bin" and e[1] == "%" and e[3] == ("num", 2):
        return _index_sum(e[2], nd)
    return None


def _colour_cond(e, nd: int):
    """colour selected by `c == (s + i0 + ...) % 2` (either side)."""
    if e[0] != "bin" or e[1] != "==":
        return None
    for a, b in ((e[2], e[3]), (e[3], e[2])):
        if a[0] == "num" and isinstance(a[1], int):
            s = _parity_expr(b, nd)
            if s is not None:
                return (a[1] - s) % 2
    return None


def _lower_cond(e):
    """d if e == (i_d > 0)."""
    if e[0] == "bin" and e[1] == ">" and e[2][0] == "id" and re.fullmatch(r"i[012]", e[2][1]) and e[3] == ("num", 0):
        return int(e[2][1][1])
    return None
