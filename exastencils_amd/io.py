"""Field output / input of the ExaSlang-4 surface: `printField`, `writeField`, `readField`.

Reference: IR_PrintField with the "lock" interface (Compiler/src/exastencils/field/ir/IR_PrintField.scala:38-110): one
line per point, node position per dimension (std::defaultfloat) then the value (std::scientific), each followed by
`separator`; points DLB..DRE, or GLB..GRE with includeGhostLayers; optional condition.  Binary mode / writeField: the same points as raw
doubles, x fastest (Compiler/src/exastencils/io/ir/IR_FileAccess_Locking.scala).  Data leave the device once, through the
kernel layer's `to_host` (not performance-relevant: checkpoint/restart and visual verification).
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from .field import Field


def _region(field: Field, include_ghost: bool):
    lay = field.layout
    lo = "GLB" if include_ghost else "DLB"
    hi = "GRE" if include_ghost else "DRE"
    b = [lay.idx(lo, d) if d < lay.nd else 0 for d in range(3)]
    e = [lay.idx(hi, d) if d < lay.nd else 1 for d in range(3)]
    sl = tuple(slice(b[d] + lay.ref(d), e[d] + lay.ref(d)) for d in (2, 1, 0))
    return b, e, sl


def print_field(filename: str, field: Field, ops, domain, slot: Optional[int] = None, include_ghost: bool = False,
                separator: str = " ", condition: Optional[Callable[[int, int, int], bool]] = None, append: bool = False,
                precision: int = -1):
    """printField ( filename, field ) / writeField_lock in ascii mode: one line per point of this block,

        x<sep>[y<sep>[z<sep>]]value<sep>\n

    positions with std::defaultfloat, the value with std::scientific, both at the stream's precision
    (Knowledge.field_printFieldPrecision, -1 = the iostream default 6): IR_PrintField.scala:62-72 puts
    `std::defaultfloat, pos_d, sep, ...` in front of what IR_Iostream.printBufferAscii (io/ir/IR_Iostream.scala:25-41) prints per
    point -- `std::scientific, value, sep, newline`; the precision is set once per file (IR_FileAccess_Locking.scala:154-160).
    Every value, the last one too, is followed by the separator."""
    b, e, sl = _region(field, include_ghost)
    a = field.host_array(ops, slot)[sl]
    g = domain.geom(field.level)
    nd = field.layout.nd
    p = 6 if precision < 0 else int(precision)
    with open(filename, "a" if append else "w") as f:
        for k in range(a.shape[0]):
            for j in range(a.shape[1]):
                for i in range(a.shape[2]):
                    i0, i1, i2 = b[0] + i, b[1] + j, b[2] + k
                    if condition is not None and not condition(i0, i1, i2):
                        continue
                    pos = [i0 * g.h[0] + g.pos_begin[0], i1 * g.h[1] + g.pos_begin[1], i2 * g.h[2] + g.pos_begin[2]][:nd]
                    f.write("".join("%.*g%s" % (p, x, separator) for x in pos) + "%.*e%s\n" % (p, float(a[k, j, i]), separator))


def read_field_ascii(filename: str, field: Field, ops, slot: Optional[int] = None, include_ghost: bool = False,
                     separator: str = " ", condition: Optional[Callable[[int, int, int], bool]] = None):
    """readField_lock in ascii mode: the inverse of print_field for the same region and condition (positions are skipped, the
    last number of a line is the value); points outside keep their values."""
    b, e, sl = _region(field, include_ghost)
    full = field.host_array(ops, slot).copy()
    view = full[sl]
    with open(filename) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip()]
    it = iter(lines)
    for k in range(view.shape[0]):
        for j in range(view.shape[1]):
            for i in range(view.shape[2]):
                if condition is not None and not condition(b[0] + i, b[1] + j, b[2] + k):
                    continue
                try:
                    ln = next(it)
                except StopIteration:
                    raise ValueError("%s ends before the field region is filled" % filename)
                toks = [t for t in (ln.split(separator) if separator.strip() else ln.split()) if t.strip()]
                view[k, j, i] = float(toks[-1])
    field.set_host_array(ops, full, slot)


def write_field(filename: str, field: Field, ops, slot: Optional[int] = None, include_ghost: bool = False):
    """writeField ( filename, field ) in binary mode: raw doubles of DLB..DRE (or GLB..GRE), x fastest."""
    _, _, sl = _region(field, include_ghost)
    a = field.host_array(ops, slot)[sl]
    np.ascontiguousarray(a, dtype=np.float64).tofile(filename)


def read_field(filename: str, field: Field, ops, slot: Optional[int] = None, include_ghost: bool = False):
    """readField ( filename, field ): the inverse of write_field; points outside the region keep their values."""
    _, _, sl = _region(field, include_ghost)
    full = field.host_array(ops, slot).copy()
    want = full[sl].shape
    a = np.fromfile(filename, dtype=np.float64)
    if a.size != int(np.prod(want)):
        raise ValueError("%s holds %d values, the field region needs %d" % (filename, a.size, int(np.prod(want))))
    full[sl] = a.reshape(want)
    field.set_host_array(ops, full, slot)
