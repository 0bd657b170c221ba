"""`external Field <name> <layout> => <field>@<level>`: hand-over of field data to and from a host application.

Reference: the generator emits `void get<name>(double* dest, int slot)` / `set<name>(double* src, int slot)` that copy
between a caller-owned array in the *external* layout and the internal field, matching ghost widths
(Compiler/src/exastencils/interfacing/ir/IR_CopyToExternalField.scala:31-90, IR_CopyFromExternalField.scala;
declaration Compiler/src/exastencils/interfacing/l4/L4_ExternalFieldDecl.scala:29-41).  Here the copy is a device kernel
(examg_copy_to_external / examg_copy_from_external); host arrays cross PCIe once, through a device staging tensor.
"""
from __future__ import annotations

import numpy as np

from .field import Field
from .layout import FieldLayout


class ExternalField:
    def __init__(self, name: str, layout: FieldLayout, target: Field, ops):
        self.name, self.layout, self.target, self.ops = name, layout, target, ops
        self.lc = layout.c_struct()
        self._stage = None

    def _staging(self):
        if self._stage is None:
            self._stage = self.ops.new_array(self.layout.size)
        return self._stage

    # get<name>(dest, slot)
    def get(self, slot=None, out: np.ndarray = None) -> np.ndarray:
        """Copy the internal field (slot) into an array in the external layout, returned in [z, y, x] shape."""
        st = self._staging()
        self.ops.copy_to_external(self.target.lc, self.target.data(slot), self.lc, st)
        self.ops.synchronize()
        a = self.ops.to_host(st).reshape(self.layout.shape_zyx)
        if out is not None:
            out[...] = a
            return out
        return a.copy()

    # set<name>(src, slot)
    def set(self, src: np.ndarray, slot=None):
        a = np.ascontiguousarray(src, dtype=np.float64).reshape(-1)
        if a.size != self.layout.size:
            raise ValueError("external array has %d values, layout needs %d" % (a.size, self.layout.size))
        st = self.ops.from_host(a)
        self.ops.copy_from_external(self.lc, st, self.target.lc, self.target.data(slot))
