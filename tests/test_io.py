"""printField / writeField / readField on the CPU kernel layer."""
import numpy as np

from oracle_ops import OracleOps

from exastencils_amd import io as xio
from exastencils_amd.domain import RectDomain
from exastencils_amd.field import Field
from exastencils_amd.layout import FieldLayout


def test_write_read_print(tmp_path):
    ops = OracleOps()
    dom = RectDomain(2, (1, 1, 1), 0)
    lay = FieldLayout.node(2, dom.ncells(2), 1)
    F = Field("Solution", 2, lay, ops)
    ops.fill_random(F.data(), 3)
    before = F.data().clone()
    xio.write_field(str(tmp_path / "f.bin"), F, ops)
    assert (tmp_path / "f.bin").stat().st_size == 5 * 5 * 8            # DLB..DRE of a 4x4-cell node field
    F.data().zero_()
    xio.read_field(str(tmp_path / "f.bin"), F, ops)
    a, b = F.data().numpy().reshape(lay.shape_zyx), before.numpy().reshape(lay.shape_zyx)
    assert np.array_equal(a[:, 1:6, 1:6], b[:, 1:6, 1:6]) and (a[:, 0, :] == 0).all()
    xio.print_field(str(tmp_path / "f.txt"), F, ops, dom)
    lines = (tmp_path / "f.txt").read_text().splitlines()
    assert len(lines) == 25
    x, y, v = lines[7].split(" ")                                     # point (i0=2, i1=1)
    assert (float(x), float(y)) == (0.5, 0.25) and abs(float(v) - a[0, 2, 3]) < 1e-5 * max(1.0, abs(a[0, 2, 3]))
    xio.print_field(str(tmp_path / "g.txt"), F, ops, dom, include_ghost=True, condition=lambda i0, i1, i2: i0 == i1)
    assert len((tmp_path / "g.txt").read_text().splitlines()) == 7
