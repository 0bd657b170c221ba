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
    assert all(ln.endswith(" ") for ln in lines)                      # every value is followed by the separator
    x, y, v = lines[7].split(" ")[:3]                                 # point (i0=2, i1=1)
    assert (float(x), float(y)) == (0.5, 0.25) and abs(float(v) - a[0, 2, 3]) < 1e-5 * max(1.0, abs(a[0, 2, 3]))
    import re
    assert re.fullmatch(r"-?\d\.\d{6}e[+-]\d{2}", v)                  # std::scientific at the default precision 6
    assert x == "0.5" and y == "0.25"                                 # std::defaultfloat
    xio.print_field(str(tmp_path / "g.txt"), F, ops, dom, include_ghost=True, condition=lambda i0, i1, i2: i0 == i1)
    assert len((tmp_path / "g.txt").read_text().splitlines()) == 7


def test_print_field_matches_iostream_format(tmp_path):
    """The exact characters std::ofstream produces for `out << std::scientific; out << std::defaultfloat << x << sep << y << sep
    << std::scientific << v << sep << std::endl` (IR_PrintField.scala:62-72 + IR_Iostream.scala:25-41), checked on values with
    known iostream renderings, and the ascii read-back."""
    ops = OracleOps()
    dom = RectDomain(2, (1, 1, 1), 0)
    lay = FieldLayout.node(2, dom.ncells(1), 0)
    F = Field("f", 1, lay, ops)
    v = F.data().numpy().reshape(lay.shape_zyx)
    v[0, :, :] = [[0.1234567891, -2.5, 1e-12], [123456.789, 0.0, 3.0], [1.0 / 3.0, -1e100, 7.25]]
    xio.print_field(str(tmp_path / "p.csv"), F, ops, dom, separator=",")
    assert (tmp_path / "p.csv").read_text().splitlines() == [
        "0,0,1.234568e-01,", "0.5,0,-2.500000e+00,", "1,0,1.000000e-12,",
        "0,0.5,1.234568e+05,", "0.5,0.5,0.000000e+00,", "1,0.5,3.000000e+00,",
        "0,1,3.333333e-01,", "0.5,1,-1.000000e+100,", "1,1,7.250000e+00,"]
    xio.print_field(str(tmp_path / "q.txt"), F, ops, dom, precision=3)
    assert (tmp_path / "q.txt").read_text().splitlines()[0] == "0 0 1.235e-01 "
    G = Field("g", 1, lay, ops)
    xio.read_field_ascii(str(tmp_path / "p.csv"), G, ops, separator=",")
    g = G.data().numpy().reshape(lay.shape_zyx)
    assert np.allclose(g, v, rtol=1e-6, atol=0)
