"""SURVEY.md 8 row f-2, colour-packed layouts: scalar fields under the layout transformation

    LayoutTransformations { transform Solution@finest with [x, y, z] => [x / 2, y, z, x % 2] }

of the reference (Testing/LayoutTrafo/rbgs.exa4:2; Compiler/src/exastencils/layoutTransformation/ir/IR_LayoutTransformStatement.scala: an
affine map on array indices, new extents = its image) -- EXAMG_LAYOUT_SPLIT_X in include/examg.h.  A transformation changes where a value
lives and never a value: every loop kind run on split fields must leave, after transforming back, the bits the oracle's loops leave on the
plain layout.  Through the C ABI on the GPU; the oracle knows plain layouts only."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle_ops import OracleOps  # noqa: E402

from exastencils_amd.field import Stencil, laplace_fd  # noqa: E402
from exastencils_amd.layout import FieldLayout  # noqa: E402

pytestmark = pytest.mark.gpu

APPLY, RESIDUAL, SMOOTH = 0, 1, 2


@pytest.fixture(scope="module")
def hip():
    from exastencils_amd.ops import HipOps

    return HipOps(0)


@pytest.fixture(scope="module")
def orc():
    return OracleOps()


def to_split(hip, lay, x):
    """Device array x (plain layout `lay`) under the colour split."""
    ls = lay.split_x()
    y = hip.new_array(ls.size)
    hip.transform_field(lay.c_struct(), x, ls.c_struct(), y)
    return y


def to_plain(hip, lay, y):
    x = hip.new_array(lay.size)
    hip.transform_field(lay.split_x().c_struct(), y, lay.c_struct(), x)
    hip.synchronize()
    return hip.to_host(x)


def same(a, b, what):
    assert a.shape == b.shape, what
    if not np.array_equal(a, b):
        d = np.abs(a - b)
        raise AssertionError("%s: %d of %d values differ, max abs %.3e" % (what, int((d > 0).sum()), d.size, d.max()))


@pytest.mark.parametrize("nd,shape,ghost", [(3, (12, 7, 5), 1), (3, (13, 6, 4), 0), (2, (16, 9), 1), (2, (15, 8), 2)])
def test_transform_places_every_value_where_the_map_says(hip, nd, shape, ghost):
    """examg_transform_field against the restated index map (FieldLayout.linear): [x, y, z] => [x / 2, y, z, x % 2] on ARRAY indices, extents
    ceil(TOTx / 2), TOTy, TOTz, 2; and back: the identity.  Odd and even row lengths."""
    lay = FieldLayout.node(nd, shape, ghost)
    ls = lay.split_x()
    hx = (lay.tot(0) + 1) // 2
    assert ls.size == 2 * hx * lay.tot(1) * lay.tot(2) == int(hip.L.examg_layout_size(ls.c_struct())) and lay.size == int(hip.L.examg_layout_size(lay.c_struct()))
    a = np.arange(lay.size, dtype=np.float64) + 1.0
    x = hip.from_host(a)
    y = to_split(hip, lay, x)
    hip.synchronize()
    got = hip.to_host(y)
    want = np.zeros(ls.size)
    for i2 in range(-lay.ref(2), lay.tot(2) - lay.ref(2)):
        for i1 in range(-lay.ref(1), lay.tot(1) - lay.ref(1)):
            for i0 in range(-lay.ref(0), lay.tot(0) - lay.ref(0)):
                want[ls.linear(i0, i1, i2)] = a[lay.linear(i0, i1, i2)]
    same(got, want, "split image")
    same(to_plain(hip, lay, y), a, "round trip")


def _st_const(nd, shape, kind):
    if kind == "star":
        return laplace_fd(nd, tuple(1.0 / s for s in shape))
    offs = [(0, 0, 0)] + [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) != (0, 0, 0)]
    return Stencil(offs, [26.0] + [-1.0 / (1 + abs(o[0]) + abs(o[1]) + abs(o[2])) for o in offs[1:]])


@pytest.mark.parametrize("mode,colour", [(SMOOTH, 0), (SMOOTH, 1), (SMOOTH, -1), (RESIDUAL, -1), (APPLY, -1)])
@pytest.mark.parametrize("nd,shape,kind,b,e", [
    (3, (40, 20, 12), "star", None, None),            # 3-D 7-point
    (3, (33, 17, 9), "star", [0, 1, 0], [34, 17, 10]),  # odd rows, loop over duplicate planes (interior faces)
    (3, (200, 12, 40), "star", None, None),           # several 64-pair windows, z chunks (k_rbgs_half_split7)
    (3, (131, 9, 7), "star", [0, 0, 0], [132, 10, 8]),  # every face an interior face: first column pair at array index 1
    (3, (70, 65, 9), "star", None, None),             # 16 row groups: the XCD band order of the workgroups
    (3, (24, 12, 10), "27", None, None),              # 27-point constant stencil
    (2, (64, 48), "star", None, None),                # 2-D 5-point
    (3, (24, 14, 8), "field7", None, None),           # 7-entry stencil field (coefficients in the plain layout)
])
def test_stencil_loops_on_split_fields(hip, orc, nd, shape, kind, b, e, mode, colour):
    """`loop over` with a stencil convolution -- half sweeps of both colours, Jacobi step, residual, A * u -- with Solution, RHS and the
    destination under the colour split: transformed back, the bits of the oracle's loop on the plain layout."""
    if kind == "27" and colour >= 0:
        pytest.skip("an in-place colour loop of a 27-point stencil reads points of its own colour: order-dependent on any layout")
    lu, lf = FieldLayout.node(nd, shape, 1), FieldLayout.node(nd, shape, 0)
    if b is None:
        b = [1 if d < nd else 0 for d in range(3)]
        e = [shape[d] if d < nd else 1 for d in range(3)]

    def case(ops, split):
        u, f, dst = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size)
        ops.fill_random(u, 12345)
        ops.fill_random(f, 4711)
        ops.fill_random(dst, 5)
        if kind == "field7":
            offs = [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
            cf = ops.new_array(7 * lf.size)
            ops.fill_random(cf, 99)
            cf += 3.0
            st, w = Stencil(offs, [], cf, lf), 0.8
        else:
            st = _st_const(nd, shape, kind)
            w = 0.8 / st.diag
        Lu, Lf = lu, lf
        if split:
            u, f, dst = to_split(ops, lu, u), to_split(ops, lf, f), to_split(ops, lu, dst)
            Lu, Lf = lu.split_x(), lf.split_x()
        if colour >= 0:
            ops.stencil_op(mode, Lu.c_struct(), u, Lf.c_struct(), f, Lu.c_struct(), u, st, w, colour, b, e)
            out = u
        else:
            ops.stencil_op(mode, Lu.c_struct(), u, Lf.c_struct(), f, Lu.c_struct(), dst, st, w, -1, b, e)
            out = dst
        return to_plain(ops, lu, out) if split else ops.to_host(out)

    got = case(hip, True)
    same(got, case(orc, False), "split layout vs oracle")
    hip.synchronize()
    same(got, case(hip, False), "split layout vs plain layout on the GPU")


def test_transfers_blas_and_boundary_loops_on_split_fields(hip, orc):
    """Restriction (fine residual split, coarse right-hand side split), correction (fine split), set / axpby / dot, `apply bc`, pack and
    unpack: split == plain == oracle, bit for bit (the dot to rounding: its tree sums in another order)."""
    from exastencils_amd.lib import GeomC

    shape, cs = (34, 18, 10), (17, 9, 5)
    lfi, lco, lrh = FieldLayout.node(3, shape, 1), FieldLayout.node(3, cs, 1), FieldLayout.node(3, cs, 0)
    bf, ef, bc, ec = [1, 1, 1], list(shape), [1, 1, 1], list(cs)
    g = GeomC()
    for d in range(3):
        g.h[d], g.pos_begin[d] = 1.0 / shape[d], 0.0

    def case(ops, split):
        r, fc, uc, uf, y = ops.new_array(lfi.size), ops.new_array(lrh.size), ops.new_array(lco.size), ops.new_array(lfi.size), ops.new_array(lfi.size)
        for i, t in enumerate((r, fc, uc, uf, y)):
            ops.fill_random(t, 1 + i)
        L = (lambda l: l.split_x()) if split else (lambda l: l)
        if split:
            r, fc, uc, uf, y = to_split(ops, lfi, r), to_split(ops, lrh, fc), to_split(ops, lco, uc), to_split(ops, lfi, uf), to_split(ops, lfi, y)
        ops.restrict(L(lfi).c_struct(), r, L(lrh).c_struct(), fc, 1.0, bc, ec)
        ops.prolong_add(L(lco).c_struct(), uc, L(lfi).c_struct(), uf, bf, ef)
        ops.axpby(L(lfi).c_struct(), r, L(lfi).c_struct(), y, 0.5, -2.0, bf, ef)
        ops.set(L(lco).c_struct(), uc, 3.25, [1, 1, 1], [cs[0], cs[1], 3])
        ops.apply_dirichlet(L(lfi).c_struct(), uf, g, 1, (), 63)          # x^2 - y^2/2 - z^2/2 on all six faces
        buf = ops.new_array(6 * 18 * 10)
        ops.pack(L(lfi).c_struct(), uf, buf, [3, 0, 0], [9, 18, 10])
        ops.unpack(L(lfi).c_struct(), y, buf, [20, 0, 0], [26, 18, 10])
        d = ops.dot(L(lfi).c_struct(), r, L(lfi).c_struct(), y, bf, ef)
        outs = [(lrh, fc), (lfi, uf), (lfi, y), (lco, uc)]
        return [to_plain(ops, l, t) if split else ops.to_host(t) for l, t in outs] + [np.array([ops.scalar_value(d)])]

    got, want = case(hip, True), case(orc, False)
    for i, (a, b_) in enumerate(zip(got[:-1], want[:-1])):
        same(a, b_, "array %d" % i)
    assert abs(got[-1][0] - want[-1][0]) <= 1e-13 * abs(want[-1][0])
    hip.synchronize()
    plain = case(hip, False)
    for i, (a, b_) in enumerate(zip(got[:-1], plain[:-1])):
        same(a, b_, "array %d vs plain on the GPU" % i)


def test_one_pass_entry_points_take_their_loops_on_split_fields(hip, orc):
    """The one-pass entry points (fused red-black sweep, residual + restriction) have no kernel for transformed layouts: they run the
    loops they stand for -- same bits; the one-kernel coarse solver refuses a transformed field with a message."""
    from exastencils_amd.lib import ExamgError, GeomC

    shape, cs = (72, 20, 12), (36, 10, 6)
    lu, lf, lc = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0), FieldLayout.node(3, cs, 0)
    st = laplace_fd(3, tuple(1.0 / s for s in shape))
    w = 0.8 / st.diag
    b, e, cb, ce = [1, 1, 1], list(shape), [1, 1, 1], list(cs)

    def case(ops, split):
        u, f, out, r, fc = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lc.size)
        ops.fill_random(u, 3)
        ops.fill_random(f, 4)
        L = (lambda l: l.split_x()) if split else (lambda l: l)
        if split:
            u, f, out, r, fc = (to_split(ops, l, t) for l, t in ((lu, u), (lf, f), (lu, out), (lu, r), (lc, fc)))
            ops.rbgs_sweep_fused(L(lu).c_struct(), u, out, L(lf).c_struct(), f, st, w, 0, b, e)
            ops.residual_restrict(L(lu).c_struct(), out, L(lf).c_struct(), f, L(lu).c_struct(), r, st, L(lc).c_struct(), fc, 1.0, b, e, cb, ce)
            return [to_plain(ops, lu, out)[...], to_plain(ops, lc, fc)]
        work = u.clone()
        for col in (0, 1):
            ops.stencil_op(SMOOTH, lu.c_struct(), work, lf.c_struct(), f, lu.c_struct(), work, st, w, col, b, e)
        ops.stencil_op(RESIDUAL, lu.c_struct(), work, lf.c_struct(), f, lu.c_struct(), r, st, 0.0, -1, b, e)
        ops.restrict(lu.c_struct(), r, lc.c_struct(), fc, 1.0, cb, ce)
        return [ops.to_host(work), ops.to_host(fc)]

    got, want = case(hip, True), case(orc, False)
    sl = tuple(slice(b[d] + 1, e[d] + 1) for d in (2, 1, 0))
    same(got[0].reshape(lu.shape_zyx)[sl], want[0].reshape(lu.shape_zyx)[sl], "sweep on split fields (box)")
    same(got[1], want[1], "residual + restriction on split fields")
    ls = lu.split_x()
    x = hip.new_array(ls.size)
    info = hip.new_array(4)
    g = GeomC()
    with pytest.raises(ExamgError, match="layout transformation"):
        hip.cg_coarse(ls.c_struct(), x, lf.c_struct(), hip.new_array(lf.size), lf.c_struct(), hip.new_array(lf.size), lu.c_struct(), hip.new_array(lu.size),
                      lf.c_struct(), hip.new_array(lf.size), st, g, 63, 8, 1e-3, b, e, info)


SPLIT_BLOCK = "LayoutTransformations {\n  transform u@(finest, (finest - 1)), f@finest with [x, y, z] => [x / 2, y, z, x % 2]\n}\n\n"


def test_program_with_the_colour_split_prints_the_same_values(hip):
    """examples/exa4/poisson3d_rbgs.exa4 with the reference's directive in front (Testing/LayoutTrafo/rbgs.exa4:1-3): the interpreter
    APPLIES it -- u on the two finest levels and f on the finest live under the colour split, every loop of the program reaches them
    through the transformed index -- and every printed value is bit-identical to the program without the directive."""
    from exastencils_amd import exa4

    with open(os.path.join(ROOT, "examples", "exa4", "poisson3d_rbgs.exa4")) as fh:
        text = fh.read()
    k = dict(dimensionality=3, minLevel=2, maxLevel=5)
    plain = exa4.Exa4Program(text, k, ops=hip, fuse=False, fuse_coarse_solver=True)
    plain.run()
    P = exa4.Exa4Program(SPLIT_BLOCK + text, k, ops=hip)
    assert P._split_fields == {("u", 5), ("u", 4), ("f", 5)}
    assert P.fields[("u", 5)].layout.transform == 1 and P.fields[("u", 3)].layout.transform == 0
    P.run()
    assert P.printed_values == plain.printed_values and len(P.printed_values) > 3
    a = P.fields[("u", 5)].host_array(hip)
    b = plain.fields[("u", 5)].host_array(hip)
    assert np.array_equal(a, b)
