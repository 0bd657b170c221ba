"""GPU parity: every HIP kernel, called through the C ABI (ctypes -> libexamg.so), against the CPU oracle on
the same seeded inputs.  Point-wise kernels: bit-exact.  Reductions: 1e-13 relative (summation order).
Analytic fills with libm calls: 4 ulp."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle_ops import OracleOps

from exastencils_amd.field import (FN_KAPPA_COEF, FN_KAPPA_EXPSOL, FN_KAPPA_RHS, FN_POLY3D, FN_TRIG2D_SOL, FN_XSQ,
                                   FN_TRIG3D_SOL, Stencil, laplace_fd, laplace_unit, stencil_field_offsets)
from exastencils_amd.layout import FieldLayout
from exastencils_amd.lib import GeomC

APPLY, RESIDUAL, SMOOTH = 0, 1, 2


@pytest.fixture(scope="module")
def hip():
    from exastencils_amd.ops import HipOps

    return HipOps(0)


@pytest.fixture(scope="module")
def hipd():
    """Kernel layer bound to the debug build (libexamg_dbg.so, -DEXAMG_DEBUG_HOOKS): same kernels plus the examg_debug_* entry
    points that force the generic / alternative variants.  The product library does not export them (tests/test_abi.py)."""
    from exastencils_amd import lib
    from exastencils_amd.ops import HipOps

    return HipOps(0, lib.DBG_LIB_PATH)


@pytest.fixture(scope="module")
def orc():
    return OracleOps()


def both(hip, orc, fn):
    """Run fn(ops) on both back ends; returns (gpu arrays, cpu arrays) as numpy."""
    g = fn(hip)
    hip.synchronize()
    c = fn(orc)
    return [hip.to_host(t) for t in g], [orc.to_host(t) for t in c]


def assert_same(g, c, what=""):
    for i, (a, b) in enumerate(zip(g, c)):
        if not np.array_equal(a, b):
            d = np.abs(a - b)
            raise AssertionError("%s[%d]: %d of %d values differ, max abs %.3e" % (what, i, int((d > 0).sum()), d.size, d.max()))


def geom(nd, n, lo=0.0):
    g = GeomC()
    for d in range(3):
        g.pos_begin[d] = lo if d < nd else 0.0
        g.h[d] = 1.0 / n if d < nd else 0.0
    return g


def box(nd, n, lo=1, hi=None):
    hi = n if hi is None else hi
    return [lo if d < nd else 0 for d in range(3)], [hi if d < nd else 1 for d in range(3)]


def test_fill_random_same_bits(hip, orc):
    def f(ops):
        x = ops.new_array(100003)
        ops.fill_random(x, 12345)
        return [x]

    g, c = both(hip, orc, f)
    assert_same(g, c, "fill_random")
    assert -1.0 <= g[0].min() and g[0].max() < 1.0


def _stencil_case(ops, nd, shape, st, mode, colour, b, e, ghost=1, align=0, cfn=None, entry_fastest=False):
    lu = FieldLayout.node(nd, shape, ghost, align=align)
    lf = FieldLayout.node(nd, shape, 0, align=align)
    u, f, dst = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size)
    ops.fill_random(u, 12345)
    ops.fill_random(f, 4711)
    if cfn is not None:
        K = len(st.offsets)
        cf = ops.new_array(K * lf.size)
        ops.fill_random(cf, 99)
        cf += 3.0       # keep the diagonal away from zero
        st = Stencil(st.offsets, [], cf, lf)
        if entry_fastest:
            # `transform <coefficients> with [x, y, z, i] => [i, x, y, z]`: same values, entries of a point contiguous (the oracle's
            # kernel layer has no transformed layouts and keeps the planes -- the results must not depend on where values live)
            st = st.entry_fastest(ops)
    w = 0.8 / st.diag if cfn is None else 0.8
    if colour >= 0:
        ops.stencil_op(mode, lu.c_struct(), u, lf.c_struct(), f, lu.c_struct(), u, st, w, colour, b, e)
        return [u]
    ops.stencil_op(mode, lu.c_struct(), u, lf.c_struct(), f, lu.c_struct(), dst, st, w, -1, b, e)
    return [dst]


@pytest.mark.parametrize("mode", [APPLY, RESIDUAL, SMOOTH])
@pytest.mark.parametrize("order", ["mp", "pm"])
@pytest.mark.parametrize("n", [64, 96, 130])
def test_stencil7_fast_path_bit_exact(hip, orc, mode, order, n):
    """3-D 7-point constant coefficients, both entry orders of the reference programs; 96 and 130 leave ragged
    tiles in x (128 per wave), y and z."""
    st = laplace_fd(3, (1.0 / n,) * 3, order)
    b, e = box(3, n)
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, (n, n, n), st, mode, -1, b, e))
    assert_same(g, c, "stencil7 mode %d" % mode)


@pytest.mark.parametrize("mode", [APPLY, RESIDUAL, SMOOTH])
@pytest.mark.parametrize("case", [
    # (cells per dim, begin, end, align): rows of 400 .. 512 points take the row-marching kernel (a wave owns whole rows and stores
    # whole, 128-byte-aligned lines through its LDS strip); every row / plane starts at another offset within its line
    ((512, 72, 20), [1, 1, 1], [512, 72, 20], 0),        # level-9 rows, odd count (511)
    ((512, 72, 20), [0, 0, 0], [513, 73, 21], 0),        # interior faces on every side: 513 points -> the window kernel takes it
    ((512, 66, 18), [0, 1, 1], [512, 66, 18], 0),        # 512 points, even count: the last pair needs its right neighbour from memory
    ((420, 70, 17), [1, 1, 1], [420, 70, 17], 0),        # last segment mostly empty, ragged tiles in y and z
    ((400, 64, 16), [0, 1, 0], [401, 64, 17], 0),        # 401 points
    ((512, 72, 20), [1, 1, 1], [512, 72, 20], 16),       # padded rows (544 doubles)
])
def test_stencil7_row_marching_kernel_bit_exact(hip, hipd, orc, mode, case):
    shape, b, e, align = case
    for order in ("mp", "pm"):
        st = laplace_fd(3, tuple(1.0 / n for n in shape), order)
        g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, shape, st, mode, -1, b, e, align=align))
        assert_same(g, c, "row-marching kernel, mode %d order %s" % (mode, order))
    # the same box through the 128-point-window kernel (row-marching kernel switched off in the debug build): same bits
    hipd.L.examg_debug_rowmarch(0, -1, -1)
    try:
        st = laplace_fd(3, tuple(1.0 / n for n in shape), "mp")
        g2 = [hipd.to_host(t) for t in _stencil_case(hipd, 3, shape, st, mode, -1, b, e, align=align)]
    finally:
        hipd.L.examg_debug_rowmarch(-1, -1, -1)
    g1 = [hip.to_host(t) for t in _stencil_case(hip, 3, shape, st, mode, -1, b, e, align=align)]
    assert_same(g1, g2, "row-marching vs window kernel")


def test_stencil7_anisotropic_box_and_interior_faces(hip, orc):
    """Non-cubic fragment, loop bounds of a block with neighbours on some faces (begin 0 / end n+1)."""
    shape = (160, 40, 24)
    st = laplace_unit(3)
    b, e = [0, 1, 0], [161, 40, 24]
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, shape, st, SMOOTH, -1, b, e))
    assert_same(g, c, "anisotropic")


def test_stencil7_padded_layout(hip, orc):
    st = laplace_fd(3, (1.0 / 64,) * 3)
    b, e = box(3, 64)
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, (64, 64, 64), st, SMOOTH, -1, b, e, align=16))
    assert_same(g, c, "padded")


@pytest.mark.parametrize("colour", [0, 1])
@pytest.mark.parametrize("order", ["mp", "pm"])
@pytest.mark.parametrize("n", [33, 64, 65, 130])
def test_rbgs_half_sweep_bit_exact(hip, orc, colour, order, n):
    """n >= 65 takes the coloured z-march fast path (in place), 33 the generic kernel."""
    st = laplace_fd(3, (1.0 / n,) * 3, order)
    b, e = box(3, n)
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, (n, n, n), st, SMOOTH, colour, b, e))
    assert_same(g, c, "rbgs colour %d" % colour)


def test_rbgs_interior_faces_and_odd_box(hip, orc):
    """Half sweeps on a block with neighbours (loop starts at the duplicate node, index 0) and odd extents."""
    shape = (97, 40, 24)
    st = laplace_unit(3)
    for colour in (0, 1):
        b, e = [0, 1, 0], [98, 40, 24]
        g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, shape, st, SMOOTH, colour, b, e))
        assert_same(g, c, "rbgs interior faces")


def _two_stage_case(ops, kind, shape, st, b, e, first=0):
    lu = FieldLayout.node(3, shape, 1)
    lf = FieldLayout.node(3, shape, 0)
    u, f, out, tmp = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size), ops.new_array(lu.size)
    ops.fill_random(u, 12345)
    ops.fill_random(f, 4711)
    ops.fill_random(out, 5)          # whatever was in the output array outside the box must survive
    w = 0.8 / st.diag
    if kind == "rbgs":
        ops.rbgs_sweep_fused(lu.c_struct(), u, out, lf.c_struct(), f, st, w, first, b, e)
    else:
        ops.jacobi2(lu.c_struct(), u, out, tmp, lf.c_struct(), f, st, w, b, e)
    return [out, u]


def _two_stage_reference(orc, kind, shape, st, b, e, first=0):
    """Two separate loops on the CPU; outside the box the output array keeps its previous content."""
    lu = FieldLayout.node(3, shape, 1)
    lf = FieldLayout.node(3, shape, 0)
    u, f, out, tmp = orc.new_array(lu.size), orc.new_array(lf.size), orc.new_array(lu.size), orc.new_array(lu.size)
    orc.fill_random(u, 12345)
    orc.fill_random(f, 4711)
    orc.fill_random(out, 5)
    w = 0.8 / st.diag
    L, Fl = lu.c_struct(), lf.c_struct()
    if kind == "rbgs":
        work = u.clone()
        for c in (first, 1 - first):
            orc.stencil_op(SMOOTH, L, work, Fl, f, L, work, st, w, c, b, e)
    else:
        tmp.copy_(u)
        orc.stencil_op(SMOOTH, L, u, Fl, f, L, tmp, st, w, -1, b, e)
        work = u.clone()
        orc.stencil_op(SMOOTH, L, tmp, Fl, f, L, work, st, w, -1, b, e)
    orc.axpby(L, work, L, out, 1.0, 0.0, b, e)      # box only
    return [orc.to_host(out), orc.to_host(u)]


@pytest.fixture(params=[None, 5, 8, 83], ids=["product", "lds5", "lds8", "lds8x3"])
def two_stage_variant(request, hip, hipd):
    """The product library with its own choice of workgroup shape, then every shape of the two-stage kernel (5 or 8 waves with two rows
    each, 8 waves with three rows each -- plain passes -- share a row stack through LDS) forced through the debug build; yields the kernel
    layer to use."""
    import ctypes as C

    if request.param is None:
        yield hip
        return
    hipd.L.examg_debug_two_stage_lds.argtypes = [C.c_int]
    hipd.L.examg_debug_two_stage_lds(request.param)
    yield hipd
    hipd.L.examg_debug_two_stage_lds(-1)     # back to the shipped default


@pytest.mark.parametrize("kind", ["rbgs", "jacobi2"])
@pytest.mark.parametrize("order", ["mp", "pm"])
@pytest.mark.parametrize("n,first", [(65, 0), (130, 1), (165, 0), (200, 0), (256, 1)])
def test_two_stage_kernel_bit_exact(orc, two_stage_variant, kind, order, n, first):
    """Fused red-black sweep / two Jacobi steps in one pass == the two loops run one after the other, bit for bit;
    130 ... 256 leave ragged 124-point x windows, row groups and z chunks."""
    hip = two_stage_variant
    st = laplace_fd(3, (1.0 / n,) * 3, order)
    b, e = box(3, n)
    g = _two_stage_case(hip, kind, (n, n, n), st, b, e, first)
    hip.synchronize()
    c = _two_stage_reference(orc, kind, (n, n, n), st, b, e, first)
    assert_same([hip.to_host(t) for t in g], c, "two-stage " + kind)


@pytest.fixture
def hip3(hipd):
    """The debug build with the size bound of the three-stage pass lowered to 2^20 points (the product takes it from 8e6 points on): the
    parity cases reach the kernel on boxes the oracle sweeps in a moment."""
    import ctypes as C

    hipd.L.examg_debug_three_stage.argtypes = [C.c_int] * 2
    hipd.L.examg_debug_three_stage(2, -1)
    yield hipd
    hipd.L.examg_debug_three_stage(0, -1)


@pytest.mark.parametrize("order", ["mp", "pm"])
@pytest.mark.parametrize("shape,b,e", [((130, 130, 130), None, None), ((165, 150, 140), None, None), ((256, 256, 64), None, None),
                                       ((200, 72, 90), None, None), ((136, 120, 120), [0, 1, 0], [137, 120, 121]),
                                       ((150, 130, 110), [3, 2, 5], [148, 127, 108]), ((96, 96, 96), None, None)])
def test_three_stage_kernel_bit_exact(hip3, orc, shape, b, e, order):
    hip = hip3
    """examg_jacobi3: three Jacobi steps in one pass (k_three_stage7_lds: 120-point x windows, 20-row groups, z chunks with three halo
    planes) == three loops one after the other, bit for bit; ragged windows, row groups and chunks; a box over the duplicate planes of a
    block with neighbours (its input halo in the ghost layer) and a box inside the inner points; 96^3 is below the (lowered) size bound
    and takes a step + a pair."""
    st = laplace_fd(3, tuple(1.0 / n for n in shape), order)
    if b is None:
        b, e = [1, 1, 1], list(shape)

    def f(ops):
        lu, lf = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0)
        u, fr, out, tmp = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size), ops.new_array(lu.size)
        ops.fill_random(u, 12345)
        ops.fill_random(fr, 4711)
        ops.fill_random(out, 5)          # whatever was in the output array outside the box must survive
        ops.jacobi3(lu.c_struct(), u, out, tmp, lf.c_struct(), fr, st, 0.8 / st.diag, b, e)
        return [out, u]

    g, c = both(hip, orc, f)
    assert_same(g, c, "jacobi3")


@pytest.mark.parametrize("order", ["mp", "pm"])
@pytest.mark.parametrize("first", [0, 1])
@pytest.mark.parametrize("shape,b,e", [((130, 130, 130), None, None), ((165, 150, 141), None, None), ((256, 256, 64), None, None),
                                       ((136, 120, 120), [0, 1, 0], [137, 120, 121]), ((150, 130, 110), [3, 2, 5], [148, 127, 108]),
                                       ((96, 96, 96), None, None)])
def test_three_colour_loops_in_one_pass(hip3, orc, shape, b, e, first, order):
    hip = hip3
    """examg_rbgs_colours3: colour `first`, the other colour, `first` again in one pass (k_three_stage7_lds<COL>) == three coloured loops
    in place, bit for bit; both first colours (all tile / plane parities), odd and even box origins, ragged tiles and chunks; 96^3 takes
    the copy + three loops."""
    st = laplace_fd(3, tuple(1.0 / n for n in shape), order)
    if b is None:
        b, e = [1, 1, 1], list(shape)

    def f(ops):
        lu, lf = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0)
        u, fr, out = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size)
        ops.fill_random(u, 12345)
        ops.fill_random(fr, 4711)
        ops.fill_random(out, 5)
        ops.rbgs_colours3(lu.c_struct(), u, out, lf.c_struct(), fr, st, 0.8 / st.diag, first, b, e)
        return [out, u]

    def ref(ops):     # outside the box the output keeps what it held (the one-pass kernel stores inside the box only)
        lu, lf = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0)
        u, fr, out = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size)
        ops.fill_random(u, 12345)
        ops.fill_random(fr, 4711)
        ops.fill_random(out, 5)
        ops.rbgs_colours3(lu.c_struct(), u, out, lf.c_struct(), fr, st, 0.8 / st.diag, first, b, e)
        return [out, u]

    got = f(hip)
    hip.synchronize()
    want = [orc.to_host(t) for t in ref(orc)]
    got = [hip.to_host(t) for t in got]
    if hip.three_stage_eligible(FieldLayout.node(3, shape, 1).c_struct(), FieldLayout.node(3, shape, 0).c_struct(), st, b, e):
        assert_same(got, want, "rbgs_colours3")
    else:
        # the fallback copies the box with its shell into the output first: compare on the box
        lu = FieldLayout.node(3, shape, 1)
        g3, w3 = got[0].reshape(lu.tot(2), lu.tot(1), lu.tot(0)), want[0].reshape(lu.tot(2), lu.tot(1), lu.tot(0))
        r = [lu.ref(d) for d in range(3)]
        sl = tuple(slice(b[d] + r[d], e[d] + r[d]) for d in (2, 1, 0))
        assert np.array_equal(g3[sl], w3[sl])
        assert np.array_equal(got[1], want[1])


def test_two_three_colour_passes_equal_three_sweeps(hip3, orc):
    hip = hip3
    """Two passes of three colour loops (first colour 0, then 1) == three fused red-black sweeps == six coloured loops of the oracle."""
    shape = (160, 140, 130)
    st = laplace_fd(3, tuple(1.0 / n for n in shape), "mp")
    b, e = [1, 1, 1], list(shape)
    lu, lf = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0)
    w = 0.8 / st.diag

    def start(ops):
        u, fr, a, c = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size), ops.new_array(lu.size)
        ops.fill_random(u, 1)
        ops.fill_random(fr, 2)
        a.copy_(u)
        c.copy_(u)
        return u, fr, a, c

    u, fr, a, c = start(hip)
    hip.rbgs_colours3(lu.c_struct(), u, a, lf.c_struct(), fr, st, w, 0, b, e)
    hip.rbgs_colours3(lu.c_struct(), a, c, lf.c_struct(), fr, st, w, 1, b, e)
    two = hip.to_host(c)
    u, fr, a, c = start(hip)
    hip.rbgs_sweep_fused(lu.c_struct(), u, a, lf.c_struct(), fr, st, w, 0, b, e)
    hip.rbgs_sweep_fused(lu.c_struct(), a, c, lf.c_struct(), fr, st, w, 0, b, e)
    hip.rbgs_sweep_fused(lu.c_struct(), c, a, lf.c_struct(), fr, st, w, 0, b, e)
    three = hip.to_host(a)
    u, fr, a, c = start(orc)
    for _ in range(3):
        for col in (0, 1):
            orc.stencil_op(SMOOTH, lu.c_struct(), u, lf.c_struct(), fr, lu.c_struct(), u, st, w, col, b, e)
    assert np.array_equal(two, three) and np.array_equal(two, orc.to_host(u))


@pytest.mark.parametrize("kind", ["jacobi3", "colours3_0", "colours3_1"])
def test_three_stage_pass_on_the_product_library(hip, orc, kind):
    """The product library takes the three-stage pass from 8e6 points on: 264 x 200 x 170 (three x windows, ten row groups, four chunks)
    against the oracle's three loops."""
    shape = (264, 200, 170)
    st = laplace_fd(3, tuple(1.0 / n for n in shape), "mp")
    b, e = [1, 1, 1], list(shape)
    lu, lf = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0)
    assert hip.three_stage_eligible(lu.c_struct(), lf.c_struct(), st, b, e)

    def f(ops):
        u, fr, out, tmp = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size), ops.new_array(lu.size)
        ops.fill_random(u, 3)
        ops.fill_random(fr, 4)
        ops.fill_random(out, 5)
        if kind == "jacobi3":
            ops.jacobi3(lu.c_struct(), u, out, tmp, lf.c_struct(), fr, st, 0.8 / st.diag, b, e)
        else:
            ops.rbgs_colours3(lu.c_struct(), u, out, lf.c_struct(), fr, st, 0.8 / st.diag, int(kind[-1]), b, e)
        return [out, u]

    g, c = both(hip, orc, f)
    assert_same(g, c, kind)


def test_three_stage_kernel_chunk_lengths(hipd, orc):
    """The three-step pass with forced z chunks of 5, 8, 33 and 200 planes (debug build): partial last chunks, one chunk for the whole box."""
    import ctypes as C

    shape = (140, 100, 131)
    st = laplace_fd(3, tuple(1.0 / n for n in shape), "mp")
    hipd.L.examg_debug_three_stage.argtypes = [C.c_int] * 2
    b, e = [1, 1, 1], list(shape)

    def f(ops):
        lu, lf = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0)
        u, fr, out, tmp = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size), ops.new_array(lu.size)
        ops.fill_random(u, 77)
        ops.fill_random(fr, 78)
        ops.fill_random(out, 79)
        ops.jacobi3(lu.c_struct(), u, out, tmp, lf.c_struct(), fr, st, 0.8 / st.diag, b, e)
        return [out]

    want = [orc.to_host(t) for t in f(orc)]
    try:
        for zc in (5, 8, 33, 200):
            hipd.L.examg_debug_three_stage(2, zc)
            got = f(hipd)
            hipd.synchronize()
            assert_same([hipd.to_host(t) for t in got], want, "jacobi3, chunks of %d planes" % zc)
    finally:
        hipd.L.examg_debug_three_stage(0, -1)


SMALL_BOXES = [
    ((64, 64, 64), None, None),                         # level 6 of config 3: 63-point rows, 16 x 16 tiles of 4 x 4 rows
    ((32, 32, 32), None, None),                         # level 5
    ((16, 16, 16), None, None),
    ((8, 8, 8), None, None),
    ((4, 4, 4), None, None),                            # 3 x 3 x 3 points: one partly filled tile
    ((62, 30, 22), None, None),                         # ragged tiles in y and z
    ((50, 21, 13), [0, 1, 0], [51, 21, 14]),            # interior faces: loop over duplicate planes, input halo in the ghost layer
    ((40, 20, 20), [3, 2, 5], [38, 17, 18]),            # box inside the inner points, mixed parities
]


@pytest.mark.parametrize("order", ["mp", "pm"])
@pytest.mark.parametrize("first", [0, 1])
@pytest.mark.parametrize("shape,b,e", SMALL_BOXES)
def test_small_level_sweep_bit_exact(hip, orc, shape, b, e, first, order):
    """Rows shorter than 64 points (the launch-bound levels): examg_rbgs_sweep_fused runs k_small_rbgs -- both colour loops of a
    sweep in one launch, staged through LDS (csrc/kernels_small.hip) -- and must leave in u_out, on the box, the bits of the two
    colour loops run one after the other; u_out outside the box and u_in stay as they were."""
    st = laplace_fd(3, tuple(1.0 / s for s in shape), order)
    if b is None:
        b, e = [1, 1, 1], list(shape)
    assert hip.two_stage_eligible(FieldLayout.node(3, shape, 1).c_struct(), FieldLayout.node(3, shape, 0).c_struct(), st, b, e, b, e)
    g = _two_stage_case(hip, "rbgs", shape, st, b, e, first)
    hip.synchronize()
    c = _two_stage_reference(orc, "rbgs", shape, st, b, e, first)
    assert_same([hip.to_host(t) for t in g], c, "small-level sweep")


@pytest.mark.parametrize("first", [0, 1])
@pytest.mark.parametrize("shape,b,e", SMALL_BOXES[:6] + SMALL_BOXES[7:])
def test_small_level_folded_forms_bit_exact(hip, orc, shape, b, e, first):
    """The folded forms on the launch-bound levels: correction + sweep (examg_rbgs_sweep_fused_prolong) and the sweep of the zero field
    (examg_rbgs_sweep_fused_zero) as ONE k_small_rbgs launch each, against the separate loops of the oracle; box values and everything
    outside the box."""
    st = laplace_fd(3, tuple(1.0 / s for s in shape))
    if b is None:
        b, e = [1, 1, 1], list(shape)
    g = _prolong_fold_case(hip, True, "rbgs", shape, st, b, e, first)
    hip.synchronize()
    c = _prolong_fold_case(orc, False, "rbgs", shape, st, b, e, first)
    assert_same([hip.to_host(t) for t in g], [orc.to_host(t) for t in c], "small-level correction + sweep")
    lu, lf = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0, True, False)
    w = 0.8 / st.diag

    def zero_case(ops, fused):
        f, out = ops.new_array(lf.size), ops.new_array(lu.size)
        ops.fill_random(f, 4711)
        if fused:
            ops.rbgs_sweep_fused_zero(lu.c_struct(), out, lf.c_struct(), f, st, w, first, b, e)
        else:
            for col in (first, 1 - first):
                ops.stencil_op(SMOOTH, lu.c_struct(), out, lf.c_struct(), f, lu.c_struct(), out, st, w, col, b, e)
        return [out]

    g = zero_case(hip, True)
    hip.synchronize()
    assert_same([hip.to_host(t) for t in g], [orc.to_host(t) for t in zero_case(orc, False)], "small-level sweep of the zero field")


@pytest.mark.parametrize("order", ["mp", "pm"])
@pytest.mark.parametrize("shape,scale", [((64, 64, 64), 1.0), ((32, 32, 32), 1.0), ((16, 16, 16), 4.0), ((8, 8, 8), 1.0), ((60, 28, 20), 1.0)])
def test_small_level_residual_restrict_bit_exact(hip, orc, shape, scale, order):
    """Coarse rows shorter than 32 points: examg_residual_restrict runs k_small_residual_restrict (one coarse point per thread, the 27
    residuals recomputed, none stored) -- the oracle's residual loop + restriction loop, bit for bit."""
    st = laplace_fd(3, tuple(1.0 / s for s in shape), order)
    cs = tuple(s // 2 for s in shape)
    lu, lf, lc = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0, True, False), FieldLayout.node(3, cs, 0, True, False)
    fb, fe, cb, ce = [1, 1, 1], list(shape), [1, 1, 1], list(cs)
    import ctypes as C

    from exastencils_amd.lib import ivec

    sc = st.c_struct(hip.ptr)
    assert hip.L.examg_residual_restrict_one_pass(C.byref(lu.c_struct()), C.byref(lf.c_struct()), C.byref(sc), C.byref(lc.c_struct()), ivec(fb), ivec(fe),
                                                  ivec(cb), ivec(ce)) == 1

    def f(ops):
        u, fr, r, fc = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size), ops.new_array(lc.size)
        ops.fill_random(u, 21)
        ops.fill_random(fr, 22)
        ops.fill_random(fc, 23)
        ops.residual_restrict(lu.c_struct(), u, lf.c_struct(), fr, lu.c_struct(), r, st, lc.c_struct(), fc, scale, fb, fe, cb, ce)
        return [fc]

    g, c = both(hip, orc, f)
    assert_same(g, c, "small-level residual + restriction")


@pytest.mark.parametrize("kind", ["rbgs", "jacobi2"])
def test_two_stage_interior_faces_anisotropic(orc, two_stage_variant, kind):
    """Block with neighbours on some faces: the loop includes the duplicate planes (begin 0 / end n+1), so the
    two-point input halo reaches the ghost layer and beyond the allocation (guarded)."""
    hip = two_stage_variant
    shape = (150, 36, 20)
    st = laplace_unit(3)
    b, e = [0, 1, 0], [151, 36, 21]
    g = _two_stage_case(hip, kind, shape, st, b, e)
    hip.synchronize()
    c = _two_stage_reference(orc, kind, shape, st, b, e)
    assert_same([hip.to_host(t) for t in g], c, "two-stage faces " + kind)


def _prolong_fold_case(ops, fused, kind, shape, st, b, e, first=0, align=0):
    """`u += P uc` on the box, then one red-black sweep / two Jacobi steps: one call (fused) or the separate loops."""
    lu, lf = FieldLayout.node(3, shape, 1, True, True, align), FieldLayout.node(3, shape, 0, True, False, align)
    lc = FieldLayout.node(3, tuple(s // 2 for s in shape), 1, True, True, align)
    u, f, out, tmp, uc = (ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size), ops.new_array(lu.size),
                          ops.new_array(lc.size))
    ops.fill_random(u, 12345)
    ops.fill_random(f, 4711)
    ops.fill_random(out, 5)
    ops.fill_random(uc, 99)
    w = 0.8 / st.diag
    L, Fl, Lc = lu.c_struct(), lf.c_struct(), lc.c_struct()
    if fused:
        if kind == "rbgs":
            ops.rbgs_sweep_fused_prolong(L, u, out, Fl, f, st, w, first, b, e, Lc, uc)
        else:
            ops.jacobi2_prolong(L, u, out, tmp, Fl, f, st, w, b, e, Lc, uc)
        return [out, u]
    work = u.clone()
    ops.prolong_add(Lc, uc, L, work, b, e)
    if kind == "rbgs":
        for c in (first, 1 - first):
            ops.stencil_op(SMOOTH, L, work, Fl, f, L, work, st, w, c, b, e)
    else:
        tmp.copy_(work)
        ops.stencil_op(SMOOTH, L, work, Fl, f, L, tmp, st, w, -1, b, e)
        ops.stencil_op(SMOOTH, L, tmp, Fl, f, L, work, st, w, -1, b, e)
    ops.axpby(L, work, L, out, 1.0, 0.0, b, e)      # box only
    return [out, u]


@pytest.mark.parametrize("kind", ["rbgs", "jacobi2"])
@pytest.mark.parametrize("shape,b,e,first,align", [
    ((66, 66, 66), None, None, 0, 0),
    ((130, 130, 130), None, None, 1, 0),
    ((200, 200, 200), None, None, 0, 0),
    ((256, 256, 64), None, None, 0, 16),                     # padded layout: aligned windows start one point further left
    ((150, 36, 20), [0, 1, 0], [151, 36, 21], 0, 0),         # loop over duplicate planes: even window / row / plane starts
    ((128, 40, 38), [1, 2, 3], [126, 39, 37], 1, 0),         # box inside the inner points, mixed parities
    ((40, 20, 20), None, None, 0, 0),                        # short rows: the copy + plain loops path
])
def test_prolongation_folded_into_the_pass_bit_exact(orc, two_stage_variant, kind, shape, b, e, first, align):
    """examg_rbgs_sweep_fused_prolong / examg_jacobi2_prolong == correction loop + the smoother loops one after the other on
    the CPU, bit for bit (same interpolation terms in the same order); u_in untouched, u_out untouched outside the box."""
    hip = two_stage_variant
    st = laplace_fd(3, tuple(1.0 / s for s in shape))
    if b is None:
        b, e = [1, 1, 1], list(shape)
    g = _prolong_fold_case(hip, True, kind, shape, st, b, e, first, align)
    hip.synchronize()
    c = _prolong_fold_case(orc, False, kind, shape, st, b, e, first, align)
    got, want = [hip.to_host(t) for t in g], [orc.to_host(t) for t in c]
    if shape[0] < 64:     # the copy path also brings u_in's values to the box's one-point shell (include/examg.h): compare the box
        lu = FieldLayout.node(3, shape, 1, True, True, align)
        sl = tuple(slice(b[d] + 1, e[d] + 1) for d in (2, 1, 0))
        got[0], want[0] = got[0].reshape(lu.shape_zyx)[sl], want[0].reshape(lu.shape_zyx)[sl]
    assert_same(got, want, "prolongation fold " + kind)


@pytest.mark.parametrize("n,first,order", [(66, 0, "mp"), (130, 1, "pm"), (200, 0, "mp"), (40, 0, "mp")])
def test_sweep_of_the_zero_field_bit_exact(orc, two_stage_variant, n, first, order):
    """examg_rbgs_sweep_fused_zero (the input field is the constant 0.0, nothing is loaded for it) == the two colour loops on a
    zeroed array; 40^3 takes the zero + plain loops path."""
    hip = two_stage_variant
    st = laplace_fd(3, (1.0 / n,) * 3, order)
    b, e = box(3, n)
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
    w = 0.8 / st.diag

    def run(ops, fused):
        f, out = ops.new_array(lf.size), ops.new_array(lu.size)
        ops.fill_random(f, 4711)
        if fused:
            ops.rbgs_sweep_fused_zero(lu.c_struct(), out, lf.c_struct(), f, st, w, first, b, e)
        else:
            for c in (first, 1 - first):
                ops.stencil_op(SMOOTH, lu.c_struct(), out, lf.c_struct(), f, lu.c_struct(), out, st, w, c, b, e)
        return [out]

    g = run(hip, True)
    hip.synchronize()
    c = run(orc, False)
    assert_same([hip.to_host(t) for t in g], [orc.to_host(t) for t in c], "sweep of the zero field")


@pytest.mark.parametrize("shape,b,e,b2,e2", [
    ((150, 36, 40), [0, 1, 0], [151, 36, 41], [1, 1, 1], [150, 36, 40]),     # neighbours in x and z: both dup planes excluded
    ((130, 70, 33), [1, 0, 1], [130, 71, 33], [1, 1, 1], [130, 71, 33]),     # lower y neighbour only
    ((40, 20, 20), [0, 1, 1], [41, 20, 20], [1, 1, 1], [41, 20, 20]),        # small rows: fallback path
])
def test_jacobi2_boxes_bit_exact(orc, two_stage_variant, shape, b, e, b2, e2):
    """Two Jacobi steps with the first step on the loop's box and the second on the box without the duplicate planes at
    interior faces (what a block with neighbours runs, exastencils_amd/smoothers.py)."""
    hip = two_stage_variant
    st = laplace_unit(3)

    def f(ops):
        lu, lf = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0)
        u, fr, out, tmp = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size), ops.new_array(lu.size)
        ops.fill_random(u, 12345)
        ops.fill_random(fr, 4711)
        ops.fill_random(out, 5)
        ops.jacobi2_boxes(lu.c_struct(), u, out, tmp, lf.c_struct(), fr, st, 0.8 / st.diag, b, e, b2, e2)
        return [out]

    g, c = both(hip, orc, f)
    assert_same(g, c, "jacobi2_boxes")


@pytest.mark.parametrize("first", [0, 1])
@pytest.mark.parametrize("shape,b1,e1,b2,e2", [
    ((150, 36, 40), [1, 1, 1], [150, 36, 40], [2, 1, 2], [149, 36, 39]),     # neighbours in x and z
    ((130, 70, 33), [1, 1, 1], [130, 71, 33], [1, 2, 1], [130, 71, 33]),     # lower y neighbour only
    ((40, 20, 20), [1, 1, 1], [41, 20, 20], [2, 1, 1], [41, 20, 20]),        # short rows: fallback path
])
def test_rbgs_sweep_fused_boxes_bit_exact(orc, two_stage_variant, shape, b1, e1, b2, e2, first):
    """Red-black sweep of a block with neighbours: first colour on the box shrunk by one point at interior faces, second
    colour on the box shrunk by two; u_out is written on the second box only (exastencils_amd/smoothers.py: rbgs_sweep)."""
    hip = two_stage_variant
    st = laplace_fd(3, tuple(1.0 / s_ for s_ in shape))

    def f(ops):
        lu, lf = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0)
        u, fr, out, tmp = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size), ops.new_array(lu.size)
        ops.fill_random(u, 12345)
        ops.fill_random(fr, 4711)
        ops.fill_random(out, 5)
        ops.rbgs_sweep_fused_boxes(lu.c_struct(), u, out, tmp, lf.c_struct(), fr, st, 0.8 / st.diag, first, b1, e1, b2, e2)
        return [out]

    g, c = both(hip, orc, f)
    assert_same(g, c, "rbgs_sweep_fused_boxes")


def test_two_stage_fallback_small_and_2d(hip, orc):
    """Boxes the fused kernel does not take (rows < 64 points) go through copy + two loops: same result."""
    n = 24
    st = laplace_fd(3, (1.0 / n,) * 3)
    b, e = box(3, n)
    for kind in ("rbgs", "jacobi2"):
        g = _two_stage_case(hip, kind, (n, n, n), st, b, e)
        hip.synchronize()
        c = _two_stage_reference(orc, kind, (n, n, n), st, b, e)
        got = [hip.to_host(t) for t in g]
        lu = FieldLayout.node(3, (n, n, n), 1)
        sl = (slice(2, n + 1),) * 3          # the box (array index = iterator + 1)
        assert np.array_equal(got[0].reshape(lu.shape_zyx)[sl], c[0].reshape(lu.shape_zyx)[sl]), kind


def test_generic_path_equals_fast_path(hipd, orc):
    hip = hipd
    n = 96
    st = laplace_fd(3, (1.0 / n,) * 3)
    b, e = box(3, n)
    fast = _stencil_case(hip, 3, (n, n, n), st, SMOOTH, -1, b, e)
    old = hip.L.examg_debug_force_generic(1)
    try:
        gen = _stencil_case(hip, 3, (n, n, n), st, SMOOTH, -1, b, e)
    finally:
        hip.L.examg_debug_force_generic(old)
    hip.synchronize()
    assert_same([hip.to_host(fast[0])], [hip.to_host(gen[0])], "generic vs fast")
    c = _stencil_case(orc, 3, (n, n, n), st, SMOOTH, -1, b, e)
    assert_same([hip.to_host(gen[0])], [orc.to_host(c[0])], "generic vs oracle")


@pytest.mark.parametrize("mode", [APPLY, RESIDUAL, SMOOTH])
@pytest.mark.parametrize("colour", [-1, 0, 1])
def test_stencil_2d_5point(hip, orc, mode, colour):
    if colour >= 0 and mode != SMOOTH:
        pytest.skip("colours only with in-place smoothing")
    n = 257
    st = laplace_fd(2, (1.0 / n, 1.0 / n, 0.0))
    b, e = box(2, n)
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 2, (n, n, 0), st, mode, colour, b, e))
    assert_same(g, c, "5-point")


@pytest.mark.parametrize("mode", [APPLY, RESIDUAL, SMOOTH])
@pytest.mark.parametrize("n", [65, 130])
def test_stencil_field_7_entries_fast_path(hip, orc, mode, n):
    """Rows >= 64 points take the z-march stencil-field kernel; ragged tiles; and a block with interior faces."""
    st = Stencil(stencil_field_offsets(3), [])
    b, e = box(3, n)
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, (n, n, n), st, mode, -1, b, e, cfn=True))
    assert_same(g, c, "stencil field fast path")
    b, e = [0, 1, 0], [n + 1, n, n + 1]
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, (n, n, n), st, mode, -1, b, e, cfn=True))
    assert_same(g, c, "stencil field fast path, interior faces")


@pytest.mark.parametrize("mode", [APPLY, RESIDUAL, SMOOTH])
def test_stencil_field_7_entries(hip, orc, mode):
    n = 48
    st = Stencil(stencil_field_offsets(3), [])
    b, e = box(3, n)
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, (n, n, n), st, mode, -1, b, e, cfn=True))
    assert_same(g, c, "stencil field")


def test_stencil_27_entries_const_and_field(hip, orc):
    n = 40
    offs = [(0, 0, 0)] + [(a, b_, c_) for a in (-1, 0, 1) for b_ in (-1, 0, 1) for c_ in (-1, 0, 1) if (a, b_, c_) != (0, 0, 0)]
    co = [26.0] + [-1.0 / (1 + abs(o[0]) + abs(o[1]) + abs(o[2])) for o in offs[1:]]
    st = Stencil(offs, co)
    b, e = box(3, n)
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, (n, n, n), st, SMOOTH, -1, b, e))
    assert_same(g, c, "27-point const")
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, (n, n, n), Stencil(offs, []), SMOOTH, -1, b, e, cfn=True))
    assert_same(g, c, "27-entry stencil field")


@pytest.mark.parametrize("mode", [APPLY, RESIDUAL, SMOOTH])
@pytest.mark.parametrize("n", [66, 130])
def test_stencil_field_27_entries_long_rows(hip, orc, mode, n):
    """27-entry stencil field in the entry order of examg_init_helmholtz27 on long rows; also a block with interior
    faces (loop starts on the duplicate plane, edge and corner ghosts are read)."""
    from exastencils_amd.field import helmholtz27_offsets

    st = Stencil(helmholtz27_offsets(), [])
    b, e = box(3, n)
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, (n, n, n), st, mode, -1, b, e, cfn=True))
    assert_same(g, c, "27-entry stencil field fast path")
    b, e = [0, 1, 0], [n + 1, n, n + 1]
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, (n, n, n), st, mode, -1, b, e, cfn=True))
    assert_same(g, c, "27-entry stencil field fast path, interior faces")


@pytest.mark.parametrize("mode", [APPLY, RESIDUAL, SMOOTH])
@pytest.mark.parametrize("case", [((66, 66, 66), [1, 1, 1], [66, 66, 66]), ((130, 40, 20), [0, 1, 0], [131, 40, 21]),
                                  ((200, 24, 12), [1, 1, 1], [200, 24, 12]),
                                  # a box that holds the LAST allocated point of the coefficient layout (no ghost / pad layers behind it):
                                  # the record kernel's clamped 16-byte loads would shift its last entry -- the dispatch must not take it
                                  ((70, 12, 9), [0, 0, 0], [71, 13, 10])])
def test_stencil_field_27_entries_under_the_entry_fastest_layout_transformation(hip, orc, mode, case):
    """Config 4's operator with its coefficient field transformed by `[x, y, z, i] => [i, x, y, z]` (the reference's
    LayoutTransformations mechanism, Compiler/src/exastencils/layoutTransformation/): one 216-byte record per point, read as ONE
    stream and transposed through LDS (k_stencilfield27_rec).  Bit-identical to the oracle on the untransformed planes."""
    from exastencils_amd.field import helmholtz27_offsets

    shape, b, e = case
    st = Stencil(helmholtz27_offsets(), [])
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 3, shape, st, mode, -1, b, e, cfn=True, entry_fastest=True))
    assert_same(g, c, "27-entry stencil field, entry-fastest coefficients")


def _sf27_pair_case(ops, shape, b1, e1, b2, e2, kind, entry_fastest=True, cghost=0):
    """Two Jacobi steps (kind 'pair': step 1 on box 1, step 2 on box 2) or one step + residual ('residual') on a 27-entry stencil field
    with random positive-diagonal coefficients; one call on the GPU, the separate loops through the oracle."""
    from exastencils_amd.field import helmholtz27_offsets

    lu, lf = FieldLayout.node(3, shape, 1 + cghost), FieldLayout.node(3, shape, cghost)
    u, f, out, tmp, res = (ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size), ops.new_array(lu.size),
                           ops.new_array(lu.size))
    ops.fill_random(u, 12345)
    ops.fill_random(f, 4711)
    ops.fill_random(out, 5)
    ops.fill_random(res, 6)
    cf = ops.new_array(27 * lf.size)
    ops.fill_random(cf, 99)
    cf[:lf.size] += 8.0                      # diagonal entry (planes layout: entry 0 first)
    st = Stencil(helmholtz27_offsets(), [], cf, lf)
    if entry_fastest:
        st = st.entry_fastest(ops)
    L, Fl = lu.c_struct(), lf.c_struct()
    if kind == "pair":
        if not entry_fastest:                # the separate loops: step 1 on box 1 into a copy of u (points outside keep u), step 2 on box 2
            tmp.copy_(u)
            ops.stencil_op(SMOOTH, L, u, Fl, f, L, tmp, st, 0.8, -1, b1, e1)
            ops.stencil_op(SMOOTH, L, tmp, Fl, f, L, out, st, 0.8, -1, b2, e2)
        elif (b1, e1) == (b2, e2):
            ops.jacobi2(L, u, out, tmp, Fl, f, st, 0.8, b2, e2)
        else:
            ops.jacobi2_boxes(L, u, out, tmp, Fl, f, st, 0.8, b1, e1, b2, e2)
        # compare on box 2 only (outside it the fallback and the pass leave different things)
        keep = ops.new_array(lu.size)
        ops.axpby(L, out, L, keep, 1.0, 0.0, b2, e2)
        return [keep, u]
    out.copy_(u)       # the residual reads the shell of the box from u_out: both arrays hold the same boundary / ghost values
    if not entry_fastest:
        ops.stencil_op(SMOOTH, L, u, Fl, f, L, out, st, 0.8, -1, b2, e2)
        ops.stencil_op(RESIDUAL, L, out, Fl, f, L, res, st, 0.0, -1, b2, e2)
    else:
        ops.jacobi_residual(L, u, out, Fl, f, L, res, st, 0.8, b2, e2)
    ko, kr = ops.new_array(lu.size), ops.new_array(lu.size)
    ops.axpby(L, out, L, ko, 1.0, 0.0, b2, e2)
    ops.axpby(L, res, L, kr, 1.0, 0.0, b2, e2)
    return [ko, kr, u]


SF27_PAIR_BOXES = [
    ((66, 66, 66), [1, 1, 1], [66, 66, 66], [1, 1, 1], [66, 66, 66], 0),       # two x windows, ragged row groups
    ((130, 40, 70), [1, 1, 1], [130, 40, 70], [1, 1, 1], [130, 40, 70], 0),    # three windows
    ((130, 40, 20), [0, 1, 0], [131, 40, 21], [0, 1, 0], [131, 40, 21], 0),    # interior faces: the halo reaches the ghost layers (one row per wave only)
    ((200, 24, 36), [0, 0, 0], [201, 25, 37], [1, 1, 1], [200, 24, 36], 0),    # separate stage boxes (block with neighbours)
    ((130, 60, 70), [1, 1, 1], [130, 60, 70], [1, 1, 1], [130, 60, 70], 0),    # two rows per wave: last window moved left, tiles of 14 + 14 + 14 + 14 + 3 rows
    ((136, 47, 9), [0, 0, 0], [137, 48, 10], [1, 1, 1], [136, 47, 9], 1),      # two rows per wave, separate stage boxes, coefficients with ghost layers
    ((64, 48, 5), [2, 1, 1], [63, 47, 4], [2, 3, 1], [63, 47, 4], 0),          # one window, boxes off the array edges
]


@pytest.mark.parametrize("kind", ["pair", "residual"])
@pytest.mark.parametrize("rows,zc", [(1, 0), (1, 7), (2, 0), (2, 6)], ids=["one-row", "one-row-chunks", "two-rows", "two-rows-chunks"])
@pytest.mark.parametrize("shape,b1,e1,b2,e2,cghost", SF27_PAIR_BOXES)
def test_two_jacobi_steps_on_a_27_entry_field_in_one_pass(hipd, orc, kind, rows, zc, shape, b1, e1, b2, e2, cghost):
    """Temporal blocking on config 4's operator (csrc/kernels_sf27pair.hip): both steps of a point share its 216 B of coefficients.
    Same 27 products in the same order as the one-step loops: bit-identical to running them one after the other (oracle).  Both kernels
    (one row per wave / two rows per wave with the records by LDS-DMA; the second keeps to the first where a box reaches the first or last
    array column), forced through the debug build at sizes the product library leaves to two launches, with and without short z chunks."""
    import ctypes as C

    if kind == "residual" and (b1, e1) != (b2, e2):
        pytest.skip("one step + residual has a single box")
    hipd.L.examg_debug_sf27_pair.argtypes = [C.c_int, C.c_int]
    hipd.L.examg_debug_sf27_pair(10 + rows, zc)
    try:
        g = _sf27_pair_case(hipd, shape, b1, e1, b2, e2, kind, cghost=cghost)
        hipd.synchronize()
    finally:
        hipd.L.examg_debug_sf27_pair(1, 0)
    c = _sf27_pair_case(orc, shape, b1, e1, b2, e2, kind, entry_fastest=False, cghost=cghost)
    assert_same([hipd.to_host(t) for t in g], [orc.to_host(t) for t in c], "27-entry pair " + kind)


@pytest.mark.parametrize("kind", ["pair", "residual"])
@pytest.mark.parametrize("shape,b1,e1,b2,e2,cghost", [
    ((130, 100, 90), [1, 1, 1], [130, 100, 90], [1, 1, 1], [130, 100, 90], 0),     # 1.14 * 10^6 points: the product library takes the pass
    ((136, 100, 84), [0, 0, 0], [137, 101, 85], [1, 1, 1], [136, 100, 84], 1),     # ... with separate stage boxes
])
def test_two_jacobi_steps_on_a_27_entry_field_product_library(hip, orc, kind, shape, b1, e1, b2, e2, cghost):
    """The same pass as the product library launches it (from 2^20 points; its own choice of kernel and chunk length)."""
    if kind == "residual" and (b1, e1) != (b2, e2):
        pytest.skip("one step + residual has a single box")
    g = _sf27_pair_case(hip, shape, b1, e1, b2, e2, kind, cghost=cghost)
    hip.synchronize()
    c = _sf27_pair_case(orc, shape, b1, e1, b2, e2, kind, entry_fastest=False, cghost=cghost)
    assert_same([hip.to_host(t) for t in g], [orc.to_host(t) for t in c], "27-entry pair " + kind)


@pytest.mark.parametrize("nd,shape", [(3, (70, 20, 12)), (3, (24, 10, 9)), (2, (80, 33, 0))])
def test_stencil_field_entry_fastest_transformation_other_entry_counts(hip, orc, nd, shape):
    """2d+1-entry stencil fields under the same transformation (generic kernel), all three loop kinds and a coloured half sweep."""
    st = Stencil(stencil_field_offsets(nd), [])
    b = [1 if d < nd else 0 for d in range(3)]
    e = [shape[d] if d < nd else 1 for d in range(3)]
    for mode in (APPLY, RESIDUAL, SMOOTH):
        g, c = both(hip, orc, lambda ops: _stencil_case(ops, nd, shape, st, mode, -1, b, e, cfn=True, entry_fastest=True))
        assert_same(g, c, "%d-entry stencil field, entry-fastest, mode %d" % (2 * nd + 1, mode))
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, nd, shape, st, SMOOTH, 1, b, e, cfn=True, entry_fastest=True))
    assert_same(g, c, "entry-fastest, half sweep")


@pytest.mark.parametrize("K,shape", [(7, (70, 20, 12)), (27, (130, 24, 10))])
@pytest.mark.parametrize("entry_fastest", [False, True])
def test_stencil_field_smoother_weight_written_as_omega_over_diag(hip, orc, K, shape, entry_fastest):
    """`Solution += (0.8 / diag(A)) * (RHS - A * Solution)` on a stencil field (Testing/PolyExpl/RBGS3Dvc.exa4:52): the weight is evaluated
    per point as w / c_diag (EXAMG_WEIGHT_DIVIDE), not as (1.0 / c_diag) * w -- bit-identical to the oracle's loop with the same form, and
    different from the other form somewhere (the two round differently)."""
    import dataclasses

    from exastencils_amd.field import helmholtz27_offsets

    offs = helmholtz27_offsets() if K == 27 else stencil_field_offsets(3)
    b, e = [1, 1, 1], list(shape)

    def case(ops, wform):
        lu, lf = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0)
        u, f, dst = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size)
        ops.fill_random(u, 12345)
        ops.fill_random(f, 4711)
        cf = ops.new_array(K * lf.size)
        ops.fill_random(cf, 99)
        cf += 3.0
        st = Stencil(offs, [], cf, lf)
        if entry_fastest:
            st = st.entry_fastest(ops)
        st = dataclasses.replace(st, wform=wform)
        ops.stencil_op(SMOOTH, lu.c_struct(), u, lf.c_struct(), f, lu.c_struct(), dst, st, 0.8, -1, b, e)
        return [dst]

    g, c = both(hip, orc, lambda ops: case(ops, 1))
    assert_same(g, c, "omega / diag(A)")
    g0 = [hip.to_host(t) for t in case(hip, 0)]
    assert not np.array_equal(g[0], g0[0])


def test_stencil_field_transformation_round_trip(hip):
    l = FieldLayout.node(3, (37, 11, 6), 0)
    for K in (7, 27):
        a = hip.new_array(K * l.size)
        hip.fill_random(a, 5)
        b_, c_ = hip.new_array(K * l.size), hip.new_array(K * l.size)
        hip.transform_stencilfield(l.c_struct(), K, a, b_, True)
        hip.transform_stencilfield(l.c_struct(), K, b_, c_, False)
        ah, bh = hip.to_host(a).reshape(K, l.size), hip.to_host(b_).reshape(l.size, K)
        assert np.array_equal(bh, ah.T) and np.array_equal(hip.to_host(c_), hip.to_host(a))


def test_empty_iteration_space_is_a_noop(hip, orc):
    """minLevel 0 on one fragment: `loop over` has no inner points (Examples/Poisson/2D_FD_Poisson_fromL4.knowledge:3)."""
    st = laplace_fd(2, (1.0, 1.0, 0.0))
    g, c = both(hip, orc, lambda ops: _stencil_case(ops, 2, (1, 1, 0), st, SMOOTH, -1, [1, 1, 0], [1, 1, 1]))
    assert_same(g, c, "empty")


@pytest.mark.parametrize("nd,n", [(3, 64), (3, 50), (3, 130), (3, 200), (2, 256)])
@pytest.mark.parametrize("scale", [1.0, 4.0])
def test_restrict_and_prolong_bit_exact(hip, orc, nd, n, scale):
    shape_f = tuple(n if d < nd else 0 for d in range(3))
    shape_c = tuple(n // 2 if d < nd else 0 for d in range(3))

    def f(ops):
        lfi, lco = FieldLayout.node(nd, shape_f, 1), FieldLayout.node(nd, shape_c, 1)
        lrh = FieldLayout.node(nd, shape_c, 0)
        r, fc, uc, uf = ops.new_array(lfi.size), ops.new_array(lrh.size), ops.new_array(lco.size), ops.new_array(lfi.size)
        ops.fill_random(r, 1)
        ops.fill_random(uc, 2)
        ops.fill_random(uf, 3)
        bc, ec = box(nd, n // 2)
        ops.restrict(lfi.c_struct(), r, lrh.c_struct(), fc, scale, bc, ec)
        bf, ef = box(nd, n)
        ops.prolong_add(lco.c_struct(), uc, lfi.c_struct(), uf, bf, ef)
        return [fc, uf]

    g, c = both(hip, orc, f)
    assert_same(g, c, "transfer")


@pytest.mark.parametrize("order", ["mp", "pm"])
@pytest.mark.parametrize("shape,scale,align", [((64, 64, 64), 1.0, 0), ((130, 130, 130), 1.0, 0), ((200, 200, 200), 4.0, 0),
                                               ((256, 72, 44), 1.0, 16), ((140, 396, 36), 1.0, 0), ((134, 70, 40), 1.0, 0),
                                               ((152, 50, 36), 1.0, 0), ((182, 44, 36), 1.0, 0), ((262, 40, 36), 1.0, 0)])
def test_residual_restrict_fused_bit_exact(hip, orc, shape, scale, align, order):
    """`Residual = RHS - A * Solution` + restriction in one pass (fine residual never stored) against the oracle's two loops;
    64^3: short coarse rows, the two-kernel path through the residual array; anisotropic blocks, a padded layout, the
    restriction scaled by 4 (the generated-from-L3 programs).  Coarse rows of 64 / 127 / 69 / 66 / 75 / 90 / 130 points: the columns
    that the 63-point tiles leave over run in narrow windows of 2 / 2 / 8 / 4 / 16 / 32 / 8 lanes (several row groups side by side
    in a wave, the last wave partly empty); 99 points: 36 left over, a whole tile."""
    st = laplace_fd(3, tuple(1.0 / s for s in shape), order)

    def f(ops):
        cs = tuple(s // 2 for s in shape)
        lu, lf, lc = (FieldLayout.node(3, shape, 1, True, True, align), FieldLayout.node(3, shape, 0, True, False, align),
                      FieldLayout.node(3, cs, 0, True, False, align))
        u, fr, r, fc = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size), ops.new_array(lc.size)
        ops.fill_random(u, 21)
        ops.fill_random(fr, 22)
        ops.fill_random(fc, 23)          # the coarse array outside the restriction's box must survive
        fb, fe = [1, 1, 1], list(shape)
        cb, ce = [1, 1, 1], list(cs)
        ops.residual_restrict(lu.c_struct(), u, lf.c_struct(), fr, lu.c_struct(), r, st, lc.c_struct(), fc, scale, fb, fe, cb, ce)
        return [fc]

    g, c = both(hip, orc, f)
    assert_same(g, c, "residual_restrict")


@pytest.mark.parametrize("rows", [1, 2])
@pytest.mark.parametrize("shape", [(134, 70, 40), (256, 62, 44), (152, 38, 36), (182, 44, 20)])
def test_residual_restrict_narrow_windows_forced_variants(hipd, orc, shape, rows):
    """k_residual_restrict3 on the debug build: one and TWO coarse rows per wave (the product takes two from 8e6 coarse points on; an odd
    count of coarse rows leaves the last row group with one) x narrow windows / whole tile for the left-over columns / x tiles fastest --
    each against the oracle's two loops."""
    import ctypes as C

    st = laplace_fd(3, tuple(1.0 / s for s in shape), "mp")
    L = hipd.L
    L.examg_debug_residual_restrict.argtypes = [C.c_int] * 2
    L.examg_debug_rr_order.argtypes = [C.c_int]

    def f(ops):
        cs = tuple(s // 2 for s in shape)
        lu, lf, lc = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0, True, False), FieldLayout.node(3, cs, 0, True, False)
        u, fr, r, fc = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size), ops.new_array(lc.size)
        ops.fill_random(u, 41)
        ops.fill_random(fr, 42)
        ops.fill_random(fc, 43)
        ops.residual_restrict(lu.c_struct(), u, lf.c_struct(), fr, lu.c_struct(), r, st, lc.c_struct(), fc, 1.0, [1, 1, 1], list(shape), [1, 1, 1], list(cs))
        return [fc]

    want = [orc.to_host(t) for t in f(orc)]
    try:
        L.examg_debug_residual_restrict(0, (1000 if rows == 2 else 2000) + 8)
        for order in (0, 2, 1):
            L.examg_debug_rr_order(order)
            got = f(hipd)
            hipd.synchronize()
            assert_same([hipd.to_host(t) for t in got], want, "residual_restrict, %d rows per wave, order %d" % (rows, order))
    finally:
        L.examg_debug_rr_order(0)
        L.examg_debug_residual_restrict(0, 8)


@pytest.mark.parametrize("shape,order", [((130, 130, 130), "mp"), ((200, 72, 44), "pm"), ((256, 256, 64), "mp"), ((40, 40, 40), "mp")])
def test_residual_norm_in_one_pass(hip, orc, shape, order):
    """examg_residual_norm2 (the residual's squares summed where the residual loop would store it) against the oracle's
    residual loop + reduction loop: 1e-13 relative (summation order), the same bits on a second call; 40^3 takes the
    two-kernel path through the residual array; the box leaves out the upper planes like a reduction box with neighbours."""
    st = laplace_fd(3, tuple(1.0 / s for s in shape), order)
    lu, lf = FieldLayout.node(3, shape, 1), FieldLayout.node(3, shape, 0, True, False)
    b, e = [1, 1, 1], [shape[0], shape[1] - 1, shape[2]]

    def f(ops):
        u, fr, r = ops.new_array(lu.size), ops.new_array(lf.size), ops.new_array(lu.size)
        ops.fill_random(u, 31)
        ops.fill_random(fr, 32)
        return u, fr, r

    u, fr, r = f(hip)
    got = [hip.scalar_value(hip.residual_norm2(lu.c_struct(), u, lf.c_struct(), fr, st, b, e, lu.c_struct(), r)) for _ in range(2)]
    u, fr, r = f(orc)
    orc.stencil_op(RESIDUAL, lu.c_struct(), u, lf.c_struct(), fr, lu.c_struct(), r, st, 0.0, -1, b, e)
    want = orc.scalar_value(orc.dot(lu.c_struct(), r, lu.c_struct(), r, b, e))
    assert got[0] == got[1]
    assert abs(got[0] - want) <= 1e-13 * want, (got, want)


def test_restrict_on_a_block_with_interior_faces(hip, orc):
    """Coarse loop bounds of a block with neighbours (iteration offsets 0): the fine footprint reaches the ghost layers;
    long rows take the wide kernel, a partial last tile included."""
    n = 136

    def f(ops):
        lfi, lrh = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n // 2,) * 3, 0)
        r, fc = ops.new_array(lfi.size), ops.new_array(lrh.size)
        ops.fill_random(r, 11)
        ops.restrict(lfi.c_struct(), r, lrh.c_struct(), fc, 1.0, [0, 1, 0], [n // 2 + 1, n // 2, n // 2 + 1])
        return [fc]

    g, c = both(hip, orc, f)
    assert_same(g, c, "restrict, interior faces")


def test_blas1_forms_bit_exact(hip, orc):
    n = 40

    def f(ops):
        lx, ly = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
        outs = []
        b, e = box(3, n)
        for a, b_ in [(1.0, 0.0), (2.5, 0.0), (0.37, 1.0), (-0.37, 1.0), (1.0, 0.81), (1.5, -0.25)]:
            x, y = ops.new_array(lx.size), ops.new_array(ly.size)
            ops.fill_random(x, 5)
            ops.fill_random(y, 6)
            ops.axpby(lx.c_struct(), x, ly.c_struct(), y, a, b_, b, e)
            outs.append(y)
        z = ops.new_array(lx.size)
        ops.fill_random(z, 7)
        ops.set(lx.c_struct(), z, 0.0, b, e)
        outs.append(z)
        return outs

    g, c = both(hip, orc, f)
    assert_same(g, c, "blas1")


def test_axpby_dev_matches_host_scalars(hip):
    n = 32
    lx = FieldLayout.node(3, (n, n, n), 1)
    b, e = box(3, n)
    x, y1, y2 = hip.new_array(lx.size), hip.new_array(lx.size), hip.new_array(lx.size)
    hip.fill_random(x, 5)
    hip.fill_random(y1, 6)
    hip.fill_random(y2, 6)
    num, den = hip.from_host(np.array([3.7])), hip.from_host(np.array([1.9]))
    hip.axpby(lx.c_struct(), x, lx.c_struct(), y1, -(3.7 / 1.9), 1.0, b, e)
    hip.axpby_dev(lx.c_struct(), x, lx.c_struct(), y2, 0.0, 1.0, 0, -1.0, num, den, b, e)
    hip.synchronize()
    assert np.array_equal(hip.to_host(y1), hip.to_host(y2))
    hip.axpby(lx.c_struct(), x, lx.c_struct(), y1, 1.0, 3.7 / 1.9, b, e)
    hip.axpby_dev(lx.c_struct(), x, lx.c_struct(), y2, 1.0, 0.0, 1, 1.0, num, den, b, e)
    hip.synchronize()
    assert np.array_equal(hip.to_host(y1), hip.to_host(y2))


@pytest.mark.parametrize("n", [16, 64, 129])
def test_reductions(hip, orc, n):
    def f(ops):
        lx, ly = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
        x, y = ops.new_array(lx.size), ops.new_array(ly.size)
        ops.fill_random(x, 11)
        ops.fill_random(y, 12)
        b, e = box(3, n)
        d1 = ops.dot(lx.c_struct(), x, lx.c_struct(), x, b, e)
        d2 = ops.dot(lx.c_struct(), x, ly.c_struct(), y, b, e)
        m = ops.max_err_fn(lx.c_struct(), x, geom(3, n), FN_POLY3D, (), b, e)
        d0 = ops.dot(lx.c_struct(), x, lx.c_struct(), x, [1, 1, 1], [1, 1, 1])
        return [d1, d2, m, d0]

    g, c = both(hip, orc, f)
    assert abs(g[0][0] - c[0][0]) <= 1e-13 * abs(c[0][0])
    assert abs(g[1][0] - c[1][0]) <= 1e-12 * (n ** 3) ** 0.5
    assert g[2][0] == c[2][0]          # max of |x - poly| has no summation order
    assert g[3][0] == 0.0 and c[3][0] == 0.0
    # deterministic: same launch twice, same bits
    g2, _ = both(hip, orc, f)
    assert g[0][0] == g2[0][0] and g[1][0] == g2[1][0]


def _ulp_close(a, b, ulps=4):
    """Device libm vs glibc: a few ulps of the operands' magnitude (differences such as cos(..) - sin(..) cancel)."""
    scale = np.maximum(np.abs(b), 1.0)
    return np.all(np.abs(a - b) <= ulps * np.spacing(scale))


# All 17 named point functions of the reference's programs.  The library knows expression programs only (examg_expr_t;
# exastencils_amd/field.py:FN_PROGRAMS spells each function as the postfix program of its expression tree); the oracle keeps
# its own closed forms (orc_eval_fn).  Polynomial ones must agree bit for bit, those through libm within a few ulp.
@pytest.mark.parametrize("fn,nd,exact", [(0, 3, True), (FN_POLY3D, 3, True), (FN_TRIG2D_SOL, 2, False), (3, 2, False), (4, 3, True),
                                         (FN_KAPPA_RHS, 3, True), (FN_KAPPA_EXPSOL, 3, False), (7, 3, False), (FN_TRIG3D_SOL, 3, False),
                                         (9, 3, False), (10, 2, True), (11, 2, True), (12, 2, False), (13, 2, False), (14, 2, True),
                                         (15, 2, False), (16, 2, True)])
def test_fill_fn_and_dirichlet(hip, orc, fn, nd, exact):
    n = 24
    shape = tuple(n if d < nd else 0 for d in range(3))

    def f(ops):
        l = FieldLayout.node(nd, shape, 1)
        x, y = ops.new_array(l.size), ops.new_array(l.size)
        ops.fill_random(x, 3)
        ops.fill_random(y, 3)
        b, e = box(nd, n)
        ops.fill_fn(l.c_struct(), x, geom(nd, n, 0.25), fn, (10.0,), b, e)
        ops.apply_dirichlet(l.c_struct(), y, geom(nd, n, 0.25), fn, (10.0,), (1 << (2 * nd)) - 1 - 2)  # all faces but x+
        return [x, y]

    g, c = both(hip, orc, f)
    if exact:
        assert_same(g, c, "fill_fn")
    else:
        for a, b_ in zip(g, c):
            assert _ulp_close(a, b_, 8)
    # the +x face was masked out: its duplicate plane must still hold the random fill
    l = FieldLayout.node(nd, shape, 1)
    v = g[1].reshape(l.shape_zyx)
    r = orc.new_array(l.size)
    orc.fill_random(r, 3)
    rv = orc.to_host(r).reshape(l.shape_zyx)
    # interior rows (not on another face's plane) of the x+ duplicate plane
    sl = (slice(None) if nd == 2 else slice(3, -3), slice(3, -3), l.ref(0) + n)
    assert np.array_equal(v[sl], rv[sl])


def test_init_varcoeff7(hip, orc):
    n = 20

    def f(ops):
        l = FieldLayout.node(3, (n, n, n), 0)
        cf = ops.new_array(7 * l.size)
        b, e = box(3, n)
        ops.init_varcoeff7(l.c_struct(), cf, geom(3, n), FN_KAPPA_COEF, (10.0,), b, e)
        return [cf]

    g, c = both(hip, orc, f)
    assert np.allclose(g[0], c[0], rtol=1e-14, atol=0.0)


def test_pack_unpack_ranges(hip, orc):
    """All duplicate/ghost send and receive boxes of a 3-D fragment (communication/ir/IR_PackInfo*.scala)."""
    from exastencils_amd.comm import Communicator

    n = 20
    lay = FieldLayout.node(3, (n, n + 4, n - 4), 1)

    def f(ops):
        x, y = ops.new_array(lay.size), ops.new_array(lay.size)
        ops.fill_random(x, 21)
        outs = []
        for d in range(3):
            (sb, rb) = Communicator.dup_ranges(lay, 3, d)
            boxes = [sb, rb]
            for side in (-1, 1):
                boxes += list(Communicator.ghost_ranges(lay, 3, d, side))
            for bx in boxes:
                cnt = Communicator._count(bx)
                buf = ops.new_array(cnt)
                ops.pack(lay.c_struct(), x, buf, bx[0], bx[1])
                ops.unpack(lay.c_struct(), y, buf, bx[0], bx[1])
                outs.append(buf)
        outs.append(y)
        return outs

    g, c = both(hip, orc, f)
    assert_same(g, c, "pack/unpack")


def test_bad_arguments_fail_loudly(hip):
    from exastencils_amd.lib import ExamgError

    n = 16
    lu = FieldLayout.node(3, (n, n, n), 1)
    st = laplace_unit(3)
    u, f = hip.new_array(lu.size), hip.new_array(lu.size)
    with pytest.raises(ExamgError):      # box + stencil reach outside the allocation
        hip.stencil_op(SMOOTH, lu.c_struct(), u, lu.c_struct(), f, lu.c_struct(), f, st, 0.1, -1, [-1, 1, 1], [n, n, n])
    with pytest.raises(ExamgError):      # in-place without a colour
        hip.stencil_op(SMOOTH, lu.c_struct(), u, lu.c_struct(), f, lu.c_struct(), u, st, 0.1, -1, [1, 1, 1], [n, n, n])
    with pytest.raises(ExamgError):      # residual without rhs
        hip.stencil_op(RESIDUAL, lu.c_struct(), u, None, None, lu.c_struct(), f, st, 0.1, -1, [1, 1, 1], [n, n, n])


def test_bad_arguments_of_the_one_pass_forms_fail_loudly(hip):
    """The folded-correction, zero-field and residual + norm entry points: aliasing arrays, a coarse field too small for the
    correction's footprint, a fallback without its scratch array, and null pointers are errors with a message, not launches."""
    import ctypes as C

    from exastencils_amd import lib
    from exastencils_amd.lib import ExamgError

    n = 128
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, True, False)
    lc, lc_small = FieldLayout.node(3, (n // 2,) * 3, 1), FieldLayout.node(3, (n // 4,) * 3, 1)
    st = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / st.diag
    u, out, f, uc = hip.new_array(lu.size), hip.new_array(lu.size), hip.new_array(lf.size), hip.new_array(lc.size)
    b, e = [1, 1, 1], [n, n, n]
    with pytest.raises(ExamgError, match="out of place"):
        hip.rbgs_sweep_fused_prolong(lu.c_struct(), u, u, lf.c_struct(), f, st, w, 0, b, e, lc.c_struct(), uc)
    with pytest.raises(ExamgError, match="coarse footprint"):
        hip.rbgs_sweep_fused_prolong(lu.c_struct(), u, out, lf.c_struct(), f, st, w, 0, b, e, lc_small.c_struct(), uc)
    with pytest.raises(ExamgError, match="first colour"):
        hip.rbgs_sweep_fused_zero(lu.c_struct(), out, lf.c_struct(), f, st, w, 2, b, e)
    with pytest.raises(ExamgError, match="out of place"):
        hip.jacobi2_prolong(lu.c_struct(), u, u, None, lf.c_struct(), f, st, w, b, e, lc.c_struct(), uc)
    # 2-D stencil: residual + norm needs the residual array for its two-kernel path
    l2, st2 = FieldLayout.node(2, (64, 64, 1), 1), laplace_unit(2)
    u2, f2 = hip.new_array(l2.size), hip.new_array(l2.size)
    with pytest.raises(ExamgError, match="needs the residual array"):
        hip.residual_norm2(l2.c_struct(), u2, l2.c_struct(), f2, st2, [1, 1, 0], [64, 64, 1])
    r2 = hip.new_array(l2.size)
    hip.fill_random(u2, 3)
    got = hip.scalar_value(hip.residual_norm2(l2.c_struct(), u2, l2.c_struct(), f2, st2, [1, 1, 0], [64, 64, 1], l2.c_struct(), r2))
    hip.stencil_op(RESIDUAL, l2.c_struct(), u2, l2.c_struct(), f2, l2.c_struct(), r2, st2, 0.0, -1, [1, 1, 0], [64, 64, 1])
    want = hip.scalar_value(hip.dot(l2.c_struct(), r2, l2.c_struct(), r2, [1, 1, 0], [64, 64, 1]))
    assert got == want
    L = hip.L
    assert L.examg_crand_seed(None, 1) != 0 and b"null" in L.examg_last_error()
    st_ = lib.CrandStateC()
    assert L.examg_crand_seed(C.byref(st_), 1) == 0
    assert L.examg_crand_draw_host(C.byref(st_), None, 4) != 0
    buf = (C.c_double * 3)()
    assert L.examg_crand_draw_host(C.byref(st_), buf, 3) == 0
    assert [int(round(v * 2147483647.0)) for v in buf] == [1804289383, 846930886, 1681692777]


def test_external_field_copy(hip):
    """get<Name>/set<Name>: internal NodeWithComm field <-> external layout without ghost layers and with padding."""
    from exastencils_amd.external import ExternalField
    from exastencils_amd.field import Field

    n = 12
    internal = FieldLayout.node(3, (n, n + 2, n - 2), 1)
    F = Field("Solution", 3, internal, hip)
    hip.fill_random(F.data(), 9)
    for ext in (FieldLayout.node(3, (n, n + 2, n - 2), 0, align=4), FieldLayout.node(3, (n, n + 2, n - 2), 1)):
        E = ExternalField("extSol", ext, F, hip)
        got = E.get()
        full = hip.to_host(F.data()).reshape(internal.shape_zyx)
        g = min(ext.ghost[0], 1)
        iz = slice(internal.ref(2) - g, internal.ref(2) + (n - 2) + 1 + g)
        iy = slice(internal.ref(1) - g, internal.ref(1) + (n + 2) + 1 + g)
        ix = slice(internal.ref(0) - g, internal.ref(0) + n + 1 + g)
        ez = slice(ext.ref(2) - g, ext.ref(2) + (n - 2) + 1 + g)
        ey = slice(ext.ref(1) - g, ext.ref(1) + (n + 2) + 1 + g)
        ex = slice(ext.ref(0) - g, ext.ref(0) + n + 1 + g)
        assert np.array_equal(got[ez, ey, ex], full[iz, iy, ix])
        # round trip through set<Name>
        new = np.arange(ext.size, dtype=np.float64).reshape(ext.shape_zyx)
        E.set(new)
        hip.synchronize()
        full2 = hip.to_host(F.data()).reshape(internal.shape_zyx)
        assert np.array_equal(full2[iz, iy, ix], new[ez, ey, ex])


def test_init_helmholtz27(hip, orc):
    n = 24

    def f(ops):
        l = FieldLayout.node(3, (n, n, n), 0)
        cf = ops.new_array(27 * l.size)
        b, e = box(3, n)
        ops.init_helmholtz27(l.c_struct(), cf, geom(3, n), FN_KAPPA_COEF, (10.0, 2.0), b, e)
        return [cf]

    g, c = both(hip, orc, f)
    assert np.allclose(g[0], c[0], rtol=1e-13, atol=0.0)      # exp() through device libm


def test_fused_kernels_on_a_large_odd_block(hipd):
    """640^3 cells (2.1 GB per array, byte offsets beyond 2^31, a size that is no power of two): the fused two-step kernel,
    the fused red-black sweep and the wide restriction equal their unfused forms bit for bit (GPU against GPU; the oracle
    covers the unfused forms at sizes it finishes in seconds)."""
    hip = hipd
    import torch

    n = 640
    lu, lf, lc = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0), FieldLayout.node(3, (n // 2,) * 3, 0)
    u, a, b_, t = (hip.new_array(lu.size) for _ in range(4))
    f, fc1, fc2 = hip.new_array(lf.size), hip.new_array(lc.size), hip.new_array(lc.size)
    hip.fill_random(u, 1)
    hip.fill_random(f, 2)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = box(3, n)
    L, F, Lc = lu.c_struct(), lf.c_struct(), lc.c_struct()
    t.copy_(u)
    a.copy_(u)
    b_.copy_(u)
    hip.stencil_op(SMOOTH, L, u, F, f, L, t, A, w, -1, b, e)
    hip.stencil_op(SMOOTH, L, t, F, f, L, a, A, w, -1, b, e)
    hip.jacobi2(L, u, b_, None, F, f, A, w, b, e)
    assert torch.equal(a, b_)
    a.copy_(u)
    for c in (0, 1):
        hip.stencil_op(SMOOTH, L, a, F, f, L, a, A, w, c, b, e)
    hip.rbgs_sweep_fused(L, u, b_, F, f, A, w, 0, b, e)
    assert torch.equal(a, b_)
    bc, ec = box(3, n // 2)
    hip.restrict(L, u, Lc, fc1, 1.0, bc, ec)
    hip.L.examg_debug_restrict(0)
    try:
        hip.restrict(L, u, Lc, fc2, 1.0, bc, ec)
    finally:
        hip.L.examg_debug_restrict(1)
    assert torch.equal(fc1, fc2)


@pytest.mark.parametrize("nd,shape,mask", [(3, (12, 10, 8), 63), (3, (12, 10, 8), 0b100110), (2, (20, 14, 0), 15), (3, (6, 6, 6), 0)])
def test_fill_of_the_duplicate_planes_of_all_physical_faces_in_one_launch(hip, orc, nd, shape, mask):
    """examg_fill_dup_faces_expr (`loop over F only dup [dir] on boundary`, the SetFuncDir loops of Testing/FMG/3D_Trigonometric.exa4) equals
    one fill per face: duplicate plane, tangentially DLB..DRE -- ghost layers and interior untouched."""
    l = FieldLayout.node(nd, shape, 2)
    g = geom(nd, max(shape), 0.25)

    def run(ops):
        x = ops.new_array(l.size)
        ops.fill_random(x, 3)
        ops.fill_dup_faces(l.c_struct(), x, g, FN_POLY3D if nd == 3 else FN_XSQ, (), mask)
        return ops.to_host(x).copy()

    assert np.array_equal(run(hip), run(orc))


def test_expression_programs_on_device(hip, orc):
    """examg_fill_expr / examg_apply_dirichlet_expr / examg_max_err_expr: a polynomial program gives the bits of the built-in
    function with the same expression tree; a transcendental one follows numpy to a few ulp; malformed programs are refused."""
    from exastencils_amd.lib import ExamgError, ExprC

    n = 24
    l = FieldLayout.node(3, (n, n, n), 1)
    g = geom(3, n, 0.25)
    b, e = box(3, n)
    poly = ExprC.from_program([("x", None), ("x", None), ("*", None), ("const", 0.5), ("y", None), ("*", None), ("y", None), ("*", None),
                               ("-", None), ("const", 0.5), ("z", None), ("*", None), ("z", None), ("*", None), ("-", None)])
    trig = ExprC.from_program([("x", None), ("tan", None), ("y", None), ("z", None), ("*", None), ("exp", None), ("+", None),
                               ("const", 2.0), ("x", None), ("pow", None), ("-", None)])

    def run(ops, expr):
        x, y = ops.new_array(l.size), ops.new_array(l.size)
        ops.fill_expr(l.c_struct(), x, g, expr, b, e)
        ops.apply_dirichlet_expr(l.c_struct(), y, g, expr, 63)
        err = ops.scalar_value(ops.max_err_expr(l.c_struct(), y, g, expr, b, e))
        return ops.to_host(x).copy(), ops.to_host(y).copy(), err

    gx, gy, gerr = run(hip, poly)
    x2, y2 = hip.new_array(l.size), hip.new_array(l.size)
    hip.fill_fn(l.c_struct(), x2, g, FN_POLY3D, (), b, e)
    hip.apply_dirichlet(l.c_struct(), y2, g, FN_POLY3D, (), 63)
    assert np.array_equal(gx, hip.to_host(x2)) and np.array_equal(gy, hip.to_host(y2))
    assert gerr == hip.scalar_value(hip.max_err_fn(l.c_struct(), y2, g, FN_POLY3D, (), b, e))
    cx, cy, cerr = run(orc, poly)
    assert np.array_equal(gx, cx) and np.array_equal(gy, cy) and gerr == cerr
    gx, gy, gerr = run(hip, trig)
    cx, cy, cerr = run(orc, trig)
    assert _ulp_close(gx, cx, 8) and _ulp_close(gy, cy, 8) and abs(gerr - cerr) <= 1e-12 * max(1.0, cerr)
    bad = ExprC.from_program([("x", None), ("+", None)])
    with pytest.raises(ExamgError, match="underflow"):
        hip.fill_expr(l.c_struct(), hip.new_array(l.size), g, bad, b, e)
