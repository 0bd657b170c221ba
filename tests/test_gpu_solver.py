"""GPU parity of whole solves: the HIP path must print the reference's golden convergence histories and
follow the CPU oracle's residual history within 1e-10 relative (BASELINE.json north_star tolerance)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from golden_cases import CASES, golden_text, oracle_program
from oracle import mg
from test_host_logic import product_program

from exastencils_amd.solver import ConfigL4, SolverFromL4

RTOL = 1e-10


@pytest.fixture(scope="module")
def hip():
    from exastencils_amd.ops import HipOps

    return HipOps(0)


def _close(a, b, rtol=RTOL):
    """Residual histories agree within 1e-10 relative per iterate; where the problem data pass through libm
    (trig / exp boundary values, right-hand sides, coefficients: device libm and glibc differ in the last ulp)
    the converged iterates carry that data perturbation, bounded here by 1e-13 of the starting residual."""
    assert len(a) == len(b), (a, b)
    floor = 1e-13 * abs(b[0]) if len(b) else 0.0
    for x, y in zip(a, b):
        assert abs(x - y) <= rtol * abs(y) + floor, (a, b)


@pytest.mark.parametrize("name", ["CommBasic_PureMPI", "Poisson_2D_FD_Poisson_fromL4", "SISC_3D_ConstCoeff",
                                  "SISC_3D_VarCoeff", "FMG_3D_Trigonometric", "FMG_3D_VarCoeff"])
def test_golden_histories_on_gpu(hip, name):
    P = product_program(name, hip)
    P.setup()
    P.Solve()
    assert mg.compare_with_golden(P.log, golden_text(name)) == [], P.log
    O = oracle_program(name)
    O.setup()
    O.Solve()
    assert P.iterations == O.iterations
    _close(P.res_history, O.res_history)
    # max-norm errors: exact solution through device libm, converged iterates differ at rounding level:
    # compare absolutely, relative to the solution's magnitude (O(1))
    assert len(P.err_history) == len(O.err_history)
    for x, y in zip(P.err_history, O.err_history):
        assert abs(x - y) <= 1e-9 * abs(y) + 1e-12, (P.err_history, O.err_history)


def test_config0_2d_poisson_at_256_squared_on_gpu(hip):
    """BASELINE.json configs[0] at its own size: Examples/Poisson/2D_FD_Poisson_fromL4.exa4 on ONE 256 x 256-cell patch (levels 0..8,
    5-point stencil, red-black V(3,3), CG on the coarsest level, stop at 1e-10) -- the reference ships a golden for this program on
    4 x 4 fragments only (1024^2, above); at 256^2 the oracle's history is the yardstick: printed text identical, every residual
    within 1e-10, max-norm errors within 1e-9."""
    P = product_program("Poisson_2D_FD_Poisson_fromL4", hip, frag_len=(1, 1, 1))
    assert P.domain.ncells(8)[:2] == (256, 256)
    P.setup()
    P.Solve()
    O = oracle_program("Poisson_2D_FD_Poisson_fromL4", single_len=(1, 1, 1))
    O.setup()
    O.Solve()
    assert P.iterations == O.iterations and P.iterations >= 5
    assert mg.compare_with_golden(P.log, "\n".join(O.log)) == [], (P.log, O.log)
    _close(P.res_history, O.res_history)
    for x, y in zip(P.err_history, O.err_history):
        assert abs(x - y) <= 1e-9 * abs(y) + 1e-12, (P.err_history, O.err_history)


@pytest.mark.parametrize("name", ["Opts_seq", "Misc_inlining", "Opts_par"])
def test_golden_random_start_on_gpu(hip, name):
    """The reference's random-start goldens (Testing/Opts/{seq,par}.results, Testing/Misc/inlining.results; 64^3, 512^3 and 256^3):
    start values from libexamg's restatement of glibc's rand(), one generator per process of the reference's grid, uploaded;
    V(3,3) Jacobi cycles as two-step passes.  The printed history is the reference's; at the sizes the oracle runs in seconds
    every norm also agrees with the oracle (which calls the C library's rand()) within 1e-10."""
    P = product_program(name, hip, temporal_blocking=True, fused_residual_restrict=True)
    P.setup()
    P.Solve()
    assert mg.compare_with_golden(P.log, golden_text(name)) == [], P.log
    if name != "Opts_par":
        O = oracle_program(name)
        O.setup()
        O.Solve()
        assert P.iterations == O.iterations
        _close(P.res_history, O.res_history)


def test_golden_rbgs_576_on_gpu(hip):
    """Testing/Smoothers/RBGS.results at its full size (576^3) -- only the golden text is checked here, the
    oracle run of this size is in the CPU suite."""
    P = product_program("Smoothers_RBGS", hip)
    P.setup()
    P.Solve()
    assert mg.compare_with_golden(P.log, golden_text("Smoothers_RBGS")) == [], P.log


def test_golden_jac_576_on_gpu(hip):
    P = product_program("Smoothers_Jac", hip)
    P.setup()
    P.Solve()
    assert mg.compare_with_golden(P.log, golden_text("Smoothers_Jac")) == [], P.log


def _l4(hip, **kw):
    base = dict(nd=3, min_level=2, max_level=7, tol=1e-6)
    base.update(kw)
    P = SolverFromL4(ConfigL4(**base), hip)
    P.setup()
    return P


def test_benchmark_program_vs_oracle_128(hip):
    """Benchmark/Poisson3D program (config 3's algorithm) at 128^3, levels 2..7: fused coarse CG kernel,
    unfused coarse CG, and the oracle agree on the residual history."""
    O = mg.ProgramA(mg.ConfigA(nd=3, min_level=2, max_level=7, tol=1e-6))
    O.setup()
    O.Solve()
    for fused in (True, False):
        P = _l4(hip, fused_coarse=fused)
        P.Solve()
        assert P.iterations == O.iterations
        _close(P.res_history, O.res_history)


def test_fused_coarse_cg_matches_unfused(hip):
    a, b = _l4(hip, fused_coarse=True, max_level=5), _l4(hip, fused_coarse=False, max_level=5)
    for P in (a, b):
        P.mgCycle(5)
    hip.synchronize()
    for l in a.levels:
        x, y = hip.to_host(a.Solution[l].data()), hip.to_host(b.Solution[l].data())
        assert np.allclose(x, y, rtol=1e-12, atol=1e-13 * np.abs(y).max())
    info = hip.to_host(a._cg_info)
    assert info[0] == b.cg_iters[0] and info[2] <= 1e-3 * info[1]


def test_graph_replay_equals_eager(hip):
    a, b = _l4(hip), _l4(hip)
    a.capture_cycle()           # one warm-up cycle runs; the captured one is only recorded
    b.mgCycle(7)
    a.replay_cycle()
    b.mgCycle(7)
    hip.synchronize()
    assert np.array_equal(hip.to_host(a.Solution[7].data()), hip.to_host(b.Solution[7].data()))


def test_graph_solve_after_reset_equals_eager(hip):
    """bench.py's totalTimeSolve path: capture, reset() to the initial state, Solve with graph replays -- same history
    as the eager, unfused program."""
    a = _l4(hip, fused_rbgs=True, max_level=6)
    a.capture_cycle()
    a.reset()
    a.Solve(use_graph=True)
    b = _l4(hip, fused_rbgs=False, fused_coarse=False, max_level=6)
    b.Solve()
    assert a.iterations == b.iterations
    _close(a.res_history, b.res_history, 1e-12)


def test_fused_rbgs_sweep_bit_exact(hip):
    a, b = _l4(hip, fused_rbgs=True), _l4(hip, fused_rbgs=False)
    for P in (a, b):
        P.mgCycle(7)
    hip.synchronize()
    for l in a.levels:
        assert np.array_equal(hip.to_host(a.Solution[l].data()), hip.to_host(b.Solution[l].data())), l


def test_temporal_blocking_jacobi_bit_exact(hip):
    """SolverFromL3 with pairs of Jacobi steps fused (examg_jacobi2) follows the plain program bit for bit."""
    from exastencils_amd.solver import ConfigL3, SolverFromL3

    res = []
    for tb in (False, True):
        P = SolverFromL3(ConfigL3(nd=3, min_level=2, max_level=7, smoother="jacobi", omega=0.8, stencil="unit",
                                  restrict_scale=4.0, tol=1e-5, cg_max=512, temporal_blocking=tb), hip)
        P.setup()
        P.Solve()
        res.append(P)
    assert res[0].res_history == res[1].res_history
    S0, S1 = res[0].Solution[7], res[1].Solution[7]
    assert np.array_equal(hip.to_host(S0.data()), hip.to_host(S1.data()))


class _LoopbackComm:
    """Stands in for the block neighbours on ONE GPU: every interior face receives this block's own opposite inner
    planes (pack -> unpack on the current stream).  Exercises the pack/unpack kernels, the thin face launches and the
    stream/event logic of jacobi_pair without a second GPU."""

    def __init__(self, domain, ops):
        from exastencils_amd.comm import Communicator

        self.domain, self.ops, self.C = domain, ops, Communicator
        self.calls = 0

    def exchange(self, f, slot=None, what="all", axis_only=False):
        lay, nd = f.layout, self.domain.nd
        x = f.data(slot)
        for d in range(nd):
            for side in (-1, 1):
                if self.domain.neighbor(d, side) is None:
                    continue
                sbox, rbox = self.C.ghost_ranges(lay, nd, d, side)
                buf = self.ops.new_array(self.C._count(sbox))
                self.ops.pack(f.lc, x, buf, sbox[0], sbox[1])
                self.ops.unpack(f.lc, x, buf, rbox[0], rbox[1])
                self.calls += 1


@pytest.mark.parametrize("rank", [0, 5])
def test_jacobi_pair_overlap_equals_sequential(hip, rank):
    """jacobi_pair on a block with interior faces (2x2x2 decomposition, loop-back neighbours): halo traffic on the side
    stream overlapped with the interior kernel gives the same bits as the sequential order and as two single steps."""
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.field import Field, laplace_unit
    from exastencils_amd.layout import FieldLayout
    from exastencils_amd.smoothers import jacobi_pair

    dom = RectDomain(3, (2, 2, 2), rank)
    L = 7
    lay, layf = FieldLayout.node(3, dom.ncells(L), 1), FieldLayout.node(3, dom.ncells(L), 0, False, False)
    A = laplace_unit(3)
    w = 0.8 / A.diag
    outs = []
    for mode in ("overlap", "sequential", "single"):
        S, F, T = Field("S", L, lay, hip, 2, None), Field("F", L, layf, hip, 1, None), Field("T", L, lay, hip, 1, None)
        hip.fill_random(S.data(0), 1)
        S.data(1).copy_(S.data(0))      # both slots and the scratch field carry the same Dirichlet shell,
        T.data().copy_(S.data(0))       # as `apply bc` leaves them in the programs
        hip.fill_random(F.data(), 3)
        comm = _LoopbackComm(dom, hip)
        for _ in range(2):
            if mode == "single":
                for _k in range(2):
                    comm.exchange(S, S.active, "ghost")
                    b, e = dom.loop_bounds(lay)
                    hip.stencil_op(2, S.lc, S.data(S.active), F.lc, F.data(), S.lc, S.data(S.next), A, w, -1, b, e)
                    S.advance()
            else:
                jacobi_pair(hip, comm, dom, S, F, A, w, T, overlap=(mode == "overlap"))
        hip.synchronize()
        b, e = dom.loop_bounds(lay)
        v = hip.to_host(S.data()).reshape(lay.shape_zyx)
        outs.append(v[b[2] + 1:e[2] + 1, b[1] + 1:e[1] + 1, b[0] + 1:e[0] + 1].copy())
        assert comm.calls > 0
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0], outs[2])


@pytest.mark.parametrize("rank", [0, 5])
def test_rbgs_sweep_overlap_equals_sequential(hip, rank):
    """rbgs_sweep on a block with interior faces (2x2x2 decomposition, loop-back neighbours): fused deep interior with the
    shell's exchanges on the side stream == sequential order == the two in-place half sweeps with an exchange before each."""
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.field import Field, laplace_fd
    from exastencils_amd.layout import FieldLayout
    from exastencils_amd.smoothers import rbgs_sweep

    dom = RectDomain(3, (2, 2, 2), rank)
    L = 7
    lay, layf = FieldLayout.node(3, dom.ncells(L), 1), FieldLayout.node(3, dom.ncells(L), 0, True, False)
    A = laplace_fd(3, dom.h(L))
    w = 0.8 / A.diag
    b, e = dom.loop_bounds(lay)
    outs = []
    for mode in ("overlap", "sequential", "plain"):
        S, F, T = Field("S", L, lay, hip, 1, None), Field("F", L, layf, hip, 1, None), Field("T", L, lay, hip, 1, None)
        hip.fill_random(S.data(), 1)
        alt = S.data().clone()          # the second array and the scratch field carry the same boundary planes
        hip.fill_random(F.data(), 3)
        comm = _LoopbackComm(dom, hip)
        for _ in range(3):
            if mode == "plain":
                for colour in (0, 1):
                    comm.exchange(S, None, "ghost")
                    hip.stencil_op(2, S.lc, S.data(), F.lc, F.data(), S.lc, S.data(), A, w, colour, b, e)
            else:
                alt = rbgs_sweep(hip, comm, dom, S, F, A, w, alt, T, 0, overlap=(mode == "overlap"))
        hip.synchronize()
        v = hip.to_host(S.data()).reshape(lay.shape_zyx)
        outs.append(v[b[2] + 1:e[2] + 1, b[1] + 1:e[1] + 1, b[0] + 1:e[0] + 1].copy())
        assert comm.calls > 0
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0], outs[2])


def _permuted7(h):
    """A 7-point constant stencil declared in an entry order that is neither of the two the one-pass kernel knows
    (c, -y, +y, -x, +x, -z, +z): valid program text, different summation order, so the fused kernels must not be taken."""
    from exastencils_amd.field import Stencil

    offs = [(0, 0, 0), (0, -1, 0), (0, 1, 0), (-1, 0, 0), (1, 0, 0), (0, 0, -1), (0, 0, 1)]
    co = [2.0 / h[0] ** 2 + 2.0 / h[1] ** 2 + 2.0 / h[2] ** 2] + [-1.0 / h[d] ** 2 for d in (1, 1, 0, 0, 2, 2)]
    return Stencil(offs, co)


@pytest.mark.parametrize("kind", ["jacobi_pair", "rbgs_sweep"])
def test_non_canonical_entry_order_with_neighbours(hip, kind):
    """A 7-point stencil in a non-canonical entry order on a block with loop-back neighbours: the kernel layer reports the
    one-pass kernel as not eligible (examg_two_stage_eligible), jacobi_pair / rbgs_sweep then run interior and shell in
    sequence through the fallback -- no race on the scratch field, no 'needs a distinct tmp' error -- and give the bits of
    the plain statement-by-statement form."""
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.field import Field
    from exastencils_amd.layout import FieldLayout
    from exastencils_amd.smoothers import jacobi_pair, rbgs_sweep

    dom = RectDomain(3, (2, 2, 2), 5)
    L = 7
    lay, layf = FieldLayout.node(3, dom.ncells(L), 1), FieldLayout.node(3, dom.ncells(L), 0, True, False)
    A = _permuted7(dom.h(L))
    w = 0.8 / A.diag
    b, e = dom.loop_bounds(lay)
    b1 = [b[d] + (1 if dom.neighbor(d, -1) is not None else 0) for d in range(3)]
    e1 = [e[d] - (1 if dom.neighbor(d, +1) is not None else 0) for d in range(3)]
    b2 = [b[d] + (2 if dom.neighbor(d, -1) is not None else 0) for d in range(3)]
    e2 = [e[d] - (2 if dom.neighbor(d, +1) is not None else 0) for d in range(3)]
    assert not hip.two_stage_eligible(lay.c_struct(), layf.c_struct(), A, b1, e1, b2, e2)
    from exastencils_amd.field import laplace_fd
    assert hip.two_stage_eligible(lay.c_struct(), layf.c_struct(), laplace_fd(3, dom.h(L)), b1, e1, b2, e2)
    outs = []
    for mode in ("fused-api", "plain"):
        nslots = 2 if kind == "jacobi_pair" else 1
        S, F, T = Field("S", L, lay, hip, nslots, None), Field("F", L, layf, hip, 1, None), Field("T", L, lay, hip, 1, None)
        hip.fill_random(S.data(0), 1)
        if nslots == 2:
            S.data(1).copy_(S.data(0))
        T.data().copy_(S.data(0))
        alt = S.data(0).clone()
        hip.fill_random(F.data(), 3)
        comm = _LoopbackComm(dom, hip)
        for _ in range(2):
            if kind == "jacobi_pair":
                if mode == "plain":
                    for _k in range(2):
                        comm.exchange(S, S.active, "ghost")
                        hip.stencil_op(2, S.lc, S.data(S.active), F.lc, F.data(), S.lc, S.data(S.next), A, w, -1, b, e)
                        S.advance()
                else:
                    jacobi_pair(hip, comm, dom, S, F, A, w, T, overlap=True)
            else:
                if mode == "plain":
                    for colour in (0, 1):
                        comm.exchange(S, None, "ghost")
                        hip.stencil_op(2, S.lc, S.data(), F.lc, F.data(), S.lc, S.data(), A, w, colour, b, e)
                else:
                    alt = rbgs_sweep(hip, comm, dom, S, F, A, w, alt, T, 0, overlap=True)
        hip.synchronize()
        v = hip.to_host(S.data()).reshape(lay.shape_zyx)
        outs.append(v[b[2] + 1:e[2] + 1, b[1] + 1:e[1] + 1, b[0] + 1:e[0] + 1].copy())
    assert np.array_equal(outs[0], outs[1])


def test_lds_resident_coarse_cg_equals_global_memory_solver():
    """Coarsest grids of up to 4096 points run the CG with its vectors in registers / LDS: same iterates, bit for bit, as
    the global-memory form of the kernel (3-D 16^3 and 8^3 coarsest grids, 2-D 16^2)."""
    from exastencils_amd import lib
    from exastencils_amd.ops import HipOps
    from exastencils_amd.solver import ConfigL4, SolverFromL4

    hip = HipOps(0, lib.DBG_LIB_PATH)      # debug build: examg_debug_cg selects the form of the solver
    for kw in (dict(nd=3, min_level=4, max_level=6), dict(nd=3, min_level=3, max_level=5),
               dict(nd=2, min_level=4, max_level=7, bc_fn=2, rhs_fn=3)):
        hist = []
        for lds in (0, 1):
            hip.L.examg_debug_cg(lds)
            try:
                P = SolverFromL4(ConfigL4(tol=1e-8, **kw), hip)
                P.setup()
                P.Solve()
            finally:
                hip.L.examg_debug_cg(1)
            hist.append((P.res_history, hip.to_host(P.Solution[kw["max_level"]].data()).copy()))
        assert hist[0][0] == hist[1][0]
        assert np.array_equal(hist[0][1], hist[1][1])


def test_coarse_cg_variant_of_the_layer3_generator():
    """examg_cg_coarse_variant(ALPHA_FROM_NORM | NO_BC) -- VCycle_0@coarsest of Testing/Smoothers/Jac.exa4:75-109: alpha from the
    squared norm, no `apply bc` -- against the oracle's loops in that order, with NON-ZERO boundary planes in Solution and in the
    search direction (the FMG start keeps Dirichlet values in Solution@coarsest): both forms of the kernel (LDS-resident and
    global-memory) agree bit for bit with each other and to rounding with the oracle; boundary planes stay untouched."""
    from exastencils_amd import lib
    from exastencils_amd.field import laplace_fd
    from exastencils_amd.layout import FieldLayout
    from exastencils_amd.ops import HipOps
    from oracle_ops import OracleOps

    hip, orc = HipOps(0, lib.DBG_LIB_PATH), OracleOps()
    for n in (4, 16, 32):        # 3^3 and 15^3 unknowns: LDS form available; 31^3: global-memory form only
        lu, ln = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
        A = laplace_fd(3, (1.0 / n,) * 3, "pm", "mul")
        b, e = [1, 1, 1], [n, n, n]
        from exastencils_amd.domain import RectDomain

        geom, mask = RectDomain(3).geom(2), 63
        results = []
        for ops, lds in ((orc, None), (hip, 1), (hip, 0)):
            if lds is not None:
                hip.L.examg_debug_cg(lds)
            sol, rhs, res, p, ap = (ops.new_array(lu.size), ops.new_array(ln.size), ops.new_array(lu.size), ops.new_array(lu.size),
                                    ops.new_array(ln.size))
            info = ops.new_array(4)
            ops.fill_random(sol, 5)        # whole arrays, boundary and ghost planes included
            ops.fill_random(rhs, 6)
            ops.fill_random(p, 7)
            ops.fill_random(res, 8)
            before = [ops.to_host(t).copy() for t in (sol, p, res)]
            ops.cg_coarse(lu.c_struct(), sol, ln.c_struct(), rhs, lu.c_struct(), res, lu.c_struct(), p, ln.c_struct(), ap, A, geom, mask,
                          200, 1e-3, b, e, info, flags=lib.CG_ALPHA_FROM_NORM | lib.CG_NO_BC)
            ops.synchronize()
            results.append(([ops.to_host(t).copy() for t in (sol, p, res)], before, float(ops.to_host(info)[0])))
        hip.L.examg_debug_cg(1)
        (o_out, o_in, o_it), (l_out, l_in, l_it), (g_out, g_in, g_it) = results
        assert o_it == l_it == g_it and 1 <= o_it < 200, (n, o_it, l_it, g_it)
        inner = tuple(slice(2, n + 1) for _ in range(3))
        for k in range(3):
            assert np.array_equal(l_out[k], g_out[k])                            # the two forms of the kernel
            scale = np.abs(o_out[k]).max()
            assert np.allclose(l_out[k], o_out[k], rtol=1e-10, atol=1e-12 * scale)
            shell_after, shell_before = l_out[k].reshape(lu.shape_zyx).copy(), l_in[k].reshape(lu.shape_zyx).copy()
            shell_after[inner] = 0.0
            shell_before[inner] = 0.0
            assert np.array_equal(shell_after, shell_before)                     # no `apply bc`: nothing outside the box is written
    # the iteration limit: info[3] counts the solves whose loop ran out (where the generated function prints its message)
    n = 16
    lu, ln = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
    A = laplace_fd(3, (1.0 / n,) * 3, "pm", "mul")
    for lds in (1, 0):
        hip.L.examg_debug_cg(lds)
        sol, rhs, res, p, ap, info = (hip.new_array(lu.size), hip.new_array(ln.size), hip.new_array(lu.size), hip.new_array(lu.size),
                                      hip.new_array(ln.size), hip.new_array(4))
        hip.fill_random(rhs, 6)
        for flags, limit, want in ((0, 2, 1.0), (3, 2, 2.0), (0, 200, 2.0)):
            hip.cg_coarse(lu.c_struct(), sol, ln.c_struct(), rhs, lu.c_struct(), res, lu.c_struct(), p, ln.c_struct(), ap, A, geom, 63,
                          limit, 1e-3, [1, 1, 1], [n, n, n], info, flags=flags)
            hip.synchronize()
            got = hip.to_host(info)
            assert got[3] == want and got[0] <= limit, (lds, flags, limit, got)
    hip.L.examg_debug_cg(1)


def test_coarse_solver_starting_from_the_zero_field():
    """EXAMG_CG_ZERO_START: `Solution@coarsest = 0` rides along with the one-kernel coarse solve -- the solver does not read `sol` and the
    zeroing loop does not run: the same bits as examg_set + the solver, in both forms of the kernel (search direction in LDS / global)."""
    from exastencils_amd import lib
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.field import laplace_fd
    from exastencils_amd.layout import FieldLayout
    from exastencils_amd.ops import HipOps

    hip = HipOps(0, lib.DBG_LIB_PATH)      # debug build: examg_debug_cg selects the form of the solver
    n = 16
    lu, ln = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
    A = laplace_fd(3, (1.0 / n,) * 3)
    geom, b, e = RectDomain(3).geom(4), [1, 1, 1], [n, n, n]
    for lds in (1, 0):
        hip.L.examg_debug_cg(lds)
        outs = []
        for zero_flag in (False, True):
            sol, rhs, res, p, ap, info = (hip.new_array(lu.size), hip.new_array(ln.size), hip.new_array(ln.size), hip.new_array(lu.size),
                                          hip.new_array(ln.size), hip.new_array(4))
            hip.fill_random(rhs, 6)
            if zero_flag:
                hip.fill_random(sol, 9)          # garbage at the inner points: never read
                hip.set(lu.c_struct(), sol, 0.0, [-1, -1, -1], [n + 2, n + 2, 1])      # boundary planes hold 0 (coarse levels: homogeneous Dirichlet)
                hip.set(lu.c_struct(), sol, 0.0, [-1, -1, n], [n + 2, n + 2, n + 2])
                hip.set(lu.c_struct(), sol, 0.0, [-1, -1, 1], [n + 2, 1, n])
                hip.set(lu.c_struct(), sol, 0.0, [-1, n, 1], [n + 2, n + 2, n])
                hip.set(lu.c_struct(), sol, 0.0, [-1, 1, 1], [1, n, n])
                hip.set(lu.c_struct(), sol, 0.0, [n, 1, 1], [n + 2, n, n])
            hip.cg_coarse(lu.c_struct(), sol, ln.c_struct(), rhs, ln.c_struct(), res, lu.c_struct(), p, ln.c_struct(), ap, A, geom, 63, 128, 1e-3, b, e,
                          info, flags=lib.CG_ZERO_START if zero_flag else 0)
            hip.synchronize()
            outs.append([hip.to_host(t).copy() for t in (sol, res, p, info)])
        for x, y in zip(*outs):
            assert np.array_equal(x, y)
        assert outs[0][3][0] >= 1
    hip.L.examg_debug_cg(1)


def test_fmg_driver_one_pass_forms_on_gpu(hip):
    """The FMG driver with every one-pass form (tests/test_host_logic.py::test_fmg_driver_with_one_pass_forms) on the HIP kernels at
    256^3: folded correction and zero-field sweeps change no bit; residual + norm and the one-call coarse solve sum in another order --
    same iteration count, histories within 1e-10 of the statement-by-statement driver and of the oracle program."""
    from exastencils_amd.solver import ConfigL3, SolverFromL3

    kw = dict(nd=3, min_level=2, max_level=8, smoother="rbgs", omega=1.0, stencil="scaled", restrict_scale=1.0, tol=1e-8, bc_fn=1, fmg=True)
    base = dict(fused_rbgs=True, fused_residual_restrict=True)
    runs = {}
    for name, extra in (("plain", {}), ("sweeps", dict(base)),
                        ("fold_zero", dict(base, fused_prolong_min_points=1, fused_zero_start=True)),
                        ("all", dict(base, fused_prolong_min_points=1, fused_zero_start=True, fused_residual_norm=True, fused_coarse=True))):
        P = SolverFromL3(ConfigL3(**kw, **extra), hip)
        P.setup()
        P.Solve()
        runs[name] = (P.iterations, P.res_history, hip.to_host(P.Solution[8].data()).copy(), P)
    P = runs["all"][3]
    # every level above the coarsest: the two-stage kernel from 64-point rows, the small-level kernel (csrc/kernels_small.hip) below
    assert P._folds_prolongation(8) and P._folds_prolongation(7) and P._starts_from_zero(7) and P._starts_from_zero(6) and P._folds_prolongation(4)
    assert runs["fold_zero"][1] == runs["sweeps"][1] and np.array_equal(runs["fold_zero"][2], runs["sweeps"][2])
    O = mg.ProgramB(mg.ConfigB(**kw))
    O.setup()
    O.Solve()
    for name in ("plain", "sweeps", "all"):
        assert runs[name][0] == O.iterations, name
        _close(runs[name][1], O.res_history)
    # FMG start and VCycle@finest replayed from hipGraphs (nothing in them returns to the host): the same launches, the same bits
    G = SolverFromL3(ConfigL3(**kw, **dict(base, fused_prolong_min_points=1, fused_zero_start=True, fused_residual_norm=True,
                                           fused_coarse=True)), hip)
    G.setup()
    G.capture()
    G.Solve(use_graph=True)
    assert G.res_history == runs["all"][1]
    assert np.array_equal(hip.to_host(G.Solution[8].data()), runs["all"][2])
    G.reset()
    G.Solve(use_graph=True)           # ... and again after a reset
    assert G.res_history == runs["all"][1]


def test_folded_prolongation_changes_no_bit(hip):
    """V-cycles with the correction loop folded into the first post-smoothing sweep (examg_rbgs_sweep_fused_prolong on every
    level with rows of 64 points or more) leave the same residual history and the same solution, bit for bit."""
    from exastencils_amd.solver import ConfigL4, SolverFromL4

    hist, sols, folds = [], [], []
    for min_points in (0, 1):       # ... and (second run) `Solution@coarser = 0` left to the first sweep of the coarser level
        P = SolverFromL4(ConfigL4(nd=3, min_level=2, max_level=8, tol=1e-6, fused_rbgs=True, fused_residual_restrict=True,
                                  fused_prolong_min_points=min_points, fused_zero_start=bool(min_points), fused_rbgs3=False), hip)
        P.setup()
        P.Solve()
        hist.append(P.res_history)
        sols.append(hip.to_host(P.Solution[8].data()).copy())
        folds.append([P._folds_prolongation(l) for l in range(3, 9)])
    assert folds == [[False] * 6, [True] * 6]       # rows of 64 points and more: the two-stage kernel; shorter rows: the small-level kernel
    # the product's default on top: three sweeps of the 256^3 level as two passes of three colour loops, its correction a loop of its own
    P = SolverFromL4(ConfigL4(nd=3, min_level=2, max_level=8, tol=1e-6, fused_rbgs=True, fused_residual_restrict=True, fused_prolong_min_points=1,
                              fused_zero_start=True), hip)
    P.setup()
    P.Solve()
    assert P._three_colour_passes(8) and not P._folds_prolongation(8) and P._folds_prolongation(7)
    assert P.res_history == hist[0] and np.array_equal(hip.to_host(P.Solution[8].data()).view(np.uint64), sols[0].view(np.uint64))
    assert hist[0] == hist[1] and len(hist[0]) > 4
    assert np.array_equal(sols[0].view(np.uint64), sols[1].view(np.uint64))
    from exastencils_amd.solver import ConfigL3, SolverFromL3

    hist, sols = [], []
    for min_points in (0, 1):        # slotted Jacobi program: Correction folded into the first pair of post-smoothing steps
        P = SolverFromL3(ConfigL3(nd=3, min_level=1, max_level=7, temporal_blocking=True, fused_residual_restrict=True,
                                  fused_prolong_min_points=min_points), hip)
        P.setup()
        P.Solve()
        assert P._folds_prolongation(7) == bool(min_points) and P._folds_prolongation(6) == bool(min_points)
        hist.append(P.res_history)
        S = P.Solution[7]
        sols.append(hip.to_host(S.data(S.active)).copy())
    assert hist[0] == hist[1] and len(hist[0]) > 4
    assert np.array_equal(sols[0].view(np.uint64), sols[1].view(np.uint64))


def test_residual_norm_in_one_pass_keeps_the_history(hip):
    """Solve@finest with `Residual = ...` + ResNorm() as one pass that never stores the residual: same iteration count, every
    norm within 1e-13 (summation order of the reduction), same solution bits (the cycle never reads Residual@finest)."""
    from exastencils_amd.solver import ConfigL4, SolverFromL4

    hist, sols = [], []
    for fused in (False, True):
        P = SolverFromL4(ConfigL4(nd=3, min_level=2, max_level=7, tol=1e-6, fused_rbgs=True, fused_residual_restrict=True,
                                  fused_residual_norm=fused), hip)
        P.setup()
        P.Solve()
        hist.append(P.res_history)
        sols.append(hip.to_host(P.Solution[7].data()).copy())
    assert len(hist[0]) == len(hist[1]) > 4
    for a, b in zip(*hist):
        assert abs(a - b) <= 1e-13 * a, (a, b)
    assert np.array_equal(sols[0].view(np.uint64), sols[1].view(np.uint64))


def test_fused_residual_restrict_changes_no_bit(hip):
    """V-cycles with residual + restriction as one pass (fine residual never stored) print the same history, bit for bit."""
    from exastencils_amd.solver import ConfigL4, SolverFromL4

    hist = []
    for fused in (False, True):
        P = SolverFromL4(ConfigL4(nd=3, min_level=2, max_level=7, tol=1e-6, fused_rbgs=True, fused_residual_restrict=fused), hip)
        P.setup()
        P.Solve()
        hist.append(P.res_history)
    assert hist[0] == hist[1]
    assert len(hist[0]) > 4
    from exastencils_amd.solver import ConfigL3, SolverFromL3

    hist = []
    for fused in (False, True):      # slotted Jacobi program, restriction scaled by 4, two-step passes
        P = SolverFromL3(ConfigL3(nd=3, min_level=1, max_level=7, temporal_blocking=True, fused_residual_restrict=fused), hip)
        P.setup()
        P.Solve()
        hist.append(P.res_history)
    assert hist[0] == hist[1]


def test_config4_helmholtz27_on_gpu(hip):
    """27-entry variable-coefficient Helmholtz V-cycles (config 4's operator) against the oracle program."""
    from test_host_logic import HELMHOLTZ27

    from exastencils_amd.solver import ConfigL3, SolverFromL3

    kw = dict(HELMHOLTZ27, max_level=5)
    O = mg.ProgramB(mg.ConfigB(nfrag=(1, 1, 1), frag_len=(2, 2, 2), **kw))
    O.setup()
    O.Solve()
    P = SolverFromL3(ConfigL3(frag_len=(2, 2, 2), **kw), hip)
    P.setup()
    P.Solve()
    assert P.iterations == O.iterations
    _close(P.res_history, O.res_history, 1e-9)       # coefficients pass through device exp()
    assert P.err_history[-1] < 1e-8
    # what bench.py runs: coefficient records, pairs of Jacobi steps and the last pre-smoothing step + residual as one pass each
    # (csrc/kernels_sf27pair.hip on the levels with rows of 32 points and more): the same statements -- the same bits
    Q = SolverFromL3(ConfigL3(frag_len=(2, 2, 2), coef_entry_fastest=True, temporal_blocking=True, fused_smooth_residual=True, **kw), hip)
    Q.setup()
    Q.Solve()
    assert Q.res_history == P.res_history and Q.err_history == P.err_history


def test_cpp_host_program_matches_oracle():
    """examples/poisson3d_host.cpp: a C++ host in the shape of the generated program, linked against libexamg only
    (no Python, no PyTorch in that process), prints the oracle's residual history."""
    import os
    import subprocess

    import __graft_entry__ as ge

    exe = ge.build_example()
    out = subprocess.run([exe, "6", "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    hist = [float(l[2:]) for l in out.stdout.splitlines() if l.startswith("# ")]
    O = mg.ProgramA(mg.ConfigA(nd=3, min_level=2, max_level=6, tol=1e-6))
    O.setup()
    O.Solve()
    _close(hist, O.res_history)
    printed = [l for l in out.stdout.splitlines() if l and not l.startswith("#") and not l.startswith("iterations")]
    assert mg.compare_with_golden(printed, "\n".join(O.log)) == []


@pytest.mark.parametrize("fold_min_points", [1, 50_000_000])
def test_cpp_fast_host_program_matches_oracle(fold_min_points):
    """examples/poisson3d_fast_host.cpp: the same program on the one-pass entry points (fused sweeps, residual + restriction,
    zero-field first sweep, correction folded into the first post-smoothing sweep, residual + norm) with the cycle replayed
    from a hipGraph, all from C++: the oracle's history within 1e-10, the same iteration count."""
    import subprocess

    import __graft_entry__ as ge

    exe = ge.build_example(name="poisson3d_fast_host")
    out = subprocess.run([exe, "7", "2", str(fold_min_points)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    hist = [float(l[2:]) for l in out.stdout.splitlines() if l.startswith("# ")]
    O = mg.ProgramA(mg.ConfigA(nd=3, min_level=2, max_level=7, tol=1e-6))
    O.setup()
    O.Solve()
    _close(hist, O.res_history)
    assert "iterations %d" % O.iterations in out.stdout
    assert any(l.startswith("vcycle_ms") for l in out.stdout.splitlines())


def test_config3_512_properties(hip):
    """Config 3 (512^3, levels 4..9) at full size, through size-independent properties: the V-cycle contracts
    the residual by the factor the oracle shows at 128^3 (multigrid convergence is h-independent), the history
    is reproducible bit for bit, and Dirichlet values are untouched."""
    P = _l4(hip, min_level=4, max_level=9, tol=1e-3)
    P.Solve()
    O = mg.ProgramA(mg.ConfigA(nd=3, min_level=2, max_level=7, tol=1e-3))
    O.setup()
    O.Solve()
    assert P.iterations == O.iterations
    fp = [P.res_history[i + 1] / P.res_history[i] for i in range(P.iterations)]
    fo = [O.res_history[i + 1] / O.res_history[i] for i in range(O.iterations)]
    for x, y in zip(fp, fo):
        assert abs(x - y) < 0.25 * y, (fp, fo)
    Q = _l4(hip, min_level=4, max_level=9, tol=1e-3)
    Q.Solve()
    assert Q.res_history == P.res_history
    S = P.Solution[9]
    v = hip.to_host(S.data()).reshape(S.layout.shape_zyx)
    n = 512
    x = np.arange(-1, n + 2) / n
    face = x[None, :] ** 2 - 0.5 * (x[:, None] ** 2) - 0.5 * 0.0       # z = 0 face: x^2 - y^2/2
    assert np.allclose(v[1, :, :], face, rtol=0, atol=1e-15)


def test_reference_large_hybrid_on_one_gpu():
    """Testing/Large/Hybrid (2560 x 1024 x 2048 cells, 5.4e9 unknowns; the reference runs it on 4 MPI ranks x 10 fragments)
    on one MI355X: ~200 GB resident (~250 GB with two-step passes).  Opt-in: EXAMG_LARGE=1 (needs the whole HBM)."""
    import json
    import os
    import subprocess
    import sys

    if not os.environ.get("EXAMG_LARGE"):
        pytest.skip("set EXAMG_LARGE=1 (uses 200-250 GB of HBM)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ([], ["--pairs"]):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "large_hybrid.py")] + extra, capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        assert json.loads(r.stdout.strip().splitlines()[-1])["matches_reference_results"]


def test_config4_algorithm_fmg_with_red_black_cycles(hip):
    """BASELINE configs[4] names FMG on the Poisson3D problem: nested iteration from the coarsest level (boundary values of
    every level from the function, zero right-hand side, cycle, prolongation to the next level) followed by red-black
    V(3,3) cycles -- the FMG driver of Testing/FMG/3D_Trigonometric.exa4:189-242 with the smoother of
    Testing/Smoothers/RBGS.exa4:125-133 on the h-scaled Laplacian, against the oracle program at 64^3."""
    from exastencils_amd.solver import ConfigL3, SolverFromL3

    kw = dict(nd=3, min_level=2, max_level=6, smoother="rbgs", omega=1.0, stencil="scaled", restrict_scale=1.0, tol=1e-8, cg_max=512,
              bc_fn=1, fmg=True)
    O = mg.ProgramB(mg.ConfigB(**kw))
    O.setup()
    O.Solve()
    P = SolverFromL3(ConfigL3(**kw), hip)
    P.setup()
    P.Solve()
    assert P.iterations == O.iterations
    _close(P.res_history, O.res_history)
    plain = mg.ProgramB(mg.ConfigB(**dict(kw, fmg=False)))
    plain.setup()
    plain.Solve()
    assert O.res_history[1] < 0.2 * plain.res_history[1]          # the nested start pays off: first V-cycle starts far closer
    # red-black sweeps as single out-of-place passes (second array's boundary planes follow SetFuncDir / ResetBC): same bits
    Q = SolverFromL3(ConfigL3(**dict(kw, max_level=7), fused_rbgs=True, fused_residual_restrict=True), hip)
    Q.setup()
    Q.Solve()
    R = SolverFromL3(ConfigL3(**dict(kw, max_level=7)), hip)
    R.setup()
    R.Solve()
    assert Q.res_history == R.res_history
