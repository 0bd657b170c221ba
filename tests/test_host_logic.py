"""The product's host driver (exastencils_amd.solver: the mirror of the generated mgCycle/Solve functions)
run on the CPU with the oracle's loops standing in for the HIP kernels: it must print the reference's
golden histories too, and agree with the oracle's own driver to the last bit (same loops, same order)."""
import pytest

from golden_cases import CASES, golden_text, oracle_program
from oracle import mg
from oracle_ops import OracleOps

from exastencils_amd.domain import RectDomain
from exastencils_amd.layout import FieldLayout
from exastencils_amd.solver import ConfigL3, ConfigL4, SolverFromL3, SolverFromL4


def product_program(name, ops, domain=None, comm=None, **override):
    c = dict(CASES[name])
    c.update(override)
    prog = c.pop("program")
    c.pop("frags")
    c.pop("frag_len")
    flen = c.pop("single_len")
    c["frag_len"] = override.get("frag_len", flen)
    if prog == "A":
        c.pop("kappa", None)
        return SolverFromL4(ConfigL4(fused_coarse=False, **c), ops, domain, comm)
    return SolverFromL3(ConfigL3(**c), ops, domain, comm)


@pytest.mark.parametrize("name", ["CommBasic_PureMPI", "Poisson_2D_FD_Poisson_fromL4", "SISC_3D_ConstCoeff",
                                  "SISC_3D_VarCoeff", "FMG_3D_Trigonometric", "Opts_seq", "Misc_inlining"])
def test_host_driver_reproduces_goldens(name):
    P = product_program(name, OracleOps())
    P.setup()
    P.Solve()
    assert mg.compare_with_golden(P.log, golden_text(name)) == [], P.log
    O = oracle_program(name)
    O.setup()
    O.Solve()
    assert P.res_history == O.res_history
    assert P.err_history == O.err_history


def test_folded_prolongation_in_the_host_driver():
    """mgCycle with the correction loop folded into the first post-smoothing sweep (one kernel-layer call instead of
    prolong_add + sweep; here the oracle's loops stand in): same history as the statement-by-statement cycle."""
    hist = []
    for min_points in (0, 1):
        P = SolverFromL4(ConfigL4(nd=3, min_level=1, max_level=4, tol=1e-8, fused_coarse=False, fused_rbgs=True,
                                  fused_prolong_min_points=min_points, fused_zero_start=bool(min_points), fused_residual_restrict=True,
                                  fused_residual_norm=bool(min_points)), OracleOps())
        P.setup()
        P.Solve()
        assert P._folds_prolongation(4) == bool(min_points)
        hist.append(P.res_history)
    assert hist[0] == hist[1] and len(hist[0]) > 3
    hist = []
    for min_points in (0, 1):        # the slotted Jacobi program with pairs of steps: Correction rides on the first pair
        P = SolverFromL3(ConfigL3(nd=3, min_level=1, max_level=4, temporal_blocking=True, fused_prolong_min_points=min_points), OracleOps())
        P.setup()
        P.Solve()
        assert P._folds_prolongation(4) == bool(min_points)
        hist.append(P.res_history)
    assert hist[0] == hist[1] and len(hist[0]) > 3


def test_three_sweeps_as_two_passes_of_three_colour_loops_in_the_host_driver():
    """mgCycle with three plain sweeps issued as two examg_rbgs_colours3 calls (first colour 0, then 1) and the correction as a loop of its
    own where those passes apply -- the host logic of SolverFromL4._smooth / _folds_prolongation, with the oracle's coloured loops standing
    in for the one-pass kernel: the same history, to the last bit, as a call per sweep; the pass count per cycle and level is odd on the
    levels that start from the zero field (zero sweep + two sweeps + two passes), which is what capture_cycle's two recordings are for."""
    class Ops(OracleOps):
        calls = 0

        def three_stage_eligible(self, lu, lf, st, begin, end):      # the kernel layer's answer: levels 4 and 3 of this small hierarchy
            return (end[0] - begin[0]) >= 7

        def rbgs_colours3(self, *a):
            Ops.calls += 1
            return OracleOps.rbgs_colours3(self, *a)

    hist = []
    for three in (False, True):
        P = SolverFromL4(ConfigL4(nd=3, min_level=1, max_level=4, tol=1e-8, fused_coarse=False, fused_rbgs=True, fused_prolong_min_points=1,
                                  fused_zero_start=True, fused_residual_restrict=True, fused_residual_norm=True, fused_rbgs3=three), Ops())
        P.setup()
        assert P._three_colour_passes(4) == three and P._three_colour_passes(3) == three and not P._three_colour_passes(2)
        # the correction is folded where a sweep takes one pass each; a loop of its own where three sweeps are two passes
        assert P._folds_prolongation(4) == (not three) and P._folds_prolongation(3) == (not three) and P._folds_prolongation(2)
        roles = [(P.Solution[l].slots[0].data_ptr(), P._sol_alt[l].data_ptr()) for l in (3, 4)]
        P.mgCycle(4)
        after = [(P.Solution[l].slots[0].data_ptr(), P._sol_alt[l].data_ptr()) for l in (3, 4)]
        if three:       # level 4: 2 + 2 passes (even); level 3: zero sweep + 2 sweeps + 2 passes (odd: its arrays have changed roles)
            assert after[1] == roles[1] and after[0] == (roles[0][1], roles[0][0])
        else:
            assert after == roles
        P.reset()
        P.Solve()
        hist.append(P.res_history)
    assert Ops.calls > 0 and hist[0] == hist[1] and len(hist[0]) > 3


def test_fmg_driver_with_one_pass_forms():
    """The layer-3 style driver (FMG start of Testing/FMG/3D_Trigonometric.exa4:189-242 with the red-black smoother of
    Testing/Smoothers/RBGS.exa4:125-133) with every one-pass form switched on -- correction folded into the first post-smoothing sweep,
    zero field left to the first pre-smoothing sweep, residual + norm, VCycle_0 as one call (alpha from the squared norm, no `apply bc`)
    -- against the statement-by-statement driver; on the CPU the oracle's loops stand in, so the histories are equal to the last bit."""
    kw = dict(nd=3, min_level=2, max_level=5, smoother="rbgs", omega=1.0, stencil="scaled", restrict_scale=1.0, tol=1e-9, bc_fn=1, fmg=True)
    plain = SolverFromL3(ConfigL3(**kw), OracleOps())
    plain.setup()
    plain.Solve()
    fused = SolverFromL3(ConfigL3(**kw, fused_rbgs=True, fused_residual_restrict=True, fused_prolong_min_points=1, fused_zero_start=True,
                                  fused_residual_norm=True, fused_coarse=True), OracleOps())
    fused.setup()
    assert fused._folds_prolongation(5) and fused._starts_from_zero(4) and not fused._starts_from_zero(2)
    fused.Solve()
    assert fused.iterations == plain.iterations and len(plain.res_history) > 2
    assert fused.res_history == plain.res_history
    O = mg.ProgramB(mg.ConfigB(**kw))
    O.setup()
    O.Solve()
    assert plain.res_history == O.res_history


def test_fmg_start_without_a_following_cycle_still_applies_its_last_correction():
    """The FMG start leaves Correction@finest (and ResetBC of the level below) to the first sweep of the cycle that follows; when no
    cycle follows (the loop condition of Solve is false at once) they run where the program has them: same Solution as the plain driver."""
    import numpy as np

    kw = dict(nd=3, min_level=2, max_level=5, smoother="rbgs", omega=1.0, stencil="scaled", restrict_scale=1.0, tol=2.0, bc_fn=1, fmg=True)
    plain = SolverFromL3(ConfigL3(**kw), OracleOps())
    plain.setup()
    plain.Solve()
    fused = SolverFromL3(ConfigL3(**kw, fused_rbgs=True, fused_residual_restrict=True, fused_prolong_min_points=1, fused_zero_start=True), OracleOps())
    fused.setup()
    assert fused._folds_prolongation(5)
    fused.Solve()
    assert plain.iterations == 0 and fused.iterations == 0 and getattr(fused, "_deferred_correction", None) is None
    for l in (4, 5):
        assert np.array_equal(plain.ops.to_host(plain.Solution[l].data()), fused.ops.to_host(fused.Solution[l].data()))


def test_one_call_coarse_solve_keeps_the_iteration_limit_message():
    """`print ( "Maximum number of cgs iterations (", n, ") was exceeded" )` after the CG loop: the one-call coarse solve counts the
    event on the device and the driver appends the message when Solve ends -- as often as the statement-by-statement driver prints it."""
    msgs = []
    for fused in (False, True):
        P = SolverFromL4(ConfigL4(nd=3, min_level=2, max_level=4, tol=1e-6, cg_max=2, max_it=3, fused_coarse=fused), OracleOps())
        P.setup()
        P.Solve()
        msgs.append([m for m in P.log if m.startswith("Maximum number of cgs iterations (2)")])
    assert msgs[0] == msgs[1] and len(msgs[0]) == 3


def test_layout_matches_reference_sizes():
    # 512^3 NodeWithComm => TOT = 2^L + 3 = 515 per dim, NodeNoGhost => 513
    l = FieldLayout.node(3, (512, 512, 512), 1)
    assert [l.tot(d) for d in range(3)] == [515, 515, 515] and l.ref(0) == 1 and l.size == 515 ** 3
    l = FieldLayout.node(3, (512, 512, 512), 0)
    assert [l.tot(d) for d in range(3)] == [513, 513, 513] and l.ref(0) == 0
    l = FieldLayout.node(2, (256, 256, 0), 1)
    assert l.shape_zyx == (1, 259, 259)
    # IR_AddPaddingToFieldLayouts with vector size 4: first dup point aligned, row length a multiple
    l = FieldLayout.node(3, (8, 8, 8), 1, align=4)
    assert l.pad_l[0] == 3 and l.ref(0) == 4 and l.tot(0) % 4 == 0 and l.tot(1) == 11
    assert l.idx("GLB", 0) == -1 and l.idx("DRE", 0) == 9 and l.idx("GRE", 0) == 10


def test_domain_rank_mapping_and_offsets():
    assert RectDomain.blocks_for(8, 3) == (1, 2, 4)      # the unit-stride dimension stays undivided
    assert RectDomain.blocks_for(4, 3) == (1, 2, 2)
    assert RectDomain.blocks_for(2, 3) == (1, 1, 2)
    assert RectDomain.blocks_for(4, 2) == (1, 4, 1)
    d = RectDomain(3, (2, 2, 2), rank=5)            # x-fastest: 5 = 1 + 2*(0 + 2*1)
    assert d.pos == (1, 0, 1)
    assert d.neighbor(0, -1) == 4 and d.neighbor(0, +1) is None
    assert d.neighbor(1, +1) == 7 and d.neighbor(2, -1) == 1
    lay = FieldLayout.node(3, d.ncells(3), 1)
    b, e = d.loop_bounds(lay)
    assert b == [0, 1, 0] and e == [8, 9, 8]       # interior faces include the dup node, physical faces do not
    b, e = d.loop_bounds(lay, reduction=True)
    assert b == [1, 1, 1]
    assert d.face_mask() == 0b100110
    g = d.geom(3)
    assert g.pos_begin[0] == 0.5 and g.pos_begin[1] == 0.0 and abs(g.h[0] - 1.0 / 16) < 1e-16


HELMHOLTZ27 = dict(nd=3, min_level=1, max_level=4, smoother="jacobi", omega=0.8, stencil="helmholtz27", restrict_scale=1.0, tol=1e-8,
                   cg_max=512, bc_fn=0, sol_fn=9, coef_fn=7, kappa=10.0, ksq=2.0, rhs_from_solution=True)


def test_config4_helmholtz27_program():
    """BASELINE.json config 4 (27-entry variable-coefficient Helmholtz; no golden in the reference -- parity unpinned):
    the product driver and the oracle program agree bit for bit, the V-cycle converges and the discrete manufactured
    solution sin(pi x) sin(pi y) sin(pi z) is recovered."""
    O = mg.ProgramB(mg.ConfigB(nfrag=(1, 1, 1), frag_len=(2, 2, 2), **HELMHOLTZ27))
    O.setup()
    O.Solve()
    P = SolverFromL3(ConfigL3(frag_len=(2, 2, 2), **HELMHOLTZ27), OracleOps())
    P.setup()
    P.Solve()
    assert P.res_history == O.res_history and P.err_history == O.err_history
    assert P.iterations <= 8 and P.err_history[-1] < 1e-8
    # pairs of Jacobi steps and the last pre-smoothing step + residual as single calls: the same statements, the same bits
    Q = SolverFromL3(ConfigL3(frag_len=(2, 2, 2), temporal_blocking=True, fused_smooth_residual=True, **HELMHOLTZ27), OracleOps())
    Q.setup()
    Q.Solve()
    assert Q.res_history == P.res_history and Q.err_history == P.err_history
    # the stencil field is symmetric positive: row sums equal -k^2 in the interior (constant functions are in the kernel
    # of the stiffness part), diagonal positive
    import numpy as np

    st = O.Laplace[4][(0, 0, 0)]
    lay = st.clayout
    cf = st.cfield.reshape(27, lay.tot(2), lay.tot(1), lay.tot(0))
    inner = (slice(None), slice(2, -2), slice(2, -2), slice(2, -2))
    rows = cf[inner].sum(axis=0)
    assert np.allclose(rows, -2.0, atol=1e-9 * np.abs(cf[0]).max())
    assert (cf[0][2:-2, 2:-2, 2:-2] > 0).all()


def test_pairs_and_step_with_residual_under_an_fmg_start_change_no_bit():
    """Testing/FMG/3D_VarCoeff.exa4's shape (stencil field, FMG start: SetFuncDir / ResetBC rewrite the boundary planes of EVERY slot, which is
    what the one-pass step + residual relies on) with Jacobi pairs and the last pre-smoothing step + residual as single calls: the same
    histories as statement by statement."""
    kw = dict(nd=3, min_level=1, max_level=4, smoother="jacobi", omega=0.85, stencil="varcoeff", restrict_scale=1.0, tol=1e-5,
              cg_max=1024, bc_fn=6, rhs_fn=5, sol_fn=6, coef_fn=7, kappa=10.0, fmg=True)
    P = SolverFromL3(ConfigL3(frag_len=(2, 2, 2), **kw), OracleOps())
    P.setup()
    P.Solve()
    Q = SolverFromL3(ConfigL3(frag_len=(2, 2, 2), temporal_blocking=True, fused_smooth_residual=True, **kw), OracleOps())
    Q.setup()
    Q.Solve()
    assert Q.res_history == P.res_history and Q.err_history == P.err_history and Q.iterations == P.iterations


def test_block_decompositions_of_a_node():
    """bench.py --blocks: 'zy' (default: 8 -> 1 x 2 x 4, the unit-stride dimension undivided), 'cube' (8 -> 2 x 2 x 2, SURVEY.md 8e /
    domain/ir/IR_ConnectFragments.scala:46-52) or explicit; rank = bx + nbx * (by + nby * bz); neighbours and iteration offsets follow."""
    import pytest

    from exastencils_amd.domain import RectDomain
    from exastencils_amd.layout import FieldLayout

    assert [RectDomain.parse_blocks("", n, 3) for n in (1, 2, 4, 8)] == [(1, 1, 1), (1, 1, 2), (1, 2, 2), (1, 2, 4)]
    assert [RectDomain.parse_blocks("cube", n, 3) for n in (2, 4, 8)] == [(2, 1, 1), (2, 2, 1), (2, 2, 2)]
    assert RectDomain.parse_blocks("2,2,2", 8, 3) == (2, 2, 2) and RectDomain.parse_blocks("1x2x4", 8, 3) == (1, 2, 4)
    assert RectDomain.parse_blocks("4,2", 8, 2) == (4, 2, 1)
    for bad, n, nd in (("2,2,2", 4, 3), ("3,3,1", 8, 3), ("2,2,2", 8, 2), ("a,b", 8, 3), ("0,8,1", 8, 3)):
        with pytest.raises(ValueError):
            RectDomain.parse_blocks(bad, n, nd)
    # 2 x 2 x 2: rank 5 = (1, 0, 1): neighbours across all three axes, one physical face per axis
    d = RectDomain(3, (2, 2, 2), 5)
    assert d.pos == (1, 0, 1)
    assert [d.neighbor(ax, s) for ax in range(3) for s in (-1, +1)] == [4, None, None, 7, 1, None]
    lay = FieldLayout.node(3, d.ncells(4), 1)
    b, e = d.loop_bounds(lay)
    assert b == [0, 1, 0] and e == [16, 17, 16]           # iteration offsets 0 at interior faces, 1 / -1 at physical ones
    rb, _ = d.loop_bounds(lay, reduction=True)
    assert rb == [1, 1, 1]                                # reductions leave the lower duplicate planes to the neighbour
