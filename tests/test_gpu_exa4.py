"""ExaSlang-4 programs interpreted onto libexamg on the MI355X: every `loop over` of the example
programs is one HIP kernel launch through the C ABI; histories are checked against the CPU oracle programs."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


def _close(got, want, r0):
    assert len(got) == len(want), (got, want)
    for x, y in zip(got, want):
        assert abs(x - y) <= 1e-10 * abs(y) + 1e-13 * r0, (got, want)


@pytest.fixture(scope="module")
def hip():
    from exastencils_amd.ops import HipOps

    return HipOps(0)


def test_rbgs_program_on_gpu(hip):
    from test_exa4 import _oracle_a, example

    P = example("poisson3d_rbgs.exa4", 2, 7, ops=hip)
    P.run()
    O = _oracle_a(2, 7)
    _close(P.printed_values, O.res_history, O.res_history[0])   # norms: the device reduction tree sums in another order
    assert P.launches > 100                                      # every launch a kernel-layer call (6 cycles of ~40 launches)
    plain = example("poisson3d_rbgs.exa4", 2, 7, ops=hip, fuse=False, fuse_coarse_solver=True)
    plain.run()
    assert plain.printed_values == P.printed_values              # fused sweeps / cross-statement fusions change no bit
    assert plain.launches > P.launches
    # the coarsest level's CG statement by statement (its dots through examg_dot: another, equally fixed summation tree): to rounding
    stmt = example("poisson3d_rbgs.exa4", 2, 7, ops=hip, fuse=False)
    stmt.run()
    _close(stmt.printed_values, P.printed_values, P.printed_values[0])
    assert P.fusions["residual_restrict"] > 0
    # opt-in: residual + norm in one pass -- the squares are summed in the residual kernel's order: same fields, norms to rounding
    N = example("poisson3d_rbgs.exa4", 2, 7, ops=hip)
    N.fuse_residual_norm = True
    N.run()
    assert N.fusions["residual_norm"] > 0
    _close(N.printed_values, P.printed_values, P.printed_values[0])
    assert np.array_equal(hip.to_host(N.fields[("u", 7)].data()), hip.to_host(P.fields[("u", 7)].data()))


def test_cycle_function_replays_from_a_graph(hip):
    """auto_graph: `Cycle@finest ( )` -- no reduction, print or builtin anywhere below it, the coarsest level's generated CG as one
    kernel -- is recorded into a hipGraph at its second call of the Solve loop and replayed from the third on: the printed residuals and
    the fields are those of the interpreted program, bit for bit; functions that return to the host (Norm) are never recorded."""
    from test_exa4 import example

    plain = example("poisson3d_rbgs.exa4", 2, 7, ops=hip, auto_graph=False)
    plain.run()
    P = example("poisson3d_rbgs.exa4", 2, 7, ops=hip)
    assert P.auto_graph
    P.run()
    assert P.printed_values == plain.printed_values and len(P.printed_values) > 4
    assert P.graph_replays >= len(P.printed_values) - 2
    assert isinstance(P._auto_graphs.get(("Cycle", 7)), dict) and ("Norm", 7) not in P._auto_graphs
    assert abs(P.launches - plain.launches) <= len(P.printed_values)      # recorded launches are counted per replay (a recorded function ends with its pending loop issued)
    assert np.array_equal(hip.to_host(P.fields[("u", 7)].data()), hip.to_host(plain.fields[("u", 7)].data()))


def test_native_rand_fill_on_gpu(hip):
    """The host-generated std::rand() start values (tests/test_exa4.py: glibc's sequence) reach the device field unchanged."""
    from oracle_ops import OracleOps
    from test_exa4 import RANDOM_START

    from exastencils_amd import exa4

    outs = []
    for ops in (hip, OracleOps()):
        P = exa4.Exa4Program(RANDOM_START, dict(dimensionality=3, minLevel=2, maxLevel=5), ops=ops)
        P.run()
        outs.append((P.printed_values[0], ops.to_host(P.fields[("u", 5)].data()).copy()))
    assert np.array_equal(outs[0][1], outs[1][1])
    assert abs(outs[0][0] - outs[1][0]) <= 1e-13 * outs[1][0]


def test_jacobi_program_on_gpu(hip):
    from test_exa4 import _oracle_b, example

    P = example("jacobi3d_slots.exa4", 1, 6, ops=hip)
    P.run()
    O = _oracle_b(1, 6)
    _close(P.printed_values, O.res_history, O.res_history[0])
    assert P.out == O.log
    plain = example("jacobi3d_slots.exa4", 1, 6, ops=hip, fuse=False, fuse_coarse_solver=True)      # the same coarse solver on both sides
    plain.run()
    assert plain.printed_values == P.printed_values              # paired Jacobi steps change no bit


def test_varcoeff_program_on_gpu(hip):
    from test_exa4 import VARCOEFF, _oracle_b, example

    P = example("varcoeff3d.exa4", 1, 6, ops=hip)
    P.run()
    O = _oracle_b(1, 6, **VARCOEFF)
    res, err = P.printed_values[:1] + P.printed_values[1::2], P.printed_values[2::2]
    _close(res, O.res_history, O.res_history[0])          # exp() in coefficients and boundary values: device libm
    assert len(err) == len(O.err_history)
    for x, y in zip(err, O.err_history):
        assert abs(x - y) <= 1e-12
    assert P.out[-1] == str(O.iterations)


def test_command_line_runs_a_program():
    r = subprocess.run([sys.executable, "-m", "exastencils_amd.exa4", os.path.join(ROOT, "examples", "exa4", "poisson3d_rbgs.exa4"),
                        "--set", "dimensionality=3", "--set", "minLevel=2", "--set", "maxLevel=5"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "initial residual 39640.6"
    assert lines[-1].startswith("cycle 6 residual 0.0207")


def test_captured_cycle_replays_like_direct_calls(hip):
    """One `Cycle@finest` of the red-black program recorded into a hipGraph (fused sweeps, coarse CG as one kernel):
    replays reproduce the residuals of direct calls."""
    from test_exa4 import example

    def residuals(use_graph):
        P = example("poisson3d_rbgs.exa4", 3, 7, ops=hip)
        P._apply_bc(P.fields[("u", 7)], 0)
        out = []
        if use_graph:
            g = P.capture("Cycle", 7)           # two cycles have run when this returns (recording executes nothing)
        else:
            for _ in range(2):
                P.call("Cycle", 7)
        for _ in range(3):
            g.replay() if use_graph else P.call("Cycle", 7)
            P.call("Defect", 7)
            out.append(P.call("Norm", 7))
        return out

    a, b = residuals(False), residuals(True)
    assert a == b
    assert a[2] < 2e-2 * a[0]


def test_periodic_direction_on_gpu(hip):
    """`domain_rect_periodic_z`: the block is its own neighbour along z (self-exchange through pack/unpack); the GPU run
    follows the CPU-ops run of the same interpreted program."""
    from oracle_ops import OracleOps
    from test_exa4 import example

    from exastencils_amd.domain import RectDomain

    def run(ops):
        P = example("jacobi3d_slots.exa4", 1, 6, ops=ops, domain=RectDomain(3, (1, 1, 1), 0, (1, 1, 1), periodic=(False, False, True)))
        P.run()
        return P

    G, C_ = run(hip), run(OracleOps())
    _close(G.printed_values, C_.printed_values, C_.printed_values[0])
    assert G.out[-1] == C_.out[-1]
    assert G.comm.stats["messages"] == 0          # no other block: local copies only


def test_program_path_of_boundary_values_equals_built_in_path(hip, monkeypatch):
    """The red-black example with its boundary polynomial forced through the expression-program kernels: same bits."""
    from test_exa4 import example

    from exastencils_amd import exa4

    ref = example("poisson3d_rbgs.exa4", 2, 6, ops=hip)
    ref.run()

    def refuse(self, e, lvl):
        if e == ("num", 0.0):
            return 0, ()
        raise exa4.Exa4Unsupported("forced")

    monkeypatch.setattr(exa4.Exa4Program, "_recognise", refuse)
    P = example("poisson3d_rbgs.exa4", 2, 6, ops=hip)
    assert not isinstance(P.fields[("u", 6)].bc_fn, int)
    P.run()
    assert P.printed_values == ref.printed_values


def test_stencil_field_entries_as_programs_equal_the_dedicated_kernel(hip, monkeypatch):
    """Seven coefficient expressions filled plane by plane by the expression kernel == examg_init_varcoeff7, bit for bit."""
    from test_exa4 import example

    from exastencils_amd import exa4

    ref = example("varcoeff3d.exa4", 1, 5, ops=hip)
    ref.run()

    def refuse(self, got, want, lvl):
        raise exa4.Exa4Unsupported("forced")

    monkeypatch.setattr(exa4.Exa4Program, "_varcoeff_function", refuse)
    P = example("varcoeff3d.exa4", 1, 5, ops=hip)
    P.run()
    assert P.printed_values == ref.printed_values


def test_layout_transformation_is_applied_to_the_coefficient_field(hip):
    """SURVEY.md f-2, the reference's `LayoutTransformations` mechanism (Compiler/src/exastencils/layoutTransformation/,
    Testing/LayoutTrafo/*.exa4): `transform coeff with [x, y, z, i] => [i, x, y, z]` re-lays the stencil field's coefficients out
    with the entries of a point contiguous and every loop reads them through the transformed index -- the program prints what the
    untransformed program prints, bit for bit (the reference tests its transformed programs against the same .results files)."""
    from test_exa4 import example, transformed_varcoeff

    P = transformed_varcoeff(1, 5, ops=hip)
    P.run()
    assert P._sf_rec and all(t.ctransform == 1 for t in P._sf_rec.values()) and len(P._sf_rec) == 5
    Q = example("varcoeff3d.exa4", 1, 5, ops=hip)
    Q.run()
    assert P.printed_values == Q.printed_values


def test_field_io_round_trip_on_gpu(hip, tmp_path):
    """examples/exa4/iotest3d.exa4 (shape of the reference's Testing/IOTest/3D_Scalar_CheckEquality_ReadAfterWrite.exa4)
    through the interpreter on device fields: ascii write / read with ghost layers, raw-double write / read, printField --
    data cross through to_host / from_host; the binary round trip is exact, the ascii one within its 7 digits."""
    from test_exa4 import _iotest

    P = _iotest(hip, tmp_path, level=4)
    assert P.launches >= 6


def test_field_io_files_equal_the_cpu_run_byte_for_byte(hip, tmp_path):
    """What writeField / printField put on disk from DEVICE fields against the files the same program writes with the oracle's loops
    as kernel layer: ascii with ghost layers (`lock` interface), raw doubles (file per process) and the printField table -- byte for
    byte (the field is a polynomial of the node position: the device fill is bit-exact).  Not a round trip: two independent writers."""
    from oracle_ops import OracleOps
    from test_exa4 import _iotest

    (tmp_path / "gpu").mkdir()
    (tmp_path / "cpu").mkdir()
    _iotest(hip, tmp_path / "gpu", level=4)
    _iotest(OracleOps(), tmp_path / "cpu", level=4)
    for name in ("src_lock.txt", "src_fpp_0.bin", "src_vis.csv"):
        a = (tmp_path / "gpu" / "data" / name).read_bytes()
        b = (tmp_path / "cpu" / "data" / name).read_bytes()
        assert len(a) > 0 and a == b, "%s differs between the device run and the CPU run" % name


def test_contracting_loop_three_plus_two_steps_on_gpu(hip):
    """The same program on a 256^3 block (the three-stage pass applies from 8e6 points on): the five contracted steps of a Smoother call run
    as a pass of three (examg_jacobi3) and a pass of two -- two launches instead of five, the printed norms bit-identical."""
    from test_exa4 import example

    P = example("jacobi3d_contraction.exa4", 8, 8, ops=hip)
    P.run()
    plain = example("jacobi3d_contraction.exa4", 8, 8, ops=hip, fuse=False)
    plain.run()
    assert P.printed_values == plain.printed_values and len(P.printed_values) > 0
    assert plain.launches - P.launches >= 9      # three Smoother calls: 2 instead of 5 launches each (pairs alone would save 6)


def test_contracting_loop_on_gpu(hip):
    """`repeat 5 times with contraction [1, 1, 1]` (node-grid reduction of the reference's Testing/PolyExpl/Jac3Dcc.exa4:1-33) on the
    device: two two-step passes + one step per Smoother call, same bits as five plain launches, norms as the CPU ops print them."""
    from oracle_ops import OracleOps
    from test_exa4 import example

    P = example("jacobi3d_contraction.exa4", 6, 6, ops=hip)
    P.run()
    plain = example("jacobi3d_contraction.exa4", 6, 6, ops=hip, fuse=False)
    plain.run()
    assert P.printed_values == plain.printed_values and P.launches < plain.launches
    O = example("jacobi3d_contraction.exa4", 6, 6, ops=OracleOps(), fuse=False)
    O.run()
    for x, y in zip(P.printed_values, O.printed_values):
        assert abs(x - y) <= 1e-13 * abs(y), (P.printed_values, O.printed_values)
