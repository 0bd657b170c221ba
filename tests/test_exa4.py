"""ExaSlang-4 subset reader / interpreter (exastencils_amd/exa4.py) on the CPU: the interpreter
issues kernel-layer calls, the oracle's loops stand in for the HIP kernels (tests/oracle_ops.py).

 * the reference's own programs (read from /root/reference when it is present -- they are its sources and are not
   copied into this repository) reproduce the reference's checked-in *.results files line by line;
 * this repository's example programs (examples/exa4/*.exa4) reproduce the oracle programs bit for bit;
 * parser details and the refusal of constructs outside the subset."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from golden_cases import golden_text  # noqa: E402
from oracle_ops import OracleOps  # noqa: E402

from exastencils_amd import exa4, knowledge  # noqa: E402
from exastencils_amd.domain import RectDomain  # noqa: E402

REF = "/root/reference"
EX = os.path.join(ROOT, "examples", "exa4")

# golden name -> (program, knowledge) relative to the reference checkout
REFERENCE_PROGRAMS = {
    "Poisson_2D_FD_Poisson_fromL4": ("Examples/Poisson/2D_FD_Poisson_fromL4.exa4", "Testing/Poisson/2D_FD_Poisson_fromL4.knowledge"),
    "CommBasic_PureMPI": ("Testing/CommBasic/PureMPI.exa4", "Testing/CommBasic/PureMPI.knowledge"),
    "SISC_3D_ConstCoeff": ("Testing/SISC/3D_ConstCoeff.exa4", "Testing/SISC/3D_ConstCoeff.knowledge"),
    "SISC_3D_VarCoeff": ("Testing/SISC/3D_VarCoeff.exa4", "Testing/SISC/3D_VarCoeff.knowledge"),
    "FMG_3D_Trigonometric": ("Testing/FMG/3D_Trigonometric.exa4", "Testing/FMG/3D_Trigonometric.knowledge"),
    "FMG_3D_VarCoeff": ("Testing/FMG/3D_VarCoeff.exa4", "Testing/FMG/3D_VarCoeff.knowledge"),
    "Smoothers_RBGS": ("Testing/Smoothers/RBGS.exa4", "Testing/Smoothers/RBGS.knowledge"),
    "Smoothers_Jac": ("Testing/Smoothers/Jac.exa4", "Testing/Smoothers/Jac.knowledge"),
}
SLOW = {"Smoothers_RBGS", "Smoothers_Jac"}      # 576^3 on the CPU

# Further tests of the reference's suite inside the subset: program, knowledge and expected output are all read from the
# reference checkout (Testing/<name>.{exa4,knowledge,results}); compared under the harness' own rule.
MORE_REFERENCE_TESTS = ["Misc/MathFunctionEvaluation", "Misc/inlining", "BC/2D_Polynomial", "BC/2D_Trigonometric", "BC/3D_Polynomial", "BC/3D_Trigonometric", "CommBasic/2D",
                        "CommBasic/PureOMP", "FMG/2D_ConstCoeff", "FMG/2D_Polynomial", "SISC/2D_ConstCoeff", "SISC/2D_VarCoeff",
                        "CUDA/2D_VarCoeff", "BC/2D_Periodic", "BC/3D_Periodic", "CommBasic/Hybrid", "CommBasic/Strategy26",
                        "CommBasic/Summarize"]
MORE_SLOW = {"CommBasic/Hybrid", "CommBasic/Strategy26", "CommBasic/Summarize"}      # 10-20 s each on the CPU


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
@pytest.mark.parametrize("name", MORE_REFERENCE_TESTS)
def test_more_reference_tests_reproduce_their_results_files(name):
    from oracle import mg

    if name in MORE_SLOW and not os.environ.get("EXAMG_SLOW"):
        pytest.skip("set EXAMG_SLOW=1")
    base = os.path.join(REF, "Testing", name)
    k = knowledge.parse_file(base + ".knowledge")
    k["testing_enabled"] = True
    with open(base + ".exa4") as f:
        P = exa4.Exa4Program(f.read(), k, ops=OracleOps())
    out = P.run()
    with open(base + ".results") as f:
        want = f.read()
    assert mg.compare_with_golden(out, want) == []      # Testing/run_test.py:12-42


# programs whose expected output lives under another test's name (the reference's CI pairs them the same way, .gitlab-ci.yml:760-767):
# the generator-optimisation tests (one program, many knowledge files, one results file) and the layout-transformation tests
SHARED_RESULTS = [
    ("Testing/Opts/base.exa4", "Testing/Opts/seq_naive.knowledge", "Testing/Opts/seq.results"),
    ("Testing/Opts/base.exa4", "Testing/Opts/seq_all.knowledge", "Testing/Opts/seq.results"),
    ("Testing/LayoutTrafo/opts.exa4", "Testing/LayoutTrafo/seq_naive.knowledge", "Testing/Opts/seq.results"),
    # 2 x 2 x 2 processes of 256^3 cells, each with std::srand(mpiRank): run as one merged block, one generator per former process
    ("Testing/Opts/base_par.exa4", "Testing/Opts/par_naive.knowledge", "Testing/Opts/par.results"),
    ("Testing/LayoutTrafo/opts.exa4", "Testing/LayoutTrafo/par_all.knowledge", "Testing/Opts/par.results"),
    # the reference's own temporal-blocking tests (.gitlab-ci.yml:437-483,521-542): the Opts program with its smoother calls
    # under `repeat .. with contraction`, checked against the results of the untransformed program
    ("Testing/Opts/tempBlock.exa4", "Testing/Opts/seq_all.knowledge", "Testing/Opts/seq.results"),
    ("Testing/Opts/tempBlock.exa4", "Testing/Opts/seq_poly.knowledge", "Testing/Opts/seq.results"),
    ("Testing/Opts/tempBlock_par.exa4", "Testing/Opts/par_all.knowledge", "Testing/Opts/par.results"),
    # the reference's CUDA CI jobs (.gitlab-ci.yml:787-797): the programs of Testing/CUDA with `cuda_enabled` knowledge must print the
    # SISC results files -- its own CPU <-> GPU parity precedent; here the same pairs through the interpreter
    ("Testing/CUDA/2D_ConstCoeff.exa4", "Testing/CUDA/2D_ConstCoeff.knowledge", "Testing/SISC/2D_ConstCoeff.results"),
    ("Testing/CUDA/2D_ConstCoeff.exa4", "Testing/CUDA/2D_ConstCoeff_VarFieldSize.knowledge", "Testing/SISC/2D_ConstCoeff.results"),
    ("Testing/CUDA/2D_VarCoeff.exa4", "Testing/CUDA/2D_VarCoeff_VarFieldSize.knowledge", "Testing/CUDA/2D_VarCoeff.results"),
    ("Testing/CUDA/3D_ConstCoeff.exa4", "Testing/CUDA/3D_ConstCoeff.knowledge", "Testing/SISC/3D_ConstCoeff.results"),
    ("Testing/CUDA/3D_ConstCoeff.exa4", "Testing/CUDA/3D_ConstCoeff_VarFieldSize.knowledge", "Testing/SISC/3D_ConstCoeff.results"),
    ("Testing/CUDA/3D_VarCoeff.exa4", "Testing/CUDA/3D_VarCoeff.knowledge", "Testing/SISC/3D_VarCoeff.results"),
    ("Testing/CUDA/3D_VarCoeff.exa4", "Testing/CUDA/3D_VarCoeff_VarFieldSize.knowledge", "Testing/SISC/3D_VarCoeff.results"),
    # common-subexpression-elimination tests of the generator (.gitlab-ci.yml:732-746): knowledge switches only, same results
    ("Testing/SISC/2D_VarCoeff.exa4", "Testing/CSE/2D_VarCoeff_conv.knowledge", "Testing/SISC/2D_VarCoeff.results"),
    ("Testing/SISC/2D_VarCoeff.exa4", "Testing/CSE/2D_VarCoeff_lc.knowledge", "Testing/SISC/2D_VarCoeff.results"),
    ("Testing/SISC/3D_VarCoeff.exa4", "Testing/CSE/3D_VarCoeff_both.knowledge", "Testing/SISC/3D_VarCoeff.results"),
]
SHARED_SLOW = {"Testing/Opts/par_naive.knowledge", "Testing/LayoutTrafo/par_all.knowledge", "Testing/Opts/par_all.knowledge"}      # 512^3 on the CPU, ~1 min each


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
@pytest.mark.parametrize("prog,know,res", SHARED_RESULTS)
def test_opts_and_layout_transformation_programs(prog, know, res):
    """Testing/Opts: `Solution = native("((double)std::rand()/RAND_MAX)")` in a `sequentially` loop -- the C library's generator,
    restated (glibc TYPE_3) -- and every optimisation switch of the generator in the knowledge file (none changes a result);
    Testing/LayoutTrafo: the same program with a `LayoutTransformations` block (storage directives: recorded, not applied)."""
    from oracle import mg

    if not os.path.exists(os.path.join(REF, know)):
        pytest.skip("knowledge file not in this checkout")
    if know in SHARED_SLOW and not os.environ.get("EXAMG_SLOW"):
        pytest.skip("512^3 on the CPU: set EXAMG_SLOW=1")
    k = knowledge.parse_file(os.path.join(REF, know))
    k["testing_enabled"] = True
    with open(os.path.join(REF, prog)) as f:
        P = exa4.Exa4Program(f.read(), k, ops=OracleOps())
    out = P.run()
    with open(os.path.join(REF, res)) as f:
        assert mg.compare_with_golden(out, f.read()) == []


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_layout_transformed_smoother_program_equals_the_plain_one():
    """Testing/LayoutTrafo/rbgs.exa4 (transform / concat / rename directives on Solution, RHS, Residual) prints what
    Testing/Smoothers/RBGS.exa4 prints -- the reference compares it with Smoothers/RBGS.results; here on one 64^3 block
    (EXAMG_SLOW=1: the 576^3 decomposition of the knowledge file against that results file)."""
    from oracle import mg

    outs = []
    for prog in ("Testing/LayoutTrafo/rbgs.exa4", "Testing/Smoothers/RBGS.exa4"):
        k = knowledge.parse_file(os.path.join(REF, "Testing/LayoutTrafo/RBGS.knowledge"))
        k["testing_enabled"] = True
        if not os.environ.get("EXAMG_SLOW"):
            for a in "xyz":
                k["domain_rect_numBlocks_" + a] = 1
                k["domain_rect_numFragsPerBlock_" + a] = 1
            k["domain_numBlocks"] = k["domain_numFragmentsPerBlock"] = 1
        with open(os.path.join(REF, prog)) as f:
            P = exa4.Exa4Program(f.read(), k, ops=OracleOps())
        outs.append(P.run())
        if prog.startswith("Testing/LayoutTrafo"):
            assert len(P.ast.layout_transformations) == 4
    assert outs[0] == outs[1] and len(outs[0]) > 3
    if os.environ.get("EXAMG_SLOW"):
        with open(os.path.join(REF, "Testing/Smoothers/RBGS.results")) as f:
            assert mg.compare_with_golden(outs[0], f.read()) == []


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
@pytest.mark.parametrize("name", sorted(REFERENCE_PROGRAMS))
def test_reference_program_reproduces_its_results_file(name):
    if name in SLOW and not os.environ.get("EXAMG_SLOW"):
        pytest.skip("576^3 on the CPU: set EXAMG_SLOW=1")
    prog, know = REFERENCE_PROGRAMS[name]
    k = knowledge.parse_file(os.path.join(REF, know))
    k["testing_enabled"] = True             # what Testing/run_test.py adds to every knowledge file
    with open(os.path.join(REF, prog)) as f:
        P = exa4.Exa4Program(f.read(), k, ops=OracleOps())
    out = P.run()
    want = [l.strip() for l in golden_text(name).splitlines() if l.strip()]
    assert out == want
    assert P.launches > 100


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_benchmark_program_runs_end_to_end():
    """Benchmark/Poisson3D (the program behind BASELINE's V-cycle metric) at a reduced depth: same histories as the oracle
    program, timers and the printJSON record are there."""
    k = knowledge.parse_file(os.path.join(REF, "Benchmark/Poisson3D/3D_FD_Poisson_fromL4.knowledge"))
    k.update(minLevel=2, maxLevel=5)
    with open(os.path.join(REF, "Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4")) as f:
        P = exa4.Exa4Program(f.read(), k, ops=OracleOps())
    P.run()
    O = _oracle_a(2, 5)
    # prints: starting residual, then (residual, convergence factor) per cycle
    assert P.printed_values[:1] + P.printed_values[1::2] == O.res_history
    assert P.out[0].startswith("Starting residual") and any(l.startswith("Mean") for l in P.out)
    assert set(P.json_results["results.json"]) == {"totalTimeSolve", "totalSetupTime"}
    assert P.json_results["results.json"]["totalTimeSolve"] > 0.0


def _oracle_a(lo, hi):
    from oracle import mg

    O = mg.ProgramA(mg.ConfigA(nd=3, min_level=lo, max_level=hi, tol=1e-6))
    O.setup()
    O.Solve()
    return O


def _oracle_b(lo, hi, **kw):
    from oracle import mg

    O = mg.ProgramB(mg.ConfigB(nd=3, min_level=lo, max_level=hi, **kw))
    O.setup()
    O.Solve()
    return O


VARCOEFF = dict(omega=0.85, stencil="varcoeff", restrict_scale=1.0, cg_max=1024, bc_fn=6, rhs_fn=5, sol_fn=6, coef_fn=7, kappa=10.0)


def example(name, lo, hi, ops=None, **kw):
    with open(os.path.join(EX, name)) as f:
        return exa4.Exa4Program(f.read(), dict(dimensionality=3, minLevel=lo, maxLevel=hi), ops=ops or OracleOps(), **kw)


def test_example_rbgs_program_matches_oracle_bitwise():
    P = example("poisson3d_rbgs.exa4", 2, 5)
    P.run()
    O = _oracle_a(2, 5)
    assert P.printed_values == O.res_history
    assert P.out[0].startswith("initial residual") and P.out[-1].startswith("cycle %d residual" % O.iterations)
    assert "solve" in P.timers


def test_example_jacobi_program_matches_oracle_bitwise():
    P = example("jacobi3d_slots.exa4", 0, 4)
    P.run()
    O = _oracle_b(0, 4)
    assert P.printed_values == O.res_history
    assert P.out == O.log


TRANSFORMED = ("LayoutTransformations {\n  transform coeff@(coarsest to finest), g@finest with [x, y, z, i] => [i, x, y, z]\n"
               "  transform r with [x, y, z] => [z, y, x]\n}\n\n")


def transformed_varcoeff(lo, hi, ops=None):
    """examples/exa4/varcoeff3d.exa4 under a LayoutTransformations block that puts the entries of the coefficient field first."""
    with open(os.path.join(EX, "varcoeff3d.exa4")) as f:
        return exa4.Exa4Program(TRANSFORMED + f.read(), dict(dimensionality=3, minLevel=lo, maxLevel=hi), ops=ops or OracleOps())


def test_layout_transformation_of_a_coefficient_field_is_recognised():
    """`transform coeff@.. with [x, y, z, i] => [i, x, y, z]` selects the stencil field built on that coefficient field on every level
    named; the scalar-field directives stay recorded only.  (The CPU kernel layer has no transformed layouts: same results either way;
    the transformation is applied on the GPU, tests/test_gpu_exa4.py.)"""
    P = transformed_varcoeff(1, 4)
    assert len(P.ast.layout_transformations) == 2
    names = {n for n, _ in P._sf_entry_fastest}
    assert len(names) == 1 and {l for _, l in P._sf_entry_fastest} == {1, 2, 3, 4}
    P.run()
    Q = example("varcoeff3d.exa4", 1, 4)
    Q.run()
    assert P.printed_values == Q.printed_values


def test_example_varcoeff_program_matches_oracle_bitwise():
    P = example("varcoeff3d.exa4", 0, 4)
    P.run()
    O = _oracle_b(0, 4, **VARCOEFF)
    res, err = P.printed_values[:1] + P.printed_values[1::2], P.printed_values[2::2]
    assert res == O.res_history
    assert err == O.err_history
    assert P.out[-1] == str(O.iterations)


def test_fused_passes_change_no_bit():
    """Red-black sweeps as one out-of-place pass and slotted Jacobi steps in pairs: same printed values, fewer launches."""
    for name, lo, hi in (("poisson3d_rbgs.exa4", 2, 5), ("jacobi3d_slots.exa4", 1, 4)):
        plain = example(name, lo, hi, fuse=False)
        plain.run()
        fused = example(name, lo, hi)
        fused.fuse_min_row = 8
        fused.run()
        assert fused.printed_values == plain.printed_values
        assert fused.launches < plain.launches


@pytest.mark.parametrize("after,field", [
    ("  Defect@finest ( )\n  Var res0", "r@finest"),                      # pending residual loop
    ("  apply bc to u@coarser\n  Cycle@coarser", "u@coarser"),            # pending `u@coarser = 0.0`
    ("  apply bc to u\n  Sweeps ( )\n}\n\nFunction Cycle@coarsest", "u"),   # pending correction loop
])
def test_field_io_sees_the_pending_loop(tmp_path, after, field):
    """A host-side field access (printField / writeField) right after a loop the interpreter keeps pending for a one-pass kernel
    must see that loop's result: the files written with fuse=True equal those written with fuse=False byte for byte."""
    with open(os.path.join(EX, "poisson3d_rbgs.exa4")) as f:
        text = f.read()
    head, tail = after.split("\n", 1)
    assert text.count(after) == 1
    files = {}
    for fuse in (False, True):
        out = str(tmp_path / ("field_%d.txt" % fuse))
        prog = text.replace(after, head + '\n  printField ( "%s", %s )\n' % (out, field) + tail)
        P = exa4.Exa4Program(prog, dict(dimensionality=3, minLevel=2, maxLevel=6), ops=OracleOps(), fuse=fuse)
        P.fuse_min_row, P.fused_prolong_min_points = 8, 0
        P.run()
        files[fuse] = (open(out, "rb").read(), P.printed_values)
    assert files[True][0] == files[False][0] and len(files[True][0]) > 1000
    if field != "u@coarser":          # (that one is the zero field by construction)
        assert any(float(l.split()[-1]) != 0.0 for l in files[True][0].decode().splitlines() if l.strip())
    assert files[True][1] == files[False][1]


def test_generated_cg_solver_is_recognised():
    """`Cycle@coarsest` of the example (and the reference's mgCycle@coarsest, when present) is the generated CG solver
    and becomes one examg_cg_coarse call; so does the slotted program's CG -- the layer-3 generator's form: alpha from the squared
    norm, no `apply bc`, both updates in one loop (examg_cg_coarse_variant, flags 3)."""
    P = example("poisson3d_rbgs.exa4", 2, 5)
    assert P._coarse_cg_plan(P._resolve("Cycle", 2), 2) is not None
    P.run()
    Q = example("poisson3d_rbgs.exa4", 2, 5, fuse_coarse_solver=False)
    Q.run()
    assert P.printed_values == Q.printed_values          # on the CPU ops the fused call runs the same loops
    assert P.launches < Q.launches
    J = example("jacobi3d_slots.exa4", 1, 4)
    plan = J._coarse_cg_plan(J._resolve("Cycle", 1), 1)
    assert plan is not None and plan[-1] == 3 and plan[0].num_slots == 2
    J.run()
    K = example("jacobi3d_slots.exa4", 1, 4, fuse_coarse_solver=False)
    K.run()
    assert J.printed_values == K.printed_values and J.launches < K.launches
    if os.path.isdir(REF):
        for prog, know in (("Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4", None),
                           ("Examples/Poisson/2D_FD_Poisson_fromL4.exa4", "Testing/Poisson/2D_FD_Poisson_fromL4.knowledge")):
            k = knowledge.parse_file(os.path.join(REF, know)) if know else dict(dimensionality=3, minLevel=2, maxLevel=4)
            with open(os.path.join(REF, prog)) as f:
                R = exa4.Exa4Program(f.read(), k, ops=OracleOps())
            assert R._coarse_cg_plan(R._resolve("mgCycle", R.min_level), R.min_level) is not None
        k = knowledge.parse_file(os.path.join(REF, "Testing/Smoothers/Jac.knowledge"))
        for d in "xyz":     # one block: the one-call form needs every face on the physical boundary
            k["domain_rect_numBlocks_" + d] = k["domain_rect_numFragsPerBlock_" + d] = 1
        with open(os.path.join(REF, "Testing/Smoothers/Jac.exa4")) as f:
            R = exa4.Exa4Program(f.read(), k, ops=OracleOps())
        plan = R._coarse_cg_plan(R._resolve("VCycle_0", R.min_level), R.min_level)
        assert plan is not None and plan[-1] == 3 and plan[6] == 512 and plan[7] == 0.001


def test_coarse_solver_limit_message_survives_the_one_call_form():
    """The statement after the CG loop (a print, reached when the loop runs out of iterations) is executed as often with the
    one-call coarse solve -- which counts the event in info[3] -- as by the statement-by-statement interpretation."""
    with open(os.path.join(EX, "poisson3d_rbgs.exa4")) as f:
        text = f.read().replace("repeat 128 times {", "repeat 2 times {")
    outs = []
    for fuse in (False, True):
        P = exa4.Exa4Program(text, dict(dimensionality=3, minLevel=2, maxLevel=4), ops=OracleOps(), fuse_coarse_solver=fuse)
        P.run()
        outs.append(sorted(P.out))          # the fused form prints the message at the end of the program, not where it happened
    assert outs[0] == outs[1]
    assert sum(1 for l in outs[0] if "iteration limit reached" in l) >= 3


# -- parser -----------------------------------------------------------------------------------------------------------
def _levels(text, lo=0, hi=6, cur=None):
    pr = exa4.Parser("Function F@%s { }" % text).parse()
    P = exa4.Exa4Program.__new__(exa4.Exa4Program)
    P.min_level, P.max_level = lo, hi
    return P.levels_of(pr.functions[0].levels, cur)


def test_level_specifications():
    assert _levels("all") == [0, 1, 2, 3, 4, 5, 6]
    assert _levels("finest") == [6]
    assert _levels("(coarsest and finest)") == [0, 6]
    assert _levels("(finest, coarsest)") == [0, 6]
    assert _levels("(all but finest)") == [0, 1, 2, 3, 4, 5]
    assert _levels("(all but (finest))") == [0, 1, 2, 3, 4, 5]
    assert _levels("((coarsest + 1) to finest)") == [1, 2, 3, 4, 5, 6]
    assert _levels("(1 to (finest - 1))") == [1, 2, 3, 4, 5]
    assert _levels("(coarsest to 0)") == [0]
    assert _levels("(all but ((finest - 1)))") == [0, 1, 2, 3, 4, 6]
    assert _levels("3") == [3]


def test_expression_precedence_and_slots():
    pr = exa4.Parser("Field u< global, L, None >[2]@all\nFunction F@all { Var a : Real = -1.0 / ( h ** 2 ) + 2 * 3 % 2\n"
                     "loop over u { u<next> = u<active>@current + 0.5 * u<0> } }").parse()
    decl, loop = pr.functions[0].body
    assert decl[2] == ("bin", "+", ("bin", "/", ("num", -1.0), ("bin", "**", ("id", "h", None), ("num", 2))),
                       ("bin", "%", ("bin", "*", ("num", 2), ("num", 3)), ("num", 2)))
    st = loop[5][0]
    assert st[2] == ("fld", "u", "next", None)
    assert st[3][2] == ("fld", "u", "active", ("single", "current", 0))
    assert st[3][3][3] == ("fld", "u", 0, None)


def test_colour_conditions():
    pr = exa4.Parser("Function F { loop over u where ( 1 == ( ( ( ( 64 + i0 ) + i1 ) + i2 ) % 2 ) ) { } }").parse()
    assert exa4._colour_cond(pr.functions[0].body[0][3], 3) == 1
    pr = exa4.Parser("Function F { loop over u where 0 == ((i0 + i1 + i2) % 2) { } }").parse()
    assert exa4._colour_cond(pr.functions[0].body[0][3], 3) == 0
    pr = exa4.Parser("Function F { loop over u where ( ( i0 > 0 ) && ( i1 > 0 ) ) { } }").parse()
    assert [exa4._lower_cond(c) for c in exa4._conjuncts(pr.functions[0].body[0][3])] == [0, 1]


HEADER = """
Domain global< [0.0, 0.0, 0.0] to [1.0, 1.0, 1.0] >
Layout L< Real, Node >@all { duplicateLayers = [1, 1, 1] with communication
 ghostLayers = [1, 1, 1] with communication }
Field u< global, L, %s >@all
Field f< global, L, None >@all
Stencil A@all { [0, 0, 0] => 6.0
 [1, 0, 0] => -1.0
 [-1, 0, 0] => -1.0
 [0, 1, 0] => -1.0
 [0, -1, 0] => -1.0
 [0, 0, 1] => -1.0
 [0, 0, -1] => -1.0 }
"""


@pytest.mark.parametrize("bc,body,what", [
    ("0.0", "loop over u@finest { u@finest += 0.8 / diag ( A@finest ) * ( f@finest - A@finest * u@finest ) }", "without colouring"),
    ("0.0", "loop over u@finest { u@finest = u@finest * f@finest }", "none of the recognised kernels"),
    ("0.0", "loop over u@finest { u@finest = atan2 ( vf_nodePosition_x, 2.0 ) }", "point expression"),
    ("0.0", "repeat 2 times with contraction [1, 1, 1] { apply bc to u@finest }", "contraction"),
])
def test_constructs_outside_the_subset_are_refused(bc, body, what):
    text = HEADER % bc + "Function Application { %s }" % body
    with pytest.raises(exa4.Exa4Unsupported, match=what):
        exa4.Exa4Program(text, dict(dimensionality=3, minLevel=0, maxLevel=2), ops=OracleOps()).run()


def test_arbitrary_point_expressions_become_device_programs():
    """Boundary values and fills that are none of the built-in functions are compiled to postfix programs (examg_expr_t)
    in tree order, user functions inlined."""
    import numpy as np

    text = (HEADER % "tan ( vf_boundaryCoord_x ) + vf_boundaryCoord_y * vf_boundaryCoord_z"
            + "Globals { Val c : Real = 3.0 }\n"
            + "Function g ( a : Real, b : Real ) : Real { return exp ( a ) - b ** 2 }\n"
            + "Function Application { loop over f@finest { f@finest = c * g ( vf_nodePosition_x, vf_nodePosition_z ) }\n"
            + " apply bc to u@finest }")
    P = exa4.Exa4Program(text, dict(dimensionality=3, minLevel=0, maxLevel=3), ops=OracleOps())
    P.run()
    u, f = P.fields[("u", 3)], P.fields[("f", 3)]
    assert not isinstance(u.bc_fn, int) and [o for o, _ in u.bc_fn.program] == ["x", "tan", "y", "z", "*", "+"]
    h = 1.0 / 8
    fv = f.data().numpy().reshape(f.layout.shape_zyx)
    x, z = 3 * h, 5 * h
    assert fv[5 + 1, 2 + 1, 3 + 1] == 3.0 * (np.exp(x) - z * z)          # array index = iterator + ghost
    uv = u.data().numpy().reshape(u.layout.shape_zyx)
    assert uv[4 + 1, 2 + 1, 0 + 1] == np.tan(0.0) + (2 * h) * (4 * h)    # x = 0 face
    assert uv[4 + 1, 2 + 1, 8 + 1] == np.tan(1.0) + (2 * h) * (4 * h)    # x = 1 face
    assert uv[4 + 1, 2 + 1, 4 + 1] == 0.0                                # interior untouched


def test_expression_programs_agree_with_the_built_in_functions(monkeypatch):
    """The red-black example with its boundary polynomial forced through the program path prints the same bits."""
    ref = example("poisson3d_rbgs.exa4", 2, 4)
    ref.run()

    def refuse(self, e, lvl):
        if e == ("num", 0.0):
            return 0, ()
        raise exa4.Exa4Unsupported("forced")

    monkeypatch.setattr(exa4.Exa4Program, "_recognise", refuse)
    P = example("poisson3d_rbgs.exa4", 2, 4)
    assert not isinstance(P.fields[("u", 4)].bc_fn, int)
    P.run()
    assert P.printed_values == ref.printed_values


def test_stencil_field_entries_as_expression_programs(monkeypatch):
    """The variable-coefficient example with its seven coefficient expressions filled plane by plane through the program
    path instead of examg_init_varcoeff7: same expression tree (bit-identical on the device, tests/test_gpu_exa4.py; the CPU
    stand-ins evaluate exp() with numpy in one path and glibc in the other, an ulp apart)."""
    ref = example("varcoeff3d.exa4", 0, 3)
    ref.run()

    def refuse(self, got, want, lvl):
        raise exa4.Exa4Unsupported("forced")

    monkeypatch.setattr(exa4.Exa4Program, "_varcoeff_function", refuse)
    P = example("varcoeff3d.exa4", 0, 3)
    P.run()
    assert len(P.printed_values) == len(ref.printed_values)
    for x, y in zip(P.printed_values, ref.printed_values):
        assert abs(x - y) <= 1e-10 * abs(y) + 1e-13 * ref.printed_values[0]
    assert P.launches > ref.launches


def test_loop_over_fragments_runs_its_body_once():
    text = HEADER % "0.0" + "Function Application { loop over fragments { loop over u@finest { u@finest = 2.5 } } }"
    P = exa4.Exa4Program(text, dict(dimensionality=3, minLevel=0, maxLevel=2), ops=OracleOps())
    P.run()
    f = P.fields[("u", 2)]
    v = f.data().numpy().reshape(f.layout.shape_zyx)
    assert v[2, 2, 2] == 2.5 and v[1, 2, 2] == 0.0 and P.launches == 1


def test_syntax_errors_carry_the_line():
    with pytest.raises(exa4.Exa4SyntaxError, match="line 2"):
        exa4.Parser("Function F {\n loop under u { } }").parse()


RANDOM_START = """
Domain global< [0.0, 0.0, 0.0] to [1.0, 1.0, 1.0] >
Layout Halo< Real, Node >@all {
  duplicateLayers = [1, 1, 1] with communication
  ghostLayers     = [1, 1, 1] with communication
}
Field u< global, Halo, 0.0 >@all
Function Application {
  loop over u@finest sequentially {
    u@finest = native ( "((double)std::rand()/RAND_MAX)" )
  }
  Var s : Real = 0.0
  loop over u@finest with reduction ( + : s ) {
    s += u@finest * u@finest
  }
  print ( sqrt ( s ) )
}
"""


def test_native_rand_fill_is_glibc_rand():
    """`native("((double)std::rand()/RAND_MAX)")` in a sequential loop: the first values of glibc's rand() after the default seed
    (1804289383, 846930886, 1681692777 -- the well-known start of the TYPE_3 generator), x fastest over the loop's box."""
    P = exa4.Exa4Program(RANDOM_START, dict(dimensionality=3, minLevel=2, maxLevel=3), ops=OracleOps())
    P.run()
    u = P.fields[("u", 3)]
    v = OracleOps().to_host(u.data()).reshape(u.layout.shape_zyx)
    first = v[2, 2, 2:5] * 2147483647.0          # iterator index 1 = array index 2 (one ghost layer, then the duplicate point)
    assert [int(round(x)) for x in first] == [1804289383, 846930886, 1681692777]
    assert v[1, :, :].max() == 0.0 and v[2, 1, :].max() == 0.0       # boundary planes untouched
    assert abs(P.printed_values[0] - float((v ** 2).sum()) ** 0.5) < 1e-12


COMPARE = """
Domain global< [0.0, 0.0, 0.0] to [1.0, 1.0, 1.0] >
Layout Halo< Real, Node >@all {
  duplicateLayers = [1, 1, 1] with communication
  ghostLayers     = [1, 1, 1] with communication
}
Field a< global, Halo, None >@all
Field b< global, Halo, None >@all
Function differ@finest ( ) : Int {
  loop over b sequentially {
    Var diff : Real = fabs ( b - a )
    if ( diff > 0.001 ) {
      print ( "fields differ: a =", a, " b =", b )
      print ( "at", i0, i1, i2 )
      return -1
    }
  }
  return 0
}
Function Application {
  loop over a@finest only dup [0, 0, 0] {
    a@finest = vf_nodePos_x + 2.0 * vf_nodePos_y + 4.0 * vf_nodePos_z
  }
  loop over b@finest only dup [0, 0, 0] {
    b@finest = vf_nodePos_x + 2.0 * vf_nodePos_y + 4.0 * vf_nodePos_z
  }
  print ( differ@finest ( ) )
  loop over b@finest only inner [0, 0, 0] {
    b@finest = vf_nodePos_x + 2.0 * vf_nodePos_y + 4.0 * vf_nodePos_z + 0.5
  }
  print ( differ@finest ( ) )
}
"""


def test_compare_loop_and_region_loop():
    """`loop over f only dup [0, 0, 0]` (the whole duplicate-to-duplicate extent, boundary points included: region bounds of
    IR_LoopOverPointsInOneFragment.scala:57-72) and the sequential compare loop of the reference's IOTest programs (`Var diff =
    fabs(b - a); if (diff > tol) { print(...); return -1 }`): silent while the fields agree, the first offending point in loop
    order otherwise."""
    P = exa4.Exa4Program(COMPARE, dict(dimensionality=3, minLevel=3, maxLevel=3), ops=OracleOps())
    out = P.run()
    assert out[0] == "0" and out[-1] == "-1", out
    assert out[1].startswith("fields differ: a = ") and out[2] == "at 1 1 1", out      # the first inner point in x-fastest order
    a = P.fields[("a", 3)]
    v = OracleOps().to_host(a.data()).reshape(a.layout.shape_zyx)
    assert v[1, 1, 1] == 0.0 and abs(v[9, 9, 9] - 7.0) < 1e-14 and v[0].max() == 0.0     # duplicates written, ghost layer not


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
@pytest.mark.parametrize("prog,know", [("Jac3Dcc.exa4", "3D_f0.knowledge"), ("Jac3Dcc.exa4", "3D_f5.knowledge"),
                                       ("Jac2Dccd.exa4", "2D_f0.knowledge"), ("RBGS2Dcc.exa4", "2D_f0.knowledge")])
def test_reference_contracting_loop_programs(prog, know):
    """Testing/PolyExpl (`repeat 5 times with contraction [1, 1, 1]` on five ghost layers, Jac3Dcc.exa4:27), as they
    are: explicit `innerPoints` (a 256^3 array on level 0), `native('std::srand(42)')` and two std::rand() draws per point in one
    sequential loop, timers named by identifiers, and the program's own check of every value (no zero, infinite or NaN product of the
    two slots) evaluated on the host.  Expected output: Testing/PolyExpl/all.results (no ERROR line)."""
    from oracle import mg

    if prog.startswith(("Jac2D", "RBGS2D")) and not os.environ.get("EXAMG_SLOW"):
        pytest.skip("2-D programs of 20 s each on the CPU: set EXAMG_SLOW=1")
    base = os.path.join(REF, "Testing", "PolyExpl")
    k = knowledge.parse_file(os.path.join(base, know))
    k["testing_enabled"] = True
    with open(os.path.join(base, prog)) as f:
        P = exa4.Exa4Program(f.read(), k, ops=OracleOps())
    out = P.run()
    with open(os.path.join(base, "all.results")) as f:
        assert mg.compare_with_golden(out, f.read()) == []
    S = P.fields[("Solution", 0)]
    v = OracleOps().to_host(S.data(S.active))
    assert np.isfinite(v).all() and 0.0 < float(np.abs(v).max()) < 10.0


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_rbgs3dvc_program_runs():
    """Testing/PolyExpl/RBGS3Dvc.exa4 as it is, on a 32^3 array (its 256^3 takes minutes on the CPU): a 7-entry stencil FIELD with two
    ghost layers, coloured `where` loops in place under `repeat 2 times with contraction`, and the smoother weight written as
    `0.8 / diag(Laplace)` -- evaluated per point as w / c_diag (EXAMG_WEIGHT_DIVIDE; Testing/SISC/3D_VarCoeff writes (1.0 / diag) * w,
    which rounds differently).  The program puts a 256^3 array on level 0 of a unit cube (grid width 1): its coefficient function
    exp(kappa (x - x^2)(y - y^2)(z - z^2)) underflows to 0 away from the first planes, so the sweep divides by zero by the program's
    own arithmetic and its CheckSolution loop reports that; what is pinned here: every statement is inside the subset, the lines of
    Testing/PolyExpl/all.results appear in order, and `fuse=False` prints the same."""
    base = os.path.join(REF, "Testing", "PolyExpl")
    k = knowledge.parse_file(os.path.join(base, "3D_f0.knowledge"))
    k["testing_enabled"] = True
    with open(os.path.join(base, "RBGS3Dvc.exa4")) as f:
        src = f.read().replace("innerPoints = [ 256, 256, 256 ]", "innerPoints = [ 32, 32, 32 ]")
    outs = []
    for fuse in (True, False):
        P = exa4.Exa4Program(src, k, ops=OracleOps(), fuse=fuse)
        outs.append(P.run())
    with open(os.path.join(base, "all.results")) as f:
        want = [l for l in f.read().splitlines() if l.strip()]
    assert [l for l in outs[0] if not l.startswith("ERROR")] == want
    assert outs[0] == outs[1]


LIVE_RESIDUAL = """  loop over f@coarser {
    f@coarser = R * r
  }
  Var chk : Real = 0.0
  loop over r with reduction ( + : chk ) {
    chk += r * r
  }
"""


def test_cross_statement_fusions_of_the_interpreter():
    """exastencils_amd/exa4_fusion.py: residual + restriction, residual + norm (never-stored residual, proven dead by the liveness
    scan), `u@coarser = 0` absorbed by the first sweep, correction folded into the first post-smoothing sweep -- reached from
    examples/exa4/poisson3d_rbgs.exa4, whose statements are spread over Cycle / Sweeps / Defect / Norm.  Same printed values bit for
    bit as one launch per statement; fewer launches.  With a statement that READS the residual after the restriction the residual
    is stored again (no residual + restriction fusion), and the values stay the same."""
    P = example("poisson3d_rbgs.exa4", 2, 6)
    P.fuse_min_row, P.fused_prolong_min_points, P.fuse_residual_norm = 16, 0, True
    P.run()
    Q = example("poisson3d_rbgs.exa4", 2, 6, fuse=False)
    Q.run()
    assert P.printed_values == Q.printed_values and len(P.printed_values) >= 5
    assert all(P.fusions[k] > 0 for k in ("residual_restrict", "residual_norm", "zero_start", "folded_correction")), P.fusions
    assert P.launches < Q.launches
    with open(os.path.join(EX, "poisson3d_rbgs.exa4")) as f:
        src = f.read()
    assert "  loop over f@coarser {\n    f@coarser = R * r\n  }\n" in src
    live = src.replace("  loop over f@coarser {\n    f@coarser = R * r\n  }\n", LIVE_RESIDUAL)
    L = exa4.Exa4Program(live, dict(dimensionality=3, minLevel=2, maxLevel=6), ops=OracleOps())
    L.fuse_min_row, L.fused_prolong_min_points, L.fuse_residual_norm = 16, 0, True
    L.run()
    assert L.fusions["residual_restrict"] == 0 and L.fusions["residual_norm"] > 0
    assert L.printed_values == Q.printed_values


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
@pytest.mark.parametrize("name", ["3D_Scalar", "2D_Scalar", "2D_LayoutTrafo"])
def test_reference_io_programs_pass_their_round_trips(name, tmp_path, monkeypatch):
    """Testing/IOTest/<name>_CheckEquality_ReadAfterWrite.{exa4,knowledge}, as they are: write, read back and compare through the
    lock (ascii), fpp (binary, one file per block), hdf5, MPI-I/O, netCDF and SIONlib interfaces -- the last four carried by raw
    binary files here (those libraries are not in this image: the round trip is the reference's, the file formats are not)."""
    monkeypatch.chdir(tmp_path)
    base = os.path.join(REF, "Testing", "IOTest", name + "_CheckEquality_ReadAfterWrite")
    k = knowledge.parse_file(base + ".knowledge")
    k["testing_enabled"] = True
    if not os.environ.get("EXAMG_SLOW"):
        k["minLevel"] = k["maxLevel"] = 4      # the knowledge files say level 6 (256 x 128 x 128 cells in 3-D: 30 s of ascii I/O)
    with open(base + ".exa4") as f:
        P = exa4.Exa4Program(f.read(), k, ops=OracleOps())
    out = [l for l in P.run() if l.startswith("Passed")]
    assert out == ["Passed lock test", "Passed fpp test", "Passed hdf5 test", "Passed MPI I/O test", "Passed nc test", "Passed sion test"]
    assert os.path.getsize(tmp_path / "data" / "src_lock.txt") > 1000


# -- two blocks over gloo ------------------------------------------------------------------------------------------------
def _worker(rank, world, port, out_dir):
    import json

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import mg

    from exastencils_amd.comm import Communicator
    from exastencils_amd.domain import RectDomain

    mg.lib().orc_set_num_threads(2)
    ops = OracleOps()
    dom = RectDomain(3, (2, 1, 1), rank, (1, 2, 2))
    P = example("poisson3d_rbgs.exa4", 1, 4, ops=ops, domain=dom, comm=Communicator(dom, ops))
    P.fuse_min_row_blocks = 8   # red-black sweeps of the two finest levels as fused interior + shell (smoothers.rbgs_sweep)
    P.run()
    json.dump({"values": P.printed_values, "messages": P.comm.stats["messages"]}, open(os.path.join(out_dir, "r%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_interpreter_on_two_blocks_matches_single_block(tmp_path):
    import json

    import torch.multiprocessing as mp

    from exastencils_amd.domain import RectDomain

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    single = example("poisson3d_rbgs.exa4", 1, 4, domain=RectDomain(3, (1, 1, 1), 0, (2, 2, 2)))
    single.run()
    for r in range(2):
        meta = json.load(open(tmp_path / ("r%d.json" % r)))
        assert meta["messages"] > 0
        assert len(meta["values"]) == len(single.printed_values)
        for x, y in zip(meta["values"], single.printed_values):
            assert abs(x - y) <= 1e-10 * abs(y) + 1e-13 * single.printed_values[0]


# -- the reference's own 2 x 2 x 2 decomposition, one process per block, against the reference's results file -------------------
def _worker_inlining(rank, world, port, out_dir):
    import json

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import mg

    from exastencils_amd.comm import Communicator

    mg.lib().orc_set_num_threads(1)
    ops = OracleOps()
    base = os.path.join(REF, "Testing", "Misc", "inlining")
    k = knowledge.parse_file(base + ".knowledge")
    k["testing_enabled"] = True
    dom = knowledge.domain_for_rank(k, rank)
    with open(base + ".exa4") as f:
        P = exa4.Exa4Program(f.read(), k, ops=ops, domain=dom, comm=Communicator(dom, ops))
    out = P.run()
    json.dump({"out": out, "messages": P.comm.stats["messages"]}, open(os.path.join(out_dir, "r%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_program_on_its_own_eight_blocks_reproduces_its_results_file(tmp_path):
    """Testing/Misc/inlining.{exa4,knowledge}: 2 x 2 x 2 blocks, `Solution = std::rand()/RAND_MAX` with std::srand(mpiRank) in
    every process -- the duplicate planes of neighbouring blocks start with DIFFERENT values, so the first residual of the results
    file pins which side `communicate` lets win (own upper plane -> the upper neighbour's lower plane, axis by axis).  Eight
    processes over gloo, one per block, print the reference's file; so does the one-process run that merges the blocks
    (test_more_reference_tests_reproduce_their_results_files)."""
    import json

    import torch.multiprocessing as mp
    from oracle import mg

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_inlining, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    with open(os.path.join(REF, "Testing", "Misc", "inlining.results")) as f:
        want = f.read()
    meta = json.load(open(tmp_path / "r0.json"))
    assert meta["messages"] > 0
    assert mg.compare_with_golden(meta["out"], want) == []


# -- periodic domain on two blocks: both neighbours along z are the other rank ---------------------------------------------
def _worker_periodic(rank, world, port, out_dir):
    import json

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import mg

    from exastencils_amd.comm import Communicator
    from exastencils_amd.domain import RectDomain

    mg.lib().orc_set_num_threads(2)
    ops = OracleOps()
    dom = RectDomain(3, (1, 1, 2), rank, (2, 2, 1), periodic=(False, False, True))
    P = example("jacobi3d_slots.exa4", 1, 4, ops=ops, domain=dom, comm=Communicator(dom, ops))
    P.run()
    json.dump({"values": P.printed_values, "messages": P.comm.stats["messages"]}, open(os.path.join(out_dir, "p%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_periodic_direction_on_two_blocks_matches_single_block(tmp_path):
    import json

    import torch.multiprocessing as mp

    from exastencils_amd.domain import RectDomain

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_periodic, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    single = example("jacobi3d_slots.exa4", 1, 4, domain=RectDomain(3, (1, 1, 1), 0, (2, 2, 2), periodic=(False, False, True)))
    single.run()
    plain = example("jacobi3d_slots.exa4", 1, 4, domain=RectDomain(3, (1, 1, 1), 0, (2, 2, 2)))
    plain.run()
    assert single.printed_values != plain.printed_values          # the periodic direction changes the problem
    for r in range(2):
        meta = json.load(open(tmp_path / ("p%d.json" % r)))
        assert meta["messages"] > 0
        assert len(meta["values"]) == len(single.printed_values)
        for x, y in zip(meta["values"], single.printed_values):
            assert abs(x - y) <= 1e-10 * abs(y) + 1e-13 * single.printed_values[0]


def _iotest(ops, tmp_path, level=3):
    import numpy as np

    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        with open(os.path.join(EX, "iotest3d.exa4")) as f:
            P = exa4.Exa4Program(f.read(), dict(dimensionality=3, minLevel=level, maxLevel=level), ops=ops)
        P.run()
    finally:
        os.chdir(cwd)
    n = (1 << level) + 1
    assert P.out[0].startswith("lock ") and P.out[1].startswith("fpp ")
    lock, fpp = P.printed_values
    assert fpp == 0.0                              # raw doubles: exact
    assert 0.0 <= lock < 1e-6 * n ** 1.5           # ascii: std::scientific with 6 digits after the point
    txt = (tmp_path / "data" / "src_lock.txt").read_text().splitlines()
    assert len(txt) == (n + 2) ** 3                # GLB..GRE with ghost layers
    first = txt[0].split(",")
    assert len(first) == 5 and first[-1] == ""     # x,y,z,value, -- every entry followed by the separator
    h = 1.0 / (n - 1)
    assert [float(t) for t in first[:3]] == [float("%g" % -h)] * 3     # the lower ghost corner, positions at %g precision
    assert (tmp_path / "data" / "src_fpp_0.bin").stat().st_size == 8 * (n + 2) ** 3
    vis = (tmp_path / "data" / "src_vis.csv").read_text().splitlines()
    assert len(vis) == n ** 3 and vis[0].endswith(" ")
    x, y, z, v = [float(t) for t in vis[n * n + n + 1].split()]          # the first inner node (h, h, h): h^2 - h^2 - h^2
    assert (x, y, z) == (h, h, h) and abs(v + h * h) <= 1e-6 * h * h
    assert float(vis[-1].split()[3]) == 0.0                               # `loop over src` leaves the boundary nodes untouched
    return P


def test_field_io_program_on_cpu_ops(tmp_path):
    _iotest(OracleOps(), tmp_path)


# -- `repeat n times with contraction` (Testing/PolyExpl/Jac3Dcc.exa4:27) ---------------------------------
def _plain_repeat_program():
    with open(os.path.join(EX, "jacobi3d_contraction.exa4")) as f:
        return f.read().replace("repeat 5 times with contraction [1, 1, 1] {", "repeat 5 times {")


def test_contracting_loop_single_block():
    """On one block every face is physical: the contracting loop is the plain repeat (IR_ContractingLoop widens bounds at
    interior faces only); run as two two-step passes + one step it gives the same bits."""
    fused = example("jacobi3d_contraction.exa4", 4, 4)
    fused.run()
    plain = example("jacobi3d_contraction.exa4", 4, 4, fuse=False)
    plain.run()
    ref = exa4.Exa4Program(_plain_repeat_program(), dict(dimensionality=3, minLevel=4, maxLevel=4), ops=OracleOps(), fuse=False)
    ref.run()
    assert fused.printed_values == plain.printed_values == ref.printed_values
    assert fused.launches < plain.launches
    with pytest.raises(exa4.Exa4Unsupported):       # 5 steps need 5 ghost layers at an interior face
        bad = exa4.Exa4Program(open(os.path.join(EX, "jacobi3d_contraction.exa4")).read().replace("ghostLayers = [5, 5, 5]", "ghostLayers = [3, 3, 3]"),
                               dict(dimensionality=3, minLevel=4, maxLevel=4), ops=OracleOps(),
                               domain=RectDomain(3, (1, 1, 2), 0, (2, 2, 1)), comm=_NoComm())
        bad.run()


class _NoComm:
    """Stands in for a communicator where only the bounds logic is under test."""
    dist = None
    stats = {"messages": 0}

    def exchange(self, *a, **k):
        pass

    def allreduce(self, t, op="sum"):
        return t

    def reduce_value(self, t, op="sum"):
        return float(t.item())

    def check(self):
        pass


def _worker_contract(rank, world, port, out_dir, fuse):
    import json

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import mg

    from exastencils_amd.comm import Communicator
    from exastencils_amd.domain import RectDomain

    mg.lib().orc_set_num_threads(2)
    ops = OracleOps()
    dom = RectDomain(3, (1, 2, 2), rank, (2, 1, 1))
    P = example("jacobi3d_contraction.exa4", 3, 3, ops=ops, domain=dom, comm=Communicator(dom, ops), fuse=fuse)
    P.run()
    json.dump({"values": P.printed_values, "messages": P.comm.stats["messages"], "launches": P.launches},
              open(os.path.join(out_dir, "c%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fuse", [False, True])
def test_contracting_loop_on_four_blocks_matches_single_block(tmp_path, fuse):
    """Four blocks (1 x 2 x 2), 5 ghost layers, ONE exchange per 5 Jacobi steps: the steps in between are computed on boxes
    4, 3, 2, 1, 0 layers wider at the interior faces; norms equal the single block's (which takes the plain repeat)."""
    import json
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_contract, args=(4, port, str(tmp_path), fuse), nprocs=4, join=True)
    single = exa4.Exa4Program(_plain_repeat_program(), dict(dimensionality=3, minLevel=3, maxLevel=3), ops=OracleOps(),
                              domain=RectDomain(3, (1, 1, 1), 0, (2, 2, 2)), fuse=False)
    single.run()
    for r in range(4):
        meta = json.load(open(tmp_path / ("c%d.json" % r)))
        assert meta["messages"] > 0
        for x, y in zip(meta["values"], single.printed_values):
            assert abs(x - y) <= 1e-12 * abs(y), (meta["values"], single.printed_values)
