"""The oracle must reproduce every convergence history the reference checks in for this path
 under the reference harness' own comparison rule (Testing/run_test.py:12-42)."""
import pytest

from golden_cases import CASES, golden_text, oracle_program
from oracle import mg


def _run(name, decomposed):
    P = oracle_program(name, decomposed)
    P.setup()
    P.Solve()
    bad = mg.compare_with_golden(P.log, golden_text(name))
    assert bad == [], (name, P.log, bad)
    return P


@pytest.mark.parametrize("name", ["CommBasic_PureMPI", "Poisson_2D_FD_Poisson_fromL4", "SISC_3D_ConstCoeff",
                                  "SISC_3D_VarCoeff", "FMG_3D_Trigonometric", "FMG_3D_VarCoeff"])
def test_golden_single_fragment(name):
    _run(name, decomposed=False)


@pytest.mark.parametrize("name", ["CommBasic_PureMPI", "Poisson_2D_FD_Poisson_fromL4"])
def test_golden_reference_decomposition(name):
    """Same goldens on the reference's own blocks x fragments decomposition: exercises the
    duplicate/ghost exchange ranges and the iteration offsets."""
    _run(name, decomposed=True)


@pytest.mark.parametrize("name,decomposed", [("Opts_seq", False), ("Misc_inlining", False), ("Misc_inlining", True)])
def test_golden_random_start(name, decomposed):
    """Testing/Opts/seq.results, Testing/Misc/inlining.results: the V(3,3) Jacobi cycle started from (double)std::rand()/RAND_MAX --
    the oracle calls the C library's rand() like the generated code does, after srand(rank) per process.  The merged run fills
    block by block; the decomposed one (one fragment per process of the 2 x 2 x 2 grid) lets `communicate` decide which of two
    neighbours' different values a shared duplicate plane keeps: the first residual pins that direction."""
    _run(name, decomposed)


@pytest.mark.slow
def test_golden_random_start_512():
    _run("Opts_par", decomposed=False)


def test_golden_rbgs_576():
    """Testing/Smoothers/RBGS.results pins the red-black colour convention (colour 0 first)."""
    _run("Smoothers_RBGS", decomposed=False)


@pytest.mark.slow
def test_golden_jac_576():
    _run("Smoothers_Jac", decomposed=False)


def test_decomposition_independence_bits():
    """Jacobi V-cycles on 1 fragment and on 2x2x2 fragments agree to rounding (the point-wise loops
    are bit-identical; only the order of the reduction partials differs)."""
    kw = dict(max_level=4)
    a = oracle_program("CommBasic_PureMPI", False, single_len=(2, 2, 2), **kw)
    b = oracle_program("CommBasic_PureMPI", True, frags=(2, 2, 2), **kw)
    for P in (a, b):
        P.setup()
        P.Solve()
    assert a.iterations == b.iterations
    for x, y in zip(a.res_history, b.res_history):
        assert abs(x - y) <= 1e-12 * abs(x)
