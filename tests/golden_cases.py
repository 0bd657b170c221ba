"""Problem parameters of the reference's known-answer tests, transcribed from their
.exa4/.knowledge files (paths relative to /root/reference/Testing), and the expected
stdout (tests/golden/*.results = the reference's checked-in *.results data files).

`single` replaces the reference's blocks x fragments decomposition by one fragment with
fragLen = nFragsTotal * fragLen (same global grid; the goldens are decomposition-independent
at the printed 4 digits); `frags` is the reference's own decomposition.
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))

FN_ZERO, FN_POLY3D, FN_TRIG2D_SOL, FN_TRIG2D_RHS, FN_KAPPA_POLY, FN_KAPPA_RHS = 0, 1, 2, 3, 4, 5
FN_KAPPA_EXPSOL, FN_KAPPA_COEF, FN_TRIG3D_SOL, FN_SIN3 = 6, 7, 8, 9


def golden_text(name):
    with open(os.path.join(HERE, "golden", name + ".results")) as f:
        return f.read()


CASES = {
    # CommBasic/PureMPI.{exa4,knowledge}: 3x3x3 blocks x 1 fragment, levels 0..6 => 192^3
    "CommBasic_PureMPI": dict(
        program="B", nd=3, min_level=0, max_level=6, frags=(3, 3, 3), frag_len=(1, 1, 1), single_len=(3, 3, 3),
        smoother="jacobi", omega=0.8, stencil="unit", restrict_scale=4.0, tol=1e-5, cg_max=512, bc_fn=FN_POLY3D),
    # Smoothers/Jac.{exa4,knowledge}: 3x3x3 blocks x 3x3x3 fragments, levels 0..6 => 576^3
    "Smoothers_Jac": dict(
        program="B", nd=3, min_level=0, max_level=6, frags=(9, 9, 9), frag_len=(1, 1, 1), single_len=(9, 9, 9),
        smoother="jacobi", omega=0.8, stencil="unit", restrict_scale=4.0, tol=1e-5, cg_max=512, bc_fn=FN_POLY3D),
    # Smoothers/RBGS.exa4:125-133 (colour 0 first, omega 1.0), same knowledge as Jac
    "Smoothers_RBGS": dict(
        program="B", nd=3, min_level=0, max_level=6, frags=(9, 9, 9), frag_len=(1, 1, 1), single_len=(9, 9, 9),
        smoother="rbgs", omega=1.0, stencil="unit", restrict_scale=4.0, tol=1e-5, cg_max=512, bc_fn=FN_POLY3D),
    # Poisson/2D_FD_Poisson_fromL4.knowledge + Examples/Poisson/2D_FD_Poisson_fromL4.exa4: 2x2 blocks x 2x2 frags, 0..8 => 1024^2
    "Poisson_2D_FD_Poisson_fromL4": dict(
        program="A", nd=2, min_level=0, max_level=8, frags=(4, 4, 1), frag_len=(1, 1, 1), single_len=(4, 4, 1),
        omega=0.8, tol=1e-10, cg_max=128, bc_fn=FN_TRIG2D_SOL, rhs_fn=FN_TRIG2D_RHS, sol_fn=FN_TRIG2D_SOL),
    # SISC/3D_ConstCoeff.{exa4,knowledge}: 2x2x2 blocks x 2x2x2 frags x fragLen 2, levels 0..5 => 256^3
    "SISC_3D_ConstCoeff": dict(
        program="B", nd=3, min_level=0, max_level=5, frags=(4, 4, 4), frag_len=(2, 2, 2), single_len=(8, 8, 8),
        smoother="jacobi", omega=0.85, stencil="scaled", restrict_scale=1.0, tol=1e-5, cg_max=1024,
        bc_fn=FN_KAPPA_POLY, rhs_fn=FN_KAPPA_RHS, sol_fn=FN_KAPPA_POLY, kappa=10.0),
    "SISC_3D_VarCoeff": dict(
        program="B", nd=3, min_level=0, max_level=5, frags=(4, 4, 4), frag_len=(2, 2, 2), single_len=(8, 8, 8),
        smoother="jacobi", omega=0.85, stencil="varcoeff", restrict_scale=1.0, tol=1e-5, cg_max=1024,
        bc_fn=FN_KAPPA_EXPSOL, rhs_fn=FN_KAPPA_RHS, sol_fn=FN_KAPPA_EXPSOL, coef_fn=FN_KAPPA_COEF, kappa=10.0),
    # FMG/3D_Trigonometric.{exa4,knowledge}: 2x2x2 x 2x2x2 frags, levels 0..6 => 256^3
    "FMG_3D_Trigonometric": dict(
        program="B", nd=3, min_level=0, max_level=6, frags=(4, 4, 4), frag_len=(1, 1, 1), single_len=(4, 4, 4),
        smoother="jacobi", omega=0.8, stencil="scaled", restrict_scale=1.0, tol=1e-5, cg_max=512,
        bc_fn=FN_TRIG3D_SOL, sol_fn=FN_TRIG3D_SOL, fmg=True),
    # FMG/3D_VarCoeff.{exa4,knowledge} (knowledge identical to SISC/3D_VarCoeff)
    "FMG_3D_VarCoeff": dict(
        program="B", nd=3, min_level=0, max_level=5, frags=(4, 4, 4), frag_len=(2, 2, 2), single_len=(8, 8, 8),
        smoother="jacobi", omega=0.85, stencil="varcoeff", restrict_scale=1.0, tol=1e-5, cg_max=1024,
        bc_fn=FN_KAPPA_EXPSOL, rhs_fn=FN_KAPPA_RHS, sol_fn=FN_KAPPA_EXPSOL, coef_fn=FN_KAPPA_COEF, kappa=10.0, fmg=True),
    # Opts/base.exa4 + seq_*.knowledge (the generator-optimisation tests; results shared by Testing/LayoutTrafo/opts.exa4): the
    # program of Smoothers/Jac with homogeneous boundary values and Solution@finest = (double)std::rand()/RAND_MAX; one block, 64^3
    "Opts_seq": dict(
        program="B", nd=3, min_level=0, max_level=6, frags=(1, 1, 1), frag_len=(1, 1, 1), single_len=(1, 1, 1),
        smoother="jacobi", omega=0.8, stencil="unit", restrict_scale=4.0, tol=1e-5, cg_max=512, bc_fn=FN_ZERO, init_rand_procs=(1, 1, 1)),
    # Opts/base_par.exa4 + par_*.knowledge: 2x2x2 processes (std::srand(mpiRank) each), levels 0..8 => 512^3
    "Opts_par": dict(
        program="B", nd=3, min_level=0, max_level=8, frags=(2, 2, 2), frag_len=(1, 1, 1), single_len=(2, 2, 2),
        smoother="jacobi", omega=0.8, stencil="unit", restrict_scale=4.0, tol=1e-5, cg_max=512, bc_fn=FN_ZERO, init_rand_procs=(2, 2, 2)),
    # Misc/inlining.{exa4,knowledge}: the same cycle, 2x2x2 processes, levels 0..7 => 256^3 (its two finest levels smooth with
    # `repeat 3 times with contraction [1, 1, 1]` on three ghost layers: the same three Jacobi steps)
    "Misc_inlining": dict(
        program="B", nd=3, min_level=0, max_level=7, frags=(2, 2, 2), frag_len=(1, 1, 1), single_len=(2, 2, 2),
        smoother="jacobi", omega=0.8, stencil="unit", restrict_scale=4.0, tol=1e-5, cg_max=512, bc_fn=FN_ZERO, init_rand_procs=(2, 2, 2)),
}


def oracle_program(name, decomposed=False, **override):
    """Instantiate the oracle program for a golden case."""
    from oracle import mg

    c = dict(CASES[name])
    c.update(override)
    prog = c.pop("program")
    frags, flen, slen = c.pop("frags"), c.pop("frag_len"), c.pop("single_len")
    if decomposed:
        c["nfrag"], c["frag_len"] = frags, flen
    else:
        c["nfrag"], c["frag_len"] = (1, 1, 1), slen
    if prog == "A":
        return mg.ProgramA(mg.ConfigA(**c))
    return mg.ProgramB(mg.ConfigB(**c))
