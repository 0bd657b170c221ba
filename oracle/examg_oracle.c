/*
 * examg_oracle.c -- CPU restatement of the ExaStencils multigrid hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  It exists to (1) check the HIP kernels bit for bit and
 * (2) be timed as the "restated reference CPU path" (SURVEY.md 8c/8d).
 *
 * Parity pin: the V-cycle programs built from these loops (oracle/mg.py)
 * reproduce the reference's own checked-in convergence histories
 * (Testing/CommBasic/PureMPI.results, Testing/Smoothers/{Jac,RBGS}.results,
 * Testing/Poisson/2D_FD_Poisson_fromL4.results, Testing/SISC/3D_*.results,
 * Testing/FMG/3D_*.results; see tests/golden and tests/test_oracle_golden.py).
 *
 * The loop shape is the generator's: z-y-x nest, x fastest, linearised index
 * into the reference field layout, `#pragma omp parallel for schedule(static)`
 * on the outermost loop (Compiler/src/exastencils/baseExt/ir/
 * IR_LoopOverDimensions.scala:206-255, parallelization/api/omp/OMP_Loop.scala:98-152).
 * Built with -ffp-contract=off so every statement rounds exactly as written;
 * the HIP kernels are built the same way, which is what makes bit-exact
 * comparison of the point-wise kernels possible.
 *
 * All citations are relative to /root/reference/; C/ = Compiler/src/exastencils/.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Loop nests over (i2, i1): by default both loops are shared among the threads (collapse(2): 2-D fields and small coarse levels
 * keep all cores busy in the tests).  The generator prints the pragma on the OUTER loop only -- `omp_useCollapse` defaults to
 * false (config/Knowledge.scala:481, parallelization/api/omp/OMP_Loop.scala:47,103-110) -- and that shape is what bench.py times
 * as the CPU baseline: libexamg_oracle_gen.so is this file compiled with -DORC_OUTER_ONLY (same arithmetic, same results). */
#define ORC_PRAGMA(x) _Pragma(#x)
#ifdef ORC_OUTER_ONLY
#define ORC_PARFOR ORC_PRAGMA(omp parallel for schedule(static))
#define ORC_PARFOR_RED(r) ORC_PRAGMA(omp parallel for schedule(static) reduction(r))
#else
#define ORC_PARFOR ORC_PRAGMA(omp parallel for schedule(static) collapse(2))
#define ORC_PARFOR_RED(r) ORC_PRAGMA(omp parallel for schedule(static) collapse(2) reduction(r))
#endif

#define ORC_MAX_ENTRIES 27

/* Per-dimension regions  pad | ghost | dup | inner | dup | ghost | pad
 * (C/field/ir/IR_FieldLayout.scala:30-129).  Unused dims: inner = 1, rest 0. */
typedef struct {
  int32_t nd;
  int32_t pad_l[3], ghost_l[3], dup_l[3], inner[3], dup_r[3], ghost_r[3], pad_r[3];
} orc_layout_t;

/* Stencil: constant coefficients (C/operator/ir/IR_Stencil.scala:34-211) or a
 * stencil field whose entry index is the slowest array dimension
 * (C/stencil/ir/IR_StencilConvolution.scala:73-95). */
typedef struct {
  int32_t nent;
  int32_t diag;                       /* index of the (0,0,0) entry */
  int32_t off[ORC_MAX_ENTRIES][3];
  double coef[ORC_MAX_ENTRIES];
  const double *cfield;               /* NULL => constant coefficients */
  orc_layout_t clayout;
  int32_t wform;                      /* stencil field, smoother weight as written: 0 ((1.0 / diag) * omega), 1 (omega / diag)
                                         (Testing/SISC/3D_VarCoeff.exa4:145; Testing/PolyExpl/RBGS3Dvc.exa4:52) */
} orc_stencil_t;

static inline int lay_tot(const orc_layout_t *l, int d) {
  return l->pad_l[d] + l->ghost_l[d] + l->dup_l[d] + l->inner[d] + l->dup_r[d] + l->ghost_r[d] + l->pad_r[d];
}
/* referenceOffset = pad_l + ghost_l (C/fieldlike/ir/IR_FieldLikeLayout.scala:45-50):
 * iterator index 0 is the lower duplicate node. */
static inline int lay_ref(const orc_layout_t *l, int d) { return l->pad_l[d] + l->ghost_l[d]; }
/* x-fastest linearisation (C/baseExt/ir/IR_Linearization.scala:27-36). */
static inline ptrdiff_t lay_idx(const orc_layout_t *l, int i0, int i1, int i2) {
  return (ptrdiff_t)(i0 + lay_ref(l, 0)) +
         (ptrdiff_t)lay_tot(l, 0) * ((ptrdiff_t)(i1 + lay_ref(l, 1)) + (ptrdiff_t)lay_tot(l, 1) * (ptrdiff_t)(i2 + lay_ref(l, 2)));
}
static inline ptrdiff_t lay_size(const orc_layout_t *l) {
  return (ptrdiff_t)lay_tot(l, 0) * lay_tot(l, 1) * lay_tot(l, 2);
}

int orc_layout_tot(const orc_layout_t *l, int d) { return lay_tot(l, d); }
int orc_layout_ref(const orc_layout_t *l, int d) { return lay_ref(l, d); }
long orc_layout_size(const orc_layout_t *l) { return (long)lay_size(l); }

/* ------------------------------------------------------------------ */
/* Stencil sweeps: A*u, residual, Jacobi / coloured Gauss-Seidel       */
/* ------------------------------------------------------------------ */

enum { ORC_APPLY = 0, ORC_RESIDUAL = 1, ORC_SMOOTH = 2 };

/* One x-row of   sum_k c_k * u[i + o_k]   folded left to right in entry order
 * (IR_StencilConvolution.expand: entries.map(resolveEntry).reduceLeft(_ + _),
 * C/stencil/ir/IR_StencilConvolution.scala:65-68). */
#define ROW_BODY(NENT)                                                                                   \
  for (int i0 = b0; i0 < e0; ++i0) {                                                                     \
    if (colour >= 0 && ((i0 + i1 + i2) & 1) != colour) continue;                                          \
    const ptrdiff_t iu = ubase + i0;                                                                     \
    double acc;                                                                                          \
    if (cf) {                                                                                            \
      const ptrdiff_t ic = cbase + i0;                                                                   \
      acc = cf[ic] * u[iu + uo[0]];                                                                      \
      for (int k = 1; k < (NENT); ++k) acc = acc + cf[ic + (ptrdiff_t)k * cplane] * u[iu + uo[k]];       \
    } else {                                                                                             \
      acc = st->coef[0] * u[iu + uo[0]];                                                                 \
      for (int k = 1; k < (NENT); ++k) acc = acc + st->coef[k] * u[iu + uo[k]];                          \
    }                                                                                                    \
    double out;                                                                                          \
    if (mode == ORC_APPLY) out = acc;                                                                    \
    else if (mode == ORC_RESIDUAL) out = rhs[fbase + i0] - acc;                                          \
    else {                                                                                               \
      /* const: w is the folded constant omega/diag; stencil field:                                      \
       * ((1.0 / diag) * omega) as written in Testing/SISC/3D_VarCoeff.exa4 Smoother */                 \
      const double dg_ = cf ? cf[cbase + i0 + (ptrdiff_t)st->diag * cplane] : 1.0;                       \
      const double ww = cf ? (st->wform ? (w / dg_) : ((1.0 / dg_) * w)) : w;                            \
      out = u[iu] + ww * (rhs[fbase + i0] - acc);                                                        \
    }                                                                                                    \
    dst[dbase + i0] = out;                                                                               \
  }

void orc_stencil_op(int mode, const orc_layout_t *lu, const double *u, const orc_layout_t *lf, const double *rhs,
                    const orc_layout_t *ld, double *dst, const orc_stencil_t *st, double w, int colour,
                    const int *begin, const int *end) {
  const int nent = st->nent;
  ptrdiff_t uo[ORC_MAX_ENTRIES];
  for (int k = 0; k < nent; ++k)
    uo[k] = st->off[k][0] + (ptrdiff_t)lay_tot(lu, 0) * (st->off[k][1] + (ptrdiff_t)lay_tot(lu, 1) * st->off[k][2]);
  const double *cf = st->cfield;
  const ptrdiff_t cplane = cf ? lay_size(&st->clayout) : 0;
  const int b0 = begin[0], e0 = end[0];
ORC_PARFOR
  for (int i2 = begin[2]; i2 < end[2]; ++i2)
    for (int i1 = begin[1]; i1 < end[1]; ++i1) {
      const ptrdiff_t ubase = lay_idx(lu, 0, i1, i2);
      const ptrdiff_t fbase = rhs ? lay_idx(lf, 0, i1, i2) : 0;
      const ptrdiff_t dbase = lay_idx(ld, 0, i1, i2);
      const ptrdiff_t cbase = cf ? lay_idx(&st->clayout, 0, i1, i2) : 0;
      switch (nent) {
        case 5: ROW_BODY(5) break;
        case 7: ROW_BODY(7) break;
        case 9: ROW_BODY(9) break;
        case 27: ROW_BODY(27) break;
        default: ROW_BODY(nent) break;
      }
    }
}

/* Generator-shaped 3-D 7-point constant-coefficient Jacobi sweep with the
 * coefficients folded to literals per entry, outer loop only parallelised --
 * this is the loop the reference's OpenMP backend prints for
 *   Solution<next> = Solution<active> + (omega/diag) * (RHS - Laplace * Solution<active>)
 * (Testing/Smoothers/Jac.exa4:125-131) and is what bench.py times as
 * cpu_baseline.  Bit-identical to orc_stencil_op(ORC_SMOOTH) for nent = 7. */
void orc_jacobi7_const(const orc_layout_t *lu, const double *restrict u, const orc_layout_t *lf,
                       const double *restrict rhs, double *restrict un, const orc_stencil_t *st, double w,
                       const int *begin, const int *end) {
  ptrdiff_t o[7];
  double c[7];
  for (int k = 0; k < 7; ++k) {
    o[k] = st->off[k][0] + (ptrdiff_t)lay_tot(lu, 0) * (st->off[k][1] + (ptrdiff_t)lay_tot(lu, 1) * st->off[k][2]);
    c[k] = st->coef[k];
  }
  const double c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5], c6 = c[6];
  const ptrdiff_t o0 = o[0], o1 = o[1], o2 = o[2], o3 = o[3], o4 = o[4], o5 = o[5], o6 = o[6];
#pragma omp parallel for schedule(static)
  for (int i2 = begin[2]; i2 < end[2]; ++i2)
    for (int i1 = begin[1]; i1 < end[1]; ++i1) {
      const double *restrict up = u + lay_idx(lu, 0, i1, i2);
      const double *restrict fp = rhs + lay_idx(lf, 0, i1, i2);
      double *restrict dp = un + lay_idx(lu, 0, i1, i2);
      for (int i0 = begin[0]; i0 < end[0]; ++i0) {
        const double au = (((((c0 * up[i0 + o0] + c1 * up[i0 + o1]) + c2 * up[i0 + o2]) + c3 * up[i0 + o3]) +
                            c4 * up[i0 + o4]) + c5 * up[i0 + o5]) + c6 * up[i0 + o6];
        dp[i0] = up[i0] + w * (fp[i0] - au);
      }
    }
}

/* ------------------------------------------------------------------ */
/* Inter-grid transfer                                                 */
/* ------------------------------------------------------------------ */

/* RHS@coarser = scale * R * Residual, R = kron of [1/4, 1/2, 1/4] per dim,
 * coarse node I <- fine nodes 2I + o  (C/operator/l4/L4_DefaultRestriction.scala:29-36,63-88;
 * explicit table Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:49-77, entry
 * order: dim-0 offset outermost, -1,0,+1).  `scale` is the "4.0 *" of
 * Testing/Smoothers/Jac.exa4:151 (exact: power of two). */
void orc_restrict(const orc_layout_t *lfine, const double *rf, const orc_layout_t *lc, double *fc, double scale,
                  const int *begin, const int *end) {
  const int nd = lfine->nd;
  static const double w1[3] = {0.25, 0.5, 0.25};
ORC_PARFOR
  for (int I2 = begin[2]; I2 < end[2]; ++I2)
    for (int I1 = begin[1]; I1 < end[1]; ++I1)
      for (int I0 = begin[0]; I0 < end[0]; ++I0) {
        double acc = 0.0;
        int first = 1;
        for (int a = -1; a <= 1; ++a)
          for (int b = -1; b <= 1; ++b) {
            if (nd == 2) {
              const double wgt = scale * (w1[a + 1] * w1[b + 1]);
              const double t = wgt * rf[lay_idx(lfine, 2 * I0 + a, 2 * I1 + b, 0)];
              acc = first ? t : acc + t;
              first = 0;
            } else {
              for (int c = -1; c <= 1; ++c) {
                const double wgt = scale * ((w1[a + 1] * w1[b + 1]) * w1[c + 1]);
                const double t = wgt * rf[lay_idx(lfine, 2 * I0 + a, 2 * I1 + b, 2 * I2 + c)];
                acc = first ? t : acc + t;
                first = 0;
              }
            }
          }
        fc[lay_idx(lc, I0, I1, I2)] = acc;
      }
}

/* Solution += P@coarser * Solution@coarser, P = 2^d * R^T: per dim an even
 * fine index takes coarse i/2 with weight 1, an odd one takes (i+1)/2 and
 * (i-1)/2 with weight 1/2 each -- the 2^d parity cases of
 * C/stencil/ir/IR_FindStencilConvolutions.scala:135-156; entry order of the
 * table Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:79-107 (dim-0 outermost,
 * +1 before -1). */
void orc_prolong_add(const orc_layout_t *lc, const double *uc, const orc_layout_t *lfine, double *uf,
                     const int *begin, const int *end) {
  const int nd = lfine->nd;
ORC_PARFOR
  for (int i2 = begin[2]; i2 < end[2]; ++i2)
    for (int i1 = begin[1]; i1 < end[1]; ++i1)
      for (int i0 = begin[0]; i0 < end[0]; ++i0) {
        int n[3], ci[3][2];
        double cw[3][2];
        const int ii[3] = {i0, i1, i2};
        for (int d = 0; d < 3; ++d) {
          if (d >= nd) { n[d] = 1; ci[d][0] = 0; cw[d][0] = 1.0; continue; }
          if ((ii[d] & 1) == 0) { n[d] = 1; ci[d][0] = ii[d] / 2; cw[d][0] = 1.0; }
          else { n[d] = 2; ci[d][0] = (ii[d] + 1) / 2; ci[d][1] = (ii[d] - 1) / 2; cw[d][0] = 0.5; cw[d][1] = 0.5; }
        }
        double acc = 0.0;
        int first = 1;
        for (int a = 0; a < n[0]; ++a)
          for (int b = 0; b < n[1]; ++b)
            for (int c = 0; c < n[2]; ++c) {
              const double t = ((cw[0][a] * cw[1][b]) * cw[2][c]) * uc[lay_idx(lc, ci[0][a], ci[1][b], ci[2][c])];
              acc = first ? t : acc + t;
              first = 0;
            }
        const ptrdiff_t k = lay_idx(lfine, i0, i1, i2);
        uf[k] = uf[k] + acc;
      }
}

/* ------------------------------------------------------------------ */
/* BLAS-1 style loops of the coarse-grid CG and the cycle driver        */
/* (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:152-201, 226-229)     */
/* ------------------------------------------------------------------ */

void orc_set(const orc_layout_t *l, double *x, double v, const int *begin, const int *end) {
ORC_PARFOR
  for (int i2 = begin[2]; i2 < end[2]; ++i2)
    for (int i1 = begin[1]; i1 < end[1]; ++i1)
      for (int i0 = begin[0]; i0 < end[0]; ++i0) x[lay_idx(l, i0, i1, i2)] = v;
}

/* y = a*x + b*y over the box; the statements of the reference are the
 * special cases  copy (a=1,b=0), `y += alpha*x` (b=1), `y -= alpha*x`
 * (a=-alpha,b=1), `p = r + beta*p` (a=1).  Evaluated as written there:
 *   b == 1:  y + a*x        a == 1:  x + b*y        b == 0:  a*x */
void orc_axpby(const orc_layout_t *lx, const double *x, const orc_layout_t *ly, double *y, double a, double b,
               const int *begin, const int *end) {
ORC_PARFOR
  for (int i2 = begin[2]; i2 < end[2]; ++i2)
    for (int i1 = begin[1]; i1 < end[1]; ++i1)
      for (int i0 = begin[0]; i0 < end[0]; ++i0) {
        const double xv = x[lay_idx(lx, i0, i1, i2)];
        const ptrdiff_t k = lay_idx(ly, i0, i1, i2);
        double r;
        if (b == 0.0) r = (a == 1.0) ? xv : a * xv;
        else if (b == 1.0) r = y[k] + a * xv;
        else if (a == 1.0) r = xv + b * y[k];
        else r = a * xv + b * y[k];
        y[k] = r;
      }
}

/* sum x*y over the box.  Deterministic: one sequential partial per outermost
 * index, partials added in index order (the reference's OpenMP reduction clause,
 * omp/OMP_Loop.scala:126-127, leaves the order unspecified). */
double orc_dot(const orc_layout_t *lx, const double *x, const orc_layout_t *ly, const double *y, const int *begin,
               const int *end) {
  const int n2 = end[2] - begin[2];
  if (n2 <= 0 || end[1] <= begin[1] || end[0] <= begin[0]) return 0.0;
  const int n1 = end[1] - begin[1];
  double *part = (double *)calloc((size_t)n2 * n1, sizeof(double));
ORC_PARFOR
  for (int i2 = begin[2]; i2 < end[2]; ++i2)
    for (int i1 = begin[1]; i1 < end[1]; ++i1) {
      double s = 0.0;
      for (int i0 = begin[0]; i0 < end[0]; ++i0) s = s + x[lay_idx(lx, i0, i1, i2)] * y[lay_idx(ly, i0, i1, i2)];
      part[(size_t)(i2 - begin[2]) * n1 + (i1 - begin[1])] = s;
    }
  double tot = 0.0;
  for (int j = 0; j < n2; ++j) {
    double sp = 0.0;
    for (int i = 0; i < n1; ++i) sp = sp + part[(size_t)j * n1 + i];
    tot = tot + sp;
  }
  free(part);
  return tot;
}

/* ------------------------------------------------------------------ */
/* Analytic functions: boundary values, right-hand sides, exact         */
/* solutions and coefficient profiles of the reference programs         */
/* ------------------------------------------------------------------ */

enum {
  ORC_FN_ZERO = 0,
  ORC_FN_POLY3D = 1,      /* x^2 - y^2/2 - z^2/2      Benchmark/Poisson3D/...exa4:25, Testing/Smoothers/Jac.exa4:43 */
  ORC_FN_TRIG2D_SOL = 2,  /* cos(pi x) - sin(2 pi y)   Examples/Poisson/2D_FD_Poisson_fromL4.exa4:26 */
  ORC_FN_TRIG2D_RHS = 3,  /* pi^2 cos(pi x) - 4 pi^2 sin(2 pi y)   ...exa4:233 */
  ORC_FN_KAPPA_POLY = 4,  /* kappa (x-x^2)(y-y^2)(z-z^2)           Testing/SISC/3D_ConstCoeff.exa4:43 */
  ORC_FN_KAPPA_RHS = 5,   /* 2 kappa ((x-x^2)(y-y^2) + (x-x^2)(z-z^2) + (y-y^2)(z-z^2))  Testing/SISC/3D_VarCoeff.exa4 InitRHS */
  ORC_FN_KAPPA_EXPSOL = 6,/* 1 - exp(-kappa (x-x^2)(y-y^2)(z-z^2)) Testing/SISC/3D_VarCoeff.exa4:48 */
  ORC_FN_KAPPA_COEF = 7,  /* exp(kappa (x-x^2)(y-y^2)(z-z^2))      Testing/SISC/3D_VarCoeff.exa4 getCoefficient */
  ORC_FN_TRIG3D_SOL = 8,  /* sin(pi x) sin(pi y) sinh(sqrt(2) pi z) Testing/FMG/3D_Trigonometric.exa4:43 */
  ORC_FN_SIN3 = 9,        /* sin(pi x) sin(pi y) sin(pi z)  (config-4 manufactured solution) */
  ORC_FN_KAPPA_POLY2D = 10,/* kappa (x-x^2)(y-y^2) */
  ORC_FN_KAPPA_RHS2D = 11, /* 2 kappa ((x-x^2) + (y-y^2)) */
  ORC_FN_KAPPA_EXPSOL2D = 12,
  ORC_FN_KAPPA_COEF2D = 13,
  ORC_FN_POLY2D = 14,
  ORC_FN_SINSINH2D = 15,
  ORC_FN_XSQ = 16
};

double orc_eval_fn(int fn, const double *p, double x, double y, double z) {
  const double PI = 3.14159265358979323846;
  switch (fn) {
    case ORC_FN_ZERO: return 0.0;
    case ORC_FN_POLY3D: return ((x * x) - ((0.5 * y) * y)) - ((0.5 * z) * z);
    case ORC_FN_TRIG2D_SOL: return cos(PI * x) - sin((2.0 * PI) * y);
    case ORC_FN_TRIG2D_RHS: return (PI * PI) * cos(PI * x) - ((4.0 * (PI * PI)) * sin((2.0 * PI) * y));
    case ORC_FN_KAPPA_POLY: return p[0] * (((x - (x * x)) * (y - (y * y))) * (z - (z * z)));
    case ORC_FN_KAPPA_RHS:
      return (2.0 * p[0]) * ((((x - (x * x)) * (y - (y * y))) + ((x - (x * x)) * (z - (z * z)))) + ((y - (y * y)) * (z - (z * z))));
    case ORC_FN_KAPPA_EXPSOL: return 1.0 - exp((-1.0 * p[0]) * (((x - (x * x)) * (y - (y * y))) * (z - (z * z))));
    case ORC_FN_KAPPA_COEF: return exp(p[0] * (((x - (x * x)) * (y - (y * y))) * (z - (z * z))));
    case ORC_FN_TRIG3D_SOL: return (sin(PI * x) * sin(PI * y)) * sinh((sqrt(2.0) * PI) * z);
    case ORC_FN_SIN3: return (sin(PI * x) * sin(PI * y)) * sin(PI * z);
    case ORC_FN_KAPPA_POLY2D: return p[0] * ((x - (x * x)) * (y - (y * y)));
    case ORC_FN_KAPPA_RHS2D: return (2.0 * p[0]) * ((x - (x * x)) + (y - (y * y)));
    case ORC_FN_KAPPA_EXPSOL2D: return 1.0 - exp((-1.0 * p[0]) * ((x - (x * x)) * (y - (y * y))));
    case ORC_FN_KAPPA_COEF2D: return exp(p[0] * ((x - (x * x)) * (y - (y * y))));
    case ORC_FN_POLY2D: return (x * x) - (y * y);              /* Testing/BC/2D_Polynomial.exa4:43 */
    case ORC_FN_SINSINH2D: return sin(PI * x) * sinh(PI * y);  /* Testing/BC/2D_Trigonometric.exa4:43 */
    case ORC_FN_XSQ: return x * x;                             /* Testing/BC/2D_Periodic.exa4:43 */
    default: return NAN;
  }
}

/* Geometry of one fragment at one level: node position = index * h + fragment
 * begin (uniform grid, C/grid/ir/IR_VF_NodePosition.scala:109-111). */
typedef struct {
  double pos_begin[3];
  double h[3];
} orc_geom_t;

/* x[box] = fn(node position): Dirichlet faces (C/boundary/ir/IR_DirichletBC.scala:37-40
 * over the index ranges of IR_ApplyBCFunction.scala:53-83), InitRHS, SetFuncDir. */
void orc_fill_fn(const orc_layout_t *l, double *x, const orc_geom_t *g, int fn, const double *p, const int *begin,
                 const int *end) {
ORC_PARFOR
  for (int i2 = begin[2]; i2 < end[2]; ++i2)
    for (int i1 = begin[1]; i1 < end[1]; ++i1)
      for (int i0 = begin[0]; i0 < end[0]; ++i0) {
        const double px = i0 * g->h[0] + g->pos_begin[0];
        const double py = i1 * g->h[1] + g->pos_begin[1];
        const double pz = i2 * g->h[2] + g->pos_begin[2];
        x[lay_idx(l, i0, i1, i2)] = orc_eval_fn(fn, p, px, py, pz);
      }
}

/* max |x - fn(pos)| over the box (PrintError / NormError_0). */
double orc_max_err_fn(const orc_layout_t *l, const double *x, const orc_geom_t *g, int fn, const double *p,
                      const int *begin, const int *end) {
  double m = 0.0;
ORC_PARFOR_RED(max : m)
  for (int i2 = begin[2]; i2 < end[2]; ++i2)
    for (int i1 = begin[1]; i1 < end[1]; ++i1)
      for (int i0 = begin[0]; i0 < end[0]; ++i0) {
        const double px = i0 * g->h[0] + g->pos_begin[0];
        const double py = i1 * g->h[1] + g->pos_begin[1];
        const double pz = i2 * g->h[2] + g->pos_begin[2];
        const double e = fabs(x[lay_idx(l, i0, i1, i2)] - orc_eval_fn(fn, p, px, py, pz));
        if (e > m) m = e;
      }
  return m;
}

/* Stencil-field initialisation of Testing/SISC/3D_VarCoeff.exa4 InitLaplace:
 * entry order [0,0,0],[1,0,0],[-1,0,0],[0,1,0],[0,-1,0],[0,0,1],[0,0,-1];
 * a = coefficient profile `coef_fn` sampled at the half-way points. */
void orc_init_varcoeff7(const orc_layout_t *lc, double *cf, const orc_geom_t *g, int coef_fn, const double *p,
                        const int *begin, const int *end) {
  const int nd = lc->nd;
  const ptrdiff_t plane = lay_size(lc);
ORC_PARFOR
  for (int i2 = begin[2]; i2 < end[2]; ++i2)
    for (int i1 = begin[1]; i1 < end[1]; ++i1)
      for (int i0 = begin[0]; i0 < end[0]; ++i0) {
        const double x = i0 * g->h[0] + g->pos_begin[0];
        const double y = i1 * g->h[1] + g->pos_begin[1];
        const double z = i2 * g->h[2] + g->pos_begin[2];
        const double hx = g->h[0], hy = g->h[1], hz = g->h[2];
        const double axp = orc_eval_fn(coef_fn, p, x + (0.5 * hx), y, z), axm = orc_eval_fn(coef_fn, p, x - (0.5 * hx), y, z);
        const double ayp = orc_eval_fn(coef_fn, p, x, y + (0.5 * hy), z), aym = orc_eval_fn(coef_fn, p, x, y - (0.5 * hy), z);
        const ptrdiff_t k = lay_idx(lc, i0, i1, i2);
        if (nd == 3) {
          const double azp = orc_eval_fn(coef_fn, p, x, y, z + (0.5 * hz)), azm = orc_eval_fn(coef_fn, p, x, y, z - (0.5 * hz));
          cf[k + 0 * plane] = (((axp + axm) / (hx * hx)) + ((ayp + aym) / (hy * hy))) + ((azp + azm) / (hz * hz));
          cf[k + 1 * plane] = (-1.0 * axp) / (hx * hx);
          cf[k + 2 * plane] = (-1.0 * axm) / (hx * hx);
          cf[k + 3 * plane] = (-1.0 * ayp) / (hy * hy);
          cf[k + 4 * plane] = (-1.0 * aym) / (hy * hy);
          cf[k + 5 * plane] = (-1.0 * azp) / (hz * hz);
          cf[k + 6 * plane] = (-1.0 * azm) / (hz * hz);
        } else {
          cf[k + 0 * plane] = ((axp + axm) / (hx * hx)) + ((ayp + aym) / (hy * hy));
          cf[k + 1 * plane] = (-1.0 * axp) / (hx * hx);
          cf[k + 2 * plane] = (-1.0 * axm) / (hx * hx);
          cf[k + 3 * plane] = (-1.0 * ayp) / (hy * hy);
          cf[k + 4 * plane] = (-1.0 * aym) / (hy * hy);
        }
      }
}

/* ------------------------------------------------------------------ */
/* Halo exchange pack / unpack (C/communication/ir/IR_NoInterpPacking.scala:53-83):
 * copy the box, x fastest, to / from a contiguous buffer.              */
/* ------------------------------------------------------------------ */

void orc_pack(const orc_layout_t *l, const double *x, double *buf, const int *begin, const int *end) {
  const ptrdiff_t n0 = end[0] - begin[0], n1 = end[1] - begin[1];
ORC_PARFOR
  for (int i2 = begin[2]; i2 < end[2]; ++i2)
    for (int i1 = begin[1]; i1 < end[1]; ++i1)
      for (int i0 = begin[0]; i0 < end[0]; ++i0)
        buf[(i0 - begin[0]) + n0 * ((i1 - begin[1]) + n1 * (ptrdiff_t)(i2 - begin[2]))] = x[lay_idx(l, i0, i1, i2)];
}

void orc_unpack(const orc_layout_t *l, double *x, const double *buf, const int *begin, const int *end) {
  const ptrdiff_t n0 = end[0] - begin[0], n1 = end[1] - begin[1];
ORC_PARFOR
  for (int i2 = begin[2]; i2 < end[2]; ++i2)
    for (int i1 = begin[1]; i1 < end[1]; ++i1)
      for (int i0 = begin[0]; i0 < end[0]; ++i0)
        x[lay_idx(l, i0, i1, i2)] = buf[(i0 - begin[0]) + n0 * ((i1 - begin[1]) + n1 * (ptrdiff_t)(i2 - begin[2]))];
}

/* Deterministic synthetic fill: SplitMix64 on the linear index, U(-1,1).
 * (Used for kernel-vs-oracle equality on identical random fields; a counter-based
 * generator so that the GPU side can produce the same field without a copy.) */
void orc_fill_random(double *x, long n, uint64_t seed) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) {
    uint64_t zz = seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(i + 1);
    zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ULL;
    zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBULL;
    zz = zz ^ (zz >> 31);
    x[i] = (double)(zz >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  }
}

/* Start values of `loop over F sequentially { F = native("((double)std::rand()/RAND_MAX)") }` (Testing/Opts/base.exa4:166-170):
 * the generated nest calls the C library's rand() once per point, x fastest, after std::srand(mpiRank) in an MPI program
 * (parallelization/api/mpi/MPI_IVs.scala:41-45).  The oracle does what the generated code does: it calls the C library (glibc in
 * this image, as on the machines the reference's results files come from).  seed < 0: keep the generator's state. */
void orc_crand_fill(const orc_layout_t *l, double *x, const int32_t *begin, const int32_t *end, int seed) {
  if (seed >= 0) srand((unsigned)seed);
  for (int i2 = begin[2]; i2 < end[2]; ++i2)
    for (int i1 = begin[1]; i1 < end[1]; ++i1)
      for (int i0 = begin[0]; i0 < end[0]; ++i0) x[lay_idx(l, i0, i1, i2)] = (double)rand() / RAND_MAX;
}

#ifdef _OPENMP
#include <omp.h>
int orc_num_threads(void) { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { omp_set_num_threads(n); }
#else
int orc_num_threads(void) { return 1; }
void orc_set_num_threads(int n) { (void)n; }
#endif

/* ------------------------------------------------------------------ */
/* Config 4 (BASELINE.json): 27-entry stencil field of a variable-coefficient Helmholtz operator
 *   -div(a grad u) - k^2 u,  a = coefficient profile `coef_fn`,
 * discretised with trilinear elements (element-wise constant a sampled at the element centre, lumped
 * mass), scaled by 1/h^3 so that it matches the finite-difference scaling of the other programs.
 * The reference ships stencil fields with 2d+1 entries only (Testing/SISC/3D_VarCoeff.exa4); this is
 * the same mechanism (entry index slowest, C/stencil/ir/IR_StencilConvolution.scala:73-95) with 27
 * entries -- no golden exists for it in the reference ("parity unpinned").
 * Entry order: (0,0,0) first, then the offsets with dz = -1, 0, +1 (dz slowest, dx fastest): a pass that marches in z can then
 * finish the 27-term sum of a point plane by plane (exastencils_amd/csrc/kernels_sf27pair.hip).
 * Element matrix of the unit cube (Q1): 1/3 on the diagonal, 0 across an edge, -1/12 across a face
 * diagonal and across the body diagonal; entry(o) = sum over the elements containing both nodes.   */
void orc_init_helmholtz27(const orc_layout_t *lc, double *cf, const orc_geom_t *g, int coef_fn, const double *p,
                          const int *begin, const int *end) {
  const ptrdiff_t plane = lay_size(lc);
  const double ksq = p[1];
ORC_PARFOR
  for (int i2 = begin[2]; i2 < end[2]; ++i2)
    for (int i1 = begin[1]; i1 < end[1]; ++i1)
      for (int i0 = begin[0]; i0 < end[0]; ++i0) {
        const double h = g->h[0];
        const double x = i0 * g->h[0] + g->pos_begin[0];
        const double y = i1 * g->h[1] + g->pos_begin[1];
        const double z = i2 * g->h[2] + g->pos_begin[2];
        double ae[2][2][2];
        for (int sx = 0; sx < 2; ++sx)
          for (int sy = 0; sy < 2; ++sy)
            for (int sz = 0; sz < 2; ++sz)
              ae[sx][sy][sz] = orc_eval_fn(coef_fn, p, x + (sx ? 0.5 : -0.5) * h, y + (sy ? 0.5 : -0.5) * h, z + (sz ? 0.5 : -0.5) * h);
        const ptrdiff_t k = lay_idx(lc, i0, i1, i2);
        int ent = 1;
        for (int dz = -1; dz <= 1; ++dz)
          for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
              const int nnz = (dx != 0) + (dy != 0) + (dz != 0);
              double s = 0.0;
              int first = 1;
              for (int sx = 0; sx < 2; ++sx)
                for (int sy = 0; sy < 2; ++sy)
                  for (int sz = 0; sz < 2; ++sz) {
                    const int okx = dx == 0 || (dx > 0) == sx, oky = dy == 0 || (dy > 0) == sy, okz = dz == 0 || (dz > 0) == sz;
                    if (okx && oky && okz) { s = first ? ae[sx][sy][sz] : s + ae[sx][sy][sz]; first = 0; }
                  }
              const double kf = nnz == 0 ? (1.0 / 3.0) : (nnz == 1 ? 0.0 : (-1.0 / 12.0));
              double c = (s * kf) / (h * h);
              if (nnz == 0) { c = c - ksq; cf[k] = c; }
              else { cf[k + (ptrdiff_t)ent * plane] = c; ++ent; }
            }
      }
}
