// C++ host program on top of the C ABI (include/examg.h) in the shape of the code ExaStencils generates for
// Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4: process-global device arrays per field and level, one function per
// leveled ExaSlang function (mgCycle_<lvl>, ResNorm_<lvl>, Solve), every `loop over` a call into libexamg.
// It shows the drop-in from the reference's own host language: no Python, no PyTorch -- HIP runtime + libexamg only.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/poisson3d_host.cpp -Lexastencils_amd -lexamg -o poisson3d_host
//   LD_LIBRARY_PATH=exastencils_amd ./poisson3d_host [maxLevel=6] [minLevel=2]
//
// Prints the residual norm per V-cycle the way the generated program does with testing_enabled (4 significant digits),
// followed by the full-precision values (lines starting with '#') that tests/test_gpu_solver.py compares with the oracle.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "examg.h"

static void check(int rc, const char *what) {  // reference behaviour: print and exit (cuda/CUDA_Error.scala)
  if (rc) { std::fprintf(stderr, "%s: %s\n", what, examg_last_error()); std::exit(1); }
}
static void checkHip(hipError_t e, const char *what) {
  if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); std::exit(1); }
}

static int minLevel = 2, maxLevel = 6;
struct Level {
  examg_layout_t withComm, noGhost;          // Layout NodeWithComm / NodeNoGhost (...exa4:13-21)
  double *Solution, *RHS, *Residual;         // fieldDeviceData_<F>[lvl]
  const examg_layout_t *resLayout;
  examg_stencil_t Laplace;                   // Stencil Laplace@all (...exa4:39-47)
  examg_geom_t geom;
  int32_t begin[3], end[3];                  // loop over <field>: [1, 2^L)
  size_t nWith, nNo;
};
static std::vector<Level> L;                 // index = level
static double *cgTmp0, *cgTmp1, *scalar, *work, *cgInfo;

static examg_layout_t nodeLayout(int level, int ghost) {
  examg_layout_t l{};
  l.nd = 3;
  for (int d = 0; d < 3; ++d) {
    l.ghost_l[d] = l.ghost_r[d] = ghost;
    l.dup_l[d] = l.dup_r[d] = 1;
    l.inner[d] = (1 << level) - 1;
  }
  return l;
}
static size_t layoutSize(const examg_layout_t &l) {
  size_t n = 1;
  for (int d = 0; d < 3; ++d) n *= l.pad_l[d] + l.ghost_l[d] + l.dup_l[d] + l.inner[d] + l.dup_r[d] + l.ghost_r[d] + l.pad_r[d];
  return n;
}
static double *deviceZeros(size_t n) {
  double *p;
  checkHip(hipMalloc(&p, n * sizeof(double)), "hipMalloc");
  checkHip(hipMemset(p, 0, n * sizeof(double)), "hipMemset");   // initFieldsWithZero
  return p;
}

// boundary expressions as postfix programs (the generator inlines them into the boundary kernels):
// Field Solution's at the finest level (...exa4:24-25): x*x - 0.5*y*y - 0.5*z*z ; 0.0 everywhere else
static examg_expr_t bcSolution, bcZero;
static void initBoundaryExpressions() {
  const int ops[] = {EXAMG_OP_X, EXAMG_OP_X, EXAMG_OP_MUL, EXAMG_OP_CONST, EXAMG_OP_Y, EXAMG_OP_MUL, EXAMG_OP_Y, EXAMG_OP_MUL, EXAMG_OP_SUB,
                     EXAMG_OP_CONST, EXAMG_OP_Z, EXAMG_OP_MUL, EXAMG_OP_Z, EXAMG_OP_MUL, EXAMG_OP_SUB};
  bcSolution = examg_expr_t{};
  bcSolution.n = (int)(sizeof(ops) / sizeof(ops[0]));
  for (int i = 0; i < bcSolution.n; ++i) { bcSolution.op[i] = ops[i]; bcSolution.c[i] = ops[i] == EXAMG_OP_CONST ? 0.5 : 0.0; }
  bcZero = examg_expr_t{};
  bcZero.n = 1;
  bcZero.op[0] = EXAMG_OP_CONST;
  bcZero.c[0] = 0.0;
}
static void applyBCsSolution(int lvl) {  // apply bc to Solution: boundary function at the finest level, 0.0 elsewhere
  check(examg_apply_dirichlet_expr(&L[lvl].withComm, L[lvl].Solution, &L[lvl].geom, lvl == maxLevel ? &bcSolution : &bcZero, 63u, nullptr),
        "applyBCsSolution");
}
static void applyBCsResidual(int lvl) {
  check(examg_apply_dirichlet_expr(L[lvl].resLayout, L[lvl].Residual, &L[lvl].geom, &bcZero, 63u, nullptr), "applyBCsResidual");
}
static void updateResidual(int lvl) {  // communicate Solution (empty: one block); Residual = RHS - Laplace * Solution; apply bc
  Level &v = L[lvl];
  check(examg_residual(&v.withComm, v.Solution, &v.noGhost, v.RHS, v.resLayout, v.Residual, &v.Laplace, v.begin, v.end, nullptr), "residual");
  applyBCsResidual(lvl);
}
static double ResNorm(int lvl) {  // Function ResNorm@(coarsest and finest) (...exa4:113-119)
  Level &v = L[lvl];
  check(examg_dot(v.resLayout, v.Residual, v.resLayout, v.Residual, v.begin, v.end, scalar, work, nullptr), "ResNorm");
  double h;
  checkHip(hipMemcpy(&h, scalar, sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy");   // the 8-byte D2H copy
  return std::sqrt(h);
}
static void smoother(int lvl) {  // repeat 3 times { color with { (i0+i1+i2) % 2, ... } } (...exa4:204-213)
  Level &v = L[lvl];
  const double w = 0.8 / v.Laplace.coef[v.Laplace.diag];
  for (int it = 0; it < 3; ++it)
    for (int colour = 0; colour < 2; ++colour) {
      check(examg_rbgs_colour(&v.withComm, v.Solution, &v.noGhost, v.RHS, &v.Laplace, w, colour, v.begin, v.end, nullptr), "smoother");
      applyBCsSolution(lvl);
    }
}
static void mgCycle(int lvl) {  // Function mgCycle@(all but coarsest) / @coarsest (...exa4:152-249)
  Level &v = L[lvl];
  if (lvl == minLevel) {
    check(examg_cg_coarse(&v.withComm, v.Solution, &v.noGhost, v.RHS, v.resLayout, v.Residual, &v.withComm, cgTmp0, &v.noGhost, cgTmp1,
                          &v.Laplace, &v.geom, 63u, 128, 0.001, v.begin, v.end, cgInfo, nullptr), "mgCycle@coarsest");
    return;
  }
  Level &c = L[lvl - 1];
  smoother(lvl);
  updateResidual(lvl);
  check(examg_restrict(v.resLayout, v.Residual, &c.noGhost, c.RHS, 1.0, c.begin, c.end, nullptr), "restriction");
  check(examg_set(&c.withComm, c.Solution, 0.0, c.begin, c.end, nullptr), "Solution@coarser = 0");
  applyBCsSolution(lvl - 1);
  mgCycle(lvl - 1);
  check(examg_prolong_add(&c.withComm, c.Solution, &v.withComm, v.Solution, v.begin, v.end, nullptr), "prolongation");
  applyBCsSolution(lvl);
  smoother(lvl);
}

int main(int argc, char **argv) {
  if (argc > 1) maxLevel = std::atoi(argv[1]);
  if (argc > 2) minLevel = std::atoi(argv[2]);
  if (minLevel < 1 || maxLevel <= minLevel || maxLevel > 10) { std::fprintf(stderr, "levels out of range\n"); return 2; }
  if (examg_device_count() < 1) { std::fprintf(stderr, "no HIP device\n"); return 2; }
  L.resize(maxLevel + 1);
  for (int l = minLevel; l <= maxLevel; ++l) {   // setupBuffers()
    Level &v = L[l];
    v.withComm = nodeLayout(l, 1);
    v.noGhost = nodeLayout(l, 0);
    v.nWith = layoutSize(v.withComm);
    v.nNo = layoutSize(v.noGhost);
    v.Solution = deviceZeros(v.nWith);
    v.RHS = deviceZeros(v.nNo);
    v.resLayout = (l == minLevel) ? &v.noGhost : &v.withComm;   // Field Residual (...exa4:29-30)
    v.Residual = deviceZeros(l == minLevel ? v.nNo : v.nWith);
    const double h = 1.0 / (1 << l);
    for (int d = 0; d < 3; ++d) { v.geom.pos_begin[d] = 0.0; v.geom.h[d] = h; v.begin[d] = 1; v.end[d] = 1 << l; }
    examg_stencil_t &A = v.Laplace;
    A = examg_stencil_t{};
    A.nent = 7;
    A.diag = 0;
    const int off[7][3] = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
    for (int k = 0; k < 7; ++k) for (int d = 0; d < 3; ++d) A.off[k][d] = off[k][d];
    A.coef[0] = 2.0 / std::pow(h, 2) + 2.0 / std::pow(h, 2) + 2.0 / std::pow(h, 2);
    for (int k = 1; k < 7; ++k) A.coef[k] = -1.0 / std::pow(h, 2);
    A.cfield = nullptr;
  }
  cgTmp0 = deviceZeros(L[minLevel].nWith);
  cgTmp1 = deviceZeros(L[minLevel].nNo);
  scalar = deviceZeros(1);
  cgInfo = deviceZeros(4);
  checkHip(hipMalloc(&work, examg_reduce_work_bytes()), "hipMalloc");

  initBoundaryExpressions();
  applyBCsSolution(maxLevel);               // Function Application (...exa4:251-277)
  // Function Solve@finest (...exa4:121-150)
  updateResidual(maxLevel);
  const double initRes = ResNorm(maxLevel);
  double curRes = initRes;
  std::vector<double> history{initRes};
  std::cout.precision(4);
  std::cout << initRes << std::endl;
  int curIt = 0;
  while (!(curIt >= 100 || curRes <= 1.0E-6 * initRes)) {
    ++curIt;
    mgCycle(maxLevel);
    updateResidual(maxLevel);
    curRes = ResNorm(maxLevel);
    history.push_back(curRes);
    std::cout << curRes << std::endl;
  }
  for (double r : history) std::printf("# %.17g\n", r);
  std::printf("iterations %d\n", curIt);
  checkHip(hipDeviceSynchronize(), "sync");
  return 0;   // destroyGlobals(): process exit
}
