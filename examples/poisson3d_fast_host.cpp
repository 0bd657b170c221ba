// The program of examples/poisson3d_host.cpp (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4) with the one-pass entry points
// of libexamg and the cycle replayed from a hipGraph: what a C++ host -- the reference's own host language -- needs to reach
// the V-cycle time bench.py reports from Python.  HIP runtime + libexamg only.
//
//   * smoother:               examg_rbgs_sweep_fused (one pass per red-black sweep, out of place, pointer swap); where the rows are
//                             too short for it, the two colour loops in place
//   * residual + restriction: examg_residual_restrict (the fine residual is never stored)
//   * `Solution@coarser = 0`: left to the coarser level's first sweep (examg_rbgs_sweep_fused_zero) where that is a one-pass sweep
//   * correction:             folded into the first post-smoothing sweep (examg_rbgs_sweep_fused_prolong) from <foldMinPoints> points
//   * Solve: residual + norm: examg_residual_norm2
//   * `apply bc`:             once -- every loop of the cycle writes inner points only and the boundary values depend on the
//                             position only (both Solution arrays of a level carry them)
// Every one of these is bit-identical to the statement-by-statement program (tests/test_gpu_kernels.py); the norms differ from
// it by the summation order of the reduction (1e-15 relative).
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/poisson3d_fast_host.cpp -Lexastencils_amd -lexamg -o poisson3d_fast_host
//   LD_LIBRARY_PATH=exastencils_amd ./poisson3d_fast_host [maxLevel=9] [minLevel=4] [foldMinPoints=10000000]
//
// Prints the residual norm per V-cycle (4 significant digits), the full-precision values ('# ' lines), the iteration count, and
// `vcycle_ms` (graph replay, steady state) / `totalTimeSolve_ms` (the benchmark's own reported quantity).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <utility>
#include <vector>

#include "examg.h"

static void check(int rc, const char *what) {
  if (rc) { std::fprintf(stderr, "%s: %s\n", what, examg_last_error()); std::exit(1); }
}
static void checkHip(hipError_t e, const char *what) {
  if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); std::exit(1); }
}

static int minLevel = 4, maxLevel = 9;
static long long foldMinPoints = 10000000;
struct Level {
  examg_layout_t withComm, noGhost;
  double *Solution, *SolutionAlt, *RHS, *Residual;   // SolutionAlt: the second array of the out-of-place sweeps
  const examg_layout_t *resLayout;
  examg_stencil_t Laplace;
  examg_geom_t geom;
  int32_t begin[3], end[3];
  size_t nWith, nNo;
  bool onePass;                                      // does the kernel layer run this level's sweep as one pass?
  long long points;
};
static std::vector<Level> L;
static double *cgTmp0, *cgTmp1, *scalar, *work, *cgInfo;
static hipStream_t stream;

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
  checkHip(hipMemset(p, 0, n * sizeof(double)), "hipMemset");
  return p;
}

static examg_expr_t bcSolution;
static void initBoundaryExpression() {   // x*x - 0.5*y*y - 0.5*z*z (...exa4:24-25)
  const int ops[] = {EXAMG_OP_X, EXAMG_OP_X, EXAMG_OP_MUL, EXAMG_OP_CONST, EXAMG_OP_Y, EXAMG_OP_MUL, EXAMG_OP_Y, EXAMG_OP_MUL, EXAMG_OP_SUB,
                     EXAMG_OP_CONST, EXAMG_OP_Z, EXAMG_OP_MUL, EXAMG_OP_Z, EXAMG_OP_MUL, EXAMG_OP_SUB};
  bcSolution = examg_expr_t{};
  bcSolution.n = (int)(sizeof(ops) / sizeof(ops[0]));
  for (int i = 0; i < bcSolution.n; ++i) { bcSolution.op[i] = ops[i]; bcSolution.c[i] = ops[i] == EXAMG_OP_CONST ? 0.5 : 0.0; }
}

// repeat 3 times { color with {...} }; correctionFrom: the coarser level whose prolongation is added first (folded into the first
// sweep); zeroInput: Solution is 0 everywhere and was not written as such
static void smoother(int lvl, const Level *correctionFrom, bool zeroInput) {
  Level &v = L[lvl];
  const double w = 0.8 / v.Laplace.coef[v.Laplace.diag];
  if (!v.onePass) {
    for (int it = 0; it < 3; ++it)
      for (int colour = 0; colour < 2; ++colour)
        check(examg_rbgs_colour(&v.withComm, v.Solution, &v.noGhost, v.RHS, &v.Laplace, w, colour, v.begin, v.end, stream), "smoother");
    return;
  }
  for (int it = 0; it < 3; ++it) {
    if (it == 0 && correctionFrom)
      check(examg_rbgs_sweep_fused_prolong(&v.withComm, v.Solution, v.SolutionAlt, &v.noGhost, v.RHS, &v.Laplace, w, 0, v.begin, v.end,
                                           &correctionFrom->withComm, correctionFrom->Solution, stream), "correction + sweep");
    else if (it == 0 && zeroInput)
      check(examg_rbgs_sweep_fused_zero(&v.withComm, v.SolutionAlt, &v.noGhost, v.RHS, &v.Laplace, w, 0, v.begin, v.end, stream), "sweep of 0");
    else
      check(examg_rbgs_sweep_fused(&v.withComm, v.Solution, v.SolutionAlt, &v.noGhost, v.RHS, &v.Laplace, w, 0, v.begin, v.end, stream), "sweep");
    std::swap(v.Solution, v.SolutionAlt);      // six swaps per level and cycle: the graph's pointers stay valid
  }
}

static void mgCycle(int lvl, bool solutionIsZero) {
  Level &v = L[lvl];
  if (lvl == minLevel) {
    // the zero field as the start (`Solution@coarsest = 0` of the level above rides along: EXAMG_CG_ZERO_START)
    check(examg_cg_coarse_variant(&v.withComm, v.Solution, &v.noGhost, v.RHS, v.resLayout, v.Residual, &v.withComm, cgTmp0, &v.noGhost, cgTmp1,
                                  &v.Laplace, &v.geom, 63u, 128, 0.001, v.begin, v.end, solutionIsZero ? EXAMG_CG_ZERO_START : 0u, cgInfo, stream),
          "mgCycle@coarsest");
    return;
  }
  Level &c = L[lvl - 1];
  smoother(lvl, nullptr, solutionIsZero);
  check(examg_residual_restrict(&v.withComm, v.Solution, &v.noGhost, v.RHS, v.resLayout, v.Residual, &v.Laplace, &c.noGhost, c.RHS, 1.0,
                                v.begin, v.end, c.begin, c.end, stream), "residual + restriction");
  const bool zeroStart = lvl - 1 == minLevel || c.onePass;      // the coarsest level: the CG takes the zero field; others: their first sweep
  if (!zeroStart) check(examg_set(&c.withComm, c.Solution, 0.0, c.begin, c.end, stream), "Solution@coarser = 0");
  mgCycle(lvl - 1, zeroStart);
  // the fold pays on large levels (a read-modify-write pass less) and on launch-bound ones (rows shorter than 64 points: a kernel less)
  if (v.onePass && (v.points >= foldMinPoints || (1 << lvl) - 1 < 64)) {
    smoother(lvl, &c, false);
  } else {
    check(examg_prolong_add(&c.withComm, c.Solution, &v.withComm, v.Solution, v.begin, v.end, stream), "prolongation");
    smoother(lvl, nullptr, false);
  }
}

static double residualNorm(int lvl) {   // Residual = RHS - Laplace * Solution; ResNorm()
  Level &v = L[lvl];
  check(examg_residual_norm2(&v.withComm, v.Solution, &v.noGhost, v.RHS, &v.Laplace, v.begin, v.end, v.resLayout, v.Residual, scalar, work,
                             stream), "residual + norm");
  double h;
  checkHip(hipMemcpyAsync(&h, scalar, sizeof(double), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
  checkHip(hipStreamSynchronize(stream), "sync");
  return std::sqrt(h);
}

static void resetFields() {   // initFieldsWithZero + the boundary values (both arrays of every level)
  for (int l = minLevel; l <= maxLevel; ++l) {
    Level &v = L[l];
    checkHip(hipMemsetAsync(v.Solution, 0, v.nWith * sizeof(double), stream), "memset");
    checkHip(hipMemsetAsync(v.SolutionAlt, 0, v.nWith * sizeof(double), stream), "memset");
    checkHip(hipMemsetAsync(v.RHS, 0, v.nNo * sizeof(double), stream), "memset");
  }
  Level &f = L[maxLevel];
  for (double *p : {f.Solution, f.SolutionAlt})
    check(examg_apply_dirichlet_expr(&f.withComm, p, &f.geom, &bcSolution, 63u, stream), "apply bc");
}

int main(int argc, char **argv) {
  if (argc > 1) maxLevel = std::atoi(argv[1]);
  if (argc > 2) minLevel = std::atoi(argv[2]);
  if (argc > 3) foldMinPoints = std::atoll(argv[3]);
  if (minLevel < 1 || maxLevel <= minLevel || maxLevel > 10) { std::fprintf(stderr, "levels out of range\n"); return 2; }
  if (examg_device_count() < 1) { std::fprintf(stderr, "no HIP device\n"); return 2; }
  checkHip(hipStreamCreate(&stream), "hipStreamCreate");
  L.resize(maxLevel + 1);
  for (int l = minLevel; l <= maxLevel; ++l) {
    Level &v = L[l];
    v.withComm = nodeLayout(l, 1);
    v.noGhost = nodeLayout(l, 0);
    v.nWith = layoutSize(v.withComm);
    v.nNo = layoutSize(v.noGhost);
    v.Solution = deviceZeros(v.nWith);
    v.SolutionAlt = deviceZeros(v.nWith);
    v.RHS = deviceZeros(v.nNo);
    v.resLayout = (l == minLevel) ? &v.noGhost : &v.withComm;
    v.Residual = deviceZeros(l == minLevel ? v.nNo : v.nWith);
    const double h = 1.0 / (1 << l);
    v.points = 1;
    for (int d = 0; d < 3; ++d) { v.geom.pos_begin[d] = 0.0; v.geom.h[d] = h; v.begin[d] = 1; v.end[d] = 1 << l; v.points *= (1 << l) - 1; }
    examg_stencil_t &A = v.Laplace;
    A = examg_stencil_t{};
    A.nent = 7;
    A.diag = 0;
    const int off[7][3] = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
    for (int k = 0; k < 7; ++k) for (int d = 0; d < 3; ++d) A.off[k][d] = off[k][d];
    A.coef[0] = 2.0 / std::pow(h, 2) + 2.0 / std::pow(h, 2) + 2.0 / std::pow(h, 2);
    for (int k = 1; k < 7; ++k) A.coef[k] = -1.0 / std::pow(h, 2);
    A.cfield = nullptr;
    v.onePass = examg_two_stage_eligible(&v.withComm, &v.noGhost, &A, v.begin, v.end, v.begin, v.end) != 0;
  }
  cgTmp0 = deviceZeros(L[minLevel].nWith);
  cgTmp1 = deviceZeros(L[minLevel].nNo);
  scalar = deviceZeros(1);
  cgInfo = deviceZeros(4);
  checkHip(hipMalloc(&work, examg_reduce_work_bytes()), "hipMalloc");
  initBoundaryExpression();
  resetFields();

  // one mgCycle@finest as a hipGraph (no host round trip inside: the coarse solve decides its exit on the device)
  mgCycle(maxLevel, false);                                  // outside the capture first (lazy initialisation)
  checkHip(hipStreamSynchronize(stream), "sync");
  hipGraph_t graph;
  hipGraphExec_t cycle;
  checkHip(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
  mgCycle(maxLevel, false);
  checkHip(hipStreamEndCapture(stream, &graph), "hipStreamEndCapture");
  checkHip(hipGraphInstantiate(&cycle, graph, nullptr, nullptr, 0), "hipGraphInstantiate");

  // steady-state time of one cycle
  for (int i = 0; i < 10; ++i) checkHip(hipGraphLaunch(cycle, stream), "hipGraphLaunch");
  checkHip(hipStreamSynchronize(stream), "sync");
  auto t0 = std::chrono::steady_clock::now();
  const int nCycles = 10;
  for (int i = 0; i < nCycles; ++i) checkHip(hipGraphLaunch(cycle, stream), "hipGraphLaunch");
  checkHip(hipStreamSynchronize(stream), "sync");
  const double vcycleMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / nCycles;

  // Function Solve@finest (...exa4:121-150) from the benchmark's initial state
  resetFields();
  checkHip(hipStreamSynchronize(stream), "sync");
  t0 = std::chrono::steady_clock::now();
  const double initRes = residualNorm(maxLevel);
  double curRes = initRes;
  std::vector<double> history{initRes};
  int curIt = 0;
  while (!(curIt >= 100 || curRes <= 1.0E-6 * initRes)) {
    ++curIt;
    checkHip(hipGraphLaunch(cycle, stream), "hipGraphLaunch");
    curRes = residualNorm(maxLevel);
    history.push_back(curRes);
  }
  const double solveMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::cout.precision(4);
  for (double r : history) std::cout << r << std::endl;
  for (double r : history) std::printf("# %.17g\n", r);
  std::printf("iterations %d\n", curIt);
  std::printf("vcycle_ms %.4f\n", vcycleMs);
  std::printf("totalTimeSolve_ms %.3f\n", solveMs);
  return 0;
}
