// libexamg bodies for the reference-named kernel wrappers, exchange and boundary functions of
// Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4 (names and numbering: exa_poisson3d.h).  This file stands where the generator
// prints Kernel/Kernel_<fn>_k<NNN>.cu (cuda/CUDA_KernelFunctions.scala:100-110) and the communication / boundary functions
// (IR_CommunicateFunction.scala, IR_ApplyBCFunction.scala); the global arrays, layouts and loop bounds are the generator's.
// Errors print and exit, as the reference's CUDA_CheckError does (cuda/CUDA_Error.scala); every wrapper returns void.
#include <hip/hip_runtime.h>

#include <signal.h>
#include <unistd.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "exa_poisson3d.h"

extern "C" {
double *fieldDeviceData_Solution[EXA_NUM_LEVELS];
double *fieldDeviceData_RHS[EXA_NUM_LEVELS];
double *fieldDeviceData_Residual[EXA_NUM_LEVELS];
double *fieldDeviceData_cgTmp0[1];
double *fieldDeviceData_cgTmp1[1];
}

namespace {

void check(int rc, const char *what) {
  if (rc) { std::fprintf(stderr, "%s: %s\n", what, examg_last_error()); std::exit(1); }
}
void checkHip(hipError_t e, const char *what) {
  if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); std::exit(1); }
}

struct LevelInfo {
  examg_layout_t withComm, noGhost;   // Layout NodeWithComm / NodeNoGhost (...exa4:13-21)
  const examg_layout_t *resLayout;    // Field Residual: NodeNoGhost on the coarsest level, NodeWithComm elsewhere (:29-30)
  examg_stencil_t Laplace;            // Stencil Laplace@all (:39-47), coefficients folded as the generator folds them
  examg_geom_t geom;
  int32_t begin[3], end[3];           // loop over <node field>: [DLB + iterationOffsetBegin, DRE + iterationOffsetEnd)
  int32_t rbegin[3];                  // reduction loops skip the lower duplicate plane at interior faces
  void *wsWith, *wsNo;                // buffer_Send / buffer_Recv of the exchange functions
  size_t wsWithBytes, wsNoBytes;
};
LevelInfo g_lv[EXA_NUM_LEVELS];
int g_blocks[3] = {1, 1, 1}, g_pos[3] = {0, 0, 0}, g_rank = 0, g_size = 1;
examg_comm_t *g_comm = nullptr;
std::string g_peerFile;               // this rank's handle file of the peer-write bootstrap (removed in destroyGlobals)
examg_neighbors_t g_nb;
uint32_t g_faceMask = 63u;
double *g_scalar = nullptr, *g_work = nullptr;
examg_expr_t g_bcSolution, g_bcZero;   // Field Solution's boundary expression at the finest level (:24): x^2 - 0.5 y^2 - 0.5 z^2

LevelInfo &lv(int level) { return g_lv[level - EXA_MIN_LEVEL]; }

// ---- deferred-launch mode (EXA_DEFERRED_LAUNCH=1, one block) ---------------------------------------------------------------------
// The wrappers keep their reference names and signatures, but a wrapper whose loop one of libexamg's one-pass kernels can absorb
// RECORDS its call instead of launching; the wrapper that completes the pattern issues the one-pass kernel:
//   k000 + k001 (k006 + k007)            the two colour loops of a sweep       -> examg_rbgs_sweep_fused            24 B per update, not 48
//   k002 + k003                          residual, restriction                 -> examg_residual_restrict           the residual is never stored
//   k004 @L + k000 + k001 @L-1           Solution@coarser = 0, first sweep     -> examg_rbgs_sweep_fused_zero
//   k005 + k006 + k007                   correction, first post-sweep          -> examg_rbgs_sweep_fused_prolong
// `exch*` of a block without neighbours is empty and `applyBCs*` re-writes position-only values on planes no loop touches (they are
// applied once per array and skipped from then on): both pass while a call is pending.  ANY other entry point first runs what is
// pending as the plain statements it stands for (flushPending), so the arrays hold what the program says whenever something outside
// these patterns looks.  The one-pass sweeps are out of place: they write the level's second array and swap
// fieldDeviceData_Solution[i] with it (generated host code reads the global array at every use, as it does for slotted fields).
// Same bits as the plain wrappers (tests/test_gpu_shim.py).
enum PendingKind { P_NONE, P_HALF0, P_RES, P_ZERO, P_ZERO_HALF0, P_PROL, P_PROL_HALF0 };
struct Pending {
  PendingKind kind = P_NONE;
  int level = 0;      // the level whose Solution / Residual the recorded loop writes
} g_pending;
bool g_deferred = false;
double *g_solutionAlt[EXA_NUM_LEVELS];      // second Solution array of the out-of-place sweeps
bool g_onePass[EXA_NUM_LEVELS];             // libexamg takes the sweep and residual + restriction of this level as one pass each
bool g_residualNotStored[EXA_NUM_LEVELS];   // Residual@level was consumed inside a one-pass kernel and never written
bool g_bcDone[EXA_NUM_LEVELS][2];           // boundary values of Solution@level written into (array, second array)
bool g_bcResDone[EXA_NUM_LEVELS];           // ... of Residual@level
long g_launches = 0;                        // libexamg calls that launch kernels

examg_layout_t nodeLayout(int level, int ghost) {
  examg_layout_t l;
  std::memset(&l, 0, sizeof(l));
  l.nd = 3;
  for (int d = 0; d < 3; ++d) {
    l.ghost_l[d] = l.ghost_r[d] = ghost;
    l.dup_l[d] = l.dup_r[d] = 1;
    l.inner[d] = (1 << level) - 1;
  }
  return l;
}
size_t layoutSize(const examg_layout_t &l) {
  size_t n = 1;
  for (int d = 0; d < 3; ++d) n *= l.pad_l[d] + l.ghost_l[d] + l.dup_l[d] + l.inner[d] + l.dup_r[d] + l.ghost_r[d] + l.pad_r[d];
  return n;
}
double *deviceZeros(size_t n) {
  double *p;
  checkHip(hipMalloc(&p, n * sizeof(double)), "hipMalloc");
  checkHip(hipMemset(p, 0, n * sizeof(double)), "hipMemset");   // initFieldsWithZero
  return p;
}

void smoothColour(int level, int colour, const char *what) {   // Solution += 0.8 / diag(Laplace) * (RHS - Laplace * Solution), one colour
  LevelInfo &v = lv(level);
  const double w = 0.8 / v.Laplace.coef[v.Laplace.diag];
  ++g_launches;
  check(examg_rbgs_colour(&v.withComm, fieldDeviceData_Solution[level - EXA_MIN_LEVEL], &v.noGhost, fieldDeviceData_RHS[level - EXA_MIN_LEVEL],
                          &v.Laplace, w, colour, v.begin, v.end, nullptr), what);
}
void residual(int level, const char *what) {
  LevelInfo &v = lv(level);
  const int i = level - EXA_MIN_LEVEL;
  ++g_launches;
  g_residualNotStored[i] = false;
  check(examg_residual(&v.withComm, fieldDeviceData_Solution[i], &v.noGhost, fieldDeviceData_RHS[i], v.resLayout, fieldDeviceData_Residual[i],
                       &v.Laplace, v.begin, v.end, nullptr), what);
}
void reduceDot(const examg_layout_t *lx, const double *x, const examg_layout_t *ly, const double *y, int level, double *reductionTmp,
               const char *what) {
  // kernel + DefaultReductionKernel + the blocking copy of the result (cuda/CUDA_Kernel.scala:595-616); the host code adds the
  // MPI_Allreduce itself (exa_allreduce_sum)
  LevelInfo &v = lv(level);
  ++g_launches;
  check(examg_dot(lx, x, ly, y, v.rbegin, v.end, g_scalar, g_work, nullptr), what);
  checkHip(hipMemcpy(reductionTmp, g_scalar, sizeof(double), hipMemcpyDeviceToHost), what);
}
void exchange(const examg_layout_t *l, double *x, void *ws, size_t wsBytes, int what, const char *name) {
  if (g_size == 1) return;   // no neighbours: the generated function is empty
  ++g_launches;
  check(examg_exchange(g_comm, l, x, &g_nb, what, ws, wsBytes, nullptr), name);
}
void applyBC(const examg_layout_t *l, double *x, int level, const examg_expr_t *e, const char *name) {
  if (g_faceMask) { ++g_launches; check(examg_apply_dirichlet_expr(l, x, &lv(level).geom, e, g_faceMask, nullptr), name); }
}

// ---- the plain statements of mgCycle@(all but coarsest), and the deferred forms ---------------------------------------------------
const examg_expr_t *bcOfSolution(int level);
double smootherWeight(int level) { return 0.8 / lv(level).Laplace.coef[lv(level).Laplace.diag]; }

void restrictPlain(int L) {
  ++g_launches;
  check(examg_restrict(lv(L).resLayout, fieldDeviceData_Residual[L - EXA_MIN_LEVEL], &lv(L - 1).noGhost, fieldDeviceData_RHS[L - 1 - EXA_MIN_LEVEL], 1.0,
                       lv(L - 1).begin, lv(L - 1).end, nullptr), "mgCycle_k003");
}
void zeroPlain(int LM1) {
  ++g_launches;
  check(examg_set(&lv(LM1).withComm, fieldDeviceData_Solution[LM1 - EXA_MIN_LEVEL], 0.0, lv(LM1).begin, lv(LM1).end, nullptr), "mgCycle_k004");
}
void prolongPlain(int L) {
  ++g_launches;
  check(examg_prolong_add(&lv(L - 1).withComm, fieldDeviceData_Solution[L - 1 - EXA_MIN_LEVEL], &lv(L).withComm, fieldDeviceData_Solution[L - EXA_MIN_LEVEL],
                          lv(L).begin, lv(L).end, nullptr), "mgCycle_k005");
}

// what is recorded runs as the plain statements it stands for
void flushPending() {
  const Pending p = g_pending;
  g_pending.kind = P_NONE;
  switch (p.kind) {
    case P_NONE: break;
    case P_HALF0: smoothColour(p.level, 0, "mgCycle colour 0 (flushed)"); break;
    case P_RES: residual(p.level, "mgCycle_k002 (flushed)"); break;
    case P_ZERO: zeroPlain(p.level); break;
    case P_ZERO_HALF0: zeroPlain(p.level); smoothColour(p.level, 0, "mgCycle colour 0 (flushed)"); break;
    case P_PROL: prolongPlain(p.level); break;
    case P_PROL_HALF0: prolongPlain(p.level); smoothColour(p.level, 0, "mgCycle colour 0 (flushed)"); break;
  }
}

// the second array of a level gets the boundary values of the first once (the sweeps write inner points only)
void ensureAltShell(int L) {
  const int i = L - EXA_MIN_LEVEL;
  if (g_bcDone[i][1]) return;
  applyBC(&lv(L).withComm, g_solutionAlt[i], L, bcOfSolution(L), "applyBCsSolution (second array)");
  g_bcDone[i][1] = true;
}
void swapSolution(int L) {
  const int i = L - EXA_MIN_LEVEL;
  std::swap(fieldDeviceData_Solution[i], g_solutionAlt[i]);
  std::swap(g_bcDone[i][0], g_bcDone[i][1]);
}

// colour loops of the smoother: k000 / k006 (colour 0) and k001 / k007 (colour 1)
void colour0(int L, const char *what) {
  if (!g_deferred || !g_onePass[L - EXA_MIN_LEVEL]) { flushPending(); smoothColour(L, 0, what); return; }
  if (g_pending.kind == P_ZERO && g_pending.level == L) { g_pending.kind = P_ZERO_HALF0; return; }
  if (g_pending.kind == P_PROL && g_pending.level == L) { g_pending.kind = P_PROL_HALF0; return; }
  flushPending();
  g_pending.kind = P_HALF0;
  g_pending.level = L;
}
void colour1(int L, const char *what) {
  const int i = L - EXA_MIN_LEVEL;
  LevelInfo &v = lv(L);
  if (g_deferred && g_onePass[i] && g_pending.level == L &&
      (g_pending.kind == P_HALF0 || g_pending.kind == P_ZERO_HALF0 || g_pending.kind == P_PROL_HALF0)) {
    const PendingKind k = g_pending.kind;
    g_pending.kind = P_NONE;
    ensureAltShell(L);
    ++g_launches;
    if (k == P_HALF0)
      check(examg_rbgs_sweep_fused(&v.withComm, fieldDeviceData_Solution[i], g_solutionAlt[i], &v.noGhost, fieldDeviceData_RHS[i], &v.Laplace,
                                   smootherWeight(L), 0, v.begin, v.end, nullptr), what);
    else if (k == P_ZERO_HALF0)
      check(examg_rbgs_sweep_fused_zero(&v.withComm, g_solutionAlt[i], &v.noGhost, fieldDeviceData_RHS[i], &v.Laplace, smootherWeight(L), 0, v.begin,
                                        v.end, nullptr), what);
    else
      check(examg_rbgs_sweep_fused_prolong(&v.withComm, fieldDeviceData_Solution[i], g_solutionAlt[i], &v.noGhost, fieldDeviceData_RHS[i], &v.Laplace,
                                           smootherWeight(L), 0, v.begin, v.end, &lv(L - 1).withComm, fieldDeviceData_Solution[i - 1], nullptr), what);
    swapSolution(L);
    return;
  }
  flushPending();
  smoothColour(L, 1, what);
}
void residualDeferred(int L, const char *what) {            // k002
  flushPending();
  if (!g_deferred || !g_onePass[L - EXA_MIN_LEVEL]) { residual(L, what); return; }
  g_pending.kind = P_RES;
  g_pending.level = L;
}
void restrictDeferred(int L) {                              // k003
  const int i = L - EXA_MIN_LEVEL;
  if (g_pending.kind == P_RES && g_pending.level == L) {
    g_pending.kind = P_NONE;
    LevelInfo &v = lv(L);
    ++g_launches;
    check(examg_residual_restrict(&v.withComm, fieldDeviceData_Solution[i], &v.noGhost, fieldDeviceData_RHS[i], v.resLayout, fieldDeviceData_Residual[i],
                                  &v.Laplace, &lv(L - 1).noGhost, fieldDeviceData_RHS[i - 1], 1.0, v.begin, v.end, lv(L - 1).begin, lv(L - 1).end, nullptr),
          "mgCycle_k002 + k003");
    g_residualNotStored[i] = true;
    return;
  }
  flushPending();
  restrictPlain(L);
}
void zeroDeferred(int L) {                                  // k004 of level L: Solution@(L-1) = 0
  flushPending();
  const int LM1 = L - 1;
  // the first sweep of the coarser level takes the zero field as a constant -- if that level starts with a one-pass sweep at all and
  // its boundary values are in place (they are zero there, and the zero-start kernel does not write the shell)
  if (g_deferred && LM1 > EXA_MIN_LEVEL && g_onePass[LM1 - EXA_MIN_LEVEL] && g_bcDone[LM1 - EXA_MIN_LEVEL][0]) {
    g_pending.kind = P_ZERO;
    g_pending.level = LM1;
    return;
  }
  zeroPlain(LM1);
}
void prolongDeferred(int L) {                               // k005
  flushPending();
  if (g_deferred && g_onePass[L - EXA_MIN_LEVEL] && g_bcDone[L - EXA_MIN_LEVEL][0]) {
    g_pending.kind = P_PROL;
    g_pending.level = L;
    return;
  }
  prolongPlain(L);
}
void applyBCSolution(int L) {
  const int i = L - EXA_MIN_LEVEL;
  // deferred mode: position-only values on planes that no loop writes -- applied once per array, a repeat would re-write the same bits
  if (g_deferred && g_bcDone[i][0]) return;
  if (g_pending.kind != P_NONE && g_pending.level != L) flushPending();
  applyBC(&lv(L).withComm, fieldDeviceData_Solution[i], L, bcOfSolution(L), "applyBCsSolution");
  g_bcDone[i][0] = true;
}

const examg_expr_t *bcOfSolution(int level) { return level == EXA_MAX_LEVEL ? &g_bcSolution : &g_bcZero; }

}  // namespace

extern "C" {

long exa_shim_launches(void) { return g_launches; }
void exa_shim_flush(void) { flushPending(); }

void initGlobals(const int numBlocks[3], int mpiRank, const void *commId) {
  g_size = numBlocks[0] * numBlocks[1] * numBlocks[2];
  g_rank = mpiRank;
  for (int d = 0; d < 3; ++d) g_blocks[d] = numBlocks[d];
  g_pos[0] = mpiRank % numBlocks[0];                       // domain/ir/IR_ConnectFragments.scala:46-52: x fastest
  g_pos[1] = (mpiRank / numBlocks[0]) % numBlocks[1];
  g_pos[2] = mpiRank / (numBlocks[0] * numBlocks[1]);
  const char *defer = std::getenv("EXA_DEFERRED_LAUNCH");
  g_deferred = defer && defer[0] == '1' && g_size == 1;     // blocks with neighbours: the plain wrappers (their exchanges are not empty)
  g_faceMask = 0;
  for (int d = 0; d < 3; ++d) {
    int q[3] = {g_pos[0], g_pos[1], g_pos[2]};
    for (int s = 0; s < 2; ++s) {
      q[d] = g_pos[d] + (s ? 1 : -1);
      const bool valid = q[d] >= 0 && q[d] < numBlocks[d];  // neighbor_isValid
      g_nb.rank[d][s] = valid ? q[0] + numBlocks[0] * (q[1] + numBlocks[1] * q[2]) : -1;
      if (!valid) g_faceMask |= 1u << (2 * d + s);
      q[d] = g_pos[d];
    }
  }
  const char *peerBase = std::getenv("EXA_PEER_HANDLE_BASE");
  if (g_size > 1 && !commId && peerBase) {
    // peer-write transport (HIP IPC; also several ranks on ONE device): create, allocate the slabs for the finest level's faces,
    // all-gather the handles -- MPI_Allgather in a generated program, files <base>.<rank> here -- and map the neighbours' regions
    check(examg_comm_create_peer(&g_comm, g_size, mpiRank), "examg_comm_create_peer");
    const examg_layout_t finest = nodeLayout(EXA_MAX_LEVEL, 1);
    unsigned char mine[EXAMG_PEER_HANDLE_BYTES];
    check(examg_comm_peer_alloc(g_comm, examg_exchange_workspace_bytes(&finest) / 4, 0, mine), "examg_comm_peer_alloc");
    // A file is <pid of its writer><handle>: a reader only accepts the file of a LIVE process, so what an earlier run with the same base
    // left behind is never mapped (it waits for the new file instead), and every rank removes its own file when it is done.
    g_peerFile = std::string(peerBase) + "." + std::to_string(mpiRank);
    {
      const std::string tmp = g_peerFile + ".tmp";
      const long long pid = (long long)getpid();
      std::remove(g_peerFile.c_str());
      FILE *f = std::fopen(tmp.c_str(), "wb");
      if (!f || std::fwrite(&pid, sizeof(pid), 1, f) != 1 || std::fwrite(mine, 1, sizeof(mine), f) != sizeof(mine)) { std::fprintf(stderr, "cannot write %s\n", tmp.c_str()); std::exit(1); }
      std::fclose(f);
      std::rename(tmp.c_str(), g_peerFile.c_str());
    }
    std::vector<unsigned char> all((size_t)g_size * EXAMG_PEER_HANDLE_BYTES);
    for (int r = 0; r < g_size; ++r) {
      const std::string path = std::string(peerBase) + "." + std::to_string(r);
      bool got = false;
      for (int tries = 0; tries < 600 && !got; ++tries) {
        if (FILE *f = std::fopen(path.c_str(), "rb")) {
          long long pid = 0;
          got = std::fread(&pid, sizeof(pid), 1, f) == 1 &&
                std::fread(all.data() + (size_t)r * EXAMG_PEER_HANDLE_BYTES, 1, EXAMG_PEER_HANDLE_BYTES, f) == EXAMG_PEER_HANDLE_BYTES &&
                pid > 0 && (kill((pid_t)pid, 0) == 0 || errno == EPERM);
          std::fclose(f);
        }
        if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
      }
      if (!got) { std::fprintf(stderr, "no handle of a live rank %d in %s\n", r, path.c_str()); std::exit(1); }
    }
    check(examg_comm_peer_connect(g_comm, all.data()), "examg_comm_peer_connect");
  } else {
    check(examg_comm_create(&g_comm, commId, g_size, mpiRank), "examg_comm_create");
  }
  // boundary expressions as postfix programs (the generator inlines them into the boundary kernels)
  std::memset(&g_bcSolution, 0, sizeof(g_bcSolution));
  std::memset(&g_bcZero, 0, sizeof(g_bcZero));
  // ( ( vf_boundaryCoord_x ** 2 ) - ( 0.5 * ( vf_boundaryCoord_y ** 2 ) ) ) - ( 0.5 * ( vf_boundaryCoord_z ** 2 ) ), folded: x*x - 0.5*y*y - 0.5*z*z
  const int ops[] = {EXAMG_OP_X, EXAMG_OP_X, EXAMG_OP_MUL, EXAMG_OP_CONST, EXAMG_OP_Y, EXAMG_OP_MUL, EXAMG_OP_Y, EXAMG_OP_MUL, EXAMG_OP_SUB,
                     EXAMG_OP_CONST, EXAMG_OP_Z, EXAMG_OP_MUL, EXAMG_OP_Z, EXAMG_OP_MUL, EXAMG_OP_SUB};
  g_bcSolution.n = (int)(sizeof(ops) / sizeof(ops[0]));
  for (int i = 0; i < g_bcSolution.n; ++i) {
    g_bcSolution.op[i] = ops[i];
    g_bcSolution.c[i] = ops[i] == EXAMG_OP_CONST ? 0.5 : 0.0;
  }
  g_bcZero.n = 1;
  g_bcZero.op[0] = EXAMG_OP_CONST;
  g_bcZero.c[0] = 0.0;
}

void setupBuffers(void) {
  for (int level = EXA_MIN_LEVEL; level <= EXA_MAX_LEVEL; ++level) {
    LevelInfo &v = lv(level);
    const int i = level - EXA_MIN_LEVEL;
    v.withComm = nodeLayout(level, 1);
    v.noGhost = nodeLayout(level, 0);
    v.resLayout = (level == EXA_MIN_LEVEL) ? &v.noGhost : &v.withComm;
    fieldDeviceData_Solution[i] = deviceZeros(layoutSize(v.withComm));
    fieldDeviceData_RHS[i] = deviceZeros(layoutSize(v.noGhost));
    fieldDeviceData_Residual[i] = deviceZeros(layoutSize(*v.resLayout));
    for (int d = 0; d < 3; ++d) {
      const double width = 1.0 / g_blocks[d];                                   // unit cube, equal blocks
      v.geom.h[d] = width / (1 << level);                                       // domain/ir/IR_DomainFromAABB.scala:31-40
      v.geom.pos_begin[d] = g_pos[d] * width;
      v.begin[d] = 0 + (g_nb.rank[d][0] < 0 ? 1 : 0);                           // iterationOffsetBegin
      v.end[d] = (1 << level) + 1 + (g_nb.rank[d][1] < 0 ? -1 : 0);             // iterationOffsetEnd
      v.rbegin[d] = v.begin[d] > 1 ? v.begin[d] : 1;
    }
    examg_stencil_t &A = v.Laplace;
    std::memset(&A, 0, sizeof(A));
    A.nent = 7;
    A.diag = 0;
    const int off[7][3] = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
    for (int k = 0; k < 7; ++k) for (int d = 0; d < 3; ++d) A.off[k][d] = off[k][d];
    // 2.0 / ( vf_gridWidth_x ** 2 ) + 2.0 / ( vf_gridWidth_y ** 2 ) + 2.0 / ( vf_gridWidth_z ** 2 ) ; -1.0 / ( vf_gridWidth_d ** 2 )
    A.coef[0] = 2.0 / std::pow(v.geom.h[0], 2) + 2.0 / std::pow(v.geom.h[1], 2) + 2.0 / std::pow(v.geom.h[2], 2);
    for (int k = 1; k < 7; ++k) A.coef[k] = -1.0 / std::pow(v.geom.h[(k - 1) / 2], 2);
    v.wsWithBytes = examg_exchange_workspace_bytes(&v.withComm);
    v.wsNoBytes = examg_exchange_workspace_bytes(&v.noGhost);
    v.wsWith = deviceZeros(v.wsWithBytes / 8 + 1);
    v.wsNo = deviceZeros(v.wsNoBytes / 8 + 1);
    g_solutionAlt[i] = nullptr;
    g_onePass[i] = false;
    g_residualNotStored[i] = false;
    g_bcDone[i][0] = g_bcDone[i][1] = g_bcResDone[i] = false;
    if (g_deferred && level > EXA_MIN_LEVEL) {
      const examg_layout_t coarse = nodeLayout(level - 1, 0);
      int32_t cb[3], ce[3];
      for (int d = 0; d < 3; ++d) { cb[d] = 1; ce[d] = 1 << (level - 1); }
      g_onePass[i] = examg_two_stage_eligible(&v.withComm, &v.noGhost, &A, v.begin, v.end, v.begin, v.end) == 1 &&
                     examg_residual_restrict_one_pass(&v.withComm, &v.noGhost, &A, &coarse, v.begin, v.end, cb, ce) == 1;
      if (g_onePass[i]) g_solutionAlt[i] = deviceZeros(layoutSize(v.withComm));
    }
  }
  fieldDeviceData_cgTmp0[0] = deviceZeros(layoutSize(lv(EXA_MIN_LEVEL).withComm));
  fieldDeviceData_cgTmp1[0] = deviceZeros(layoutSize(lv(EXA_MIN_LEVEL).noGhost));
  g_scalar = deviceZeros(1);
  checkHip(hipMalloc(&g_work, examg_reduce_work_bytes()), "hipMalloc");
}

void destroyGlobals(void) {
  flushPending();
  checkHip(hipDeviceSynchronize(), "hipDeviceSynchronize");
  for (int i = 0; i < EXA_NUM_LEVELS; ++i) {
    (void)hipFree(fieldDeviceData_Solution[i]);
    if (g_solutionAlt[i]) (void)hipFree(g_solutionAlt[i]);
    (void)hipFree(fieldDeviceData_RHS[i]);
    (void)hipFree(fieldDeviceData_Residual[i]);
    (void)hipFree(g_lv[i].wsWith);
    (void)hipFree(g_lv[i].wsNo);
  }
  (void)hipFree(fieldDeviceData_cgTmp0[0]);
  (void)hipFree(fieldDeviceData_cgTmp1[0]);
  (void)hipFree(g_scalar);
  (void)hipFree(g_work);
  check(examg_comm_destroy(g_comm), "examg_comm_destroy");
  if (!g_peerFile.empty()) std::remove(g_peerFile.c_str());
}

void exa_allreduce_sum(double *x) {
  if (g_size == 1) return;
  checkHip(hipMemcpy(g_scalar, x, sizeof(double), hipMemcpyHostToDevice), "exa_allreduce_sum");
  check(examg_allreduce(g_comm, g_scalar, 1, 0, nullptr), "examg_allreduce");
  checkHip(hipMemcpy(x, g_scalar, sizeof(double), hipMemcpyDeviceToHost), "exa_allreduce_sum");
  check(examg_comm_status(g_comm, nullptr), "examg_comm_status");   // the host has waited for the device here anyway: a wait that gave up ends the run
}

#define EXA_I(L) ((L) - EXA_MIN_LEVEL)

#define EXA_KERNELS_COMMON(L)                                                                                                        \
  void EXA_CAT3(exchSolution_, L, )(int) {                                                                                           \
    if (g_size > 1) flushPending();                                                                                                  \
    exchange(&lv(L).withComm, fieldDeviceData_Solution[EXA_I(L)], lv(L).wsWith, lv(L).wsWithBytes, EXAMG_EXCH_ALL, "exchSolution"); \
  }                                                                                                                                  \
  void EXA_CAT3(exchResidual_, L, )(int) {                                                                                           \
    if (g_size > 1) flushPending();                                                                                                  \
    const bool g = lv(L).resLayout == &lv(L).withComm;                                                                               \
    exchange(lv(L).resLayout, fieldDeviceData_Residual[EXA_I(L)], g ? lv(L).wsWith : lv(L).wsNo, g ? lv(L).wsWithBytes : lv(L).wsNoBytes, \
             g ? EXAMG_EXCH_ALL : EXAMG_EXCH_DUP, "exchResidual");                                                                   \
  }                                                                                                                                  \
  void EXA_CAT3(applyBCsSolution_, L, )(int) { applyBCSolution(L); }                                                                 \
  void EXA_CAT3(applyBCsResidual_, L, )(int) {                                                                                       \
    /* zero on the boundary planes, which no loop writes: once per array in deferred mode; passes a pending residual loop */        \
    if (g_deferred && g_bcResDone[EXA_I(L)]) return;                                                                                 \
    if (g_pending.kind != P_NONE && !(g_pending.kind == P_RES && g_pending.level == L)) flushPending();                              \
    applyBC(lv(L).resLayout, fieldDeviceData_Residual[EXA_I(L)], L, &g_bcZero, "applyBCsResidual");                                  \
    g_bcResDone[EXA_I(L)] = true;                                                                                                    \
  }                                                                                                                                  \
  void EXA_CAT3(ResNorm_, L, _k000_wrapper)(double *reductionTmp) {                                                                  \
    flushPending();                                                                                                                  \
    if (g_residualNotStored[EXA_I(L)]) {                                                                                             \
      std::fprintf(stderr, "ResNorm_%d: Residual@%d was consumed by the one-pass residual + restriction and never stored (EXA_DEFERRED_LAUNCH)\n", L, L); \
      std::exit(1);                                                                                                                  \
    }                                                                                                                                \
    reduceDot(lv(L).resLayout, fieldDeviceData_Residual[EXA_I(L)], lv(L).resLayout, fieldDeviceData_Residual[EXA_I(L)], L, reductionTmp, "ResNorm_k000"); \
  }

#define EXA_KERNELS_FINE(L, LM1)                                                                                                     \
  void EXA_CAT3(mgCycle_, L, _k000_wrapper)(void) { colour0(L, "mgCycle_k000"); }                                                    \
  void EXA_CAT3(mgCycle_, L, _k001_wrapper)(void) { colour1(L, "mgCycle_k001"); }                                                    \
  void EXA_CAT3(mgCycle_, L, _k002_wrapper)(void) { residualDeferred(L, "mgCycle_k002"); }                                           \
  void EXA_CAT3(mgCycle_, L, _k003_wrapper)(void) { restrictDeferred(L); }                                                           \
  void EXA_CAT3(mgCycle_, L, _k004_wrapper)(void) { zeroDeferred(L); }                                                               \
  void EXA_CAT3(mgCycle_, L, _k005_wrapper)(void) { prolongDeferred(L); }                                                            \
  void EXA_CAT3(mgCycle_, L, _k006_wrapper)(void) { colour0(L, "mgCycle_k006"); }                                                    \
  void EXA_CAT3(mgCycle_, L, _k007_wrapper)(void) { colour1(L, "mgCycle_k007"); }

#define EXA_KERNELS_COARSEST(L)                                                                                                      \
  void EXA_CAT3(exchcgTmp0_, L, )(int) { flushPending(); exchange(&lv(L).withComm, fieldDeviceData_cgTmp0[0], lv(L).wsWith, lv(L).wsWithBytes, EXAMG_EXCH_ALL, "exchcgTmp0"); } \
  void EXA_CAT3(applyBCscgTmp0_, L, )(int) { flushPending(); applyBC(&lv(L).withComm, fieldDeviceData_cgTmp0[0], L, &g_bcZero, "applyBCscgTmp0"); }  \
  void EXA_CAT3(mgCycle_, L, _k000_wrapper)(void) { flushPending(); residual(L, "mgCycle_k000"); }                                   \
  void EXA_CAT3(mgCycle_, L, _k001_wrapper)(void) {                                                                                  \
    ++g_launches;                                                                                                                    \
    check(examg_axpby(lv(L).resLayout, fieldDeviceData_Residual[EXA_I(L)], &lv(L).withComm, fieldDeviceData_cgTmp0[0], 1.0, 0.0, lv(L).begin, \
                      lv(L).end, nullptr), "mgCycle_k001");                                                                          \
  }                                                                                                                                  \
  void EXA_CAT3(mgCycle_, L, _k002_wrapper)(void) {                                                                                  \
    ++g_launches;                                                                                                                    \
    check(examg_stencil_op(EXAMG_APPLY, &lv(L).withComm, fieldDeviceData_cgTmp0[0], nullptr, nullptr, &lv(L).noGhost, fieldDeviceData_cgTmp1[0], \
                           &lv(L).Laplace, 0.0, -1, lv(L).begin, lv(L).end, nullptr), "mgCycle_k002");                                \
  }                                                                                                                                  \
  void EXA_CAT3(mgCycle_, L, _k003_wrapper)(double *reductionTmp) {                                                                  \
    reduceDot(lv(L).resLayout, fieldDeviceData_Residual[EXA_I(L)], lv(L).resLayout, fieldDeviceData_Residual[EXA_I(L)], L, reductionTmp, "mgCycle_k003"); \
  }                                                                                                                                  \
  void EXA_CAT3(mgCycle_, L, _k004_wrapper)(double *reductionTmp) {                                                                  \
    reduceDot(&lv(L).withComm, fieldDeviceData_cgTmp0[0], &lv(L).noGhost, fieldDeviceData_cgTmp1[0], L, reductionTmp, "mgCycle_k004"); \
  }                                                                                                                                  \
  void EXA_CAT3(mgCycle_, L, _k005_wrapper)(double alpha) { ++g_launches;                                                            \
    check(examg_axpby(&lv(L).withComm, fieldDeviceData_cgTmp0[0], &lv(L).withComm, fieldDeviceData_Solution[EXA_I(L)], alpha, 1.0, lv(L).begin, \
                      lv(L).end, nullptr), "mgCycle_k005");                                                                          \
  }                                                                                                                                  \
  void EXA_CAT3(mgCycle_, L, _k006_wrapper)(double alpha) { ++g_launches;                                                            \
    check(examg_axpby(&lv(L).noGhost, fieldDeviceData_cgTmp1[0], lv(L).resLayout, fieldDeviceData_Residual[EXA_I(L)], -alpha, 1.0, lv(L).begin, \
                      lv(L).end, nullptr), "mgCycle_k006");                                                                          \
  }                                                                                                                                  \
  void EXA_CAT3(mgCycle_, L, _k007_wrapper)(double beta) { ++g_launches;                                                             \
    check(examg_axpby(lv(L).resLayout, fieldDeviceData_Residual[EXA_I(L)], &lv(L).withComm, fieldDeviceData_cgTmp0[0], 1.0, beta, lv(L).begin, \
                      lv(L).end, nullptr), "mgCycle_k007");                                                                          \
  }

#define EXA_LEVEL_KERNELS
#include "exa_levels.inc"
#undef EXA_LEVEL_KERNELS

void EXA_CAT3(Solve_, EXA_MAX_LEVEL, _k000_wrapper)(void) { flushPending(); residual(EXA_MAX_LEVEL, "Solve_k000"); }
void EXA_CAT3(Solve_, EXA_MAX_LEVEL, _k001_wrapper)(void) { flushPending(); residual(EXA_MAX_LEVEL, "Solve_k001"); }

}  // extern "C"
