// The host side of Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4 as the generator emits it (one C function per leveled
// ExaSlang function, User/User_<fn>_<L>.cpp, base/ir/IR_Function.scala:123-131), calling ONLY reference-named entry points:
// <fn>_<L>_k<NNN>_wrapper(), exch<Field>_<L>(slot), applyBCs<Field>_<L>(slot) -- declared in exa_poisson3d.h, bodies in
// exa_poisson3d_kernels.cpp (libexamg).  Nothing in this file knows about libexamg.
//
//   hipcc -O2 -DEXA_MIN_LEVEL=2 -DEXA_MAX_LEVEL=6 -Iinclude shim/exa_poisson3d_host.cpp shim/exa_poisson3d_kernels.cpp \
//         -Lexastencils_amd -lexamg -o shim/exa_poisson3d_2_6
//   shim/exa_poisson3d_2_6                                    one block
//   shim/exa_poisson3d_2_6 <bx> <by> <bz> <rank> <idfile>     one process per block / GPU; rank 0 writes the 128-byte RCCL id
//                                                             to <idfile> (the generated program would MPI_Bcast it)
//
// Prints the residual norm per V-cycle the way the generated program does with testing_enabled (4 significant digits), then
// the full-precision values (lines starting with '#') that the tests compare with the oracle.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "exa_poisson3d.h"

extern "C" {

// Function ResNorm@(coarsest and finest) : Real  (...exa4:113-119)
#define EXA_HOST_COMMON(L)                                   \
  double EXA_CAT3(ResNorm_, L, )(void) {                     \
    double norm = 0.0;                                       \
    EXA_CAT3(ResNorm_, L, _k000_wrapper)(&norm);             \
    exa_allreduce_sum(&norm);                                \
    return std::sqrt(norm);                                  \
  }

// Function mgCycle@coarsest  (...exa4:152-201)
#define EXA_HOST_COARSEST(L)                                                           \
  void EXA_CAT3(mgCycle_, L, )(void) {                                                 \
    EXA_CAT3(exchSolution_, L, )(0);                                                   \
    EXA_CAT3(mgCycle_, L, _k000_wrapper)();                                            \
    EXA_CAT3(applyBCsResidual_, L, )(0);                                               \
    double curRes = EXA_CAT3(ResNorm_, L, )();                                         \
    const double initRes = curRes;                                                     \
    EXA_CAT3(mgCycle_, L, _k001_wrapper)();                                            \
    EXA_CAT3(applyBCscgTmp0_, L, )(0);                                                 \
    for (int curStep = 0; curStep < 128; ++curStep) {                                  \
      EXA_CAT3(exchcgTmp0_, L, )(0);                                                   \
      EXA_CAT3(mgCycle_, L, _k002_wrapper)();                                          \
      double reductionVar_2 = 0.0;                                                     \
      EXA_CAT3(mgCycle_, L, _k003_wrapper)(&reductionVar_2);                           \
      exa_allreduce_sum(&reductionVar_2);                                              \
      const double alphaNom_Solution = reductionVar_2;                                 \
      double reductionVar_3 = 0.0;                                                     \
      EXA_CAT3(mgCycle_, L, _k004_wrapper)(&reductionVar_3);                           \
      exa_allreduce_sum(&reductionVar_3);                                              \
      const double alphaDenom_Solution = reductionVar_3;                               \
      const double alpha = alphaNom_Solution / alphaDenom_Solution;                    \
      EXA_CAT3(mgCycle_, L, _k005_wrapper)(alpha);                                     \
      EXA_CAT3(applyBCsSolution_, L, )(0);                                             \
      EXA_CAT3(mgCycle_, L, _k006_wrapper)(alpha);                                     \
      EXA_CAT3(applyBCsResidual_, L, )(0);                                             \
      const double nextRes = EXA_CAT3(ResNorm_, L, )();                                \
      if (nextRes <= 0.001 * initRes) return;                                          \
      const double beta = (nextRes * nextRes) / (curRes * curRes);                     \
      EXA_CAT3(mgCycle_, L, _k007_wrapper)(beta);                                      \
      EXA_CAT3(applyBCscgTmp0_, L, )(0);                                               \
      curRes = nextRes;                                                                \
    }                                                                                  \
    std::cout << "Maximum number of cgs iterations (" << 128 << ") was exceeded" << std::endl; \
  }

// Function mgCycle@(all but coarsest)  (...exa4:203-249)
#define EXA_HOST_FINE(L, LM1)                                \
  void EXA_CAT3(mgCycle_, L, )(void) {                       \
    for (int i = 0; i < 3; ++i) {                            \
      EXA_CAT3(exchSolution_, L, )(0);                       \
      EXA_CAT3(mgCycle_, L, _k000_wrapper)();                \
      EXA_CAT3(applyBCsSolution_, L, )(0);                   \
      EXA_CAT3(exchSolution_, L, )(0);                       \
      EXA_CAT3(mgCycle_, L, _k001_wrapper)();                \
      EXA_CAT3(applyBCsSolution_, L, )(0);                   \
    }                                                        \
    EXA_CAT3(exchSolution_, L, )(0);                         \
    EXA_CAT3(mgCycle_, L, _k002_wrapper)();                  \
    EXA_CAT3(applyBCsResidual_, L, )(0);                     \
    EXA_CAT3(exchResidual_, L, )(0);                         \
    EXA_CAT3(mgCycle_, L, _k003_wrapper)();                  \
    EXA_CAT3(mgCycle_, L, _k004_wrapper)();                  \
    EXA_CAT3(applyBCsSolution_, LM1, )(0);                   \
    EXA_CAT3(mgCycle_, LM1, )();                             \
    EXA_CAT3(exchSolution_, LM1, )(0);                       \
    EXA_CAT3(mgCycle_, L, _k005_wrapper)();                  \
    EXA_CAT3(applyBCsSolution_, L, )(0);                     \
    for (int i = 0; i < 3; ++i) {                            \
      EXA_CAT3(exchSolution_, L, )(0);                       \
      EXA_CAT3(mgCycle_, L, _k006_wrapper)();                \
      EXA_CAT3(applyBCsSolution_, L, )(0);                   \
      EXA_CAT3(exchSolution_, L, )(0);                       \
      EXA_CAT3(mgCycle_, L, _k007_wrapper)();                \
      EXA_CAT3(applyBCsSolution_, L, )(0);                   \
    }                                                        \
  }

#define EXA_LEVEL_HOST
#include "exa_levels.inc"
#undef EXA_LEVEL_HOST

}  // extern "C"

static std::vector<double> g_history;
static int g_iterations = 0;

// Function Solve@finest  (...exa4:121-150)
static void Solve(void) {
  EXA_CAT3(exchSolution_, EXA_MAX_LEVEL, )(0);
  EXA_CAT3(Solve_, EXA_MAX_LEVEL, _k000_wrapper)();
  EXA_CAT3(applyBCsResidual_, EXA_MAX_LEVEL, )(0);
  const double initRes = EXA_CAT3(ResNorm_, EXA_MAX_LEVEL, )();
  double curRes = initRes;
  g_history.push_back(initRes);
  std::cout << initRes << std::endl;
  int curIt = 0;
  while (!(curIt >= 100 || curRes <= 1.0E-6 * initRes)) {
    curIt += 1;
    EXA_CAT3(mgCycle_, EXA_MAX_LEVEL, )();
    EXA_CAT3(exchSolution_, EXA_MAX_LEVEL, )(0);
    EXA_CAT3(Solve_, EXA_MAX_LEVEL, _k001_wrapper)();
    EXA_CAT3(applyBCsResidual_, EXA_MAX_LEVEL, )(0);
    curRes = EXA_CAT3(ResNorm_, EXA_MAX_LEVEL, )();
    g_history.push_back(curRes);
    std::cout << curRes << std::endl;
  }
  g_iterations = curIt;
}

int main(int argc, char **argv) {
  int blocks[3] = {1, 1, 1}, rank = 0;
  unsigned char id[EXAMG_COMM_ID_BYTES];
  const void *idp = nullptr;
  if (argc >= 6) {
    for (int d = 0; d < 3; ++d) blocks[d] = std::atoi(argv[1 + d]);
    rank = std::atoi(argv[4]);
    const int nranks = blocks[0] * blocks[1] * blocks[2];
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "no HIP device\n"); return 2; }
    if (hipSetDevice(rank % ndev) != hipSuccess) { std::fprintf(stderr, "hipSetDevice failed\n"); return 2; }
    if (nranks > 1 && argc >= 7 && std::string(argv[6]) == "peer") {
      // peer-write transport: no id; initGlobals exchanges the IPC handles of the ranks through files <argv[5]>.<rank>
      setenv("EXA_PEER_HANDLE_BASE", argv[5], 1);
    } else if (nranks > 1) {   // MPI_Bcast of the communicator id in a generated program; a file here
      if (rank == 0) {
        if (examg_comm_unique_id(id)) { std::fprintf(stderr, "%s\n", examg_last_error()); return 1; }
        std::string tmp = std::string(argv[5]) + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id)) { std::fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 1; }
        std::fclose(f);
        std::rename(tmp.c_str(), argv[5]);
      } else {
        FILE *f = nullptr;
        for (int tries = 0; tries < 600 && !(f = std::fopen(argv[5], "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!f || std::fread(id, 1, sizeof(id), f) != sizeof(id)) { std::fprintf(stderr, "cannot read %s\n", argv[5]); return 1; }
        std::fclose(f);
      }
      idp = id;
    }
  }
  if (examg_device_count() < 1) { std::fprintf(stderr, "no HIP device\n"); return 2; }
  initGlobals(blocks, rank, idp);
  setupBuffers();
  EXA_CAT3(applyBCsSolution_, EXA_MAX_LEVEL, )(0);    // Function Application (...exa4:251-277): initial boundary values
  std::cout.precision(4);
  if (rank != 0) std::cout.setstate(std::ios_base::failbit);   // only the root prints (print statements are guarded by mpiRank == 0)
  Solve();
  if (rank == 0) {
    for (double r : g_history) std::printf("# %.17g\n", r);
    std::printf("iterations %d\n", g_iterations);
  }
  if (const char *tc = std::getenv("EXA_TIME_CYCLES")) {
    // the benchmark's timer around the cycle (benchmarkStart / benchmarkStop, ...exa4:263-276): n further cycles, device synchronised
    const int n = std::atoi(tc);
    EXA_CAT3(mgCycle_, EXA_MAX_LEVEL, )();
    (void)hipDeviceSynchronize();
    const long l0 = exa_shim_launches();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) EXA_CAT3(mgCycle_, EXA_MAX_LEVEL, )();
    (void)hipDeviceSynchronize();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (n > 0 ? n : 1);
    if (rank == 0) std::printf("vcycle_ms %.6f launches_per_cycle %ld\n", ms, n > 0 ? (exa_shim_launches() - l0) / n : 0);
  }
  destroyGlobals();
  return 0;
}
