/*
 * Reference-named boundary of Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4 on libexamg (MI355X).
 *
 * With cuda_enabled the ExaStencils generator prints, for every device-eligible `loop over` of a leveled function <fn>_<L>,
 *     extern "C" void <fn>_<L>_k<NNN>_wrapper(<pass-through locals>[, double* reductionTmp])
 * (Compiler/src/exastencils/parallelization/api/cuda/CUDA_Kernel.scala:546-632; numbering k000, k001, .. in the order the
 * loops appear in the function, CUDA_KernelFunctions.scala:78-82), which reads the process-global device arrays
 * fieldDeviceData_<Field>[level - minLevel] (cuda/CUDA_Memory.scala:121-141) itself, and per field and level
 *     void exch<Field>_<L>(int slot)        communication/ir/IR_CommunicateFunction.scala:473-480, IR_SetupCommunication.scala:119-147
 *     void applyBCs<Field>_<L>(int slot)    boundary/ir/IR_ApplyBCFunction.scala:53-104
 * This header declares exactly those names for the benchmark program; exa_poisson3d_kernels.cpp gives them libexamg bodies;
 * exa_poisson3d_host.cpp is the generated host side (User/User_mgCycle_<L>.cpp, ..) and calls nothing else.
 *
 * Knowledge is compile-time in generated code: -DEXA_MIN_LEVEL=.. -DEXA_MAX_LEVEL=.. (defaults 4 and 9, the CUDA benchmark's
 * 6 levels at 512^3, Benchmark/Poisson3D/3D_FD_Poisson_fromL4_CUDA.knowledge:5-6).
 *
 * Kernel numbering (order of the `loop over` statements in the program text, ...exa4:152-249):
 *   mgCycle_<L>, L > minLevel : k000 / k001 pre-smoothing colour 0 / 1 (:204-213); k002 Residual = RHS - Laplace * Solution (:216-218);
 *                               k003 RHS@coarser = NodeRestriction * Residual (:222-224); k004 Solution@coarser = 0 (:226-228);
 *                               k005 Solution += NodeProlongation@coarser * Solution@coarser (:234-236); k006 / k007 post-smoothing (:239-248)
 *   mgCycle_<minLevel>        : k000 residual (:154-156); k001 cgTmp0 = Residual (:160-162); k002 cgTmp1 = Laplace * cgTmp0 (:167-169);
 *                               k003 sum Residual^2 (:171-173); k004 sum cgTmp0 * cgTmp1 (:176-178); k005 Solution += alpha * cgTmp0 (:181-183);
 *                               k006 Residual -= alpha * cgTmp1 (:185-187); k007 cgTmp0 = Residual + beta * cgTmp0 (:194-196)
 *   ResNorm_<L>               : k000 sum Residual^2 (:115-117)
 *   Solve_<maxLevel>          : k000 / k001 residual before / inside the iteration (:123-125, :139-141)
 */
#ifndef EXA_POISSON3D_H
#define EXA_POISSON3D_H

#include "examg.h"

#ifndef EXA_MIN_LEVEL
#define EXA_MIN_LEVEL 4
#endif
#ifndef EXA_MAX_LEVEL
#define EXA_MAX_LEVEL 9
#endif
#define EXA_NUM_LEVELS (EXA_MAX_LEVEL - EXA_MIN_LEVEL + 1)

#define EXA_CAT3_(a, b, c) a##b##c
#define EXA_CAT3(a, b, c) EXA_CAT3_(a, b, c)

#ifdef __cplusplus
extern "C" {
#endif

/* Global/Global.h: one device array per field and level (one slot each in this program), reference layout */
extern double *fieldDeviceData_Solution[EXA_NUM_LEVELS];
extern double *fieldDeviceData_RHS[EXA_NUM_LEVELS];
extern double *fieldDeviceData_Residual[EXA_NUM_LEVELS];
extern double *fieldDeviceData_cgTmp0[1];
extern double *fieldDeviceData_cgTmp1[1];

/* initGlobals / setupBuffers / destroyGlobals (globals/ir/IR_AllocateDataFunction.scala:63-65) and the decomposition
 * (domain_rect_numBlocks_*, rank = mpiRank; `commId` = the 128-byte RCCL id of rank 0 when there is more than one block) */
void initGlobals(const int numBlocks[3], int mpiRank, const void *commId);
void setupBuffers(void);
void destroyGlobals(void);
/* MPI_Allreduce(MPI_IN_PLACE, x, 1, MPI_DOUBLE, MPI_SUM) on a host value, as the generated host code does after a reduction */
void exa_allreduce_sum(double *x);
/* Deferred-launch mode of the wrappers (environment EXA_DEFERRED_LAUNCH=1, one block; exa_poisson3d_kernels.cpp): a wrapper whose loop
 * a one-pass kernel of libexamg can absorb records its call, the wrapper that completes the pattern launches the one-pass kernel; any
 * other entry point runs what is recorded first.  exa_shim_flush() does that explicitly (before the host touches a field array by
 * other means); exa_shim_launches() counts the libexamg launches so far. */
void exa_shim_flush(void);
long exa_shim_launches(void);

#define EXA_DECLARE_FINE(L)                                   \
  void EXA_CAT3(mgCycle_, L, _k000_wrapper)(void);            \
  void EXA_CAT3(mgCycle_, L, _k001_wrapper)(void);            \
  void EXA_CAT3(mgCycle_, L, _k002_wrapper)(void);            \
  void EXA_CAT3(mgCycle_, L, _k003_wrapper)(void);            \
  void EXA_CAT3(mgCycle_, L, _k004_wrapper)(void);            \
  void EXA_CAT3(mgCycle_, L, _k005_wrapper)(void);            \
  void EXA_CAT3(mgCycle_, L, _k006_wrapper)(void);            \
  void EXA_CAT3(mgCycle_, L, _k007_wrapper)(void);
#define EXA_DECLARE_COARSEST(L)                                                  \
  void EXA_CAT3(mgCycle_, L, _k000_wrapper)(void);                               \
  void EXA_CAT3(mgCycle_, L, _k001_wrapper)(void);                               \
  void EXA_CAT3(mgCycle_, L, _k002_wrapper)(void);                               \
  void EXA_CAT3(mgCycle_, L, _k003_wrapper)(double *reductionTmp);               \
  void EXA_CAT3(mgCycle_, L, _k004_wrapper)(double *reductionTmp);               \
  void EXA_CAT3(mgCycle_, L, _k005_wrapper)(double alpha);                       \
  void EXA_CAT3(mgCycle_, L, _k006_wrapper)(double alpha);                       \
  void EXA_CAT3(mgCycle_, L, _k007_wrapper)(double beta);                        \
  void EXA_CAT3(exchcgTmp0_, L, )(int slot);                                     \
  void EXA_CAT3(applyBCscgTmp0_, L, )(int slot);
#define EXA_DECLARE_COMMON(L)                                 \
  void EXA_CAT3(exchSolution_, L, )(int slot);                \
  void EXA_CAT3(exchResidual_, L, )(int slot);                \
  void EXA_CAT3(applyBCsSolution_, L, )(int slot);            \
  void EXA_CAT3(applyBCsResidual_, L, )(int slot);            \
  void EXA_CAT3(ResNorm_, L, _k000_wrapper)(double *reductionTmp);

#define EXA_LEVEL_DECL
#include "exa_levels.inc"
#undef EXA_LEVEL_DECL

void EXA_CAT3(Solve_, EXA_MAX_LEVEL, _k000_wrapper)(void);
void EXA_CAT3(Solve_, EXA_MAX_LEVEL, _k001_wrapper)(void);

#ifdef __cplusplus
}
#endif
#endif
