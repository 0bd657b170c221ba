/*
 * examg.h -- C ABI of libexamg: the MI355X (gfx950) multigrid hot path that drops in where
 * ExaStencils' generated CUDA kernel wrappers sit.
 *
 * What this replaces.  For every device-eligible `loop over` the reference generator prints
 *     extern "C" void <fn>_k<NNN>_wrapper(<pass-through args>[, double* reductionTmp])
 * into Kernel/Kernel_<fn>_k<NNN>.cu (Compiler/src/exastencils/parallelization/api/cuda/
 * CUDA_Kernel.scala:546-632, naming CUDA_KernelFunctions.scala:78-110, call site
 * CUDA_ExtractDeviceCode.scala:262-280), called from generated host functions mgCycle_<lvl>(),
 * Solve_<lvl>() on process-global device arrays fieldDeviceData_<F>[lvl][slot]
 * (cuda/CUDA_Memory.scala:121-141) laid out as IR_FieldLayout prescribes
 * (field/ir/IR_FieldLayout.scala:30-129).  The <NNN> numbering is an artefact of strategy order,
 * so this header defines one *semantic* entry point per kind of emitted loop; INTEGRATION.md shows
 * the one-line `_wrapper` shims a maintainer adds to bind them.
 *
 * Conventions (same as the generated code):
 *   - all pointers are DEVICE pointers owned by the caller (setupBuffers()/destroyGlobals(),
 *     globals/ir/IR_AddInternalVariables.scala:115-182); nothing here allocates or frees;
 *   - fields are raw double arrays in the reference layout, x fastest
 *     (baseExt/ir/IR_Linearization.scala:27-36); `examg_layout_t` carries the per-dimension
 *     pad|ghost|dup|inner|dup|ghost|pad counts, referenceOffset = pad_l + ghost_l;
 *   - loop bounds `begin`/`end` are in iterator coordinates (0 = lower duplicate node), half-open,
 *     exactly the `_cu_begin_d`/`_cu_end_d` kernel arguments (cuda/CUDA_Kernel.scala:364-393) that
 *     baseExt/ir/IR_LoopOverPointsInOneFragment.scala:84-101 computes; unused dims use [0,1);
 *   - `stream` is a hipStream_t (cuda/CUDA_Stream.scala:96-202); launches are asynchronous and
 *     capturable into a hipGraph; no call synchronises the device;
 *   - return value: 0 on success, non-zero on error with a message in examg_last_error().
 *     (The reference wrappers are void and print+exit on error, cuda/CUDA_Error.scala; the
 *     reference-named shims in INTEGRATION.md reproduce that on top of these.)
 *   - arithmetic: fp64, each statement evaluated in the order the generator prints it
 *     (stencil entries folded left to right, stencil/ir/IR_StencilConvolution.scala:65-68),
 *     no FMA contraction -- point-wise results are bit-identical to the CPU path.
 */
#ifndef EXAMG_H
#define EXAMG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXAMG_MAX_ENTRIES 27

/* field/ir/IR_FieldLayout.scala:103-129 (IR_FieldLayoutPerDim). Unused dims: inner = 1, rest 0.
 * transform: the field under a layout transformation of the program's `LayoutTransformations` block (layoutTransformation/ir/
 * IR_LayoutTransformStatement.scala; Testing/LayoutTrafo/rbgs.exa4:2).  EXAMG_LAYOUT_SPLIT_X is the colour split
 * `transform <field> with [x, y, z] => [x / 2, y, z, x % 2]`: with ax = array index in x (iterator + referenceOffset) the value of a
 * point lives at  ax / 2 + H * (ay + TOTy * (az + TOTz * (ax % 2))),  H = ceil(TOTx / 2): the even and the odd columns of every row
 * are two contiguous half rows in two half arrays, so the points of one red-black colour of a row -- and their x neighbours -- are
 * contiguous (32 B per update for a half sweep instead of 48).  A transformation changes where a value lives, never a value; regions,
 * iterator coordinates and boxes are those of the untransformed layout.  Entry points that have no form for a transformed layout
 * refuse it (examg_last_error); examg_transform_field converts. */
enum { EXAMG_LAYOUT_PLAIN = 0, EXAMG_LAYOUT_SPLIT_X = 1 };
typedef struct examg_layout {
  int32_t nd;
  int32_t pad_l[3], ghost_l[3], dup_l[3], inner[3], dup_r[3], ghost_r[3], pad_r[3];
  int32_t transform; /* EXAMG_LAYOUT_* */
} examg_layout_t;

/* operator/ir/IR_Stencil.scala:34-211 (constant coefficients, entry order significant) or a
 * stencil field whose entry index is the slowest array dimension
 * (stencil/ir/IR_StencilConvolution.scala:73-95): cfield[k * size(clayout) + linear(clayout, i)].
 * ctransform: the coefficient field under a layout transformation (layoutTransformation/, the `LayoutTransformations` block of an
 * ExaSlang-4 program, Testing/LayoutTrafo/{rbgs,opts}.exa4) -- EXAMG_CLAYOUT_ENTRY_FASTEST is `transform <field> with [x, y, z, i] => [i, x, y, z]`:
 * the entries of a point are contiguous, cfield[linear(clayout, i) * nent + k]: ONE stream of 8 * nent bytes per point instead of
 * nent streams (27-entry fields: 30 -> 4 streams per sweep).  A transformation changes where values live, never a value. */
enum { EXAMG_CLAYOUT_PLANES = 0, EXAMG_CLAYOUT_ENTRY_FASTEST = 1 };
typedef struct examg_stencil {
  int32_t nent;
  int32_t diag; /* index of the (0,0,0) entry, `diag(A)` */
  int32_t off[EXAMG_MAX_ENTRIES][3];
  double coef[EXAMG_MAX_ENTRIES];
  const double *cfield; /* device pointer or NULL */
  examg_layout_t clayout;
  int32_t ctransform;   /* EXAMG_CLAYOUT_* */
  /* smoother weight of a stencil FIELD (EXAMG_SMOOTH with cfield): how the statement writes it.  EXAMG_WEIGHT_INV_TIMES:
   * `((1.0 / diag(A)) * omega)` (Testing/SISC/3D_VarCoeff.exa4:145) -- per point (1.0 / c_diag) * w;  EXAMG_WEIGHT_DIVIDE:
   * `(omega / diag(A))` (Testing/PolyExpl/RBGS3Dvc.exa4:52) -- per point w / c_diag.  Both round differently; the kernels evaluate
   * what the program says.  Constant stencils take the folded literal in `w` and ignore this. */
  int32_t wform;
} examg_stencil_t;
enum { EXAMG_WEIGHT_INV_TIMES = 0, EXAMG_WEIGHT_DIVIDE = 1 };

/* Uniform node grid of one fragment at one level: position = index * h + pos_begin
 * (grid/ir/IR_VF_NodePosition.scala:109-111, domain/ir/IR_DomainFromAABB.scala:31-40). */
typedef struct examg_geom {
  double pos_begin[3];
  double h[3];
} examg_geom_t;

typedef void *examg_stream_t; /* hipStream_t */

/* stencil loop kinds */
enum { EXAMG_APPLY = 0, EXAMG_RESIDUAL = 1, EXAMG_SMOOTH = 2 };

int examg_version(void);
const char *examg_last_error(void);
/* Number of visible HIP devices (cuda/CUDA_AddGlobals.scala:30-48 does cudaGetDeviceCount). */
int examg_device_count(void);

/* ---- K1/K2/K3: stencil loops ---------------------------------------------
 * mode EXAMG_APPLY    : dst = A*u                       (cgTmp1 = Laplace * cgTmp0, ...exa4:166-168)
 *      EXAMG_RESIDUAL : dst = rhs - A*u                 (Residual = RHS - Laplace * Solution, :215-219)
 *      EXAMG_SMOOTH   : dst = u + ww * (rhs - A*u)      (Jacobi, Testing/Smoothers/Jac.exa4:125-131;
 *                       with colour >= 0 and dst == u: one red-black half sweep, ...exa4:204-213)
 *   ww = w for constant stencils (w is the folded constant omega/diag(A));
 *   ww = (1.0 / cfield[diag]) * w for stencil fields (w = omega, Testing/SISC/3D_VarCoeff.exa4:145).
 * colour: -1 = all points, else only points with (i0+i1+i2) % 2 == colour
 *   (baseExt/l4/L4_ColorLoops.scala:44-66; condition emitted by IR_LoopOverDimensions.scala:215-216).
 * u and dst may alias only when colour >= 0 (or for disjoint boxes). */
int examg_stencil_op(int mode, const examg_layout_t *lu, const double *u, const examg_layout_t *lf, const double *rhs,
                     const examg_layout_t *ld, double *dst, const examg_stencil_t *st, double w, int colour,
                     const int32_t *begin, const int32_t *end, examg_stream_t stream);

/* Names of SURVEY.md 8b; thin forms of examg_stencil_op. */
int examg_jacobi(const examg_layout_t *lu, const double *u, double *u_next, const examg_layout_t *lf, const double *rhs,
                 const examg_stencil_t *st, double w, const int32_t *begin, const int32_t *end, examg_stream_t stream);
int examg_rbgs_colour(const examg_layout_t *lu, double *u, const examg_layout_t *lf, const double *rhs,
                      const examg_stencil_t *st, double w, int colour, const int32_t *begin, const int32_t *end,
                      examg_stream_t stream);
int examg_residual(const examg_layout_t *lu, const double *u, const examg_layout_t *lf, const double *rhs,
                   const examg_layout_t *lr, double *res, const examg_stencil_t *st, const int32_t *begin,
                   const int32_t *end, examg_stream_t stream);

/* One full red-black sweep (colour `first`, then the other) out of place, in ONE pass over HBM: the points of
 * [begin,end) of u_out receive exactly (bit for bit) what the two examg_rbgs_colour calls would leave there;
 * outside the box at most the one-stencil-reach shell is touched, and only by copying u_in's values there; the
 * caller keeps u_out's duplicate/ghost shell valid the way the program does anyway (`apply bc` / `communicate`
 * after the loop) and swaps the two pointers.  24 B per update instead of 48 B.  3-D 7-point constant stencils
 * take the two-stage kernel (writes the box only); anything else falls back to copy + two half sweeps. */
int examg_rbgs_sweep_fused(const examg_layout_t *lu, const double *u_in, double *u_out, const examg_layout_t *lf,
                           const double *rhs, const examg_stencil_t *st, double w, int first, const int32_t *begin,
                           const int32_t *end, examg_stream_t stream);

/* As examg_rbgs_sweep_fused with separate boxes, for blocks with neighbours: colour `first` on [begin1,end1) (points
 * outside keep u_in's value), then the other colour of that field on [begin2,end2), inside box 1; u_out receives both
 * colours on box 2 and is not touched elsewhere.  Box 1 = the loop's box shrunk by one point at interior faces, box 2 by
 * two: everything inside is independent of the halo exchange between the two half sweeps
 * (communicate inside `color with`, Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:204-213).  `tmp` is used by the fallback
 * path only (general stencils, short rows) and may be NULL otherwise. */
int examg_rbgs_sweep_fused_boxes(const examg_layout_t *lu, const double *u_in, double *u_out, double *tmp,
                                 const examg_layout_t *lf, const double *rhs, const examg_stencil_t *st, double w, int first,
                                 const int32_t *begin1, const int32_t *end1, const int32_t *begin2, const int32_t *end2,
                                 examg_stream_t stream);

/* Two Jacobi steps (Smoother called twice, Testing/Smoothers/Jac.exa4:125-131) in ONE pass: temporal blocking in the
 * sense of baseExt/ir/IR_ContractingLoop.scala.  u_out[box] = J(J(u_in)); bit-identical to two examg_jacobi calls
 * u_in -> tmp -> u_out.  Only valid when no halo exchange is needed between the two steps (single block, or ghost
 * layers two deep); `tmp` is used by the fallback path only (general stencils, small boxes) and may be NULL otherwise.
 * One-pass forms exist for the 3-D 7-point constant stencil and for 27-entry stencil fields in the record layout
 * (EXAMG_CLAYOUT_ENTRY_FASTEST, entry order of examg_init_helmholtz27: the two steps of a point share its 216 B of coefficients). */
int examg_jacobi2(const examg_layout_t *lu, const double *u_in, double *u_out, double *tmp, const examg_layout_t *lf,
                  const double *rhs, const examg_stencil_t *st, double w, const int32_t *begin, const int32_t *end,
                  examg_stream_t stream);

/* Three Jacobi steps in ONE pass (temporal blocking of depth 3: `repeat 5 times with contraction [1,1,1]` of
 * Testing/PolyExpl/Jac3Dcc.exa4:27 runs as 3 + 2; baseExt/ir/IR_ContractingLoop.scala:45-196): u_out[box] = J(J(J(u_in))), bit-identical
 * to three examg_jacobi calls.  Same conditions as examg_jacobi2 (no halo exchange needed in between); the one-pass form exists for the
 * 3-D 7-point constant stencil on rows of at least 64 points; otherwise a step into `tmp` and a pair from there (`tmp` must then be a
 * distinct array; NULL is allowed where the one-pass form applies). */
int examg_jacobi3(const examg_layout_t *lu, const double *u_in, double *u_out, double *tmp, const examg_layout_t *lf,
                  const double *rhs, const examg_stencil_t *st, double w, const int32_t *begin, const int32_t *end,
                  examg_stream_t stream);

/* Three colour loops of a red-black smoother in ONE pass, out of place: colour `first`, the other colour, `first` again -- three sweeps
 * (`repeat 3 times { color with { ... } }`, Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:204-213) are two such passes, the second with
 * first = 1 - first.  u_out[box] = the three loops applied to u_in; bit-identical to three examg_rbgs_colour calls in place.  One-pass
 * form: 3-D 7-point constant stencil, rows of at least 64 points (examg_three_stage_eligible); otherwise a copy and the three loops. */
int examg_rbgs_colours3(const examg_layout_t *lu, const double *u_in, double *u_out, const examg_layout_t *lf, const double *rhs,
                        const examg_stencil_t *st, double w, int first, const int32_t *begin, const int32_t *end, examg_stream_t stream);

/* 1 if examg_jacobi3 / examg_rbgs_colours3 run their one-pass kernel for this box, 0 if they run their loops one after the other. */
int examg_three_stage_eligible(const examg_layout_t *lu, const examg_layout_t *lf, const examg_stencil_t *st, const int32_t *begin,
                               const int32_t *end);

/* One Jacobi step on [begin,end) and the residual of its result in ONE pass: u_out = J(u_in), res = rhs - A u_out on the box (the
 * last pre-smoothing `Smoother@current` + `Residual@current = RHS - Laplace * Solution`, Testing/SISC/3D_VarCoeff.exa4:141-153,
 * Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:215-219).  27-entry stencil fields in the record layout (EXAMG_CLAYOUT_ENTRY_FASTEST,
 * entry order of examg_init_helmholtz27) share the 216 B of coefficients per point between the two loops; everything else runs
 * examg_jacobi, then examg_residual.  The residual reads the one-point shell of the box from u_in: bit-identical to those two calls
 * when u_out holds u_in's values there (the slots of a field after `apply bc` / `communicate`).  u_in != u_out. */
int examg_jacobi_residual(const examg_layout_t *lu, const double *u_in, double *u_out, const examg_layout_t *lf, const double *rhs,
                          const examg_layout_t *lr, double *res, const examg_stencil_t *st, double w, const int32_t *begin,
                          const int32_t *end, examg_stream_t stream);

/* As examg_jacobi2 with separate boxes: stage 1 = J on [begin1,end1) (points outside keep u_in's value), stage 2 = J of
 * that field on [begin2,end2), inside box 1, written to u_out.  With block neighbours: box 1 = the loop's box, box 2 =
 * box 1 without the duplicate planes at interior faces, whose second step needs the neighbour's first-step values
 * (halo exchange of the intermediate field, then a thin single-step launch on those planes). */
int examg_jacobi2_boxes(const examg_layout_t *lu, const double *u_in, double *u_out, double *tmp, const examg_layout_t *lf,
                        const double *rhs, const examg_stencil_t *st, double w, const int32_t *begin1, const int32_t *end1,
                        const int32_t *begin2, const int32_t *end2, examg_stream_t stream);

/* examg_rbgs_sweep_fused for a u_in that is 0.0 everywhere, boundary planes included -- a coarse level's first pre-smoothing
 * sweep after `Solution@coarser = 0` (mgCycle, Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:225-229 followed by :204-213): u_in is
 * not passed and not read (16 B per point instead of 24, and the zeroing loop need not run), the arithmetic is the same
 * expression evaluated on the constant 0.0: bit-identical.  The shell of u_out is not written (fallback path: zeroed). */
int examg_rbgs_sweep_fused_zero(const examg_layout_t *lu, double *u_out, const examg_layout_t *lf, const double *rhs,
                                const examg_stencil_t *st, double w, int first, const int32_t *begin, const int32_t *end,
                                examg_stream_t stream);

/* `Solution += Prolongation@coarser * Solution@coarser` on [begin,end) followed by the first post-smoothing pass on the same box
 * (mgCycle, Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:240-247: correction loop, `apply bc`, smoother) in ONE pass: the
 * one-pass kernels interpolate the coarse values (1/8 of the points, staged through LDS) while they load u_in, instead of a
 * separate read-modify-write loop over the fine field (16 B per point).  u_in is not modified; u_out receives on the box
 * exactly (bit for bit) what examg_prolong_add on u_in followed by examg_rbgs_sweep_fused / examg_jacobi2 would put there.
 * Single block only (the correction's `communicate` and the halo exchanges of the smoother must be empty).  Arguments that the
 * one-pass kernel does not take (examg_two_stage_eligible with both boxes = [begin,end)) run copy + the plain loops. */
int examg_rbgs_sweep_fused_prolong(const examg_layout_t *lu, const double *u_in, double *u_out, const examg_layout_t *lf,
                                   const double *rhs, const examg_stencil_t *st, double w, int first, const int32_t *begin,
                                   const int32_t *end, const examg_layout_t *lc, const double *uc, examg_stream_t stream);
int examg_jacobi2_prolong(const examg_layout_t *lu, const double *u_in, double *u_out, double *tmp, const examg_layout_t *lf,
                          const double *rhs, const examg_stencil_t *st, double w, const int32_t *begin, const int32_t *end,
                          const examg_layout_t *lc, const double *uc, examg_stream_t stream);

/* 1 if examg_jacobi2_boxes / examg_rbgs_sweep_fused_boxes will run their one-pass kernel for these arguments, 0 if they will
 * take the fallback that writes `tmp` on the launch stream (other stencils or entry orders, short rows, boxes at the edge of
 * the allocation).  A caller that overlaps the pass with work on `tmp` on another stream must ask here first. */
int examg_two_stage_eligible(const examg_layout_t *lu, const examg_layout_t *lf, const examg_stencil_t *st, const int32_t *begin1,
                             const int32_t *end1, const int32_t *begin2, const int32_t *end2);

/* `Residual = RHS - A * Solution` followed by `RHS@coarser = scale * R * Residual` (mgCycle,
 * Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:215-223) as ONE pass when nothing else reads the fine residual: it is never
 * written (48 + 8 B per fine point -> 16 B).  [fbegin,fend): the residual loop's box; [cbegin,cend): the restriction loop's
 * box, whose fine footprint must lie inside it (a block without neighbours; with neighbours the restriction reads residuals
 * on ghost points, which only an exchange of the stored field provides -- use the two calls).  3-D 7-point constant
 * stencils on long rows take the fused kernel and never touch `res`/`lr` (may be NULL); everything else runs
 * examg_residual + examg_restrict through `res`.  Bit-identical to the two calls either way. */
int examg_residual_restrict(const examg_layout_t *lu, const double *u, const examg_layout_t *lf, const double *rhs,
                            const examg_layout_t *lr, double *res, const examg_stencil_t *st, const examg_layout_t *lc,
                            double *fc, double scale, const int32_t *fbegin, const int32_t *fend, const int32_t *cbegin,
                            const int32_t *cend, examg_stream_t stream);
/* 1 if examg_residual_restrict will run its one-pass kernel for these arguments (and leave `res` untouched), else 0 */
int examg_residual_restrict_one_pass(const examg_layout_t *lu, const examg_layout_t *lf, const examg_stencil_t *st, const examg_layout_t *lc,
                                     const int32_t *fbegin, const int32_t *fend, const int32_t *cbegin, const int32_t *cend);

/* ---- K4: RHS@coarser = scale * R * Residual, R = kron [1/4 1/2 1/4]
 * (operator/l4/L4_DefaultRestriction.scala:29-36,63-88; solver/ir/IR_ResolveIntergridIndices.scala);
 * begin/end: coarse iterator box. */
int examg_restrict(const examg_layout_t *lfine, const double *res_fine, const examg_layout_t *lcoarse,
                   double *rhs_coarse, double scale, const int32_t *begin, const int32_t *end,
                   examg_stream_t stream);

/* ---- K5: Solution += P@coarser * Solution@coarser, P = 2^d R^T
 * (operator/l4/L4_DefaultProlongation.scala:30-45; parity cases
 * stencil/ir/IR_FindStencilConvolutions.scala:135-156); begin/end: fine iterator box. */
int examg_prolong_add(const examg_layout_t *lcoarse, const double *u_coarse, const examg_layout_t *lfine,
                      double *u_fine, const int32_t *begin, const int32_t *end, examg_stream_t stream);

/* ---- K6: BLAS-1 loops (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:160-198, 226-229) */
int examg_set(const examg_layout_t *l, double *x, double v, const int32_t *begin, const int32_t *end,
              examg_stream_t stream);
/* y = a*x + b*y, evaluated as the reference statements are written:
 * b==0: a*x (copy for a==1);  b==1: y + a*x;  a==1: x + b*y. */
int examg_axpby(const examg_layout_t *lx, const double *x, const examg_layout_t *ly, double *y, double a, double b,
                const int32_t *begin, const int32_t *end, examg_stream_t stream);
/* As examg_axpby with a = sign * (*num / *den) or b = (*num / *den) read from device memory
 * (CG's alpha and beta without a host round trip); which: 0 => a, 1 => b. */
int examg_axpby_dev(const examg_layout_t *lx, const double *x, const examg_layout_t *ly, double *y, double a,
                    double b, int which, double sign, const double *num, const double *den, const int32_t *begin,
                    const int32_t *end, examg_stream_t stream);

/* ---- K7: reductions (`loop over ... with reduction`, ...exa4:113-119; replaces
 * cuda/CUDA_Reduction.scala:84-131 + DefaultReductionKernel, cuda/CUDA_KernelFunctions.scala:112-238).
 * Result (one double) is written to device memory `result`; `work` is caller-owned scratch of
 * examg_reduce_work_bytes() bytes.  Deterministic (fixed tree) for a fixed box. */
size_t examg_reduce_work_bytes(void);
int examg_dot(const examg_layout_t *lx, const double *x, const examg_layout_t *ly, const double *y,
              const int32_t *begin, const int32_t *end, double *result, void *work, examg_stream_t stream);

/* `Residual = RHS - A * Solution` followed by the norm's reduction loop (`ResNorm`, Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:113-119
 * after :215-219) as ONE pass when nothing else reads the residual: *result (device) = sum over [begin,end) of (rhs - A u)^2, the
 * residual is not stored (16 B per point instead of 24 + 8).  [begin,end) is the reduction loop's box (the residual is only needed
 * there).  3-D 7-point constant stencils on long rows take the one-pass kernel and never touch `res` / `lr` (may be NULL); everything
 * else runs examg_residual + examg_dot through `res`.  Same partial sums every run; their order differs from examg_dot's (1e-15
 * relative).  `work`: examg_reduce_work_bytes() bytes. */
int examg_residual_norm2(const examg_layout_t *lu, const double *u, const examg_layout_t *lf, const double *rhs,
                         const examg_stencil_t *st, const int32_t *begin, const int32_t *end, const examg_layout_t *lr, double *res,
                         double *result, void *work, examg_stream_t stream);

/* ---- start values from the C library's generator (host side).  Programs of the reference's test suite initialise their solution
 * with `loop over F sequentially { F = native ( "((double)std::rand()/RAND_MAX)" ) }` (Testing/Opts/base.exa4:166-170), every
 * MPI process after std::srand(mpiRank) (parallelization/api/mpi/MPI_IVs.scala:41-45); their results files come from glibc's rand()
 * (TYPE_3 additive feedback generator).  examg_crand_seed / examg_crand_draw_host restate that generator, so that a host program
 * gets the reference's start values whatever C library it runs with: the next n values of (double)rand()/RAND_MAX into a HOST
 * buffer, in the order the generated loop nest draws them (x fastest; several draws per point interleave).  The caller places and
 * uploads them. */
typedef struct {
  uint32_t r[31];
  int32_t k;
} examg_crand_state_t;
int examg_crand_seed(examg_crand_state_t *state, uint32_t seed);
int examg_crand_draw_host(examg_crand_state_t *state, double *host_out, int64_t n);

/* ---- analytic expressions as stack programs: the ONE mechanism for boundary values, right-hand sides, exact solutions and
 * coefficient profiles (round 1 also had 17 built-in function ids of the reference's test programs; they are gone -- the same
 * expression trees now travel as programs, exastencils_amd/field.py:FN_PROGRAMS) --------------------------------------- ------------------------------------------------------------------------
 * The generator inlines whatever expression a program gives for boundary values, right-hand sides and exact solutions
 * into the loop body (boundary/ir/IR_DirichletBC.scala:37-40; `loop over RHS { RHS = <expr> }`).  A library cannot be
 * recompiled per program, so an expression over the node position travels as a postfix program that a generic kernel
 * evaluates per point, in the order the expression tree prescribes (same operations in the same order as the printed
 * code).  op[i]: EXAMG_OP_*; c[i]: the literal of a CONST instruction. */
#define EXAMG_MAX_EXPR 256
enum {
  EXAMG_OP_CONST = 0, EXAMG_OP_X = 1, EXAMG_OP_Y = 2, EXAMG_OP_Z = 3, EXAMG_OP_ADD = 4, EXAMG_OP_SUB = 5, EXAMG_OP_MUL = 6,
  EXAMG_OP_DIV = 7, EXAMG_OP_NEG = 8, EXAMG_OP_SIN = 9, EXAMG_OP_COS = 10, EXAMG_OP_EXP = 11, EXAMG_OP_SINH = 12,
  EXAMG_OP_COSH = 13, EXAMG_OP_SQRT = 14, EXAMG_OP_POW = 15, EXAMG_OP_TAN = 16, EXAMG_OP_LOG = 17, EXAMG_OP_FABS = 18,
  EXAMG_OP_MAX = 19, EXAMG_OP_MIN = 20, EXAMG_OP_TANH = 21
};
typedef struct {
  int32_t n;
  int32_t op[EXAMG_MAX_EXPR];
  double c[EXAMG_MAX_EXPR];
} examg_expr_t;

/* x[box] = e(node position): InitRHS, SetFuncDir, `loop over F { F = <expr> }`. */
int examg_fill_expr(const examg_layout_t *l, double *x, const examg_geom_t *g, const examg_expr_t *e, const int32_t *begin,
                    const int32_t *end, examg_stream_t stream);
/* All faces of `apply bc` in one launch (boundary/ir/IR_DirichletBC.scala:37-40 over the ranges of
 * boundary/ir/IR_ApplyBCFunction.scala:53-83): face_mask bit (2*d + (side>0)) set => that face has no neighbour
 * (IR_IV_NeighborIsValid false) and gets its duplicate plane, tangentially GLB..GRE, set to e(node position). */
int examg_apply_dirichlet_expr(const examg_layout_t *l, double *x, const examg_geom_t *g, const examg_expr_t *e,
                               uint32_t face_mask, examg_stream_t stream);
/* `loop over F only dup [dir] on boundary { F = <expr> }` for every direction of the mask in one launch (the SetFuncDir loops of
 * Testing/FMG/3D_Trigonometric.exa4:189-201; baseExt/ir/IR_LoopOverPointsInOneFragment.scala:57-70): the duplicate plane of each
 * physical face, tangentially DLB..DRE.  Same values as one examg_fill_expr per face (edges are written by two faces: same value). */
int examg_fill_dup_faces_expr(const examg_layout_t *l, double *x, const examg_geom_t *g, const examg_expr_t *e,
                              uint32_t face_mask, examg_stream_t stream);
/* max |x - e(node position)| over the box (`loop over F with reduction(max : err)`); result in device memory */
int examg_max_err_expr(const examg_layout_t *l, const double *x, const examg_geom_t *g, const examg_expr_t *e,
                       const int32_t *begin, const int32_t *end, double *result, void *work, examg_stream_t stream);

/* ---- stencil-field initialisation.  Testing/SISC/3D_VarCoeff.exa4:206-217 (2*nd+1 entries): -div(a grad u) with the
 * coefficient expression `a` evaluated half a mesh width to either side of the node. */
int examg_init_varcoeff7(const examg_layout_t *lc, double *cfield, const examg_geom_t *g, const examg_expr_t *a,
                         const int32_t *begin, const int32_t *end, examg_stream_t stream);

/* 27-entry stencil field of -div(a grad u) - k^2 u (BASELINE.json config 4): trilinear elements with element-wise
 * constant a = a(element centre) (expression program), lumped mass, scaled by 1/h^3; ksq = k^2.
 * Entry order: (0,0,0), then (dx,dy,dz) lexicographic with dx slowest.  The stencil-field mechanism is the reference's
 * (stencil/ir/IR_StencilConvolution.scala:73-95); the reference itself ships 2d+1-entry fields only. */
int examg_init_helmholtz27(const examg_layout_t *lc, double *cfield, const examg_geom_t *g, const examg_expr_t *a, double ksq,
                           const int32_t *begin, const int32_t *end, examg_stream_t stream);

/* Apply (to_entry_fastest = 1) or undo (0) the layout transformation `[x, y, z, i] => [i, x, y, z]` of a stencil field's
 * coefficient array: dst[linear * nent + k] <-> src[k * size(lc) + linear] over the whole allocation; src != dst. */
int examg_transform_stencilfield(const examg_layout_t *lc, int nent, const double *src, double *dst, int to_entry_fastest,
                                 examg_stream_t stream);

/* Copy a field between two layouts that differ in `transform` only (same regions): dst[index under ldst] = src[index under lsrc] over
 * the whole allocation; src != dst.  doubles of an array: examg_layout_size. */
int examg_transform_field(const examg_layout_t *lsrc, const double *src, const examg_layout_t *ldst, double *dst, examg_stream_t stream);
int64_t examg_layout_size(const examg_layout_t *l);

/* ---- K9: halo pack / unpack (communication/ir/IR_NoInterpPacking.scala:53-83): box <-> contiguous
 * buffer, x fastest; ranges from IR_PackInfoDuplicate.scala:15-39 / IR_PackInfoGhost.scala:13-60. */
int examg_pack(const examg_layout_t *l, const double *x, double *buf, const int32_t *begin, const int32_t *end,
               examg_stream_t stream);
int examg_unpack(const examg_layout_t *l, double *x, const double *buf, const int32_t *begin, const int32_t *end,
                 examg_stream_t stream);

/* ---- a-16: coarse-grid CG, mgCycle@coarsest (...exa4:152-201) as ONE persistent single-workgroup
 * kernel (no host round trips).  Fields in the reference layout.  info: >= 4 doubles in device memory, zeroed by the
 * caller once: info[0] = iterations, info[1] = initial residual, info[2] = final residual of this call; info[3] is
 * incremented when the loop ended at max_it without meeting the tolerance -- where the generated function prints
 * "Maximum number of cgs iterations (max_it) was exceeded" (the host reads the count when it next synchronises). */
int examg_cg_coarse(const examg_layout_t *lu, double *sol, const examg_layout_t *lf, const double *rhs,
                    const examg_layout_t *lr, double *res, const examg_layout_t *lp, double *p,
                    const examg_layout_t *lq, double *ap, const examg_stencil_t *st, const examg_geom_t *g,
                    uint32_t face_mask, int max_it, double rel_tol, const int32_t *begin, const int32_t *end,
                    double *info, examg_stream_t stream);
/* The same solver as the reference's layer-3 solver generator writes it (Function VCycle_0@coarsest, Testing/Smoothers/Jac.exa4:75-109,
 * Testing/FMG/3D_Trigonometric.exa4: the programs with slotted fields): the numerator of alpha is the SQUARE OF THE ROUNDED NORM carried
 * from the previous iteration (`alpha = res * res / alphaDenom`) instead of a fresh sum of squares, and the solver contains no
 * `apply bc` statements -- boundary planes are read as they are and never written (Solution@coarsest carries the Dirichlet values of
 * the FMG start there). */
#define EXAMG_CG_ALPHA_FROM_NORM 1u
#define EXAMG_CG_NO_BC 2u
/* `Solution@coarsest = 0` (mgCycle of the next finer level, ...exa4:225-229) rides along: the solver takes the zero field as its start --
 * `sol` is not read (inner points; its boundary planes hold 0 already) and the zeroing loop need not run.  Same expressions on the
 * constant 0.0: bit-identical to examg_set + the solver. */
#define EXAMG_CG_ZERO_START 4u
int examg_cg_coarse_variant(const examg_layout_t *lu, double *sol, const examg_layout_t *lf, const double *rhs,
                            const examg_layout_t *lr, double *res, const examg_layout_t *lp, double *p,
                            const examg_layout_t *lq, double *ap, const examg_stencil_t *st, const examg_geom_t *g,
                            uint32_t face_mask, int max_it, double rel_tol, const int32_t *begin, const int32_t *end,
                            uint32_t flags, double *info, examg_stream_t stream);

/* ---- external fields: get<Name>(dest, slot) / set<Name>(src, slot) of `external Field` declarations
 * (interfacing/ir/IR_CopyToExternalField.scala:31-90, IR_CopyFromExternalField.scala): device-to-device copy between a
 * caller-owned array in its own layout (same duplicate/inner extents, its own ghost/pad widths) and the internal field
 * over [DLB - min(ghosts), DRE + min(ghosts)) per dimension. */
int examg_copy_to_external(const examg_layout_t *l_int, const double *x_int, const examg_layout_t *l_ext, double *dest,
                           examg_stream_t stream);
int examg_copy_from_external(const examg_layout_t *l_ext, const double *src, const examg_layout_t *l_int, double *x_int,
                             examg_stream_t stream);

/* ---- a-13 / e: block-to-block transport over RCCL (one process per GPU) -----------------------------------------------
 * What the generated host calls `exch<Field>_<level>(slot)` (communication/ir/IR_CommunicateFunction.scala:194-219,412-480,
 * naming IR_SetupCommunication.scala:119-147) and the MPI_Allreduce after a reduction loop
 * (parallelization/api/mpi/MPI_Reduction.scala:100-126).  The communicator stands where MPI_COMM_WORLD does; it is created
 * from a 128-byte id that rank 0 obtains and the host distributes by its own means (MPI_Bcast in a generated program, a
 * file, torch.distributed).  All calls are asynchronous on `stream` (ncclSend / ncclRecv groups, pack / unpack kernels).  NOT for
 * stream capture: RCCL point-to-point groups hang inside a hipGraph capture on ROCm 7.2 -- the peer-write transport below is
 * the capturable one.  A communicator of one rank needs no RCCL (id may be NULL). */
typedef struct examg_comm examg_comm_t;
#define EXAMG_COMM_ID_BYTES 128
int examg_comm_unique_id(void *id /* EXAMG_COMM_ID_BYTES */);
int examg_comm_create(examg_comm_t **comm, const void *id, int nranks, int rank);   /* collective; binds the current HIP device */
int examg_comm_destroy(examg_comm_t *comm);
int examg_comm_rank(const examg_comm_t *comm);
int examg_comm_size(const examg_comm_t *comm);

/* Ranks of the axis neighbours of this block, rank[d][0] across the lower and rank[d][1] across the upper face of dimension d;
 * -1 = physical boundary (IR_IV_NeighborIsValid false, domain/ir/IR_ConnectFragments.scala:110-151); the own rank = periodic
 * dimension with one block. */
typedef struct examg_neighbors {
  int32_t rank[3][2];
} examg_neighbors_t;

/* what: EXAMG_EXCH_DUP  duplicate layers, upstream (own upper plane -> '+' neighbour's lower plane), axis by axis;
 *       EXAMG_EXCH_GHOST ghost layers both ways, axis by axis with tangential extent GLB..GRE (corner ghosts become valid);
 *       | EXAMG_EXCH_CONCURRENT_AXES: ghost layers of all axes as ONE send/recv group -- face ghosts only, for loops that read
 *       nothing else (5/7-point stencils).  Which parts a field communicates is the layout declaration's business
 *       (`ghostLayers = [..] with communication`): the caller passes `what` accordingly. */
enum { EXAMG_EXCH_DUP = 1, EXAMG_EXCH_GHOST = 2, EXAMG_EXCH_ALL = 3, EXAMG_EXCH_CONCURRENT_AXES = 4,
       /* examg_jacobi2_blocks / examg_rbgs_sweep_blocks only: tmp's duplicate planes on the physical faces already hold u_in's values
        * (position-only Dirichlet values, written once by the caller): the pass does not refresh them */
       EXAMG_PASS_TMP_PLANES_VALID = 8 };
/* caller-owned device scratch for the packed slabs (the generated program's buffer_Send / buffer_Recv arrays) */
size_t examg_exchange_workspace_bytes(const examg_layout_t *l);
int examg_exchange(examg_comm_t *comm, const examg_layout_t *l, double *x, const examg_neighbors_t *nb, int what,
                   void *workspace, size_t workspace_bytes, examg_stream_t stream);
/* Smoother passes on a block WITH neighbours, halo traffic overlapped with the interior kernel -- the reference's core / boundary
 * split (baseExt/ir/IR_LoopOverPointsInOneFragment.scala:143-222) applied to a pair of dependent sweeps, as one call:
 *   examg_jacobi2_blocks     two Jacobi steps  (`communicate ghost of u; loop; advance` twice, Testing/Smoothers/Jac.exa4:125-131)
 *   examg_rbgs_sweep_blocks  one red-black sweep (`color with { .., communicate u; loop; .. }`, ...exa4:204-213), colour `first` first
 * The loop's box [begin, end) shrunk by one point (first stage) and two (second stage) at interior faces runs as ONE two-stage kernel
 * on `stream`; meanwhile a side stream owned by the communicator exchanges the ghost layers of u_in, evaluates the first stage on
 * the three planes next to every interior face into `tmp`, exchanges tmp's ghost layers and evaluates the second stage on the two
 * planes next to every interior face into u_out; events join the streams.  Two exchanges per pass, as the plain loops have;
 * results bit-identical to them.  u_in, u_out, tmp: three arrays of layout lu (tmp: scratch; its duplicate planes on physical
 * faces are refreshed from u_in).  exchange_flags: EXAMG_EXCH_CONCURRENT_AXES or 0.  overlap = 0, or a stencil / box for which
 * the one-pass kernel is not eligible: the same steps in sequence on `stream`.  The caller swaps u_in / u_out afterwards. */
int examg_jacobi2_blocks(examg_comm_t *comm, const examg_neighbors_t *nb, const examg_layout_t *lu, const double *u_in, double *u_out,
                         double *tmp, const examg_layout_t *lf, const double *rhs, const examg_stencil_t *st, double w,
                         const int32_t *begin, const int32_t *end, int exchange_flags, void *workspace, size_t workspace_bytes,
                         int overlap, examg_stream_t stream);
int examg_rbgs_sweep_blocks(examg_comm_t *comm, const examg_neighbors_t *nb, const examg_layout_t *lu, const double *u_in, double *u_out,
                            double *tmp, const examg_layout_t *lf, const double *rhs, const examg_stencil_t *st, double w, int first,
                            const int32_t *begin, const int32_t *end, int exchange_flags, void *workspace, size_t workspace_bytes,
                            int overlap, examg_stream_t stream);
/* The transfer operators of mgCycle on a block WITH neighbours, one call each (same split, around the `communicate` statements of
 * Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:215-237):
 *   examg_residual_restrict_blocks  `communicate Solution; Residual = RHS - A * Solution; communicate Residual; RHS@coarser = R * Residual`:
 *       the one-pass residual + restriction kernel on the coarse box shrunk by one point at interior faces (reads no ghost value,
 *       stores no residual) on `stream`; on the side stream the ghost exchange of u, the residual on the two fine planes next to
 *       every interior face into `res`, the exchange of `res` and the restriction of the coarse planes on the interior faces.
 *       `res` holds the residual on that two-plane shell only afterwards.  Bit-identical to the four statements.
 *   examg_prolong_add_blocks        `communicate Solution@coarser; Solution += P * Solution@coarser`: the interpolation reads duplicate
 *       and inner coarse points only, so the ghost part of the exchange runs beside the kernel.
 * exchange_flags: EXAMG_EXCH_DUP -- also exchange the duplicate layers (first, in sequence; leave it out when both owners of a
 * shared plane are known to hold the same bits); EXAMG_EXCH_CONCURRENT_AXES -- ghost layers of u in one batch (7-point stencil). */
int examg_residual_restrict_blocks(examg_comm_t *comm, const examg_neighbors_t *nb, const examg_layout_t *lu, double *u,
                                   const examg_layout_t *lf, const double *rhs, const examg_layout_t *lr, double *res,
                                   const examg_stencil_t *st, const examg_layout_t *lc, double *fc, double scale, const int32_t *fbegin,
                                   const int32_t *fend, const int32_t *cbegin, const int32_t *cend, int exchange_flags, void *workspace,
                                   size_t workspace_bytes, int overlap, examg_stream_t stream);
int examg_prolong_add_blocks(examg_comm_t *comm, const examg_neighbors_t *nb, const examg_layout_t *lc, double *uc,
                             const examg_layout_t *lfine, double *uf, const int32_t *begin, const int32_t *end, int exchange_flags,
                             void *workspace, size_t workspace_bytes, int overlap, examg_stream_t stream);
/* MPI_Allreduce(MPI_IN_PLACE, x, n, MPI_DOUBLE, op): x is device memory; op 0 = sum, 1 = max, 2 = min */
int examg_allreduce(examg_comm_t *comm, double *x, int n, int op, examg_stream_t stream);
/* every rank's n doubles, in rank order (coarse-level agglomeration: fewer, larger collectives over xGMI) */
int examg_allgather(examg_comm_t *comm, const double *send, double *recv, int64_t n, examg_stream_t stream);

/* ---- a-13 / e, second transport: peer writes through HIP IPC (SURVEY.md section 5.8 "direct peer writes"; replaces the
 * MPI_Isend / MPI_Irecv branch of communication/ir/IR_CommunicateFunction.scala:412-471, IR_RemoteSend.scala:48-58,
 * IR_RemoteRecv.scala:50-65) -- no communication library: every rank owns a region of uncached device memory that its
 * neighbours map (hipIpcOpenMemHandle: xGMI peer mapping across the GPUs of a node, a plain mapping when ranks share a device);
 * the send kernel packs a field box straight into the neighbour's receive slab and publishes a sequence number, the receive
 * kernel waits for it and unpacks.  Ordering is done by device-side flags whose counters live in device memory, so
 * examg_exchange / examg_allreduce / examg_allgather / examg_*_blocks on such a communicator are stream-ordered, free of host
 * round trips and CAPTURABLE INTO A hipGraph (replayed by all ranks alike).  Set-up, all collective:
 *   examg_comm_create_peer(&c, nranks, rank);
 *   examg_comm_peer_alloc(c, slab_bytes, gather_bytes, handle);    slab_bytes >= the largest halo message (one face slab,
 *                                                                  examg_exchange_workspace_bytes(l) / 4 is always enough),
 *                                                                  gather_bytes >= the largest examg_allgather piece (0: none)
 *   <all-gather the EXAMG_PEER_HANDLE_BYTES of every rank by the host's own means: MPI_Allgather, torch.distributed, files>
 *   examg_comm_peer_connect(c, all_handles);
 * To grow the slabs later: synchronise the device on every rank, host barrier, examg_comm_peer_release on every rank (no region is
 * freed while a neighbour still maps it), host barrier, then alloc / gather / connect again (sequence numbers restart).  The workspace arguments of examg_exchange / examg_*_blocks are ignored (may be NULL).  A wait that sees no
 * progress for EXAMG_PEER_TIMEOUT_MS (default 120000) gives up, makes every later wait return at once and is reported by
 * examg_comm_status() -- a lost neighbour never leaves a kernel spinning. */
#define EXAMG_PEER_HANDLE_BYTES 128
int examg_comm_create_peer(examg_comm_t **comm, int nranks, int rank);
int examg_comm_peer_alloc(examg_comm_t *comm, size_t slab_bytes, size_t gather_bytes, void *handle_out /* EXAMG_PEER_HANDLE_BYTES */);
int examg_comm_peer_connect(examg_comm_t *comm, const void *all_handles /* nranks x EXAMG_PEER_HANDLE_BYTES, rank order */);
/* growing: examg_comm_peer_release on every rank (unmaps the other ranks' regions), host barrier, then alloc / gather / connect */
int examg_comm_peer_release(examg_comm_t *comm);
size_t examg_comm_peer_slab_bytes(const examg_comm_t *comm);
size_t examg_comm_peer_gather_bytes(const examg_comm_t *comm);
/* synchronises `stream`, then 0 if no wait of the peer-write transport has given up (always 0 for RCCL communicators) */
int examg_comm_status(examg_comm_t *comm, examg_stream_t stream);

/* Deterministic synthetic field (SplitMix64 of the linear index, U(-1,1)); same bits as the oracle's. */
int examg_fill_random(double *x, int64_t n, uint64_t seed, examg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EXAMG_H */
