// Coarse-grid CG as one persistent single-workgroup kernel, and the fused red-black sweep (gfx950).
//
// Reference: Function mgCycle@coarsest, Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:152-201 --
// per iteration 1 mat-vec, 2 dots, 1 norm, 3 vector updates, 5 `apply bc`, i.e. ~15 kernel launches
// and 3 host-visible reductions in the reference's CUDA build (each reduction itself a chain of
// halving kernels + an 8-byte D2H copy, parallelization/api/cuda/CUDA_KernelFunctions.scala:112-238).
// On the coarsest grid (16^3 at config 3) that is pure latency, so the whole solve runs in ONE
// workgroup of 1024 threads: phases separated by workgroup barriers, reductions through LDS with a
// fixed tree (deterministic), the data-dependent exit (`nextRes <= 0.001 * initRes`) taken on device.
// Single fragment only (every face a physical boundary): with neighbours the loop needs
// `communicate cgTmp0` and all-reduces, which the host driver does with the unfused kernels.
#include "examg_common.h"

namespace examg {

constexpr int CG_THREADS = 1024;

struct StencilCG {
  int nent, diag;
  long long uo[EXAMG_MAX_ENTRIES];
  double coef[EXAMG_MAX_ENTRIES];
  const double *cfield;
  long long cplane;
};

__device__ __forceinline__ double cg_block_sum(double v, double *sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = v + __shfl_down(v, o);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();  // protect sm reuse
  if (lane == 0) sm[wv] = v;
  __syncthreads();
  double r = sm[0];
  for (int i = 1; i < CG_THREADS / 64; ++i) r = r + sm[i];
  return r;  // same value in every thread
}

__device__ __forceinline__ void cg_unflatten(const Box &box, int t, int &i0, int &i1, int &i2) {
  const int n0 = box.n0(), n1 = box.n1();
  i0 = box.b0 + t % n0;
  const int row = t / n0;
  i1 = box.b1 + row % n1;
  i2 = box.b2 + row / n1;
}

__device__ __forceinline__ double cg_apply(const StencilCG &st, const LayoutDev &lu, const double *u, const LayoutDev &lc,
                                           int i0, int i1, int i2) {
  const long long iu = lidx(lu, i0, i1, i2);
  double acc;
  if (st.cfield) {
    const long long ic = lidx(lc, i0, i1, i2);
    acc = st.cfield[ic] * u[iu + st.uo[0]];
    for (int k = 1; k < st.nent; ++k) acc = acc + st.cfield[ic + k * st.cplane] * u[iu + st.uo[k]];
  } else {
    acc = st.coef[0] * u[iu + st.uo[0]];
    for (int k = 1; k < st.nent; ++k) acc = acc + st.coef[k] * u[iu + st.uo[k]];
  }
  return acc;
}

struct FaceBoxesCG {
  Box box[6];
  int n;
};

// x[faces] = 0 for one field (the coarse-level fields all carry a homogeneous Dirichlet BC:
// `Field Solution<global, NodeWithComm, 0.0>@(all but finest)`, Residual / cgTmp0 `0.0`, ...exa4:24-33)
__device__ __forceinline__ void cg_zero_faces(const FaceBoxesCG &fb, const LayoutDev &l, double *x) {
  for (int f = 0; f < fb.n; ++f) {
    const int total = (int)fb.box[f].count();
    for (int t = threadIdx.x; t < total; t += CG_THREADS) {
      int i0, i1, i2;
      cg_unflatten(fb.box[f], t, i0, i1, i2);
      x[lidx(l, i0, i1, i2)] = 0.0;
    }
  }
}

__global__ void __launch_bounds__(CG_THREADS)
k_cg_coarse(LayoutDev lu, double *sol, LayoutDev lf, const double *rhs, LayoutDev lr, double *res, LayoutDev lp, double *p,
            LayoutDev lq, double *ap, LayoutDev lc, StencilCG st, FaceBoxesCG fbr, FaceBoxesCG fbp, FaceBoxesCG fbs, int max_it,
            double rel_tol, Box box, double *info) {
  __shared__ double sm[CG_THREADS / 64];
  const int total = (int)box.count();
  int i0, i1, i2;

  // Residual = RHS - Laplace * Solution ; apply bc to Residual
  double s = 0.0;
  for (int t = threadIdx.x; t < total; t += CG_THREADS) {
    cg_unflatten(box, t, i0, i1, i2);
    const double r = rhs[lidx(lf, i0, i1, i2)] - cg_apply(st, lu, sol, lc, i0, i1, i2);
    res[lidx(lr, i0, i1, i2)] = r;
    s = s + r * r;
  }
  cg_zero_faces(fbr, lr, res);
  double curRes = sqrt(cg_block_sum(s, sm));
  const double initRes = curRes;
  // cgTmp0 = Residual ; apply bc to cgTmp0
  for (int t = threadIdx.x; t < total; t += CG_THREADS) {
    cg_unflatten(box, t, i0, i1, i2);
    p[lidx(lp, i0, i1, i2)] = res[lidx(lr, i0, i1, i2)];
  }
  cg_zero_faces(fbp, lp, p);
  __syncthreads();

  int it = 0;
  double nextRes = curRes;
  for (; it < max_it;) {
    // cgTmp1 = Laplace * cgTmp0 ; alphaNom = sum Residual^2 ; alphaDenom = sum cgTmp0 * cgTmp1
    double sn = 0.0, sd = 0.0;
    for (int t = threadIdx.x; t < total; t += CG_THREADS) {
      cg_unflatten(box, t, i0, i1, i2);
      const double q = cg_apply(st, lp, p, lc, i0, i1, i2);
      ap[lidx(lq, i0, i1, i2)] = q;
      const double r = res[lidx(lr, i0, i1, i2)];
      sn = sn + r * r;
      sd = sd + p[lidx(lp, i0, i1, i2)] * q;
    }
    const double alphaNom = cg_block_sum(sn, sm);
    const double alphaDenom = cg_block_sum(sd, sm);
    const double alpha = alphaNom / alphaDenom;
    // Solution += alpha * cgTmp0 ; Residual -= alpha * cgTmp1 ; nextRes = ResNorm()
    double s2 = 0.0;
    for (int t = threadIdx.x; t < total; t += CG_THREADS) {
      cg_unflatten(box, t, i0, i1, i2);
      const long long ks = lidx(lu, i0, i1, i2), kr = lidx(lr, i0, i1, i2);
      sol[ks] = sol[ks] + alpha * p[lidx(lp, i0, i1, i2)];
      const double r = res[kr] - alpha * ap[lidx(lq, i0, i1, i2)];
      res[kr] = r;
      s2 = s2 + r * r;
    }
    nextRes = sqrt(cg_block_sum(s2, sm));
    ++it;
    if (nextRes <= rel_tol * initRes) break;  // uniform: every thread holds the same sums
    const double beta = (nextRes * nextRes) / (curRes * curRes);
    // cgTmp0 = Residual + beta * cgTmp0
    for (int t = threadIdx.x; t < total; t += CG_THREADS) {
      cg_unflatten(box, t, i0, i1, i2);
      const long long kp = lidx(lp, i0, i1, i2);
      p[kp] = res[lidx(lr, i0, i1, i2)] + beta * p[kp];
    }
    curRes = nextRes;
    __syncthreads();
  }
  cg_zero_faces(fbs, lu, sol);  // apply bc to Solution (homogeneous on coarse levels)
  if (threadIdx.x == 0 && info) {
    info[0] = (double)it;
    info[1] = initRes;
    info[2] = nextRes;
  }
}

static FaceBoxesCG face_boxes(const examg_layout_t *l, uint32_t face_mask) {
  FaceBoxesCG fb;
  fb.n = 0;
  for (int d = 0; d < l->nd; ++d)
    for (int side = 0; side < 2; ++side) {
      if (!(face_mask & (1u << (2 * d + side)))) continue;
      int b[3] = {0, 0, 0}, e[3] = {1, 1, 1};
      for (int t = 0; t < l->nd; ++t) {
        if (t == d) {
          if (side == 0) { b[t] = 0; e[t] = l->dup_l[t]; }
          else { b[t] = l->dup_l[t] + l->inner[t]; e[t] = b[t] + l->dup_r[t]; }
        } else {
          b[t] = -l->ghost_l[t];
          e[t] = l->dup_l[t] + l->inner[t] + l->dup_r[t] + l->ghost_r[t];
        }
      }
      Box bx{b[0], b[1], b[2], e[0], e[1], e[2]};
      if (bx.count() == 0) continue;
      fb.box[fb.n++] = bx;
    }
  return fb;
}

}  // namespace examg

using namespace examg;

extern "C" int examg_cg_coarse(const examg_layout_t *lu_, double *sol, const examg_layout_t *lf_, const double *rhs,
                               const examg_layout_t *lr_, double *res, const examg_layout_t *lp_, double *p,
                               const examg_layout_t *lq_, double *ap, const examg_stencil_t *st, const examg_geom_t *g,
                               uint32_t face_mask, int max_it, double rel_tol, const int32_t *begin, const int32_t *end,
                               double *info, examg_stream_t stream) {
  (void)g;
  if (!lu_ || !sol || !lf_ || !rhs || !lr_ || !res || !lp_ || !p || !lq_ || !ap || !st || !begin || !end) {
    set_error("examg_cg_coarse: null argument");
    return 1;
  }
  const uint32_t all = (1u << (2 * lu_->nd)) - 1;
  if ((face_mask & all) != all) {
    set_error("examg_cg_coarse: fused coarse solve needs every face on the physical boundary (single fragment)");
    return 1;
  }
  const Box box = make_box(begin, end);
  hipStream_t s = (hipStream_t)stream;
  if (box.count() == 0) {  // e.g. minLevel 0 on one fragment: CG is a no-op
    if (info) return check_hip(hipMemsetAsync(info, 0, 3 * sizeof(double), s), "examg_cg_coarse memset");
    return 0;
  }
  if (box.count() > (1 << 22)) { set_error("examg_cg_coarse: box too large for the single-workgroup solver"); return 1; }
  const int reach = stencil_reach(st);
  if (!box_inside(lu_, box, reach) || !box_inside(lp_, box, reach) || !box_inside(lf_, box, 0) || !box_inside(lr_, box, 0) ||
      !box_inside(lq_, box, 0)) {
    set_error("examg_cg_coarse: box leaves an allocation");
    return 1;
  }
  // the mat-vec runs on Solution (residual) and cgTmp0 (A*p): both must share strides
  const LayoutDev lu = make_layout(lu_), lp = make_layout(lp_);
  if (lu.s1 != lp.s1 || lu.s2 != lp.s2) { set_error("examg_cg_coarse: Solution and cgTmp0 layouts must agree"); return 1; }
  StencilCG sd;
  sd.nent = st->nent;
  sd.diag = st->diag;
  for (int k = 0; k < st->nent; ++k) {
    sd.uo[k] = st->off[k][0] + lu.s1 * st->off[k][1] + lu.s2 * st->off[k][2];
    sd.coef[k] = st->coef[k];
  }
  sd.cfield = st->cfield;
  sd.cplane = 0;
  LayoutDev lc = lu;
  if (st->cfield) {
    lc = make_layout(&st->clayout);
    sd.cplane = lc.size;
  }
  hipLaunchKernelGGL(k_cg_coarse, dim3(1), dim3(CG_THREADS), 0, s, lu, sol, make_layout(lf_), rhs, make_layout(lr_), res, lp, p,
                     make_layout(lq_), ap, lc, sd, face_boxes(lr_, face_mask), face_boxes(lp_, face_mask),
                     face_boxes(lu_, face_mask), max_it, rel_tol, box, info);
  EXAMG_CHECK_LAUNCH("k_cg_coarse");
  return 0;
}
