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
  long long cplane;   // stride between the entries of a point
  long long cpt;      // stride between points (1: reference layout, nent: entry-fastest transformation)
};

// Block sums with ONE workgroup barrier each: the wave partials go to one of two buffers in turn (a wave can only write buffer k & 1 again,
// for sum k + 2, after the barrier of sum k + 1, which every wave reaches after it has read buffer k & 1) -- the solver is a chain of
// barriers and cross-lane steps, ~32 iterations of it per cycle at config 3.  Fixed tree, the same on every call:
//   row of 16 lanes: lane 0 += lanes 8, 4, 2, 1 apart (DPP row_shl: a few cycles per step; the ds_bpermute path of __shfl_down costs
//   ~100 each); wave: the four row sums in row order (v_readlane); workgroup: the 16 wave sums in wave order.
constexpr int CG_SM = 4 * (CG_THREADS / 64);      // two buffers of up to two sums

__device__ __forceinline__ double cg_row_shl(double v, int n) {     // lane l of a row receives lane l + n of the same row, 0.0 beyond it
  int lo = __double2loint(v), hi = __double2hiint(v);
  switch (n) {
    case 8: lo = __builtin_amdgcn_update_dpp(0, lo, 0x108, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x108, 0xF, 0xF, true); break;
    case 4: lo = __builtin_amdgcn_update_dpp(0, lo, 0x104, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x104, 0xF, 0xF, true); break;
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x102, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x102, 0xF, 0xF, true); break;
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x101, 0xF, 0xF, true); break;
  }
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double cg_wave_sum(double v) {           // the same value in every lane
  v = v + cg_row_shl(v, 8);
  v = v + cg_row_shl(v, 4);
  v = v + cg_row_shl(v, 2);
  v = v + cg_row_shl(v, 1);
  const int lo = __double2loint(v), hi = __double2hiint(v);
  double r = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  r = r + __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  r = r + __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  r = r + __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return r;
}

__device__ __forceinline__ double cg_block_sum(double v, double *sm, int &turn) {
  v = cg_wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double *b = sm + (turn & 1) * 2 * (CG_THREADS / 64);
  ++turn;
  if (lane == 0) b[wv] = v;
  __syncthreads();
  double r = b[0];
  for (int i = 1; i < CG_THREADS / 64; ++i) r = r + b[i];
  return r;  // same value in every thread
}

// two sums behind one barrier (each with the tree of cg_block_sum: same bits as two calls)
__device__ __forceinline__ void cg_block_sum2(double &v0, double &v1, double *sm, int &turn) {
  v0 = cg_wave_sum(v0);
  v1 = cg_wave_sum(v1);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double *b = sm + (turn & 1) * 2 * (CG_THREADS / 64);
  ++turn;
  if (lane == 0) {
    b[wv] = v0;
    b[CG_THREADS / 64 + wv] = v1;
  }
  __syncthreads();
  double r0 = b[0], r1 = b[CG_THREADS / 64];
  for (int i = 1; i < CG_THREADS / 64; ++i) {
    r0 = r0 + b[i];
    r1 = r1 + b[CG_THREADS / 64 + i];
  }
  v0 = r0;
  v1 = r1;
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
  const long long iu = lidx_plain(lu, i0, i1, i2);
  double acc;
  if (st.cfield) {
    const long long ic = lidx_plain(lc, i0, i1, i2);
    acc = st.cfield[ic * st.cpt] * u[iu + st.uo[0]];
    for (int k = 1; k < st.nent; ++k) acc = acc + st.cfield[ic * st.cpt + k * st.cplane] * u[iu + st.uo[k]];
  } else {
    acc = st.coef[0] * u[iu + st.uo[0]];
    for (int k = 1; k < st.nent; ++k) acc = acc + st.coef[k] * u[iu + st.uo[k]];
  }
  return acc;
}

// A * 0: the convolution's own expression on the constant 0.0 (constant coefficients; a stencil field's zero-start is not offered)
__device__ __forceinline__ double cg_apply_zero(const StencilCG &st) {
  double acc = st.coef[0] * 0.0;
  for (int k = 1; k < st.nent; ++k) acc = acc + st.coef[k] * 0.0;
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
      x[lidx_plain(l, i0, i1, i2)] = 0.0;
    }
  }
}

__global__ void __launch_bounds__(CG_THREADS)
k_cg_coarse(LayoutDev lu, double *sol, LayoutDev lf, const double *rhs, LayoutDev lr, double *res, LayoutDev lp, double *p,
            LayoutDev lq, double *ap, LayoutDev lc, StencilCG st, FaceBoxesCG fbr, FaceBoxesCG fbp, FaceBoxesCG fbs, int max_it,
            double rel_tol, Box box, double *info, unsigned flags) {
  __shared__ double sm[CG_SM];
  int turn = 0;
  const bool from_norm = flags & EXAMG_CG_ALPHA_FROM_NORM, bc = !(flags & EXAMG_CG_NO_BC);
  const int total = (int)box.count();
  int i0, i1, i2;

  // Residual = RHS - Laplace * Solution ; apply bc to Residual
  double s = 0.0;
  for (int t = threadIdx.x; t < total; t += CG_THREADS) {
    cg_unflatten(box, t, i0, i1, i2);
    const double r = rhs[lidx_plain(lf, i0, i1, i2)] - ((flags & EXAMG_CG_ZERO_START) ? cg_apply_zero(st) : cg_apply(st, lu, sol, lc, i0, i1, i2));
    if (flags & EXAMG_CG_ZERO_START) sol[lidx_plain(lu, i0, i1, i2)] = 0.0;
    res[lidx_plain(lr, i0, i1, i2)] = r;
    s = s + r * r;
  }
  if (bc) cg_zero_faces(fbr, lr, res);
  double curRes = sqrt(cg_block_sum(s, sm, turn));
  const double initRes = curRes;
  // cgTmp0 = Residual ; apply bc to cgTmp0
  for (int t = threadIdx.x; t < total; t += CG_THREADS) {
    cg_unflatten(box, t, i0, i1, i2);
    p[lidx_plain(lp, i0, i1, i2)] = res[lidx_plain(lr, i0, i1, i2)];
  }
  if (bc) cg_zero_faces(fbp, lp, p);
  __syncthreads();

  int it = 0;
  bool converged = false;
  double nextRes = curRes;
  for (; it < max_it;) {
    // cgTmp1 = Laplace * cgTmp0 ; alphaNom = sum Residual^2 ; alphaDenom = sum cgTmp0 * cgTmp1
    double sn = 0.0, sd = 0.0;
    for (int t = threadIdx.x; t < total; t += CG_THREADS) {
      cg_unflatten(box, t, i0, i1, i2);
      const double q = cg_apply(st, lp, p, lc, i0, i1, i2);
      ap[lidx_plain(lq, i0, i1, i2)] = q;
      const double r = res[lidx_plain(lr, i0, i1, i2)];
      sn = sn + r * r;
      sd = sd + p[lidx_plain(lp, i0, i1, i2)] * q;
    }
    double alphaNom, alphaDenom = sd;
    if (from_norm) {
      alphaNom = curRes * curRes;
      alphaDenom = cg_block_sum(sd, sm, turn);
    } else {
      alphaNom = sn;
      cg_block_sum2(alphaNom, alphaDenom, sm, turn);
    }
    const double alpha = alphaNom / alphaDenom;
    // Solution += alpha * cgTmp0 ; Residual -= alpha * cgTmp1 ; nextRes = ResNorm()
    double s2 = 0.0;
    for (int t = threadIdx.x; t < total; t += CG_THREADS) {
      cg_unflatten(box, t, i0, i1, i2);
      const long long ks = lidx_plain(lu, i0, i1, i2), kr = lidx_plain(lr, i0, i1, i2);
      sol[ks] = sol[ks] + alpha * p[lidx_plain(lp, i0, i1, i2)];
      const double r = res[kr] - alpha * ap[lidx_plain(lq, i0, i1, i2)];
      res[kr] = r;
      s2 = s2 + r * r;
    }
    nextRes = sqrt(cg_block_sum(s2, sm, turn));
    ++it;
    if (nextRes <= rel_tol * initRes) { converged = true; break; }  // uniform: every thread holds the same sums
    const double beta = (nextRes * nextRes) / (curRes * curRes);
    // cgTmp0 = Residual + beta * cgTmp0
    for (int t = threadIdx.x; t < total; t += CG_THREADS) {
      cg_unflatten(box, t, i0, i1, i2);
      const long long kp = lidx_plain(lp, i0, i1, i2);
      p[kp] = res[lidx_plain(lr, i0, i1, i2)] + beta * p[kp];
    }
    curRes = nextRes;
    __syncthreads();
  }
  if (bc) cg_zero_faces(fbs, lu, sol);  // apply bc to Solution (homogeneous on coarse levels)
  if (threadIdx.x == 0 && info) {
    info[0] = (double)it;
    info[1] = initRes;
    info[2] = nextRes;
    if (!converged) info[3] = info[3] + 1.0;   // the statement after the loop: print ( "Maximum number of cgs iterations (..) was exceeded" )
  }
}


// Small coarsest grids (<= 4096 points, constant coefficients, reach 1): the same solver with the vectors resident on the CU.
// Every thread owns up to four fixed points and keeps their Residual, Solution and A*p values in registers; only the search
// direction, which the mat-vec reads at neighbours, lives in LDS (with a zero halo = its homogeneous Dirichlet planes).  No
// global memory traffic inside the iteration.  Per-thread partial sums run over the same points in the same order and the
// block sums use the same tree as k_cg_coarse: bit-identical results.
constexpr int CG_PPT = 4;

struct StencilCGL {
  int nent;
  int lo[EXAMG_MAX_ENTRIES];   // offsets in the LDS array of p
  double coef[EXAMG_MAX_ENTRIES];
};

__global__ void __launch_bounds__(CG_THREADS)
k_cg_coarse_lds(LayoutDev lu, double *sol, LayoutDev lf, const double *rhs, LayoutDev lr, double *res, LayoutDev lp, double *p,
                LayoutDev lq, double *ap, StencilCG st, StencilCGL sl, FaceBoxesCG fbr, FaceBoxesCG fbp, FaceBoxesCG fbs, int max_it,
                double rel_tol, Box box, double *info, int ldx, int ldxy, int ldtot, unsigned flags) {
  extern __shared__ double P[];
  __shared__ double sm[CG_SM];
  int turn = 0;
  const int total = (int)box.count();
  const bool from_norm = flags & EXAMG_CG_ALPHA_FROM_NORM, bc = !(flags & EXAMG_CG_NO_BC);
  // halo of the search direction: the boundary planes `apply bc` keeps at zero -- or, for a solver without `apply bc` statements,
  // whatever those planes hold
  for (int t = threadIdx.x; t < ldtot; t += CG_THREADS) {
    double v = 0.0;
    if (!bc) {
      const int a = t % ldx, rw = t / ldx, b2 = rw % (ldxy / ldx), c = rw / (ldxy / ldx);
      v = p[lidx_plain(lp, box.b0 - 1 + a, box.b1 - 1 + b2, box.b2 - 1 + c)];
    }
    P[t] = v;
  }
  __syncthreads();
  double r[CG_PPT], x[CG_PPT], q[CG_PPT], pv[CG_PPT];
  int li[CG_PPT];
  long long ku[CG_PPT], kr[CG_PPT], kp[CG_PPT], kq[CG_PPT];
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < CG_PPT; ++j) {
    const int t = threadIdx.x + j * CG_THREADS;
    r[j] = x[j] = q[j] = pv[j] = 0.0;
    li[j] = 0;
    ku[j] = kr[j] = kp[j] = kq[j] = 0;
    if (t < total) {
      int i0, i1, i2;
      cg_unflatten(box, t, i0, i1, i2);
      ku[j] = lidx_plain(lu, i0, i1, i2);
      kr[j] = lidx_plain(lr, i0, i1, i2);
      kp[j] = lidx_plain(lp, i0, i1, i2);
      kq[j] = lidx_plain(lq, i0, i1, i2);
      li[j] = (i0 - box.b0 + 1) + (i1 - box.b1 + 1) * ldx + (i2 - box.b2 + 1) * ldxy;
      const bool zero = flags & EXAMG_CG_ZERO_START;      // the start is the zero field: nothing of `sol` is read
      r[j] = rhs[lidx_plain(lf, i0, i1, i2)] - (zero ? cg_apply_zero(st) : cg_apply(st, lu, sol, lu, i0, i1, i2));   // Residual = RHS - Laplace * Solution
      x[j] = zero ? 0.0 : sol[ku[j]];
      s = s + r[j] * r[j];
    }
  }
  if (bc) {
    cg_zero_faces(fbr, lr, res);   // apply bc to Residual / cgTmp0: their boundary planes in memory, as the statements leave them
    cg_zero_faces(fbp, lp, p);
  }
  double curRes = sqrt(cg_block_sum(s, sm, turn));
  const double initRes = curRes;
#pragma unroll
  for (int j = 0; j < CG_PPT; ++j) {
    pv[j] = r[j];                 // cgTmp0 = Residual
    if (threadIdx.x + j * CG_THREADS < total) P[li[j]] = pv[j];
  }
  __syncthreads();

  int it = 0;
  bool converged = false;
  double nextRes = curRes;
  for (; it < max_it;) {
    double sn = 0.0, sd = 0.0;
#pragma unroll
    for (int j = 0; j < CG_PPT; ++j) {
      if (threadIdx.x + j * CG_THREADS < total) {
        double acc = sl.coef[0] * P[li[j] + sl.lo[0]];
        for (int k = 1; k < sl.nent; ++k) acc = acc + sl.coef[k] * P[li[j] + sl.lo[k]];
        q[j] = acc;
        sn = sn + r[j] * r[j];
        sd = sd + pv[j] * acc;
      }
    }
    double alphaNom, alphaDenom = sd;
    if (from_norm) {
      alphaNom = curRes * curRes;
      alphaDenom = cg_block_sum(sd, sm, turn);
    } else {
      alphaNom = sn;
      cg_block_sum2(alphaNom, alphaDenom, sm, turn);
    }
    const double alpha = alphaNom / alphaDenom;
    double s2 = 0.0;
#pragma unroll
    for (int j = 0; j < CG_PPT; ++j) {
      if (threadIdx.x + j * CG_THREADS < total) {
        x[j] = x[j] + alpha * pv[j];
        r[j] = r[j] - alpha * q[j];
        s2 = s2 + r[j] * r[j];
      }
    }
    nextRes = sqrt(cg_block_sum(s2, sm, turn));
    ++it;
    if (nextRes <= rel_tol * initRes) { converged = true; break; }
    const double beta = (nextRes * nextRes) / (curRes * curRes);
#pragma unroll
    for (int j = 0; j < CG_PPT; ++j) {
      if (threadIdx.x + j * CG_THREADS < total) {
        pv[j] = r[j] + beta * pv[j];
        P[li[j]] = pv[j];
      }
    }
    curRes = nextRes;
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < CG_PPT; ++j) {
    if (threadIdx.x + j * CG_THREADS < total) {
      sol[ku[j]] = x[j];
      res[kr[j]] = r[j];
      p[kp[j]] = pv[j];
      ap[kq[j]] = q[j];
    }
  }
  if (bc) cg_zero_faces(fbs, lu, sol);
  if (threadIdx.x == 0 && info) {
    info[0] = (double)it;
    info[1] = initRes;
    info[2] = nextRes;
    if (!converged) info[3] = info[3] + 1.0;   // the statement after the loop: print ( "Maximum number of cgs iterations (..) was exceeded" )
  }
}

static thread_local int g_cg_lds = 1;   // examg_debug_cg(0): global-memory solver for every size

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

#ifdef EXAMG_DEBUG_HOOKS
extern "C" int examg_debug_cg(int lds) {
  examg::g_cg_lds = lds;
  return 0;
}
#endif

extern "C" int examg_cg_coarse(const examg_layout_t *lu_, double *sol, const examg_layout_t *lf_, const double *rhs,
                               const examg_layout_t *lr_, double *res, const examg_layout_t *lp_, double *p,
                               const examg_layout_t *lq_, double *ap, const examg_stencil_t *st, const examg_geom_t *g,
                               uint32_t face_mask, int max_it, double rel_tol, const int32_t *begin, const int32_t *end,
                               double *info, examg_stream_t stream) {
  return examg_cg_coarse_variant(lu_, sol, lf_, rhs, lr_, res, lp_, p, lq_, ap, st, g, face_mask, max_it, rel_tol, begin, end, 0u, info,
                                 stream);
}

extern "C" int examg_cg_coarse_variant(const examg_layout_t *lu_, double *sol, const examg_layout_t *lf_, const double *rhs,
                                       const examg_layout_t *lr_, double *res, const examg_layout_t *lp_, double *p,
                                       const examg_layout_t *lq_, double *ap, const examg_stencil_t *st, const examg_geom_t *g,
                                       uint32_t face_mask, int max_it, double rel_tol, const int32_t *begin, const int32_t *end,
                                       uint32_t flags, double *info, examg_stream_t stream) {
  (void)g;
  if (!lu_ || !sol || !lf_ || !rhs || !lr_ || !res || !lp_ || !p || !lq_ || !ap || !st || !begin || !end) {
    set_error("examg_cg_coarse: null argument");
    return 1;
  }
  if (flags & ~(EXAMG_CG_ALPHA_FROM_NORM | EXAMG_CG_NO_BC | EXAMG_CG_ZERO_START)) { set_error("examg_cg_coarse_variant: unknown flag"); return 1; }
  if (lay_split(lu_) || lay_split(lf_) || lay_split(lr_) || lay_split(lp_) || lay_split(lq_)) {
    set_error("examg_cg_coarse: fields under a layout transformation are not supported by the one-kernel solver (examg_transform_field)");
    return 1;
  }
  const uint32_t all = (1u << (2 * lu_->nd)) - 1;
  if ((face_mask & all) != all) {
    set_error("examg_cg_coarse: fused coarse solve needs every face on the physical boundary (single fragment)");
    return 1;
  }
  if ((flags & EXAMG_CG_ZERO_START) && st->cfield) { set_error("examg_cg_coarse_variant: EXAMG_CG_ZERO_START needs constant coefficients"); return 1; }
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
  sd.cpt = 1;
  LayoutDev lc = lu;
  if (st->cfield) {
    lc = make_layout(&st->clayout);
    sd.cplane = lc.size;
    if (st->ctransform == EXAMG_CLAYOUT_ENTRY_FASTEST) { sd.cplane = 1; sd.cpt = st->nent; }
  }
  const long long ldx = box.n0() + 2, ldxy = ldx * (box.n1() + 2), ldtot = ldxy * (box.n2() + 2);
  // the LDS copy of cgTmp0 has a zero halo: right when the box is the whole interior, so that its neighbours are exactly the
  // boundary planes `apply bc` keeps at zero
  bool whole = true;
  {
    const int bb[3] = {box.b0, box.b1, box.b2}, ee[3] = {box.e0, box.e1, box.e2};
    for (int d = 0; d < lp_->nd; ++d) whole = whole && bb[d] == lp_->dup_l[d] && ee[d] == lp_->dup_l[d] + lp_->inner[d];
  }
  if (g_cg_lds && whole && !st->cfield && reach <= 1 && box.count() <= CG_PPT * CG_THREADS && ldtot * 8 <= 60 * 1024) {
    StencilCGL sl;
    sl.nent = st->nent;
    for (int k = 0; k < st->nent; ++k) {
      sl.lo[k] = (int)(st->off[k][0] + ldx * st->off[k][1] + ldxy * st->off[k][2]);
      sl.coef[k] = st->coef[k];
    }
    hipLaunchKernelGGL(k_cg_coarse_lds, dim3(1), dim3(CG_THREADS), (size_t)ldtot * 8, s, lu, sol, make_layout(lf_), rhs, make_layout(lr_), res,
                       lp, p, make_layout(lq_), ap, sd, sl, face_boxes(lr_, face_mask), face_boxes(lp_, face_mask),
                       face_boxes(lu_, face_mask), max_it, rel_tol, box, info, (int)ldx, (int)ldxy, (int)ldtot, flags);
    EXAMG_CHECK_LAUNCH("k_cg_coarse_lds");
    return 0;
  }
  hipLaunchKernelGGL(k_cg_coarse, dim3(1), dim3(CG_THREADS), 0, s, lu, sol, make_layout(lf_), rhs, make_layout(lr_), res, lp, p,
                     make_layout(lq_), ap, lc, sd, face_boxes(lr_, face_mask), face_boxes(lp_, face_mask),
                     face_boxes(lu_, face_mask), max_it, rel_tol, box, info, flags);
  EXAMG_CHECK_LAUNCH("k_cg_coarse");
  return 0;
}
