// Stencil loops of the multigrid hot path for gfx950: A*u, residual, Jacobi, red-black half sweeps.
//
// Replaces the kernels the reference extracts from `loop over` bodies with an
// IR_StencilConvolution (Compiler/src/exastencils/stencil/ir/IR_StencilConvolution.scala:42-95;
// CUDA kernel object parallelization/api/cuda/CUDA_Kernel.scala:357-544: one thread per point,
// no on-chip reuse).  Two paths:
//   * k_stencil7_zmarch: 3-D 7-point constant-coefficient fast path.  A wave owns a strip of
//     128 x-points (one 16-byte load per lane) by RY y-rows and marches in z with a register
//     pipeline (u[z-1], u[z], u[z+1]); x-neighbours come from the adjacent lane
//     (wavefront-level exchange), y-halo rows from L1/L2.  Every u value is fetched from HBM once.
//   * k_stencil_generic: any dimensionality / entry list / stencil field / colour.
// Arithmetic is ordered exactly as the generator prints it (entries folded left to right) and the
// file is compiled with -ffp-contract=off, so results are bit-identical to the CPU path.
#include "examg_common.h"

namespace examg {

struct StencilDev {
  int nent, diag;
  long long uo[EXAMG_MAX_ENTRIES];  // linear offsets in the u layout
  double coef[EXAMG_MAX_ENTRIES];
  const double *cfield;
  long long cplane;  // doubles per coefficient plane
};

// ---------------------------------------------------------------------------------------------
// generic path
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256)
k_stencil_generic(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs, LayoutDev ld,
                  double *dst, LayoutDev lc, StencilDev st, double w, int colour, Box box, int row_w) {
  const long long rows = (long long)box.n1() * box.n2();
  const long long total = rows * row_w;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long row = t / row_w;
    const int c0 = (int)(t - row * row_w);
    const int i1 = box.b1 + (int)(row % box.n1());
    const int i2 = box.b2 + (int)(row / box.n1());
    int i0;
    if (colour >= 0) {
      // points of this row with (i0+i1+i2) % 2 == colour, no idle lanes
      const int first = box.b0 + (((box.b0 + i1 + i2) & 1) != colour ? 1 : 0);
      i0 = first + 2 * c0;
    } else {
      i0 = box.b0 + c0;
    }
    if (i0 >= box.e0) continue;
    const long long iu = lidx(lu, i0, i1, i2);
    double acc;
    if (st.cfield) {
      const long long ic = lidx(lc, i0, i1, i2);
      acc = st.cfield[ic] * u[iu + st.uo[0]];
      for (int k = 1; k < st.nent; ++k) acc = acc + st.cfield[ic + k * st.cplane] * u[iu + st.uo[k]];
      if (MODE == EXAMG_SMOOTH) {
        const double ww = (1.0 / st.cfield[ic + st.diag * st.cplane]) * w;
        acc = u[iu] + ww * (rhs[lidx(lf, i0, i1, i2)] - acc);
      }
    } else {
      acc = st.coef[0] * u[iu + st.uo[0]];
      for (int k = 1; k < st.nent; ++k) acc = acc + st.coef[k] * u[iu + st.uo[k]];
      if (MODE == EXAMG_SMOOTH) acc = u[iu] + w * (rhs[lidx(lf, i0, i1, i2)] - acc);
    }
    if (MODE == EXAMG_RESIDUAL) acc = rhs[lidx(lf, i0, i1, i2)] - acc;
    dst[lidx(ld, i0, i1, i2)] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// 3-D 7-point constant-coefficient z-marching kernel
// ---------------------------------------------------------------------------------------------
typedef double d2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(8))) d2u { double a, b; };  // 16-byte load at 8-byte alignment

__device__ __forceinline__ d2 load2(const double *p) {
  const d2u v = *reinterpret_cast<const d2u *>(p);
  d2 r;
  r.x = v.a;
  r.y = v.b;
  return r;
}
__device__ __forceinline__ void store2(double *p, d2 v) {
  d2u s;
  s.a = v.x;
  s.b = v.y;
  *reinterpret_cast<d2u *>(p) = s;
}

struct Coef7 {
  double c[7];
};

// ORDER 0: entries c,-x,+x,-y,+y,-z,+z (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:39-47)
// ORDER 1: entries c,+x,-x,+y,-y,+z,-z (Testing/Smoothers/Jac.exa4:55-63)
template <int ORDER>
__device__ __forceinline__ double conv7(const Coef7 &k, double c, double xm, double xp, double ym, double yp, double zm,
                                        double zp) {
  double acc = k.c[0] * c;
  if (ORDER == 0) {
    acc = acc + k.c[1] * xm;
    acc = acc + k.c[2] * xp;
    acc = acc + k.c[3] * ym;
    acc = acc + k.c[4] * yp;
    acc = acc + k.c[5] * zm;
    acc = acc + k.c[6] * zp;
  } else {
    acc = acc + k.c[1] * xp;
    acc = acc + k.c[2] * xm;
    acc = acc + k.c[3] * yp;
    acc = acc + k.c[4] * ym;
    acc = acc + k.c[5] * zp;
    acc = acc + k.c[6] * zm;
  }
  return acc;
}

template <int MODE>
__device__ __forceinline__ double finish(double u, double acc, double f, double w) {
  if (MODE == EXAMG_APPLY) return acc;
  if (MODE == EXAMG_RESIDUAL) return f - acc;
  return u + w * (f - acc);
}

struct ZMarchGeom {
  int ntx, nty, ntz;  // tiles per dim
  int zc;             // planes per z chunk
  int nblocks;        // ntx * nty * ntz
};

// XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (blocks b and b+8 share one
// L2), so give each XCD a contiguous run of tiles -- y-adjacent tiles then share halo rows in L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int per = nblocks >> 3;
  const int full = per << 3;
  if (bid >= full) return bid;  // tail blocks keep their id
  return (bid & 7) * per + (bid >> 3);
}

template <int MODE, int ORDER, int RY, int WY, bool NT>
__global__ void __launch_bounds__(64 * WY)
k_stencil7_zmarch(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs, LayoutDev ld,
                  double *__restrict__ dst, Coef7 k, double w, Box box, ZMarchGeom g) {
  const int lane = threadIdx.x;   // 0..63
  const int wv = threadIdx.y;     // wave in block
  int t = xcd_remap(blockIdx.x, g.nblocks);
  const int tx = t % g.ntx;
  t /= g.ntx;
  const int ty = t % g.nty;
  const int tz = t / g.nty;

  const int x = box.b0 + tx * 128 + lane * 2;
  const int yw = box.b1 + (ty * WY + wv) * RY;  // first row of this wave
  const int zb = box.b2 + tz * g.zc;
  const int ze = min(zb + g.zc, box.e2);
  if (yw >= box.e1) return;  // wave-uniform
  const bool va = x < box.e0, vb = x + 1 < box.e0;
  // right neighbour of b comes from lane+1 unless that lane is past the box
  const bool rload = vb && (lane == 63 || x + 2 >= box.e0);
  const bool lload = va && lane == 0;

  int yr[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) yr[r] = min(yw + r, box.e1);  // clamped rows re-read the upper halo row
  const int yhm = yw - 1, yhp = min(yw + RY, box.e1);
  const int xs = va ? x : box.b0;  // safe column for idle lanes (never stored)

  d2 um[RY], uc[RY], up[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    um[r] = load2(u + lidx(lu, xs, yr[r], zb - 1));
    uc[r] = load2(u + lidx(lu, xs, yr[r], zb));
  }
  for (int z = zb; z < ze; ++z) {
    d2 f[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      up[r] = load2(u + lidx(lu, xs, yr[r], z + 1));
      if (MODE != EXAMG_APPLY) f[r] = load2(rhs + lidx(lf, xs, yr[r], z));
    }
    const d2 hm = load2(u + lidx(lu, xs, yhm, z));
    const d2 hp = load2(u + lidx(lu, xs, yhp, z));
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      // wavefront-level x-halo exchange
      double xl = __shfl_up(uc[r].y, 1);
      double xr = __shfl_down(uc[r].x, 1);
      if (lload) xl = u[lidx(lu, x - 1, yr[r], z)];
      if (rload) xr = u[lidx(lu, x + 2, yr[r], z)];
      const d2 ym = (r == 0) ? hm : uc[r == 0 ? 0 : r - 1];
      const d2 yp = (r == RY - 1) ? hp : uc[r == RY - 1 ? r : r + 1];
      const double acc_a = conv7<ORDER>(k, uc[r].x, xl, uc[r].y, ym.x, yp.x, um[r].x, up[r].x);
      const double acc_b = conv7<ORDER>(k, uc[r].y, uc[r].x, xr, ym.y, yp.y, um[r].y, up[r].y);
      d2 o;
      o.x = finish<MODE>(uc[r].x, acc_a, f[r].x, w);
      o.y = finish<MODE>(uc[r].y, acc_b, f[r].y, w);
      if (yw + r < box.e1) {
        double *q = dst + lidx(ld, x, yw + r, z);
        if (vb) {
          if (NT) {
            __builtin_nontemporal_store(o.x, q);
            __builtin_nontemporal_store(o.y, q + 1);
          } else {
            store2(q, o);
          }
        } else if (va) {
          q[0] = o.x;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      um[r] = uc[r];
      uc[r] = up[r];
    }
  }
}

// Which canonical 7-point entry order (if any) does this stencil use?  -1: none.
static int order7(const examg_stencil_t *st) {
  static const int o0[7][3] = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
  static const int o1[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  if (st->nent != 7 || st->cfield) return -1;
  bool m0 = true, m1 = true;
  for (int k = 0; k < 7; ++k)
    for (int d = 0; d < 3; ++d) {
      m0 = m0 && st->off[k][d] == o0[k][d];
      m1 = m1 && st->off[k][d] == o1[k][d];
    }
  return m0 ? 0 : (m1 ? 1 : -1);
}

static int g_force_generic = 0;  // test hook: examg_debug_force_generic

template <int MODE, int ORDER>
static void launch_zmarch(const LayoutDev &lu, const double *u, const LayoutDev &lf, const double *rhs, const LayoutDev &ld,
                          double *dst, const Coef7 &k, double w, const Box &box, hipStream_t s) {
  constexpr int RY = 2, WY = 4;
  ZMarchGeom g;
  g.ntx = (box.n0() + 127) / 128;
  g.nty = (box.n1() + RY * WY - 1) / (RY * WY);
  // enough z chunks for >= ~2048 workgroups, but at least 16 planes per chunk
  int zc = box.n2();
  const int want = 2048;
  const int xy = g.ntx * g.nty;
  int ntz = (want + xy - 1) / xy;
  if (ntz < 1) ntz = 1;
  zc = (box.n2() + ntz - 1) / ntz;
  if (zc < 16) zc = 16;
  if (zc > box.n2()) zc = box.n2();
  g.zc = zc;
  g.ntz = (box.n2() + zc - 1) / zc;
  g.nblocks = g.ntx * g.nty * g.ntz;
  dim3 block(64, WY, 1), grid(g.nblocks, 1, 1);
  hipLaunchKernelGGL((k_stencil7_zmarch<MODE, ORDER, RY, WY, false>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, k, w, box, g);
}

}  // namespace examg

using namespace examg;

extern "C" int examg_debug_force_generic(int on) {
  const int old = g_force_generic;
  g_force_generic = on;
  return old;
}

extern "C" int examg_stencil_op(int mode, const examg_layout_t *lu_, const double *u, const examg_layout_t *lf_,
                                const double *rhs, const examg_layout_t *ld_, double *dst, const examg_stencil_t *st,
                                double w, int colour, const int32_t *begin, const int32_t *end, examg_stream_t stream) {
  if (!lu_ || !u || !ld_ || !dst || !st || !begin || !end) { set_error("examg_stencil_op: null argument"); return 1; }
  if (mode < 0 || mode > 2) { set_error("examg_stencil_op: bad mode %d", mode); return 1; }
  if (mode != EXAMG_APPLY && (!rhs || !lf_)) { set_error("examg_stencil_op: rhs required for mode %d", mode); return 1; }
  if (st->nent < 1 || st->nent > EXAMG_MAX_ENTRIES) { set_error("examg_stencil_op: nent %d out of range", st->nent); return 1; }
  if (colour > 1) { set_error("examg_stencil_op: colour must be -1, 0 or 1"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;  // empty iteration space (e.g. coarsest level of one fragment with minLevel 0)
  int reach = 0;
  for (int k = 0; k < st->nent; ++k)
    for (int d = 0; d < 3; ++d) reach = reach > abs(st->off[k][d]) ? reach : abs(st->off[k][d]);
  if (!box_inside(lu_, box, reach)) { set_error("examg_stencil_op: box + stencil reach leaves the u allocation"); return 1; }
  if (!box_inside(ld_, box, 0)) { set_error("examg_stencil_op: box leaves the dst allocation"); return 1; }
  if (mode != EXAMG_APPLY && !box_inside(lf_, box, 0)) { set_error("examg_stencil_op: box leaves the rhs allocation"); return 1; }
  if (st->cfield && !box_inside(&st->clayout, box, 0)) { set_error("examg_stencil_op: box leaves the coefficient allocation"); return 1; }
  if (u == dst && colour < 0) { set_error("examg_stencil_op: in-place update needs a colour"); return 1; }

  const LayoutDev lu = make_layout(lu_), ld = make_layout(ld_);
  const LayoutDev lf = lf_ ? make_layout(lf_) : lu;
  hipStream_t s = (hipStream_t)stream;

  const int ord = order7(st);
  if (!g_force_generic && lu_->nd == 3 && ord >= 0 && colour < 0 && box.n0() >= 64) {
    Coef7 k;
    for (int i = 0; i < 7; ++i) k.c[i] = st->coef[i];
#define EXAMG_ZM(M, O) launch_zmarch<M, O>(lu, u, lf, rhs, ld, dst, k, w, box, s)
    if (ord == 0) {
      if (mode == EXAMG_APPLY) EXAMG_ZM(EXAMG_APPLY, 0);
      else if (mode == EXAMG_RESIDUAL) EXAMG_ZM(EXAMG_RESIDUAL, 0);
      else EXAMG_ZM(EXAMG_SMOOTH, 0);
    } else {
      if (mode == EXAMG_APPLY) EXAMG_ZM(EXAMG_APPLY, 1);
      else if (mode == EXAMG_RESIDUAL) EXAMG_ZM(EXAMG_RESIDUAL, 1);
      else EXAMG_ZM(EXAMG_SMOOTH, 1);
    }
#undef EXAMG_ZM
    EXAMG_CHECK_LAUNCH("k_stencil7_zmarch");
    return 0;
  }

  StencilDev sd;
  sd.nent = st->nent;
  sd.diag = st->diag;
  for (int k = 0; k < st->nent; ++k) {
    sd.uo[k] = st->off[k][0] + lu.s1 * st->off[k][1] + lu.s2 * st->off[k][2];
    sd.coef[k] = st->coef[k];
  }
  sd.cfield = st->cfield;
  LayoutDev lc = lu;
  sd.cplane = 0;
  if (st->cfield) {
    lc = make_layout(&st->clayout);
    sd.cplane = lc.size;
  }
  const int row_w = colour >= 0 ? (box.n0() + 1) / 2 : box.n0();
  const long long total = (long long)row_w * box.n1() * box.n2();
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  dim3 grid((unsigned)nb), block(256);
  if (mode == EXAMG_APPLY)
    hipLaunchKernelGGL((k_stencil_generic<EXAMG_APPLY>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, lc, sd, w, colour, box, row_w);
  else if (mode == EXAMG_RESIDUAL)
    hipLaunchKernelGGL((k_stencil_generic<EXAMG_RESIDUAL>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, lc, sd, w, colour, box, row_w);
  else
    hipLaunchKernelGGL((k_stencil_generic<EXAMG_SMOOTH>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, lc, sd, w, colour, box, row_w);
  EXAMG_CHECK_LAUNCH("k_stencil_generic");
  return 0;
}

extern "C" int examg_jacobi(const examg_layout_t *lu, const double *u, double *u_next, const examg_layout_t *lf,
                            const double *rhs, const examg_stencil_t *st, double w, const int32_t *begin,
                            const int32_t *end, examg_stream_t stream) {
  return examg_stencil_op(EXAMG_SMOOTH, lu, u, lf, rhs, lu, u_next, st, w, -1, begin, end, stream);
}

extern "C" int examg_rbgs_colour(const examg_layout_t *lu, double *u, const examg_layout_t *lf, const double *rhs,
                                 const examg_stencil_t *st, double w, int colour, const int32_t *begin,
                                 const int32_t *end, examg_stream_t stream) {
  if (colour != 0 && colour != 1) { set_error("examg_rbgs_colour: colour must be 0 or 1"); return 1; }
  return examg_stencil_op(EXAMG_SMOOTH, lu, u, lf, rhs, lu, u, st, w, colour, begin, end, stream);
}

extern "C" int examg_residual(const examg_layout_t *lu, const double *u, const examg_layout_t *lf, const double *rhs,
                              const examg_layout_t *lr, double *res, const examg_stencil_t *st, const int32_t *begin,
                              const int32_t *end, examg_stream_t stream) {
  return examg_stencil_op(EXAMG_RESIDUAL, lu, u, lf, rhs, lr, res, st, 0.0, -1, begin, end, stream);
}
