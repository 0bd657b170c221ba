// 3-D 7-entry stencil FIELD (variable coefficients) on the z-march structure of kernels_stencil.hip (gfx950).
//
// Reference: `StencilField Laplace< LaplaceCoeff => LaplaceStencil >` with `Layout NoCommSF< ColumnVector<Real,7>, Node >`
// (Testing/SISC/3D_VarCoeff.exa4:39-43,60-74; the CUDA CI target Testing/CUDA/3D_VarCoeff.exa4): the convolution is
// sum_k coeff[i][k] * u[i + o_k] in entry order c,+x,-x,+y,-y,+z,-z, coefficients stored with the entry index as the
// slowest array dimension (Compiler/src/exastencils/stencil/ir/IR_StencilConvolution.scala:73-95).
// 24 + 8*7 = 80 algorithmic bytes per update, 72 of them reads: the seven coefficient planes are streamed with one
// 16-byte load per lane and plane, u stays in the register pipeline (u[z-1], u[z], u[z+1]), x-neighbours by DPP.
#include "examg_common.h"

namespace examg {

// rhs / coefficient rows have no ghost layer: the lane holding the last point of a row must not read past it
__device__ __forceinline__ d2 sf_load2g(const double *p, bool both) {
  if (both) return load2(p);
  d2 r;
  r.x = p[0];
  r.y = 0.0;
  return r;
}

struct SFGeom {
  int ntx, nty, ntz, zc, nblocks;
};

template <int MODE, int RY, int WY, int PF>
__global__ void __launch_bounds__(64 * WY)
k_stencilfield7_zmarch(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs, LayoutDev ld,
                       double *__restrict__ dst, LayoutDev lc, const double *__restrict__ cf, double w, Box box, SFGeom g) {
  const int lane = threadIdx.x, wv = threadIdx.y;
  int t = blockIdx.x;
  const int tx = t % g.ntx;
  t /= g.ntx;
  const int ty = t % g.nty;
  const int tz = t / g.nty;
  const int x = box.b0 + tx * 128 + lane * 2;
  const int rw = box.b1 + (ty * WY + wv) * RY;
  const int mb = box.b2 + tz * g.zc;
  const int me = min(mb + g.zc, box.e2);
  if (rw >= box.e1) return;
  const bool va = x < box.e0, vb = x + 1 < box.e0;
  const bool rload = vb && (lane == 63 || x + 2 >= box.e0);
  const bool lload = va && lane == 0;
  const int xs = va ? x : box.b0;
  const long long cplane = lc.size;

  const double *ur[RY], *fr[RY], *cr[RY];
  double *dr[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    const int row = min(rw + r, box.e1);
    const int crow = min(rw + r, box.e1 - 1);   // coefficient and rhs rows exist inside the box only
    ur[r] = u + lu.origin + xs + lu.s1 * row;
    fr[r] = rhs + lf.origin + xs + lf.s1 * crow;
    cr[r] = cf + lc.origin + xs + lc.s1 * crow;
    dr[r] = dst + ld.origin + xs + ld.s1 * row;
  }
  const double *uhm = u + lu.origin + xs + lu.s1 * (rw - 1);
  const double *uhp = u + lu.origin + xs + lu.s1 * min(rw + RY, box.e1);

  d2 um[RY], uc[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    um[r] = load2(ur[r] + lu.s2 * (mb - 1));
    uc[r] = load2(ur[r] + lu.s2 * mb);
  }
  struct Stage {
    d2 up[RY], f[RY], hm, hp, c[RY][7];
  };
  auto load_stage = [&](Stage &st, int m) {
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      st.up[r] = load2(ur[r] + lu.s2 * (m + 1));
      if (MODE != EXAMG_APPLY) st.f[r] = sf_load2g(fr[r] + lf.s2 * m, vb || !va);
#pragma unroll
      for (int k = 0; k < 7; ++k) st.c[r][k] = sf_load2g(cr[r] + lc.s2 * m + cplane * k, vb || !va);
    }
    st.hm = load2(uhm + lu.s2 * m);
    st.hp = load2(uhp + lu.s2 * m);
  };
  auto compute = [&](const Stage &st, int m) {
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      double xl = lane_below(uc[r].y);
      double xr = lane_above(uc[r].x);
      if (lload) xl = ur[r][lu.s2 * m - 1];
      if (rload) xr = ur[r][lu.s2 * m + 2];
      const d2 ym = (r == 0) ? st.hm : uc[r == 0 ? 0 : r - 1];
      const d2 yp = (r == RY - 1) ? st.hp : uc[r == RY - 1 ? r : r + 1];
      const d2(&c)[7] = st.c[r];
      // entries c,+x,-x,+y,-y,+z,-z folded left to right
      double acc_a = c[0].x * uc[r].x;
      acc_a = acc_a + c[1].x * uc[r].y;
      acc_a = acc_a + c[2].x * xl;
      acc_a = acc_a + c[3].x * yp.x;
      acc_a = acc_a + c[4].x * ym.x;
      acc_a = acc_a + c[5].x * st.up[r].x;
      acc_a = acc_a + c[6].x * um[r].x;
      double acc_b = c[0].y * uc[r].y;
      acc_b = acc_b + c[1].y * xr;
      acc_b = acc_b + c[2].y * uc[r].x;
      acc_b = acc_b + c[3].y * yp.y;
      acc_b = acc_b + c[4].y * ym.y;
      acc_b = acc_b + c[5].y * st.up[r].y;
      acc_b = acc_b + c[6].y * um[r].y;
      d2 o;
      if (MODE == EXAMG_APPLY) {
        o.x = acc_a;
        o.y = acc_b;
      } else if (MODE == EXAMG_RESIDUAL) {
        o.x = st.f[r].x - acc_a;
        o.y = st.f[r].y - acc_b;
      } else {
        // ((1.0 / diag) * omega) as written in Testing/SISC/3D_VarCoeff.exa4:145
        o.x = uc[r].x + ((1.0 / c[0].x) * w) * (st.f[r].x - acc_a);
        o.y = uc[r].y + ((1.0 / c[0].y) * w) * (st.f[r].y - acc_b);
      }
      if (rw + r < box.e1) {
        double *q = dr[r] + ld.s2 * m;
        if (vb) {
          __builtin_nontemporal_store(o.x, q);
          __builtin_nontemporal_store(o.y, q + 1);
        } else if (va) {
          q[0] = o.x;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      um[r] = uc[r];
      uc[r] = st.up[r];
    }
  };
  Stage st[PF + 1];
  const int cnt = me - mb;
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (j < cnt) load_stage(st[j], mb + j);
  int q = 0;
  while (q < cnt) {
#pragma unroll
    for (int j = 0; j <= PF; ++j) {
      if (q < cnt) {
        if (q + PF < cnt) load_stage(st[(j + PF) % (PF + 1)], mb + q + PF);
        compute(st[j], mb + q);
        ++q;
      }
    }
  }
}

// tuning hook (examg_debug_stencilfield): 0 = RY1/PF1, 1 = RY2/PF0, 2 = RY2/PF1, 3 = RY1/PF2; -1 = generic kernel.
// tools/varcoeff_times.py at 512^3 (Jacobi, ms): generic 2.64, variant 0 2.31, 1 1.99, 2 2.74, 3 2.44 -> variant 1
static thread_local int g_sf_variant = 1;
static thread_local int g_sf_blocks = -1;       // -1: 2048 workgroups, 16-plane chunks from 2*10^8 points (768^3 7.24 -> 6.95 ms, 1024^3 18.1 -> 17.1 ms)

template <int MODE, int RY, int WY, int PF>
static void launch_sf(const LayoutDev &lu, const double *u, const LayoutDev &lf, const double *rhs, const LayoutDev &ld, double *dst,
                      const LayoutDev &lc, const double *cf, double w, const Box &box, hipStream_t s) {
  SFGeom g;
  g.ntx = (box.n0() + 127) / 128;
  g.nty = (box.n1() + RY * WY - 1) / (RY * WY);
  const int xy = g.ntx * g.nty;
  const int blocks_target = g_sf_blocks > 0 ? g_sf_blocks : (box.count() >= 200000000LL ? (1 << 24) : 2048);
  int ntz = (blocks_target + xy - 1) / xy;
  if (ntz < 1) ntz = 1;
  int zc = (box.n2() + ntz - 1) / ntz;
  if (zc < 16) zc = 16;
  if (zc > box.n2()) zc = box.n2();
  g.zc = zc;
  g.ntz = (box.n2() + zc - 1) / zc;
  g.nblocks = xy * g.ntz;
  hipLaunchKernelGGL((k_stencilfield7_zmarch<MODE, RY, WY, PF>), dim3(g.nblocks), dim3(64, WY, 1), 0, s, lu, u, lf, rhs, ld, dst, lc, cf, w,
                     box, g);
}

template <int MODE>
static void launch_sf_variant(const LayoutDev &lu, const double *u, const LayoutDev &lf, const double *rhs, const LayoutDev &ld,
                              double *dst, const LayoutDev &lc, const double *cf, double w, const Box &box, hipStream_t s) {
  switch (g_sf_variant) {
    case 1: launch_sf<MODE, 2, 4, 0>(lu, u, lf, rhs, ld, dst, lc, cf, w, box, s); break;
    case 2: launch_sf<MODE, 2, 4, 1>(lu, u, lf, rhs, ld, dst, lc, cf, w, box, s); break;
    case 3: launch_sf<MODE, 1, 4, 2>(lu, u, lf, rhs, ld, dst, lc, cf, w, box, s); break;
    default: launch_sf<MODE, 1, 4, 1>(lu, u, lf, rhs, ld, dst, lc, cf, w, box, s); break;
  }
}

// Is this the 7-entry stencil field in the reference's entry order, on a box the z-march kernel can take?
bool stencilfield7_ok(const examg_layout_t *lu, const examg_stencil_t *st, const Box &box, int colour) {
  static const int o1[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  if (g_sf_variant < 0 || lu->nd != 3 || st->nent != 7 || !st->cfield || st->diag != 0 || colour >= 0 || box.n0() < 64) return false;
  if (st->ctransform != EXAMG_CLAYOUT_PLANES) return false;     // transformed coefficient layouts: generic kernel
  if (st->wform != EXAMG_WEIGHT_INV_TIMES) return false;       // `omega / diag(A)`: generic kernel
  for (int k = 0; k < 7; ++k)
    for (int d = 0; d < 3; ++d)
      if (st->off[k][d] != o1[k][d]) return false;
  return true;
}

int launch_stencilfield7(int mode, const LayoutDev &lu, const double *u, const LayoutDev &lf, const double *rhs, const LayoutDev &ld,
                         double *dst, const LayoutDev &lc, const double *cf, double w, const Box &box, hipStream_t s) {
  if (mode == EXAMG_APPLY) launch_sf_variant<EXAMG_APPLY>(lu, u, lf, rhs, ld, dst, lc, cf, w, box, s);
  else if (mode == EXAMG_RESIDUAL) launch_sf_variant<EXAMG_RESIDUAL>(lu, u, lf, rhs, ld, dst, lc, cf, w, box, s);
  else launch_sf_variant<EXAMG_SMOOTH>(lu, u, lf, rhs, ld, dst, lc, cf, w, box, s);
  EXAMG_CHECK_LAUNCH("k_stencilfield7_zmarch");
  return 0;
}

}  // namespace examg

#ifdef EXAMG_DEBUG_HOOKS
extern "C" int examg_debug_stencilfield(int variant, int blocks) {
  examg::g_sf_variant = variant;   // -1 disables the fast path
  examg::g_sf_blocks = blocks > 0 ? blocks : -1;
  return 0;
}
#endif
