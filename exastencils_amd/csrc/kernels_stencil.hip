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

// shipped configuration of the fast path, from the sweeps of tools/tune_jacobi.py on MI355X (gpurun_out/tune*.log,
// summary in DESIGN.md): 2 rows per wave, 4 waves per workgroup, non-temporal stores (+12 %), loads of the next
// plane in flight while this one is computed, ~1024 workgroups, z march, plain tile order.
#ifndef EXAMG_ZM_RY
#define EXAMG_ZM_RY 2
#endif
#ifndef EXAMG_ZM_WY
#define EXAMG_ZM_WY 4
#endif
#ifndef EXAMG_ZM_NT
#define EXAMG_ZM_NT true
#endif
#ifndef EXAMG_ZM_MY
#define EXAMG_ZM_MY false
#endif
#ifndef EXAMG_ZM_PF
#define EXAMG_ZM_PF 1
#endif

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


// Stencil fields with a fixed entry count, fully unrolled: all NENT coefficient loads and all NENT u loads of a point are
// in flight together -- the run-time loop of the generic kernel keeps two loads in flight per thread, too few to cover the
// HBM latency at 8 + 8*NENT bytes per point.  Same products in the same order as the generic kernel.  Measured at 512^3,
// 27 entries (tools/varcoeff_times.py): 8.6 -> 7.3 ms per Jacobi step; non-temporal coefficient loads 7.5 ms (not used).
struct UOffsets {
  long long o[EXAMG_MAX_ENTRIES];
};

template <int MODE, int NENT>
__global__ void __launch_bounds__(256)
k_stencilfield_unrolled(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs, LayoutDev ld,
                        double *__restrict__ dst, LayoutDev lc, const double *__restrict__ cf, long long cplane, UOffsets uo, int diag,
                        double w, Box box) {
  const int n0 = box.n0();
  const long long rows = (long long)box.n1() * box.n2();
  const long long total = rows * n0;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long row = t / n0;
    const int i0 = box.b0 + (int)(t - row * n0);
    const int i1 = box.b1 + (int)(row % box.n1());
    const int i2 = box.b2 + (int)(row / box.n1());
    const long long iu = lidx(lu, i0, i1, i2), ic = lidx(lc, i0, i1, i2);
    double c[NENT], v[NENT];
#pragma unroll
    for (int k = 0; k < NENT; ++k) c[k] = cf[ic + k * cplane];
#pragma unroll
    for (int k = 0; k < NENT; ++k) v[k] = u[iu + uo.o[k]];
    double acc = c[0] * v[0];
#pragma unroll
    for (int k = 1; k < NENT; ++k) acc = acc + c[k] * v[k];
    if (MODE == EXAMG_SMOOTH) {
      // the centre entry comes first (dispatch condition): c[0] is the diagonal, v[0] the point's own value; a run-time
      // index into c[] would push the whole array to scratch memory
      const double ww = (1.0 / c[0]) * w;
      acc = v[0] + ww * (rhs[lidx(lf, i0, i1, i2)] - acc);
    }
    if (MODE == EXAMG_RESIDUAL) acc = rhs[lidx(lf, i0, i1, i2)] - acc;
    dst[lidx(ld, i0, i1, i2)] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// 3-D 7-point constant-coefficient z-marching kernel
// ---------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ double finish(double u, double acc, double f, double w) {
  if (MODE == EXAMG_APPLY) return acc;
  if (MODE == EXAMG_RESIDUAL) return f - acc;
  return u + w * (f - acc);
}

struct ZMarchGeom {
  int colour;         // COL kernels: the colour to update
  int ntx, ntt, ntm;  // tiles per dim: x, tile-row dim T, march dim M
  int mc;             // planes (or rows) per march chunk
  int nblocks;        // ntx * ntt * ntm
  int remap;          // XCD-aware tile order on/off
};

// XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (blocks b and b+8 share one
// L2), so give each XCD a contiguous run of tiles -- adjacent tiles then share halo rows in L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int per = nblocks >> 3;
  const int full = per << 3;
  if (bid >= full) return bid;  // tail blocks keep their id
  return (bid & 7) * per + (bid >> 3);
}

// Tile = 128 x-points (2 per lane) by RY rows per wave, WY waves per workgroup, marching along M.
//   MY = false: rows are y, march is z (register pipeline holds u[z-1], u[z], u[z+1]);
//   MY = true : rows are z, march is y (consecutive steps are 1 row = ~4 KB apart: TLB-friendly).
// COL: red-black half sweep in place (dst == u): of the two points a lane holds per row exactly one has
//   (i0+i1+i2) % 2 == g.colour; it is updated, the other is written back unchanged (whole 16-byte stores).
//   Race-free: only values of the other colour (never written in this sweep) and the lane's own centre are used.
template <bool ALIAS> struct Ptr { typedef double *__restrict__ out; typedef const double *__restrict__ in; };
template <> struct Ptr<true> { typedef double *out; typedef const double *in; };

template <int MODE, int ORDER, int RY, int WY, bool NT, bool MY, int PF, bool REV, bool COL>
__global__ void __launch_bounds__(64 * WY)
k_stencil7_zmarch(LayoutDev lu, typename Ptr<COL>::in u, LayoutDev lf, const double *__restrict__ rhs, LayoutDev ld,
                  typename Ptr<COL>::out dst, Coef7 k, double w, Box box, ZMarchGeom g) {
  const int lane = threadIdx.x;   // 0..63
  const int wv = threadIdx.y;     // wave in block
  int t = g.remap ? xcd_remap(blockIdx.x, g.nblocks) : (int)blockIdx.x;
  if (REV) t = g.nblocks - 1 - t;  // backward sweep: last tiles first (see launch_zmarch: Infinity-Cache reuse)
  const int tx = t % g.ntx;
  t /= g.ntx;
  const int tt = t % g.ntt;
  const int tm = t / g.ntt;

  const int bT = MY ? box.b2 : box.b1, eT = MY ? box.e2 : box.e1;
  const int bM = MY ? box.b1 : box.b2, eM = MY ? box.e1 : box.e2;
  const long long uT = MY ? lu.s2 : lu.s1, uM = MY ? lu.s1 : lu.s2;
  const long long fT = MY ? lf.s2 : lf.s1, fM = MY ? lf.s1 : lf.s2;
  const long long dT = MY ? ld.s2 : ld.s1, dM = MY ? ld.s1 : ld.s2;

  const int x = box.b0 + tx * 128 + lane * 2;
  const int rw = bT + (tt * WY + wv) * RY;  // first row of this wave
  const int mb = bM + tm * g.mc;
  const int me = min(mb + g.mc, eM);
  if (rw >= eT) return;  // wave-uniform
  const bool va = x < box.e0, vb = x + 1 < box.e0;
  // right neighbour of b comes from lane+1 unless that lane is past the box
  const bool rload = vb && (lane == 63 || x + 2 >= box.e0);
  const bool lload = va && lane == 0;
  const int xs = va ? x : box.b0;  // safe column for idle lanes (never stored)

  const double *ur[RY];
  const double *fr[RY];
  double *dr[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    const int row = min(rw + r, eT);  // clamped rows re-read the upper halo row
    ur[r] = u + lu.origin + xs + uT * row;
    fr[r] = rhs + lf.origin + xs + fT * row;
    dr[r] = dst + ld.origin + xs + dT * row;
  }
  const double *uhm = u + lu.origin + xs + uT * (rw - 1);
  const double *uhp = u + lu.origin + xs + uT * min(rw + RY, eT);

  constexpr int sg = REV ? -1 : 1;         // march direction
  const int m0 = REV ? me - 1 : mb;        // first plane; step q handles plane m0 + sg*q
  const int cnt = me - mb;
  d2 um[RY], uc[RY];                       // planes m - sg and m of the own rows
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    um[r] = load2(ur[r] + uM * (m0 - sg));
    uc[r] = load2(ur[r] + uM * m0);
  }
  // one pipeline stage = everything step m needs from memory: u[m+1] and rhs[m] of the own rows, the two
  // halo rows of plane m.  With PF the loads of step m+1 are issued before step m is computed.
  struct Stage {
    d2 up[RY], f[RY], hm, hp;
  };
  auto load_stage = [&](Stage &st, int q) {
    const int m = m0 + sg * q;
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      st.up[r] = load2(ur[r] + uM * (m + sg));
      if (MODE != EXAMG_APPLY) st.f[r] = load2(fr[r] + fM * m);
    }
    st.hm = load2(uhm + uM * m);
    st.hp = load2(uhp + uM * m);
  };
  auto compute = [&](const Stage &st, int q) {
    const int m = m0 + sg * q;
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      // wavefront-level x-halo exchange
      double xl = lane_below(uc[r].y);
      double xr = lane_above(uc[r].x);
      if (lload) xl = ur[r][uM * m - 1];
      if (rload) xr = ur[r][uM * m + 2];
      const d2 tm_ = (r == 0) ? st.hm : uc[r == 0 ? 0 : r - 1];
      const d2 tp_ = (r == RY - 1) ? st.hp : uc[r == RY - 1 ? r : r + 1];
      const d2 mm_ = REV ? st.up[r] : um[r];   // plane m-1 and m+1 along the march dimension
      const d2 mp_ = REV ? um[r] : st.up[r];
      d2 o;
      if (COL) {
        // the point of this lane's pair that carries the colour: a if (x + row + m) has that parity, else b; x = b0 +
        // 128*tx + 2*lane, so the parity is the same in every lane: scalar branch, one convolution, one lane exchange
        o = uc[r];
        if (((box.b0 + rw + r + m) & 1) == g.colour) {
          const double acc = MY ? conv7<ORDER>(k, uc[r].x, xl, uc[r].y, mm_.x, mp_.x, tm_.x, tp_.x)
                                : conv7<ORDER>(k, uc[r].x, xl, uc[r].y, tm_.x, tp_.x, mm_.x, mp_.x);
          o.x = finish<MODE>(uc[r].x, acc, st.f[r].x, w);
        } else {
          const double acc = MY ? conv7<ORDER>(k, uc[r].y, uc[r].x, xr, mm_.y, mp_.y, tm_.y, tp_.y)
                                : conv7<ORDER>(k, uc[r].y, uc[r].x, xr, tm_.y, tp_.y, mm_.y, mp_.y);
          o.y = finish<MODE>(uc[r].y, acc, st.f[r].y, w);
        }
      } else {
        double acc_a, acc_b;
        if (MY) {
          acc_a = conv7<ORDER>(k, uc[r].x, xl, uc[r].y, mm_.x, mp_.x, tm_.x, tp_.x);
          acc_b = conv7<ORDER>(k, uc[r].y, uc[r].x, xr, mm_.y, mp_.y, tm_.y, tp_.y);
        } else {
          acc_a = conv7<ORDER>(k, uc[r].x, xl, uc[r].y, tm_.x, tp_.x, mm_.x, mp_.x);
          acc_b = conv7<ORDER>(k, uc[r].y, uc[r].x, xr, tm_.y, tp_.y, mm_.y, mp_.y);
        }
        o.x = finish<MODE>(uc[r].x, acc_a, st.f[r].x, w);
        o.y = finish<MODE>(uc[r].y, acc_b, st.f[r].y, w);
      }
      if (rw + r < eT) {
        double *q = dr[r] + dM * m;
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
      uc[r] = st.up[r];
    }
  };
  // software pipeline of depth PF: the loads of step m+PF are in flight while step m is computed
  Stage st[PF + 1];
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (j < cnt) load_stage(st[j], j);
  int q = 0;
  while (q < cnt) {
#pragma unroll
    for (int j = 0; j <= PF; ++j) {
      if (q < cnt) {
        if (q + PF < cnt) load_stage(st[(j + PF) % (PF + 1)], q + PF);
        compute(st[j], q);
        ++q;
      }
    }
  }
}

static int g_force_generic = 0;  // test hook: examg_debug_force_generic
static int g_sf27_unrolled = 1;  // examg_debug_sf27(0): 27-entry stencil fields on the generic kernel

// kernels_stencilfield.hip
bool stencilfield7_ok(const examg_layout_t *lu, const examg_stencil_t *st, const Box &box, int colour);
int launch_stencilfield7(int mode, const LayoutDev &lu, const double *u, const LayoutDev &lf, const double *rhs, const LayoutDev &ld,
                         double *dst, const LayoutDev &lc, const double *cf, double w, const Box &box, hipStream_t s);


// Tuning knobs (examg_debug_tune); the defaults are the measured best on MI355X at 512^3.
struct Tune {
  int ry = 2, wy = 4, nt = 1, my = 0, pf = 1, remap = 0, blocks = 1024, minchunk = 16, dir = 0;
};
static Tune g_tune;
static bool g_dir_toggle = false;

template <int MODE, int ORDER, int RY, int WY, bool NT, bool MY, int PF>
static void launch_zmarch_t(const LayoutDev &lu, const double *u, const LayoutDev &lf, const double *rhs, const LayoutDev &ld,
                            double *dst, const Coef7 &k, double w, const Box &box, hipStream_t s, int colour = -1) {
  const int nT = MY ? box.n2() : box.n1(), nM = MY ? box.n1() : box.n2();
  ZMarchGeom g;
  g.ntx = (box.n0() + 127) / 128;
  g.ntt = (nT + RY * WY - 1) / (RY * WY);
  const int xy = g.ntx * g.ntt;
  int ntm = (g_tune.blocks + xy - 1) / xy;
  if (ntm < 1) ntm = 1;
  int mc = (nM + ntm - 1) / ntm;
  if (mc < g_tune.minchunk) mc = g_tune.minchunk;
  if (mc > nM) mc = nM;
  g.mc = mc;
  g.ntm = (nM + mc - 1) / mc;
  g.nblocks = g.ntx * g.ntt * g.ntm;
  g.remap = g_tune.remap;
  g.colour = colour;
  dim3 block(64, WY, 1), grid(g.nblocks, 1, 1);
  if (colour >= 0) {
    if (MODE == EXAMG_SMOOTH)
      hipLaunchKernelGGL((k_stencil7_zmarch<EXAMG_SMOOTH, ORDER, RY, WY, NT, MY, PF, false, true>), grid, block, 0, s, lu, u, lf,
                         rhs, ld, dst, k, w, box, g);
    return;
  }
  // Sweep direction (tuning knob `dir`; -1 alternates from launch to launch so that a sweep starts where the
  // previous one ended).  Measured on MI355X at 256^3 and 512^3: no gain -- the 256 MiB Infinity Cache does not
  // hold streamed data long enough -- so the shipped default is always forward.  Results do not depend on it.
  bool rev;
  if (g_tune.dir < 0) { rev = g_dir_toggle; g_dir_toggle = !g_dir_toggle; }
  else rev = g_tune.dir != 0;
  if (rev) hipLaunchKernelGGL((k_stencil7_zmarch<MODE, ORDER, RY, WY, NT, MY, PF, true, false>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, k, w, box, g);
  else hipLaunchKernelGGL((k_stencil7_zmarch<MODE, ORDER, RY, WY, NT, MY, PF, false, false>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, k, w, box, g);
}

#ifdef EXAMG_TUNE
// every (RY, WY, NT, MY) combination, for tools/tune_jacobi.py
template <int MODE, int ORDER, int RY, int WY>
static void launch_zmarch_ntmy(const LayoutDev &lu, const double *u, const LayoutDev &lf, const double *rhs, const LayoutDev &ld,
                               double *dst, const Coef7 &k, double w, const Box &box, hipStream_t s) {
#define EXAMG_L(NT_, MY_, PF_) launch_zmarch_t<MODE, ORDER, RY, WY, NT_, MY_, PF_>(lu, u, lf, rhs, ld, dst, k, w, box, s)
  switch (g_tune.pf) {
    case 1: if (g_tune.nt) EXAMG_L(true, false, 1); else EXAMG_L(false, false, 1); break;
    case 2: if (g_tune.nt) EXAMG_L(true, false, 2); else EXAMG_L(false, false, 2); break;
    case 3: if (g_tune.nt) EXAMG_L(true, false, 3); else EXAMG_L(false, false, 3); break;
    case 4: if (g_tune.nt) EXAMG_L(true, false, 4); else EXAMG_L(false, false, 4); break;
    default:
      if (g_tune.nt) { if (g_tune.my) EXAMG_L(true, true, 0); else EXAMG_L(true, false, 0); }
      else { if (g_tune.my) EXAMG_L(false, true, 0); else EXAMG_L(false, false, 0); }
  }
#undef EXAMG_L
}
template <int MODE, int ORDER, int RY>
static void launch_zmarch_wy(const LayoutDev &lu, const double *u, const LayoutDev &lf, const double *rhs, const LayoutDev &ld,
                             double *dst, const Coef7 &k, double w, const Box &box, hipStream_t s) {
  switch (g_tune.wy) {
    case 1: launch_zmarch_ntmy<MODE, ORDER, RY, 1>(lu, u, lf, rhs, ld, dst, k, w, box, s); break;
    case 2: launch_zmarch_ntmy<MODE, ORDER, RY, 2>(lu, u, lf, rhs, ld, dst, k, w, box, s); break;
    case 8: launch_zmarch_ntmy<MODE, ORDER, RY, 8>(lu, u, lf, rhs, ld, dst, k, w, box, s); break;
    default: launch_zmarch_ntmy<MODE, ORDER, RY, 4>(lu, u, lf, rhs, ld, dst, k, w, box, s); break;
  }
}
template <int MODE, int ORDER>
static void launch_zmarch(const LayoutDev &lu, const double *u, const LayoutDev &lf, const double *rhs, const LayoutDev &ld,
                          double *dst, const Coef7 &k, double w, const Box &box, hipStream_t s, int colour = -1) {
  if (colour >= 0) {
    launch_zmarch_t<MODE, ORDER, EXAMG_ZM_RY, EXAMG_ZM_WY, EXAMG_ZM_NT, EXAMG_ZM_MY, EXAMG_ZM_PF>(lu, u, lf, rhs, ld, dst, k, w, box, s, colour);
    return;
  }
  switch (g_tune.ry) {
    case 1: launch_zmarch_wy<MODE, ORDER, 1>(lu, u, lf, rhs, ld, dst, k, w, box, s); break;
    case 4: launch_zmarch_wy<MODE, ORDER, 4>(lu, u, lf, rhs, ld, dst, k, w, box, s); break;
    default: launch_zmarch_wy<MODE, ORDER, 2>(lu, u, lf, rhs, ld, dst, k, w, box, s); break;
  }
}
#else
template <int MODE, int ORDER>
static void launch_zmarch(const LayoutDev &lu, const double *u, const LayoutDev &lf, const double *rhs, const LayoutDev &ld,
                          double *dst, const Coef7 &k, double w, const Box &box, hipStream_t s, int colour = -1) {
  launch_zmarch_t<MODE, ORDER, EXAMG_ZM_RY, EXAMG_ZM_WY, EXAMG_ZM_NT, EXAMG_ZM_MY, EXAMG_ZM_PF>(lu, u, lf, rhs, ld, dst, k, w, box, s, colour);
}
#endif

}  // namespace examg

using namespace examg;

extern "C" int examg_debug_sf27(int unrolled) {
  g_sf27_unrolled = unrolled;
  return 0;
}

extern "C" int examg_debug_force_generic(int on) {
  const int old = g_force_generic;
  g_force_generic = on;
  return old;
}

// Tuning hook for tools/tune_jacobi.py: key in {ry, wy, nt, my, remap, blocks, minchunk}.  ry/wy/nt/my only
// take effect in a build with -DEXAMG_TUNE (all template combinations instantiated).
extern "C" int examg_debug_tune(const char *key, int value) {
  if (!key) return 1;
  if (!strcmp(key, "ry")) g_tune.ry = value;
  else if (!strcmp(key, "wy")) g_tune.wy = value;
  else if (!strcmp(key, "nt")) g_tune.nt = value;
  else if (!strcmp(key, "my")) g_tune.my = value;
  else if (!strcmp(key, "pf")) g_tune.pf = value;
  else if (!strcmp(key, "dir")) g_tune.dir = value;
  else if (!strcmp(key, "remap")) g_tune.remap = value;
  else if (!strcmp(key, "blocks")) g_tune.blocks = value;
  else if (!strcmp(key, "minchunk")) g_tune.minchunk = value;
  else return 1;
  return 0;
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
  const int reach = stencil_reach(st);
  if (!box_inside(lu_, box, reach)) { set_error("examg_stencil_op: box + stencil reach leaves the u allocation"); return 1; }
  if (!box_inside(ld_, box, 0)) { set_error("examg_stencil_op: box leaves the dst allocation"); return 1; }
  if (mode != EXAMG_APPLY && !box_inside(lf_, box, 0)) { set_error("examg_stencil_op: box leaves the rhs allocation"); return 1; }
  if (st->cfield && !box_inside(&st->clayout, box, 0)) { set_error("examg_stencil_op: box leaves the coefficient allocation"); return 1; }
  if (u == dst && colour < 0) { set_error("examg_stencil_op: in-place update needs a colour"); return 1; }

  const LayoutDev lu = make_layout(lu_), ld = make_layout(ld_);
  const LayoutDev lf = lf_ ? make_layout(lf_) : lu;
  hipStream_t s = (hipStream_t)stream;

  const int ord = canonical_order7(st);
  const bool colour_ok = colour < 0 || (mode == EXAMG_SMOOTH && u == dst && memcmp(lu_, ld_, sizeof(*lu_)) == 0);
  if (!g_force_generic && lu_->nd == 3 && ord >= 0 && colour_ok && box.n0() >= 64) {
    Coef7 k;
    for (int i = 0; i < 7; ++i) k.c[i] = st->coef[i];
#define EXAMG_ZM(M, O) launch_zmarch<M, O>(lu, u, lf, rhs, ld, dst, k, w, box, s, colour)
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

  if (!g_force_generic && stencilfield7_ok(lu_, st, box, colour)) {
    const LayoutDev lcf = make_layout(&st->clayout);
    return launch_stencilfield7(mode, lu, u, lf, rhs, ld, dst, lcf, st->cfield, w, box, s);
  }

  if (!g_force_generic && g_sf27_unrolled && st->cfield && st->nent == 27 && colour < 0 && st->diag == 0 && st->off[0][0] == 0 &&
      st->off[0][1] == 0 && st->off[0][2] == 0) {
    const LayoutDev lc27 = make_layout(&st->clayout);
    UOffsets uo;
    for (int k = 0; k < 27; ++k) uo.o[k] = st->off[k][0] + lu.s1 * st->off[k][1] + lu.s2 * st->off[k][2];
    long long nb27 = (box.count() + 255) / 256;
    if (nb27 > 16384) nb27 = 16384;
    dim3 grid27((unsigned)nb27), block27(256);
#define EXAMG_SF27(M) hipLaunchKernelGGL((k_stencilfield_unrolled<M, 27>), grid27, block27, 0, s, lu, u, lf, rhs, ld, dst, lc27, st->cfield, lc27.size, uo, 0, w, box)
    if (mode == EXAMG_APPLY) EXAMG_SF27(EXAMG_APPLY);
    else if (mode == EXAMG_RESIDUAL) EXAMG_SF27(EXAMG_RESIDUAL);
    else EXAMG_SF27(EXAMG_SMOOTH);
#undef EXAMG_SF27
    EXAMG_CHECK_LAUNCH("k_stencilfield_unrolled");
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
