// Shared host/device helpers of libexamg (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/examg.h"

namespace examg {

// Flattened layout facts a kernel needs: strides and the offset of iterator (0,0,0).
struct LayoutDev {
  int tot0, tot1, tot2;
  int ref0, ref1, ref2;
  long long s1, s2;    // strides of dim 1, dim 2 (dim 0 stride = 1); under the colour split: strides within a half array
  long long size;      // doubles per (scalar) field
  long long origin;    // linear index of iterator (0,0,0) (plain layouts)
  long long half;      // colour split (EXAMG_LAYOUT_SPLIT_X): doubles per half array, 0 for a plain layout
};

static inline int lay_tot(const examg_layout_t *l, int d) {
  return l->pad_l[d] + l->ghost_l[d] + l->dup_l[d] + l->inner[d] + l->dup_r[d] + l->ghost_r[d] + l->pad_r[d];
}

static inline bool lay_split(const examg_layout_t *l) { return l->transform == EXAMG_LAYOUT_SPLIT_X; }

static inline LayoutDev make_layout(const examg_layout_t *l) {
  LayoutDev d;
  d.tot0 = lay_tot(l, 0);
  d.tot1 = lay_tot(l, 1);
  d.tot2 = lay_tot(l, 2);
  d.ref0 = l->pad_l[0] + l->ghost_l[0];
  d.ref1 = l->pad_l[1] + l->ghost_l[1];
  d.ref2 = l->pad_l[2] + l->ghost_l[2];
  if (lay_split(l)) {
    // [x, y, z] => [x / 2, y, z, x % 2] on array indices (layoutTransformation/ir/IR_LayoutTransformStatement.scala): extents
    // ceil(TOTx / 2), TOTy, TOTz, 2 -- first index fastest
    d.s1 = (d.tot0 + 1) / 2;
    d.s2 = d.s1 * d.tot1;
    d.half = d.s2 * d.tot2;
    d.size = 2 * d.half;
    d.origin = 0;
    return d;
  }
  d.s1 = d.tot0;
  d.s2 = (long long)d.tot0 * d.tot1;
  d.size = d.s2 * d.tot2;
  d.origin = d.ref0 + d.s1 * d.ref1 + d.s2 * d.ref2;
  d.half = 0;
  return d;
}

// the index map of an UNTRANSFORMED layout: what the kernels use whose dispatch takes plain layouts only (the transformation test of lidx
// costs registers in kernels that have none to spare: k_stencilfield27_rec went from two waves per SIMD to one with it)
__host__ __device__ static inline long long lidx_plain(const LayoutDev &l, int i0, int i1, int i2) {
  return l.origin + i0 + l.s1 * i1 + l.s2 * i2;
}

__host__ __device__ static inline long long lidx(const LayoutDev &l, int i0, int i1, int i2) {
  if (l.half) {
    const int ax = i0 + l.ref0;
    return (ax >> 1) + l.s1 * (i1 + l.ref1) + l.s2 * (i2 + l.ref2) + (long long)(ax & 1) * l.half;
  }
  return l.origin + i0 + l.s1 * i1 + l.s2 * i2;
}

// iterator-coordinate box
struct Box {
  int b0, b1, b2, e0, e1, e2;
  __host__ __device__ int n0() const { return e0 - b0; }
  __host__ __device__ int n1() const { return e1 - b1; }
  __host__ __device__ int n2() const { return e2 - b2; }
  __host__ __device__ long long count() const {
    return (n0() <= 0 || n1() <= 0 || n2() <= 0) ? 0 : (long long)n0() * n1() * n2();
  }
};

static inline Box make_box(const int32_t *begin, const int32_t *end) {
  Box b{begin[0], begin[1], begin[2], end[0], end[1], end[2]};
  return b;
}

// Does the box (grown by `halo` points, in iterator coords) stay inside the allocation?
static inline bool box_inside(const examg_layout_t *l, const Box &b, int halo) {
  const int bb[3] = {b.b0, b.b1, b.b2}, ee[3] = {b.e0, b.e1, b.e2};
  for (int d = 0; d < 3; ++d) {
    const int ref = l->pad_l[d] + l->ghost_l[d];
    const int h = (d < l->nd) ? halo : 0;
    if (ee[d] <= bb[d]) continue;
    if (bb[d] - h + ref < 0) return false;
    if (ee[d] + h + ref > lay_tot(l, d)) return false;
  }
  return true;
}

void set_error(const char *fmt, ...);
int check_hip(hipError_t e, const char *what);

// one-pass kernels of the launch-bound levels (kernels_small.hip), tried by the entry points when the large-level kernels decline
bool small_two_stage_ok(const examg_layout_t *lu, const examg_layout_t *lf, const examg_stencil_t *st, const Box &box);
int launch_small_two_stage(bool col, int var, const examg_layout_t *lu, const double *u_in, double *u_out, const examg_layout_t *lf, const double *rhs,
                      const examg_stencil_t *st, double w, int first, const Box &box, const examg_layout_t *lc, const double *uc, hipStream_t s);
bool small_residual_restrict_ok(const examg_layout_t *lu, const examg_layout_t *lf, const examg_stencil_t *st, const examg_layout_t *lc, const Box &fb,
                                const Box &cb);
int launch_small_residual_restrict(const examg_layout_t *lu, const double *u, const examg_layout_t *lf, const double *rhs, const examg_layout_t *lc,
                                   double *fc, const examg_stencil_t *st, double scale, const Box &cb, hipStream_t s);

// coloured loop out of place with the other colour carried over (kernels_stencil.hip), for the shell passes of examg_comm.hip
int stencil_colour_passthrough(const examg_layout_t *lu, const double *u, const examg_layout_t *lf, const double *rhs, const examg_layout_t *ld,
                               double *dst, const examg_stencil_t *st, double w, int colour, const int32_t *begin, const int32_t *end, hipStream_t s);

#define EXAMG_CHECK_LAUNCH(name)                                   \
  do {                                                             \
    hipError_t _e = hipGetLastError();                             \
    if (_e != hipSuccess) return examg::check_hip(_e, name);       \
  } while (0)

// Wavefront-level halo exchange: value of the neighbouring lane through DPP wave shifts (one v_mov_b32_dpp per
// dword, no LDS crossbar round trip as with ds_bpermute / __shfl).  Lane 0 (resp. 63) keeps its own value.
__device__ __forceinline__ double lane_below(double v) {   // lane l receives lane l-1
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xF, 0xF, false);   // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_above(double v) {   // lane l receives lane l+1
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xF, 0xF, false);   // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// The same shifts where the end lane's result is never used: it receives 0 (bound_ctrl) instead of its own value, which
// saves the copy that ties the old value to the destination (one v_mov_b32 per dword).
__device__ __forceinline__ double lane_below0(double v) {   // lane l receives lane l-1, lane 0 receives 0.0
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x138, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0x138, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_above0(double v) {   // lane l receives lane l+1, lane 63 receives 0.0
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x130, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0x130, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// ---- shared by the 7-point kernels (kernels_stencil.hip, kernels_twostage.hip, kernels_stencilfield.hip) ----------
typedef double d2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(8))) d2u { double a, b; };  // 16-byte access at 8-byte alignment

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
// non-temporal 16-byte store at 8-byte alignment (global_store_dwordx4 ... nt)
typedef d2 d2_a8 __attribute__((aligned(8)));
__device__ __forceinline__ void store2_nt(double *p, d2 v) { __builtin_nontemporal_store(v, (d2_a8 *)p); }
// 16-byte load where both points may be read, scalar loads at the edges of an allocation / box, 0 outside
__device__ __forceinline__ d2 load2g(const double *p, bool oka, bool okb) {
  d2 r = {0.0, 0.0};
  if (oka && okb) return load2(p);
  if (oka) r.x = p[0];
  else if (okb) r.y = p[1];
  return r;
}

struct Coef7 {
  double c[7];
};

// sum_k c_k * u[i + o_k] folded left to right in entry order
// (Compiler/src/exastencils/stencil/ir/IR_StencilConvolution.scala:65-68).
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

// Which canonical 7-point entry order does a constant stencil use?  -1: none of the two.
static inline int canonical_order7(const examg_stencil_t *st) {
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

static inline int stencil_reach(const examg_stencil_t *st) {
  int reach = 0;
  for (int k = 0; k < st->nent; ++k)
    for (int d = 0; d < 3; ++d) {
      const int a = st->off[k][d] < 0 ? -st->off[k][d] : st->off[k][d];
      reach = reach > a ? reach : a;
    }
  return reach;
}

struct Params4 {
  double v[4];
};
struct Geom {
  double pb0, pb1, pb2, h0, h1, h2;
};
static inline Geom make_geom(const examg_geom_t *g) {
  return Geom{g->pos_begin[0], g->pos_begin[1], g->pos_begin[2], g->h[0], g->h[1], g->h[2]};
}
static inline Params4 make_params(const double *p) {
  Params4 q{{0, 0, 0, 0}};
  if (p) for (int i = 0; i < 4; ++i) q.v[i] = p[i];
  return q;
}

}  // namespace examg
