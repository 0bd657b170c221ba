// Two Jacobi steps on a 27-entry stencil FIELD in ONE pass over HBM (gfx950): temporal blocking for the operator whose traffic is its
// coefficients -- 216 B of them per point and sweep against 24 B of u, rhs and result (BASELINE.json configs[3]; the reference's
// IR_ContractingLoop, baseExt/ir/IR_ContractingLoop.scala, applied to a stencil-field smoother such as Testing/SISC/3D_VarCoeff.exa4:141-147
// with 27 entries).  Both steps of a point use the SAME 27 coefficients: read once per pair of steps instead of once per step.
//
// Needs the coefficient field under the layout transformation `[x, y, z, i] => [i, x, y, z]` (records, EXAMG_CLAYOUT_ENTRY_FASTEST) and the
// entry order "centre, then the offsets with dz = -1, dz = 0, dz = +1, each group with dy outer and dx inner" (examg_init_helmholtz27
// writes it): marching in z, the second step of plane q-1 is then a running sum whose first 18 terms (centre, dz = -1, dz = 0 of the
// first-step field) are known when the first step of plane q-1 is, and whose last 9 terms (dz = +1) follow one plane later -- the same
// 27 products added in the same order as the one-step kernels (bit-identical to two launches), with 9 coefficients and the partial
// sum carried in registers instead of all 27.
//
// Workgroup: 8 waves, one row of 64 points each (stage-1 rows s = 0..7), marching in z.  The input tile (10 rows x 66 columns) of planes
// q-1, q, q+1 and the stage-1 tile of planes q-1, q live in LDS; a wave's 64 records of plane q (13.5 KiB, one contiguous stream of 27
// eight-byte loads per lane, issued a whole step ahead) are transposed through its private LDS strip.  Outputs: rows 1..6, lanes 1..62.
// Two barriers per plane.  134 KB of LDS: one workgroup (8 waves, ~200 VGPRs each) per CU.
#include "examg_common.h"

namespace examg {

constexpr int S27_NW = 8;              // waves = stage-1 rows per workgroup
constexpr int S27_XO = 62;             // outputs per 64-point window
constexpr int S27_YO = S27_NW - 2;     // output rows per workgroup
constexpr int S27_UR = S27_NW + 2;     // input rows per workgroup
constexpr int S27_UC = 66;             // input columns: x = xw - 1 .. xw + 64

struct S27Geom {
  int ntx, nty, ntz, zc;
  Box box1;   // stage-1 box (contains the output box); points outside keep the input value
};

// canonical entry order (checked by the launcher): entry 0 = centre; k = 1 + 9 (dz + 1) + 3 (dy + 1) + (dx + 1) for dz = -1, the same
// numbering without the centre for dz = 0, and 18 + 3 (dy + 1) + (dx + 1) for dz = +1
__host__ __device__ constexpr int s27_dx(int k) {
  return k == 0 ? 0 : (k <= 9 ? (k - 1) % 3 - 1 : (k <= 17 ? ((k - 10) + ((k - 10) >= 4 ? 1 : 0)) % 3 - 1 : (k - 18) % 3 - 1));
}
__host__ __device__ constexpr int s27_dy(int k) {
  return k == 0 ? 0 : (k <= 9 ? (k - 1) / 3 - 1 : (k <= 17 ? ((k - 10) + ((k - 10) >= 4 ? 1 : 0)) / 3 - 1 : (k - 18) / 3 - 1));
}
__host__ __device__ constexpr int s27_dz(int k) { return k == 0 ? 0 : (k <= 9 ? -1 : (k <= 17 ? 0 : 1)); }

static thread_local int g_s27_disable = 0;    // examg_debug_sf27_pair(0 / 1, planes per chunk)
static thread_local int g_s27_zc = 0;       // 0: by the rule in launch_sf27_pair

template <int MODE2>   // second stage: EXAMG_SMOOTH (-> out) or EXAMG_RESIDUAL (first-stage field -> out, its residual -> res)
__global__ void __launch_bounds__(64 * S27_NW)
k_sf27_two_stage(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs, double *__restrict__ out, LayoutDev lr,
                 double *__restrict__ res, LayoutDev lc, const double *__restrict__ cf, double w, Box box, S27Geom g) {
  __shared__ __attribute__((aligned(16))) double strip[S27_NW][64 * 27];
  __shared__ double UB[3][S27_UR][S27_UC];
  __shared__ double VB[2][S27_NW][64];
  const int lane = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
  int t = blockIdx.x;
  const int tx = t % g.ntx;
  t /= g.ntx;
  const int ty = t % g.nty;
  const int tz = t / g.nty;
  const int xw = box.b0 - 1 + S27_XO * tx;          // x of lane 0
  const int x = xw + lane;
  const int ry0 = box.b1 - 1 + S27_YO * ty;         // stage-1 row 0
  const int gy = ry0 + wv;                          // this wave's row
  const int mb = box.b2 + tz * g.zc, me = min(mb + g.zc, box.e2);
  const Box &b1 = g.box1;
  const bool on1_xy = x >= b1.b0 && x < b1.e0 && gy >= b1.b1 && gy < b1.e1;
  const bool out_xy = lane >= 1 && lane <= 62 && x >= box.b0 && x < box.e0 && wv >= 1 && wv <= S27_NW - 2 && gy >= box.b1 && gy < box.e1;

  // ---- addressing.  Loads are unconditional with element indices clamped into the arrays (8-byte accesses: a clamped load
  // puts some other value where a point outside the allocation would be, which nothing valid depends on) ----
  const long long usz = lu.size - 1, fsz = lf.size - 1, csz = lc.size * 27 - 1;
  auto uidx = [&](int xx, int yy, int zz) { return min(max(lidx_plain(lu, xx, yy, zz), 0LL), usz); };
  // rows / planes of the coefficient field clamped into its allocation (scalar); the row's record stream starts at point xw
  const int cy = min(max(gy + lc.ref1, 0), lc.tot1 - 1);
  auto crow = [&](int zz) {
    const int cz = min(max(zz + lc.ref2, 0), lc.tot2 - 1);
    return ((long long)(xw + lc.ref0) + lc.s1 * cy + lc.s2 * cz) * 27;
  };
  double *sb = strip[wv];

  double raw[27];                 // records of (row gy, plane q+1) in flight
  auto load_raw = [&](int zz) {
    const long long r0 = crow(zz) + lane;
#pragma unroll
    for (int i = 0; i < 27; ++i) raw[i] = cf[min(max(r0 + 64 * i, 0LL), csz)];
  };
  // input tile: wave wv brings row wv + 1 (its own), columns 1..64, and through lanes 0 / 1 the columns 0 / 65; waves 0 and NW-1 also
  // bring the outer rows 0 and UR-1
  const bool has_outer = wv == 0 || wv == S27_NW - 1;
  const int orow = wv == 0 ? 0 : S27_UR - 1;
  const int hx = lane == 0 ? xw - 1 : xw + 64;      // halo column of lanes 0 / 1
  const int hc = lane == 0 ? 0 : S27_UC - 1;
  struct UIn { double c, h, oc, oh; };
  auto load_u = [&](int zz) {
    UIn v;
    v.c = u[uidx(x, gy, zz)];
    v.h = 0.0; v.oc = 0.0; v.oh = 0.0;
    if (lane < 2) v.h = u[uidx(hx, gy, zz)];
    if (has_outer) {
      const int oy = ry0 - 1 + orow;
      v.oc = u[uidx(x, oy, zz)];
      if (lane < 2) v.oh = u[uidx(hx, oy, zz)];
    }
    return v;
  };
  auto put_u = [&](const UIn &v, int zz) {
    const int pb = ((zz % 3) + 3) % 3;
    UB[pb][wv + 1][lane + 1] = v.c;
    if (lane < 2) UB[pb][wv + 1][hc] = v.h;
    if (has_outer) {
      UB[pb][orow][lane + 1] = v.oc;
      if (lane < 2) UB[pb][orow][hc] = v.oh;
    }
  };
  auto load_f = [&](int zz) { return rhs[min(max(lidx_plain(lf, x, gy, zz), 0LL), fsz)]; };

  // ---- start-up: input planes mb-2, mb-1 in LDS, plane mb and the records / rhs of plane mb-1 in flight ----
  put_u(load_u(mb - 2), mb - 2);
  put_u(load_u(mb - 1), mb - 1);
  UIn un = load_u(mb);
  load_raw(mb - 1);
  double fn = load_f(mb - 1);
  // carried from step q-1 to step q (second stage of plane q-1)
  double cR[9], P = 0.0, wwR = 0.0, fR = 0.0, vR = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) cR[k] = 0.0;

  for (int q = mb - 1; q <= me; ++q) {
    // A: input plane q+1 enters LDS (its buffer held plane q-2, last read before the second barrier of step q-1)
    put_u(un, q + 1);
    // B: this row's records of plane q through the private strip
#pragma unroll
    for (int i = 0; i < 27; ++i) sb[64 * i + lane] = raw[i];
    double c[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) c[k] = sb[27 * lane + k];
    const double f = fn;
    // C: loads of the next step
    un = load_u(q + 2);
    load_raw(q + 1);
    fn = load_f(q + 1);
    __syncthreads();
    // E: first stage on plane q
    const int p0 = (((q - 1) % 3) + 3) % 3, p1 = ((q % 3) + 3) % 3, p2 = (((q + 1) % 3) + 3) % 3;
    auto U = [&](int dx, int dy, int dz) { return UB[dz < 0 ? p0 : (dz == 0 ? p1 : p2)][wv + 1 + dy][lane + 1 + dx]; };
    const double uc = U(0, 0, 0);
    double acc = c[0] * uc;
#pragma unroll
    for (int k = 1; k < 27; ++k) acc = acc + c[k] * U(s27_dx(k), s27_dy(k), s27_dz(k));
    const double ww = (1.0 / c[0]) * w;
    const double v1n = uc + ww * (f - acc);
    const double v1 = (on1_xy && q >= b1.b2 && q < b1.e2) ? v1n : uc;
    VB[q & 1][wv][lane] = v1;
    __syncthreads();
    // G: second stage -- the last 9 terms of plane q-1, then the first 18 of plane q.  Neighbour rows / lanes outside the tile are
    // read clamped: they belong to points that are not outputs.
    const int ylo = wv > 0 ? wv - 1 : 0, yhi = wv < S27_NW - 1 ? wv + 1 : S27_NW - 1;
    const int xlo = lane > 0 ? lane - 1 : 0, xhi = lane < 63 ? lane + 1 : 63;
    auto V = [&](int buf, int dx, int dy) { return VB[buf][dy < 0 ? ylo : (dy == 0 ? wv : yhi)][dx < 0 ? xlo : (dx == 0 ? lane : xhi)]; };
    const int m = q - 1;
    if (m >= mb && m < me) {
      double a2 = P;
#pragma unroll
      for (int k = 18; k < 27; ++k) a2 = a2 + cR[k - 18] * V(q & 1, s27_dx(k), s27_dy(k));
      if (out_xy) {
        if (MODE2 == EXAMG_SMOOTH) {
          __builtin_nontemporal_store(vR + wwR * (fR - a2), out + lidx_plain(lu, x, gy, m));
        } else {
          __builtin_nontemporal_store(vR, out + lidx_plain(lu, x, gy, m));
          __builtin_nontemporal_store(fR - a2, res + lidx_plain(lr, x, gy, m));
        }
      }
    }
    if (q >= mb && q < me) {
      double a2 = c[0] * v1;
#pragma unroll
      for (int k = 1; k < 10; ++k) a2 = a2 + c[k] * V((q - 1) & 1, s27_dx(k), s27_dy(k));
#pragma unroll
      for (int k = 10; k < 18; ++k) a2 = a2 + c[k] * V(q & 1, s27_dx(k), s27_dy(k));
      P = a2;
#pragma unroll
      for (int k = 0; k < 9; ++k) cR[k] = c[18 + k];
      wwR = ww;
      fR = f;
      vR = v1;
    }
  }
}

// Can the pair kernel take these arguments?  (entry order, layouts, weight form, boxes)
static bool sf27_pair_ok(const examg_layout_t *lu, const examg_layout_t *lf, const examg_stencil_t *st, const Box &box1, const Box &box2) {
  if (g_s27_disable || lay_split(lu) || lay_split(lf) || !st->cfield || st->nent != 27 || st->diag != 0 || st->ctransform != EXAMG_CLAYOUT_ENTRY_FASTEST ||
      st->wform != EXAMG_WEIGHT_INV_TIMES || lu->nd != 3)
    return false;
  for (int k = 0; k < 27; ++k)
    if (st->off[k][0] != s27_dx(k) || st->off[k][1] != s27_dy(k) || st->off[k][2] != s27_dz(k)) return false;
  if (box2.n0() < 32 || box2.count() < 32768) return false;      // small levels: two launches of the one-step kernel
  return box_inside(lu, box1, 1) && box_inside(lf, box1, 0) && box_inside(&st->clayout, box1, 0);
}

static int launch_sf27_pair(int mode2, const examg_layout_t *lu_, const double *u, const examg_layout_t *lf_, const double *rhs, double *out,
                            const examg_layout_t *lr_, double *res, const examg_stencil_t *st, double w, const Box &box1, const Box &box2,
                            hipStream_t s) {
  const LayoutDev lu = make_layout(lu_), lf = make_layout(lf_), lr = lr_ ? make_layout(lr_) : lu, lc = make_layout(&st->clayout);
  S27Geom g;
  g.ntx = (box2.n0() + S27_XO - 1) / S27_XO;
  g.nty = (box2.n1() + S27_YO - 1) / S27_YO;
  // planes per chunk: one workgroup per CU at a time -- the chunk count that minimises  rounds of 256 workgroups x (planes per chunk +
  // 4 planes of halo and start-up); 512^3: 64 planes (measured: 16 / 32 / 64 / 128 / 256 planes 8.78 / 8.18 / 8.01 / 8.23 / 8.97 ms)
  int zc = g_s27_zc;
  if (zc <= 0) {
    const long long xy = (long long)g.ntx * g.nty;
    const int n2 = box2.n2();
    long long best = -1;
    for (int t = 1; t <= (n2 + 7) / 8; ++t) {
      const int c = (n2 + t - 1) / t;
      const long long cost = ((xy * ((n2 + c - 1) / c) + 255) / 256) * (c + 4);
      if (best < 0 || cost < best) { best = cost; zc = c; }
    }
  }
  if (zc > box2.n2()) zc = box2.n2();
  g.zc = zc;
  g.ntz = (box2.n2() + zc - 1) / zc;
  g.box1 = box1;
  const long long nblocks = (long long)g.ntx * g.nty * g.ntz;
  if (nblocks > 0x7fffffffLL) { set_error("examg 27-entry pair kernel: too many tiles"); return 1; }
  dim3 grid((unsigned)nblocks), block(64, S27_NW);
  if (mode2 == EXAMG_SMOOTH)
    hipLaunchKernelGGL((k_sf27_two_stage<EXAMG_SMOOTH>), grid, block, 0, s, lu, u, lf, rhs, out, lr, res, lc, st->cfield, w, box2, g);
  else
    hipLaunchKernelGGL((k_sf27_two_stage<EXAMG_RESIDUAL>), grid, block, 0, s, lu, u, lf, rhs, out, lr, res, lc, st->cfield, w, box2, g);
  EXAMG_CHECK_LAUNCH("k_sf27_two_stage");
  return 0;
}

// used by examg_jacobi2 / examg_jacobi2_boxes (kernels_twostage.hip): 1 = launched, 0 = not applicable, -1 = error
int sf27_jacobi2_try(const examg_layout_t *lu, const double *u_in, double *u_out, const examg_layout_t *lf, const double *rhs,
                     const examg_stencil_t *st, double w, const Box &box1, const Box &box2, hipStream_t s) {
  if (!sf27_pair_ok(lu, lf, st, box1, box2)) return 0;
  return launch_sf27_pair(EXAMG_SMOOTH, lu, u_in, lf, rhs, u_out, nullptr, nullptr, st, w, box1, box2, s) ? -1 : 1;
}

}  // namespace examg

using namespace examg;

#ifdef EXAMG_DEBUG_HOOKS
extern "C" int examg_debug_sf27_pair(int enable, int zc) {
  g_s27_disable = enable ? 0 : 1;
  g_s27_zc = zc > 0 ? zc : 0;
  return 0;
}
#endif

// One Jacobi step on [begin,end) followed by the residual of its result, in one pass: u_out = J(u_in) on the box, res = rhs - A u_out
// there (`Smoother@current` as the last pre-smoothing step + `Residual = RHS - Laplace * Solution`, Testing/SISC/3D_VarCoeff.exa4:141-153).
// The 27-entry record form shares the coefficients between the two; everything else runs the two loops.
extern "C" int examg_jacobi_residual(const examg_layout_t *lu, const double *u_in, double *u_out, const examg_layout_t *lf, const double *rhs,
                                     const examg_layout_t *lr, double *res, const examg_stencil_t *st, double w, const int32_t *begin,
                                     const int32_t *end, examg_stream_t stream) {
  if (!lu || !u_in || !u_out || !lf || !rhs || !lr || !res || !st || !begin || !end) { set_error("examg_jacobi_residual: null argument"); return 1; }
  if (u_in == u_out) { set_error("examg_jacobi_residual: out of place only"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (sf27_pair_ok(lu, lf, st, box, box) && box_inside(lr, box, 0))
    return launch_sf27_pair(EXAMG_RESIDUAL, lu, u_in, lf, rhs, u_out, lr, res, st, w, box, box, (hipStream_t)stream);
  int rc = examg_jacobi(lu, u_in, u_out, lf, rhs, st, w, begin, end, stream);
  if (rc) return rc;
  return examg_residual(lu, u_out, lf, rhs, lr, res, st, begin, end, stream);
}
