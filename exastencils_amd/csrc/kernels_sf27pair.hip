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
#include <stdint.h>

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
static thread_local int g_s27_rows = 0;     // rows per wave: 0 = by the rule, 1 / 2 forced

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

// ---- Two rows per wave, records by LDS-DMA ------------------------------------------------------------------------------------------
// The tile halo is what the pass above pays for (6 of its 8 rows are outputs: a third more coefficient reads than points).  Here a wave
// owns TWO adjacent rows and takes them one after the other through the same private strip: 16 stage-1 rows per workgroup, 14 of them
// outputs.  What made room: the records travel from memory into the strip by `global_load_lds_dwordx4` (the strip image IS the record
// stream, 13.5 pieces of 1 KiB per row) instead of through 27 registers per lane, so that the coefficients of both rows (2 x 27) can stay in
// registers from their first-stage to their second-stage use; the strip of a wave holds row a of plane q until its coefficients are
// read, then row b of plane q (requested while row a's first stage runs), then row a of plane q + 1 (requested while row b's runs).
// Barriers are bare `s_barrier`s behind `s_waitcnt lgkmcnt(0)`: a `__syncthreads()` would wait for the DMA in flight.
// 155 KB of LDS, one workgroup per CU.  Same products in the same order as the kernel above.
constexpr int S2_RPW = 2;
constexpr int S2_SR = S27_NW * S2_RPW;   // stage-1 rows per workgroup
constexpr int S2_YO = S2_SR - 2;         // output rows
constexpr int S2_UR = S2_SR + 2;         // input rows

typedef __attribute__((address_space(3))) void *s27_lptr;

// four pieces of 1 KiB (lane l: 16 bytes at base + voff + 1024 i -> LDS m0 + 16 l + 1024 i; the instruction offset moves both sides).
// Inline assembly on purpose: the compiler, told of an LDS-DMA in flight, holds every later LDS read (of ANY array) back behind
// `s_waitcnt vmcnt(0)`; the waits that order the strip are written out in the kernel.  Nothing else in these kernels uses m0.
// everything this wave has asked memory for has arrived (written as the builtin, so that the compiler's own counting of its loads knows)
__device__ __forceinline__ void s27_wait_vm() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), expcnt / lgkmcnt untouched
  asm volatile("" ::: "memory");
}

template <int PIECES>
__device__ __forceinline__ void s27_dma(unsigned lds, const double *base, unsigned voff) {
  if (PIECES == 4)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072" ::"s"(lds), "v"(voff), "s"(base) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(base) : "memory");
}

template <int MODE2>
__global__ void __launch_bounds__(64 * S27_NW)
k_sf27_two_stage_r2(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs, double *__restrict__ out, LayoutDev lr,
                    double *__restrict__ res, LayoutDev lc, const double *__restrict__ cf, double w, Box box, S27Geom g) {
  __shared__ __attribute__((aligned(16))) double strip[S27_NW][64 * 27];
  __shared__ double UB[3][S2_UR][S27_UC];
  __shared__ double VB[2][S2_SR][64];
  const int lane = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
  int t = blockIdx.x;
  const int tx = t % g.ntx;
  t /= g.ntx;
  const int ty = t % g.nty;
  const int tz = t / g.nty;
  // x of lane 0.  The last window is moved left until its 64 records end with the row of the coefficient array: every record stream stays
  // inside its row (the launcher checked that the rows are long enough), the points it shares with the window before are written there.
  const int xfirst = box.b0 + S27_XO * tx;          // first output column of this window
  const int xw = min(xfirst - 1, lc.tot0 - 64 - lc.ref0);
  const int x = xw + lane;
  const int ry0 = box.b1 - 1 + S2_YO * ty;          // stage-1 row 0
  const int rw = S2_RPW * wv;                       // tile row of this wave's first row
  const int mb = box.b2 + tz * g.zc, me = min(mb + g.zc, box.e2);
  const Box &b1 = g.box1;
  const bool col_out = lane >= 1 && lane <= 62 && x >= xfirst && x < box.e0;
  bool on1_xy[S2_RPW], out_xy[S2_RPW];
#pragma unroll
  for (int j = 0; j < S2_RPW; ++j) {
    const int gy = ry0 + rw + j;
    on1_xy[j] = x >= b1.b0 && x < b1.e0 && gy >= b1.b1 && gy < b1.e1;
    out_xy[j] = col_out && rw + j >= 1 && rw + j <= S2_SR - 2 && gy >= box.b1 && gy < box.e1;
  }

  const long long usz = lu.size - 1, fsz = lf.size - 1;
  auto uidx = [&](int xx, int yy, int zz) { return min(max(lidx_plain(lu, xx, yy, zz), 0LL), usz); };
  double *sb = strip[wv];
  const unsigned sb_lds = (unsigned)(uintptr_t)(s27_lptr)sb;
  const unsigned voff = 16u * lane;
  // records of (row j, plane zz) -> strip, no registers in between: 13 whole pieces and half a piece (rows / planes clamped into the array).
  // Not asked for: rows and planes outside the first-stage box (their points keep the input value, whatever the strip holds), and in the
  // moved last window the leading pieces, whose records belong to lanes left of this window's own columns (groups of four pieces).
  const int first_piece = (27 * (xfirst - 1 - xw)) >> 7;
  auto request = [&](int j, int zz) {
    const int gy = ry0 + rw + j;
    if (gy < b1.b1 || gy >= b1.e1 || zz < b1.b2 || zz >= b1.e2) return;
    const int cy = min(max(gy + lc.ref1, 0), lc.tot1 - 1);
    const int cz = min(max(zz + lc.ref2, 0), lc.tot2 - 1);
    const double *src = cf + ((long long)(xw + lc.ref0) + lc.s1 * cy + lc.s2 * cz) * 27;
    if (first_piece < 4) s27_dma<4>(sb_lds, src, voff);
    if (first_piece < 8) s27_dma<4>(sb_lds + 4096, src + 512, voff);
    if (first_piece < 12) s27_dma<4>(sb_lds + 8192, src + 1024, voff);
    s27_dma<1>(sb_lds + 12288, src + 1536, voff);
    if (lane < 32) s27_dma<1>(sb_lds + 13312, src + 1664, voff);
  };
  // input tile: wave wv brings its two rows (tile rows rw + 1, rw + 2 of UB), columns 1..64, and through lanes 0 / 1 the columns 0 / 65;
  // waves 0 and NW-1 also bring the outer rows 0 and UR-1
  const bool has_outer = wv == 0 || wv == S27_NW - 1;
  const int orow = wv == 0 ? 0 : S2_UR - 1;
  const int hx = lane == 0 ? xw - 1 : xw + 64;      // halo column of lanes 0 / 1
  const int hc = lane == 0 ? 0 : S27_UC - 1;
  struct UIn { double c[S2_RPW], h[S2_RPW], oc, oh; };
  auto load_u = [&](int zz) {
    UIn v;
#pragma unroll
    for (int j = 0; j < S2_RPW; ++j) {
      v.c[j] = u[uidx(x, ry0 + rw + j, zz)];
      v.h[j] = 0.0;
      if (lane < 2) v.h[j] = u[uidx(hx, ry0 + rw + j, zz)];
    }
    v.oc = 0.0; v.oh = 0.0;
    if (has_outer) {
      const int oy = ry0 - 1 + orow;
      v.oc = u[uidx(x, oy, zz)];
      if (lane < 2) v.oh = u[uidx(hx, oy, zz)];
    }
    return v;
  };
  auto put_u = [&](const UIn &v, int zz) {
    const int pb = ((zz % 3) + 3) % 3;
#pragma unroll
    for (int j = 0; j < S2_RPW; ++j) {
      UB[pb][rw + j + 1][lane + 1] = v.c[j];
      if (lane < 2) UB[pb][rw + j + 1][hc] = v.h[j];
    }
    if (has_outer) {
      UB[pb][orow][lane + 1] = v.oc;
      if (lane < 2) UB[pb][orow][hc] = v.oh;
    }
  };
  auto load_f = [&](int j, int zz) { return rhs[min(max(lidx_plain(lf, x, ry0 + rw + j, zz), 0LL), fsz)]; };

  // ---- start-up: input planes mb-2, mb-1 in LDS, plane mb, the rhs of plane mb-1 and row a's records of plane mb-1 in flight ----
  put_u(load_u(mb - 2), mb - 2);
  put_u(load_u(mb - 1), mb - 1);
  UIn un = load_u(mb);
  double fn[S2_RPW];
#pragma unroll
  for (int j = 0; j < S2_RPW; ++j) fn[j] = load_f(j, mb - 1);
  request(0, mb - 1);
  // carried from step q-1 to step q (second stage of plane q-1)
  double cR[S2_RPW][9], P[S2_RPW], wwR[S2_RPW], fR[S2_RPW], vR[S2_RPW];
#pragma unroll
  for (int j = 0; j < S2_RPW; ++j) {
    P[j] = 0.0; wwR[j] = 0.0; fR[j] = 0.0; vR[j] = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) cR[j][k] = 0.0;
  }

  for (int q = mb - 1; q <= me; ++q) {
    const int p0 = (((q - 1) % 3) + 3) % 3, p1 = ((q % 3) + 3) % 3, p2 = (((q + 1) % 3) + 3) % 3;
    double c[S2_RPW][27], f[S2_RPW], v1[S2_RPW], ww[S2_RPW];
    // first stage of row j on plane q (coefficients in c[j]) -> VB
    auto stage1 = [&](int j) {
      auto U = [&](int dx, int dy, int dz) { return UB[dz < 0 ? p0 : (dz == 0 ? p1 : p2)][rw + j + 1 + dy][lane + 1 + dx]; };
      const double uc = U(0, 0, 0);
      double acc = c[j][0] * uc;
#pragma unroll
      for (int k = 1; k < 27; ++k) acc = acc + c[j][k] * U(s27_dx(k), s27_dy(k), s27_dz(k));
      ww[j] = (1.0 / c[j][0]) * w;
      const double v1n = uc + ww[j] * (f[j] - acc);
      v1[j] = (on1_xy[j] && q >= b1.b2 && q < b1.e2) ? v1n : uc;
      VB[q & 1][rw + j][lane] = v1[j];
    };
    // A: row a's records of plane q have landed (everything this wave has asked for has); out of the strip, input plane q+1 into LDS
    s27_wait_vm();
#pragma unroll
    for (int k = 0; k < 27; ++k) c[0][k] = sb[27 * lane + k];
    put_u(un, q + 1);
#pragma unroll
    for (int j = 0; j < S2_RPW; ++j) f[j] = fn[j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // B: row b's records of plane q into the strip; ordinary loads of the next step
    request(1, q);
    un = load_u(q + 2);
#pragma unroll
    for (int j = 0; j < S2_RPW; ++j) fn[j] = load_f(j, q + 1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // C: first stage of row a; then row b's records (landed), row a's of the next plane requested, first stage of row b
    stage1(0);
    s27_wait_vm();
#pragma unroll
    for (int k = 0; k < 27; ++k) c[1][k] = sb[27 * lane + k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (q < me) request(0, q + 1);
    stage1(1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // D: second stage -- the last 9 terms of plane q-1, then the first 18 of plane q.  Neighbour rows / lanes outside the tile are
    // read clamped: they belong to points that are not outputs.
    const int xlo = lane > 0 ? lane - 1 : 0, xhi = lane < 63 ? lane + 1 : 63;
    const int m = q - 1;
#pragma unroll
    for (int j = 0; j < S2_RPW; ++j) {
      const int r = rw + j;
      const int ylo = r > 0 ? r - 1 : 0, yhi = r < S2_SR - 1 ? r + 1 : S2_SR - 1;
      auto V = [&](int buf, int dx, int dy) { return VB[buf][dy < 0 ? ylo : (dy == 0 ? r : yhi)][dx < 0 ? xlo : (dx == 0 ? lane : xhi)]; };
      if (m >= mb && m < me) {
        double a2 = P[j];
#pragma unroll
        for (int k = 18; k < 27; ++k) a2 = a2 + cR[j][k - 18] * V(q & 1, s27_dx(k), s27_dy(k));
        if (out_xy[j]) {
          const int gy = ry0 + r;
          if (MODE2 == EXAMG_SMOOTH) {
            __builtin_nontemporal_store(vR[j] + wwR[j] * (fR[j] - a2), out + lidx_plain(lu, x, gy, m));
          } else {
            __builtin_nontemporal_store(vR[j], out + lidx_plain(lu, x, gy, m));
            __builtin_nontemporal_store(fR[j] - a2, res + lidx_plain(lr, x, gy, m));
          }
        }
      }
      if (q >= mb && q < me) {
        double a2 = c[j][0] * v1[j];
#pragma unroll
        for (int k = 1; k < 10; ++k) a2 = a2 + c[j][k] * V((q - 1) & 1, s27_dx(k), s27_dy(k));
#pragma unroll
        for (int k = 10; k < 18; ++k) a2 = a2 + c[j][k] * V(q & 1, s27_dx(k), s27_dy(k));
        P[j] = a2;
#pragma unroll
        for (int k = 0; k < 9; ++k) cR[j][k] = c[j][18 + k];
        wwR[j] = ww[j];
        fR[j] = f[j];
        vR[j] = v1[j];
      }
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
  // small levels: two launches of the one-step kernel (64^3: 2 x 20 us against 48 us for the pass; 128^3: 2 x 105 against 136)
  if (box2.n0() < 32 || (box2.count() < (1 << 20) && !g_s27_rows)) return false;      // (a forced variant -- debug build -- runs at any size)
  return box_inside(lu, box1, 1) && box_inside(lf, box1, 0) && box_inside(&st->clayout, box1, 0);
}

static int launch_sf27_pair(int mode2, const examg_layout_t *lu_, const double *u, const examg_layout_t *lf_, const double *rhs, double *out,
                            const examg_layout_t *lr_, double *res, const examg_stencil_t *st, double w, const Box &box1, const Box &box2,
                            hipStream_t s) {
  const LayoutDev lu = make_layout(lu_), lf = make_layout(lf_), lr = lr_ ? make_layout(lr_) : lu, lc = make_layout(&st->clayout);
  // two rows per wave where the taller tile does not waste what it saves (rows beyond the box in the last tile)
  // (and where every 64-record stream fits the rows of the coefficient array: see the kernel's last window)
  const bool r2_fits = lc.tot0 >= 64 && box2.b0 - 1 + lc.ref0 >= 0 && box2.e0 + lc.ref0 <= lc.tot0 - 1;
  const int rpw = !r2_fits ? 1 : (g_s27_rows ? g_s27_rows : (box2.n1() >= 3 * S2_YO ? 2 : 1));
  const int yo = rpw == 2 ? S2_YO : S27_YO;
  S27Geom g;
  g.ntx = (box2.n0() + S27_XO - 1) / S27_XO;
  g.nty = (box2.n1() + yo - 1) / yo;
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
  if (rpw == 2 && mode2 == EXAMG_SMOOTH)
    hipLaunchKernelGGL((k_sf27_two_stage_r2<EXAMG_SMOOTH>), grid, block, 0, s, lu, u, lf, rhs, out, lr, res, lc, st->cfield, w, box2, g);
  else if (rpw == 2)
    hipLaunchKernelGGL((k_sf27_two_stage_r2<EXAMG_RESIDUAL>), grid, block, 0, s, lu, u, lf, rhs, out, lr, res, lc, st->cfield, w, box2, g);
  else if (mode2 == EXAMG_SMOOTH)
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
extern "C" int examg_debug_sf27_pair(int enable, int zc) {      // enable: 0 off, 1 on (rule), 11 / 12: on with one / two rows per wave
  g_s27_disable = enable ? 0 : 1;
  g_s27_rows = enable >= 11 ? enable - 10 : 0;
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
