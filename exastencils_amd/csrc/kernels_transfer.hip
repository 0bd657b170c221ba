// Inter-grid transfer loops for gfx950: full-weighting restriction and tri-linear prolongation+correction.
//
// Reference: `RHS@coarser = NodeRestriction * Residual` / `Solution += NodeProlongation@coarser * Solution@coarser`
// (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:221-237); weights
// Compiler/src/exastencils/operator/l4/L4_DefaultRestriction.scala:29-36,63-88 and
// L4_DefaultProlongation.scala:30-45; index mapping solver/ir/IR_ResolveIntergridIndices.scala;
// the reference's prolongation kernel is 2^d guarded statements per thread
// (stencil/ir/IR_FindStencilConvolutions.scala:135-156) -- here parity is resolved per thread
// without divergence in the memory pattern.  Summation order = the oracle's (entry-table order).
#include "examg_common.h"

namespace examg {

// One thread per coarse point; x fastest.  Reads the 3^d fine neighbourhood of 2I.
template <int ND>
__global__ void __launch_bounds__(256)
k_restrict(LayoutDev lfine, const double *__restrict__ rf, LayoutDev lc, double *__restrict__ fc, double scale, Box box) {
  const long long total = box.count();
  const int n0 = box.n0(), n1 = box.n1();
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int I0 = box.b0 + (int)(t % n0);
    const long long row = t / n0;
    const int I1 = box.b1 + (int)(row % n1);
    const int I2 = box.b2 + (int)(row / n1);
    const double w1[3] = {0.25, 0.5, 0.25};
    const long long base = lidx(lfine, 2 * I0, 2 * I1, ND == 3 ? 2 * I2 : 0);
    double acc = 0.0;
    bool first = true;
#pragma unroll
    for (int a = -1; a <= 1; ++a)
#pragma unroll
      for (int b = -1; b <= 1; ++b) {
        if (ND == 2) {
          const double wgt = scale * (w1[a + 1] * w1[b + 1]);
          const double tv = wgt * rf[lfine.half ? lidx(lfine, 2 * I0 + a, 2 * I1 + b, 0) : base + a + lfine.s1 * b];
          acc = first ? tv : acc + tv;
          first = false;
        } else {
#pragma unroll
          for (int c = -1; c <= 1; ++c) {
            const double wgt = scale * ((w1[a + 1] * w1[b + 1]) * w1[c + 1]);
            const double tv = wgt * rf[lfine.half ? lidx(lfine, 2 * I0 + a, 2 * I1 + b, 2 * I2 + c) : base + a + lfine.s1 * b + lfine.s2 * c];
            acc = first ? tv : acc + tv;
            first = false;
          }
        }
      }
    fc[lidx(lc, I0, I1, I2)] = acc;
  }
}


// 3-D restriction on long rows: one coarse point per lane, the fine pair (2I, 2I+1) of each of the nine fine rows with
// one 16-byte load, the (2I-1) column from the lane below (DPP; lane 0 reads it).  A wave owns 64 coarse points of one
// coarse row and marches in coarse z: fine plane 2K+1 serves coarse planes K and K+1, so a step loads 6 rows, not 9.
// Same 27 products in the same order as k_restrict (x offset outermost, then y, then z).
// RW = 2: a wave owns TWO consecutive coarse rows (five fine rows 2J-1 .. 2J+3 instead of 2 x 3: the fine row between them is read
// once -- fabric traffic 1.51 x -> 1.25 x compulsory).
template <int RW>
__global__ void __launch_bounds__(256)
k_restrict3_wide(LayoutDev lfine, const double *__restrict__ rf, LayoutDev lc, double *__restrict__ fc, double scale, Box box,
                 int ntx, int zc, int nwaves) {
  constexpr int NR = 2 * RW + 1;      // fine rows of the wave
  const int lane = threadIdx.x;
  const int n1w = (box.n1() + RW - 1) / RW;
  long long wg = blockIdx.x;
  // XCD bands of y-adjacent workgroups within a z chunk (see k_residual_restrict3): 512^3 -> 256^3 0.229 -> 0.225 ms, 256^3 -> 128^3 0.026 -> 0.024
  if (((ntx * n1w) & 3) == 0) {
    const int wpl = (ntx * n1w) >> 2, per = wpl >> 3;
    const long long lz = wg / wpl;
    const int r = (int)(wg - lz * wpl);
    wg = lz * wpl + (r < (per << 3) ? (r & 7) * per + (r >> 3) : r);
  }
  long long t = wg * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y);
  if (t >= nwaves) return;
  const int tx = (int)(t % ntx);
  t /= ntx;
  const int I1 = box.b1 + (int)(t % n1w) * RW;
  const int kb = box.b2 + (int)(t / n1w) * zc;
  const int ke = min(kb + zc, box.e2);
  int I0 = box.b0 + tx * 64 + lane;
  const bool valid = I0 < box.e0;
  if (!valid) I0 = box.e0 - 1;
  const bool second = RW == 2 && I1 + 1 < box.e1;      // wave-uniform: the box has this wave's second coarse row
  const double *row[NR];
#pragma unroll
  for (int b = 0; b < NR; ++b) row[b] = rf + lfine.origin + 2 * I0 + lfine.s1 * (2 * I1 + ((b < 3 || second) ? b : 2) - 1);
  double *out = fc + lc.origin + I0 + lc.s1 * I1;
  const double w1[3] = {0.25, 0.5, 0.25};
  d2 P[3][NR];      // [plane slot: 2K-1, 2K, 2K+1][fine row]
  double E[3][NR];  // lane 0: the column 2I-1
#pragma unroll
  for (int b = 0; b < NR; ++b) {
    P[0][b] = load2(row[b] + lfine.s2 * (2 * kb - 1));
    E[0][b] = lane == 0 ? row[b][lfine.s2 * (2 * kb - 1) - 1] : 0.0;
  }
  for (int K = kb; K < ke; ++K) {
#pragma unroll
    for (int c = 1; c < 3; ++c)
#pragma unroll
      for (int b = 0; b < NR; ++b) P[c][b] = load2(row[b] + lfine.s2 * (2 * K + c - 1));
    if (lane == 0) {
#pragma unroll
      for (int c = 1; c < 3; ++c)
#pragma unroll
        for (int b = 0; b < NR; ++b) E[c][b] = row[b][lfine.s2 * (2 * K + c - 1) - 1];
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      double acc = 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            double v;
            if (a == 0) {
              v = lane_below(P[c][2 * r + b].y);
              if (lane == 0) v = E[c][2 * r + b];
            } else {
              v = a == 1 ? P[c][2 * r + b].x : P[c][2 * r + b].y;
            }
            const double wgt = scale * ((w1[a] * w1[b]) * w1[c]);
            const double tv = wgt * v;
            acc = (a == 0 && b == 0 && c == 0) ? tv : acc + tv;
          }
      if (valid && (r == 0 || second)) out[lc.s1 * r + lc.s2 * K] = acc;
    }
#pragma unroll
    for (int b = 0; b < NR; ++b) {
      P[0][b] = P[2][b];
      E[0][b] = E[2][b];
    }
  }
}

// RHS@coarser = scale * R * (rhs - A u) in ONE pass: residual (3-D 7-point constant coefficients) and full-weighting
// restriction fused, the fine residual is never written or re-read (48 + 8 B per fine point -> 16 B).  Lane l of a tile
// computes the residuals of the fine pair (2I, 2I+1), I = first + l - 1, on the three fine rows 2J-1..2J+1 of its coarse row J,
// marching in fine z with u[p-1], u[p], u[p+1] in registers; fine plane 2K+1 serves coarse planes K and K+1.  Lanes 1..63
// produce a coarse value (lane 0 only supplies r(2I-1) to lane 1 through the DPP shift), tiles advance by 63 points.
// Every residual is the expression of the residual kernel, the 27 products are added in k_restrict's order: bit-identical
// to examg_residual followed by examg_restrict.
// RW = 2: a wave owns TWO consecutive coarse rows: seven fine rows of u and five of rhs for four fine rows of progress instead of
// 2 x (5 + 3), five residual rows instead of 2 x 3 (the row between the two coarse rows is evaluated once).
template <int ORDER, int RW>
__global__ void __launch_bounds__(256)
k_residual_restrict3(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ f, LayoutDev lc,
                     double *__restrict__ fc, Coef7 k, double scale, Box box, int ntx, int zc, int nwaves, int order) {
  constexpr int NR = 2 * RW + 1;     // residual (and rhs) rows of the wave: 2J-1 .. 2J+2RW-1
  constexpr int NU = NR + 2;         // u rows: 2J-2 .. 2J+2RW
  const int lane = threadIdx.x;
  const int n1w = (box.n1() + RW - 1) / RW;
  int J, tx, kb;
  if (order == 1) {
    // workgroups are dealt round-robin to the 8 XCDs: within a z layer every XCD takes a band of consecutive (x tile fastest, then y)
    // workgroups -- the fine rows that y-adjacent row groups share AND the cache lines that x-adjacent tiles share (a tile's rows start
    // 1008 B after its neighbour's: 9 lines for 1 KiB) then meet in one L2 while both are being worked on
    const int ngy = (n1w + 3) >> 2, wpl = ntx * ngy, per = wpl >> 3;
    const int lz = blockIdx.x / wpl;
    int r = blockIdx.x - lz * wpl;
    if (r < (per << 3)) r = (r & 7) * per + (r >> 3);
    tx = r % ntx;
    const int jw = (r / ntx) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y);
    if (jw >= n1w) return;
    J = box.b1 + jw * RW;
    kb = box.b2 + lz * zc;
  } else {
    long long wg = blockIdx.x;
    // within a layer (one x tile, one z chunk) every XCD takes a band of y-adjacent workgroups, whose shared fine rows then meet in one
    // L2 (512^3: 0.462 -> 0.455 ms)
    if ((n1w & 3) == 0) {
      const int wpl = n1w >> 2, per = wpl >> 3;
      const long long lz = wg / wpl;
      const int r = (int)(wg - lz * wpl);
      wg = lz * wpl + (r < (per << 3) ? (r & 7) * per + (r >> 3) : r);
    }
    long long t = wg * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y);
    if (t >= nwaves) return;
    J = box.b1 + (int)(t % n1w) * RW;   // consecutive waves of a workgroup: consecutive coarse rows (shared fine rows)
    t /= n1w;
    tx = (int)(t % ntx);
    kb = box.b2 + (int)(t / ntx) * zc;
  }
  const int ke = min(kb + zc, box.e2);
  int I = box.b0 + tx * 63 + lane - 1;
  const bool valid = lane >= 1 && I < box.e0;
  if (I >= box.e0) I = box.e0 - 1;
  const bool edge = lane == 63 || I == box.e0 - 1;   // the x+1 neighbour of the pair's second point is in no lane
  const bool second = RW == 2 && J + 1 < box.e1;     // wave-uniform: the box has this wave's second coarse row
  // rows of a second coarse row that the box does not have are read as the last rows of the first (nothing is stored for them)
  const double *ur[NU], *fr[NR];
#pragma unroll
  for (int r = 0; r < NU; ++r) ur[r] = u + lu.origin + 2 * I + lu.s1 * (2 * J - 2 + ((r < 5 || second) ? r : 4));
#pragma unroll
  for (int r = 0; r < NR; ++r) fr[r] = f + lf.origin + 2 * I + lf.s1 * (2 * J - 1 + ((r < 3 || second) ? r : 2));
  double *out = fc + lc.origin + I + lc.s1 * J;
  const double w1[3] = {0.25, 0.5, 0.25};

  d2 U[3][NU];   // fine planes p-1, p, p+1
  d2 R[3][NR];   // residual planes 2K-1, 2K, 2K+1
  auto load_plane = [&](d2 (&P)[NU], int p) {
#pragma unroll
    for (int r = 0; r < NU; ++r) P[r] = load2(ur[r] + lu.s2 * p);
  };
  auto residual_plane = [&](d2 (&out_r)[NR], int p) {      // needs U = planes p-1, p, p+1
#pragma unroll
    for (int b = 0; b < NR; ++b) {
      const d2 c = U[1][b + 1], ym = U[1][b], yp = U[1][b + 2], zm = U[0][b + 1], zp = U[2][b + 1];
      const d2 fv = load2(fr[b] + lf.s2 * p);
      const double xl = lane_below(c.y);
      double xr = lane_above(c.x);
      if (edge) xr = ur[b + 1][lu.s2 * p + 2];
      const double acc_a = conv7<ORDER>(k, c.x, xl, c.y, ym.x, yp.x, zm.x, zp.x);
      const double acc_b = conv7<ORDER>(k, c.y, c.x, xr, ym.y, yp.y, zm.y, zp.y);
      out_r[b].x = fv.x - acc_a;      // lane 0's first point has no valid x-1 neighbour: its value is never used
      out_r[b].y = fv.y - acc_b;
    }
  };
  // residual plane 2kb-1
  load_plane(U[0], 2 * kb - 2);
  load_plane(U[1], 2 * kb - 1);
  load_plane(U[2], 2 * kb);
  residual_plane(R[0], 2 * kb - 1);
  for (int K = kb; K < ke; ++K) {
#pragma unroll
    for (int c = 1; c < 3; ++c) {
      const int p = 2 * K + c - 1;                        // fine plane of this residual plane
#pragma unroll
      for (int r = 0; r < NU; ++r) {
        U[0][r] = U[1][r];
        U[1][r] = U[2][r];
      }
      load_plane(U[2], p + 1);
      residual_plane(R[c], p);
    }
#pragma unroll
    for (int q = 0; q < RW; ++q) {
      double acc = 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const d2 rv = R[c][2 * q + b];
            const double v = a == 0 ? lane_below(rv.y) : (a == 1 ? rv.x : rv.y);
            const double wgt = scale * ((w1[a] * w1[b]) * w1[c]);
            const double tv = wgt * v;
            acc = (a == 0 && b == 0 && c == 0) ? tv : acc + tv;
          }
      if (valid && (q == 0 || second)) out[lc.s1 * q + lc.s2 * K] = acc;
    }
#pragma unroll
    for (int b = 0; b < NR; ++b) R[0][b] = R[2][b];
  }
}

// One thread per fine point.
template <int ND>
__global__ void __launch_bounds__(256)
k_prolong_add(LayoutDev lc, const double *__restrict__ uc, LayoutDev lfine, double *__restrict__ uf, Box box) {
  const long long total = box.count();
  const int n0 = box.n0(), n1 = box.n1();
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int i0 = box.b0 + (int)(t % n0);
    const long long row = t / n0;
    const int i1 = box.b1 + (int)(row % n1);
    const int i2 = box.b2 + (int)(row / n1);
    const int ii[3] = {i0, i1, i2};
    int n[3], ci[3][2];
    double cw[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (d >= ND) { n[d] = 1; ci[d][0] = 0; ci[d][1] = 0; cw[d][0] = 1.0; cw[d][1] = 0.0; continue; }
      if ((ii[d] & 1) == 0) { n[d] = 1; ci[d][0] = ii[d] / 2; ci[d][1] = ii[d] / 2; cw[d][0] = 1.0; cw[d][1] = 0.0; }
      else { n[d] = 2; ci[d][0] = (ii[d] + 1) / 2; ci[d][1] = (ii[d] - 1) / 2; cw[d][0] = 0.5; cw[d][1] = 0.5; }
    }
    double acc = 0.0;
    bool first = true;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (a < n[0] && b < n[1] && c < n[2]) {
            const double tv = ((cw[0][a] * cw[1][b]) * cw[2][c]) * uc[lidx(lc, ci[0][a], ci[1][b], ci[2][c])];
            acc = first ? tv : acc + tv;
            first = false;
          }
        }
    const long long kf = lidx(lfine, i0, i1, i2);
    uf[kf] = uf[kf] + acc;
  }
}

// 3-D fast path: one thread per pair of fine points (2j, 2j+1) -- 16-byte load/store of the fine row; the
// coarse values (1/8 of the fine traffic) come through L2.  Same terms, same order as k_prolong_add.
struct __attribute__((packed, aligned(8))) pd2 { double a, b; };

__global__ void __launch_bounds__(256)
k_prolong_add3_pairs(LayoutDev lc, const double *__restrict__ uc, LayoutDev lfine, double *__restrict__ uf, Box box, int x0, int npairs,
                     int zb) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  const int i1 = box.b1 + blockIdx.y * 4 + threadIdx.y;
  if (p >= npairs || i1 >= box.e1) return;
  const int xa = x0 + 2 * p;              // even fine index; xb = xa + 1 is odd
  const int j = xa >> 1;                  // coarse column of xa; xb interpolates (j+1, j)
  const bool va = xa >= box.b0 && xa < box.e0, vb = xa + 1 >= box.b0 && xa + 1 < box.e0;
  if (!va && !vb) return;
  // rows: even -> one coarse row with weight 1; odd -> rows (i1+1)/2 then (i1-1)/2 with weight 1/2 each
  const bool odd1 = i1 & 1;
  const int r0 = odd1 ? (i1 + 1) / 2 : i1 / 2, r1 = (i1 - 1) / 2;
  const double wr = odd1 ? 0.5 : 1.0;
  const int z_begin = box.b2 + blockIdx.z * zb, z_end = min(z_begin + zb, box.e2);
  for (int i2 = z_begin; i2 < z_end; ++i2) {
    const bool odd2 = i2 & 1;
    const int q0 = odd2 ? (i2 + 1) / 2 : i2 / 2, q1 = (i2 - 1) / 2;
    const double wq = odd2 ? 0.5 : 1.0;
    // coarse values c[column][row slot][plane slot]; column 0 = j, 1 = j + 1
    double c[2][2][2];
#pragma unroll
    for (int rs = 0; rs < 2; ++rs)
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        if ((rs == 1 && !odd1) || (qs == 1 && !odd2)) { c[0][rs][qs] = 0.0; c[1][rs][qs] = 0.0; continue; }
        const long long k = lidx_plain(lc, j, rs ? r1 : r0, qs ? q1 : q0);
        c[0][rs][qs] = uc[k];
        c[1][rs][qs] = vb ? uc[k + 1] : 0.0;
      }
    // point a (even x): dim-0 case has the single entry (j, weight 1); point b: entries (j+1, 1/2), (j, 1/2)
    double acc_a = 0.0, acc_b = 0.0;
    bool fa = true, fb = true;
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      if (rs == 1 && !odd1) continue;
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        if (qs == 1 && !odd2) continue;
        const double ta = ((1.0 * wr) * wq) * c[0][rs][qs];
        acc_a = fa ? ta : acc_a + ta;
        fa = false;
      }
    }
#pragma unroll
    for (int cs = 1; cs >= 0; --cs)
#pragma unroll
      for (int rs = 0; rs < 2; ++rs) {
        if (rs == 1 && !odd1) continue;
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
          if (qs == 1 && !odd2) continue;
          const double tb = ((0.5 * wr) * wq) * c[cs][rs][qs];
          acc_b = fb ? tb : acc_b + tb;
          fb = false;
        }
      }
    double *q = uf + lidx_plain(lfine, xa, i1, i2);
    if (va && vb) {
      pd2 v = *reinterpret_cast<pd2 *>(q);
      v.a = v.a + acc_a;
      v.b = v.b + acc_b;
      *reinterpret_cast<pd2 *>(q) = v;
    } else if (va) {
      q[0] = q[0] + acc_a;
    } else {
      q[1] = q[1] + acc_b;
    }
  }
}

static thread_local int g_restrict_wide = 1;
// residual_restrict: wave count target, minimum coarse planes per wave.  MI355X, 512^3 -> 256^3 (tools/sweep_rr.py), ms by target:
// 4096: 0.560, 9216: 0.511, 18432: 0.486, 24576: 0.479, 36864: 0.481 -- many short waves keep the tail of the launch short
// chunks of 8 coarse planes at any size: with a fixed count of 24576 waves the chunks of blocks larger than 512^3 grew long and the
// front wide (tools/sweep_big_others.py: 768^3 1.86 -> 1.75 ms, 1024^3 4.69 -> 4.16 ms; 512^3 0.527 -> 0.523, there 13 planes before)
static thread_local int g_rr_waves = 1 << 22, g_rr_minzc = 8, g_rr_rows = 0;   // examg_debug_residual_restrict(waves, minzc [+ 1000: two coarse rows per wave, + 2000: one]); 0 = by size
static thread_local int g_rr_order = 0;          // examg_debug_rr_order: 1 = x tiles fastest inside the XCD bands of a z layer (measured equal: 0.491 / 0.493 ms at 512^3), 0 = bands of row groups per x tile
static thread_local int g_restrict_rows = 2;     // examg_debug_restrict(-1 / -2): coarse rows per wave of the wide restriction kernel
static thread_local int g_restrict_waves = -1;   // examg_debug_restrict(n > 1): wave count target of the wide restriction kernel
static thread_local int g_prolong_zb = -1;    // planes per workgroup of the pair prolongation (examg_debug_prolong)   // examg_debug_restrict(0): one-thread-per-point kernel everywhere

static inline dim3 grid_for(long long total) {
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  return dim3((unsigned)nb);
}

}  // namespace examg

using namespace examg;

#ifdef EXAMG_DEBUG_HOOKS
extern "C" int examg_debug_residual_restrict(int waves, int minzc) {
  examg::g_rr_waves = waves > 0 ? waves : (1 << 22);
  examg::g_rr_rows = minzc >= 2000 ? 1 : (minzc >= 1000 ? 2 : 0);
  minzc %= 1000;
  if (minzc > 0) examg::g_rr_minzc = minzc;
  return 0;
}

extern "C" int examg_debug_rr_order(int order) {
  examg::g_rr_order = order ? 1 : 0;
  return 0;
}

extern "C" int examg_debug_prolong(int zb) {
  examg::g_prolong_zb = zb;
  return 0;
}
extern "C" int examg_debug_restrict(int wide) {
  if (wide < 0) { examg::g_restrict_rows = wide == -1 ? 1 : 2; return 0; }
  examg::g_restrict_wide = wide != 0;
  examg::g_restrict_waves = wide > 1 ? wide : -1;
  return 0;
}
#endif

extern "C" int examg_restrict(const examg_layout_t *lfine_, const double *rf, const examg_layout_t *lc_, double *fc,
                              double scale, const int32_t *begin, const int32_t *end, examg_stream_t stream) {
  if (!lfine_ || !rf || !lc_ || !fc || !begin || !end) { set_error("examg_restrict: null argument"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (!box_inside(lc_, box, 0)) { set_error("examg_restrict: box leaves the coarse allocation"); return 1; }
  // fine footprint: [2*b - 1, 2*(e-1) + 1]
  Box fb = box;
  fb.b0 = 2 * box.b0; fb.e0 = 2 * (box.e0 - 1) + 1;
  fb.b1 = 2 * box.b1; fb.e1 = 2 * (box.e1 - 1) + 1;
  if (lfine_->nd == 3) { fb.b2 = 2 * box.b2; fb.e2 = 2 * (box.e2 - 1) + 1; }
  if (!box_inside(lfine_, fb, 1)) { set_error("examg_restrict: fine footprint leaves the fine allocation"); return 1; }
  const LayoutDev lf = make_layout(lfine_), lc = make_layout(lc_);
  hipStream_t s = (hipStream_t)stream;
  const bool transformed = lay_split(lfine_) || lay_split(lc_);      // colour-split fields: the generic kernel (transformed index per point)
  if (lfine_->nd == 3 && box.n0() >= 32 && g_restrict_wide && !transformed) {
    const int ntx = (box.n0() + 63) / 64;
    const int rw = g_restrict_rows == 1 ? 1 : 2;      // coarse rows per wave
    const long long cols = (long long)ntx * ((box.n1() + rw - 1) / rw);
    // ~4096 waves up to 512^3 -> 256^3; beyond that shorter chunks (tools/sweep_big_others2.py: 768^3 -> 384^3 0.90 -> 0.83 ms,
    // 1024^3 -> 512^3 2.35 -> 2.14 ms with 65536 waves)
    const int waves_target = g_restrict_waves > 0 ? g_restrict_waves : (box.count() >= 25000000LL ? 65536 : 4096);
    int ntz = (int)((waves_target + cols - 1) / cols);
    if (ntz < 1) ntz = 1;
    int zc = (box.n2() + ntz - 1) / ntz;
    if (zc < 8) zc = 8;
    if (zc > box.n2()) zc = box.n2();
    ntz = (box.n2() + zc - 1) / zc;
    const long long nwaves = cols * ntz;
    if (rw == 2)
      hipLaunchKernelGGL((k_restrict3_wide<2>), dim3((unsigned)((nwaves + 3) / 4)), dim3(64, 4, 1), 0, s, lf, rf, lc, fc, scale, box, ntx, zc,
                         (int)nwaves);
    else
      hipLaunchKernelGGL((k_restrict3_wide<1>), dim3((unsigned)((nwaves + 3) / 4)), dim3(64, 4, 1), 0, s, lf, rf, lc, fc, scale, box, ntx, zc,
                         (int)nwaves);
  } else if (lfine_->nd == 3) hipLaunchKernelGGL((k_restrict<3>), grid_for(box.count()), dim3(256), 0, s, lf, rf, lc, fc, scale, box);
  else if (lfine_->nd == 2) hipLaunchKernelGGL((k_restrict<2>), grid_for(box.count()), dim3(256), 0, s, lf, rf, lc, fc, scale, box);
  else { set_error("examg_restrict: nd must be 2 or 3"); return 1; }
  EXAMG_CHECK_LAUNCH("k_restrict");
  return 0;
}

// RHS@coarser = scale * R * (rhs - A * u): `Residual = RHS - A * Solution` followed by the restriction loop, with the fine
// residual field left out when nothing else reads it (mgCycle, Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:215-223, on a
// block without neighbours).  [fbegin,fend): the residual loop's box, [cbegin,cend): the restriction loop's box.
// 3-D 7-point constant stencils on long rows take the fused kernel and never touch `res`; everything else runs the two
// kernels through `res` (then required).  Bit-identical to examg_residual + examg_restrict either way.
// does examg_residual_restrict run its one-pass kernel (which never touches the residual array) for these arguments?
static bool residual_restrict_one_pass(const examg_layout_t *lu_, const examg_layout_t *lf_, const examg_stencil_t *st, const examg_layout_t *lc_,
                                       const int32_t *fbegin, const int32_t *fend, const int32_t *cbegin, const int32_t *cend) {
  const Box cb = make_box(cbegin, cend), fb = make_box(fbegin, fend);
  if (cb.count() == 0) return false;
  const int ord = canonical_order7(st);
  // fine footprint of the restriction: [2*cb - 1, 2*(ce-1) + 1] must lie inside the residual loop's box
  const bool inside = 2 * cb.b0 - 1 >= fb.b0 && 2 * (cb.e0 - 1) + 1 < fb.e0 && 2 * cb.b1 - 1 >= fb.b1 && 2 * (cb.e1 - 1) + 1 < fb.e1 &&
                      2 * cb.b2 - 1 >= fb.b2 && 2 * (cb.e2 - 1) + 1 < fb.e2;
  const bool left_ok = 2 * (cb.b0 - 1) >= -(lu_->pad_l[0] + lu_->ghost_l[0]) && 2 * (cb.b0 - 1) >= -(lf_->pad_l[0] + lf_->ghost_l[0]);
  if (lay_split(lu_) || lay_split(lf_) || lay_split(lc_)) return false;         // colour-split fields: the two loops
  if (small_residual_restrict_ok(lu_, lf_, st, lc_, fb, cb)) return true;      // launch-bound levels: kernels_small.hip
  return g_restrict_wide && lu_->nd == 3 && ord >= 0 && cb.n0() >= 32 && inside && left_ok && box_inside(lu_, fb, 1) && box_inside(lf_, fb, 0) &&
         box_inside(lc_, cb, 0);
}

extern "C" int examg_residual_restrict_one_pass(const examg_layout_t *lu, const examg_layout_t *lf, const examg_stencil_t *st, const examg_layout_t *lc,
                                                const int32_t *fbegin, const int32_t *fend, const int32_t *cbegin, const int32_t *cend) {
  if (!lu || !lf || !st || !lc || !fbegin || !fend || !cbegin || !cend) return 0;
  return residual_restrict_one_pass(lu, lf, st, lc, fbegin, fend, cbegin, cend) ? 1 : 0;
}

extern "C" int examg_residual_restrict(const examg_layout_t *lu_, const double *u, const examg_layout_t *lf_, const double *rhs,
                                       const examg_layout_t *lr_, double *res, const examg_stencil_t *st,
                                       const examg_layout_t *lc_, double *fc, double scale, const int32_t *fbegin,
                                       const int32_t *fend, const int32_t *cbegin, const int32_t *cend, examg_stream_t stream) {
  if (!lu_ || !u || !lf_ || !rhs || !st || !lc_ || !fc || !fbegin || !fend || !cbegin || !cend) { set_error("examg_residual_restrict: null argument"); return 1; }
  const Box cb = make_box(cbegin, cend);
  if (cb.count() == 0) return 0;
  const int ord = canonical_order7(st);
  if (!lay_split(lu_) && !lay_split(lf_) && !lay_split(lc_) && small_residual_restrict_ok(lu_, lf_, st, lc_, make_box(fbegin, fend), cb))
    return launch_small_residual_restrict(lu_, u, lf_, rhs, lc_, fc, st, scale, cb, (hipStream_t)stream);
  if (residual_restrict_one_pass(lu_, lf_, st, lc_, fbegin, fend, cbegin, cend)) {
    const LayoutDev lu = make_layout(lu_), lf = make_layout(lf_), lc = make_layout(lc_);
    Coef7 k;
    for (int i = 0; i < 7; ++i) k.c[i] = st->coef[i];
    const int ntx = (cb.n0() + 62) / 63;
    // coarse rows per wave: two from 512^3 -> 256^3 (0.520 -> 0.496 ms; 242 VGPRs, two waves per SIMD), one below (256^3 -> 128^3: 0.067 ms
    // against 0.074 with two)
    const int rw = g_rr_rows > 0 ? g_rr_rows : (cb.count() >= 8000000LL ? 2 : 1);
    const long long cols = (long long)ntx * ((cb.n1() + rw - 1) / rw);
    int ntz = (int)((g_rr_waves + cols - 1) / cols);
    if (ntz < 1) ntz = 1;
    int zc = (cb.n2() + ntz - 1) / ntz;
    if (zc < g_rr_minzc) zc = g_rr_minzc;
    if (zc > cb.n2()) zc = cb.n2();
    ntz = (cb.n2() + zc - 1) / zc;
    const long long nwaves = cols * ntz;
    const int order = g_rr_order;
    const long long nwg = order == 1 ? (long long)ntz * ntx * (((cb.n1() + rw - 1) / rw + 3) / 4) : (nwaves + 3) / 4;
    dim3 grid((unsigned)nwg), block(64, 4, 1);
    hipStream_t s = (hipStream_t)stream;
#define EXAMG_RR(O, W) hipLaunchKernelGGL((k_residual_restrict3<O, W>), grid, block, 0, s, lu, u, lf, rhs, lc, fc, k, scale, cb, ntx, zc, (int)nwaves, order)
    if (rw == 2) {
      if (ord == 0) EXAMG_RR(0, 2);
      else EXAMG_RR(1, 2);
    } else {
      if (ord == 0) EXAMG_RR(0, 1);
      else EXAMG_RR(1, 1);
    }
#undef EXAMG_RR
    EXAMG_CHECK_LAUNCH("k_residual_restrict3");
    return 0;
  }
  if (!lr_ || !res) { set_error("examg_residual_restrict: this stencil / box needs the residual array"); return 1; }
  int rc = examg_residual(lu_, u, lf_, rhs, lr_, res, st, fbegin, fend, stream);
  if (rc) return rc;
  return examg_restrict(lr_, res, lc_, fc, scale, cbegin, cend, stream);
}

extern "C" int examg_prolong_add(const examg_layout_t *lc_, const double *uc, const examg_layout_t *lfine_, double *uf,
                                 const int32_t *begin, const int32_t *end, examg_stream_t stream) {
  if (!lfine_ || !uf || !lc_ || !uc || !begin || !end) { set_error("examg_prolong_add: null argument"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (!box_inside(lfine_, box, 0)) { set_error("examg_prolong_add: box leaves the fine allocation"); return 1; }
  Box cb = box;  // coarse footprint [b/2 (floor), ceil((e-1)/2)]
  cb.b0 = box.b0 / 2; cb.e0 = (box.e0 - 1 + 1) / 2 + 1;
  cb.b1 = box.b1 / 2; cb.e1 = (box.e1 - 1 + 1) / 2 + 1;
  if (lfine_->nd == 3) { cb.b2 = box.b2 / 2; cb.e2 = (box.e2 - 1 + 1) / 2 + 1; }
  if (box.b0 < 0 || box.b1 < 0 || box.b2 < 0) { set_error("examg_prolong_add: negative fine index"); return 1; }
  if (!box_inside(lc_, cb, 0)) { set_error("examg_prolong_add: coarse footprint leaves the coarse allocation"); return 1; }
  const LayoutDev lf = make_layout(lfine_), lc = make_layout(lc_);
  hipStream_t s = (hipStream_t)stream;
  if (lfine_->nd == 3 && box.n0() >= 32 && !lay_split(lfine_) && !lay_split(lc_)) {
    const int x0 = box.b0 & ~1;
    const int npairs = (box.e0 - x0 + 1) / 2;
    const int zb = g_prolong_zb > 0 ? g_prolong_zb : (box.count() >= 200000000LL ? 1 : 2);    // 768^3: 1.73 -> 1.68 ms with one plane pair per workgroup
    dim3 grid((npairs + 63) / 64, (box.n1() + 3) / 4, (box.n2() + zb - 1) / zb), block(64, 4, 1);
    hipLaunchKernelGGL(k_prolong_add3_pairs, grid, block, 0, s, lc, uc, lf, uf, box, x0, npairs, zb);
  } else if (lfine_->nd == 3) hipLaunchKernelGGL((k_prolong_add<3>), grid_for(box.count()), dim3(256), 0, s, lc, uc, lf, uf, box);
  else if (lfine_->nd == 2) hipLaunchKernelGGL((k_prolong_add<2>), grid_for(box.count()), dim3(256), 0, s, lc, uc, lf, uf, box);
  else { set_error("examg_prolong_add: nd must be 2 or 3"); return 1; }
  EXAMG_CHECK_LAUNCH("k_prolong_add");
  return 0;
}
