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
          const double tv = wgt * rf[base + a + lfine.s1 * b];
          acc = first ? tv : acc + tv;
          first = false;
        } else {
#pragma unroll
          for (int c = -1; c <= 1; ++c) {
            const double wgt = scale * ((w1[a + 1] * w1[b + 1]) * w1[c + 1]);
            const double tv = wgt * rf[base + a + lfine.s1 * b + lfine.s2 * c];
            acc = first ? tv : acc + tv;
            first = false;
          }
        }
      }
    fc[lidx(lc, I0, I1, I2)] = acc;
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
        const long long k = lidx(lc, j, rs ? r1 : r0, qs ? q1 : q0);
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
    double *q = uf + lidx(lfine, xa, i1, i2);
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

static inline dim3 grid_for(long long total) {
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  return dim3((unsigned)nb);
}

}  // namespace examg

using namespace examg;

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
  if (lfine_->nd == 3) hipLaunchKernelGGL((k_restrict<3>), grid_for(box.count()), dim3(256), 0, s, lf, rf, lc, fc, scale, box);
  else if (lfine_->nd == 2) hipLaunchKernelGGL((k_restrict<2>), grid_for(box.count()), dim3(256), 0, s, lf, rf, lc, fc, scale, box);
  else { set_error("examg_restrict: nd must be 2 or 3"); return 1; }
  EXAMG_CHECK_LAUNCH("k_restrict");
  return 0;
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
  if (lfine_->nd == 3 && box.n0() >= 32) {
    const int x0 = box.b0 & ~1;
    const int npairs = (box.e0 - x0 + 1) / 2;
    const int zb = 8;
    dim3 grid((npairs + 63) / 64, (box.n1() + 3) / 4, (box.n2() + zb - 1) / zb), block(64, 4, 1);
    hipLaunchKernelGGL(k_prolong_add3_pairs, grid, block, 0, s, lc, uc, lf, uf, box, x0, npairs, zb);
  } else if (lfine_->nd == 3) hipLaunchKernelGGL((k_prolong_add<3>), grid_for(box.count()), dim3(256), 0, s, lc, uc, lf, uf, box);
  else if (lfine_->nd == 2) hipLaunchKernelGGL((k_prolong_add<2>), grid_for(box.count()), dim3(256), 0, s, lc, uc, lf, uf, box);
  else { set_error("examg_prolong_add: nd must be 2 or 3"); return 1; }
  EXAMG_CHECK_LAUNCH("k_prolong_add");
  return 0;
}
