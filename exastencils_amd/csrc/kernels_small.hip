// One-pass kernels for the launch-bound levels of the cycle (rows shorter than 64 points: 64^3 and 32^3 at config 3) -- gfx950.
//
// On these levels every loop is a few microseconds of work and the cycle's time is the NUMBER of kernels: 16 launches per level
// and V(3,3) cycle when every `loop over` of mgCycle (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:203-249) is one kernel -- the two
// colour loops of a sweep (baseExt/l4/L4_ColorLoops.scala:44-66), residual, restriction, `Solution@coarser = 0`, correction.
// (What does NOT pay on this part: one persistent kernel with device-wide barriers between the loops -- a barrier over 64 .. 256
// workgroups costs 6 .. 21 us against 2.3 .. 4.7 us for a kernel boundary inside a hipGraph, profiles/r04_barrier_probe.txt.)
// The forms below are the small-level counterparts of the one-pass kernels of the large levels, behind the SAME entry points:
//   k_small_two_stage<COL, VAR> both colour loops of a red-black sweep (COL) or two Jacobi steps, out of place (examg_rbgs_sweep_fused,
//                               examg_jacobi2; VAR 1: the correction `Solution += P * Solution@coarser` folded in, examg_*_prolong;
//                               VAR 2: the zero field as input, examg_rbgs_sweep_fused_zero).  A workgroup owns SY x SZ whole rows:
//                               it stages them with a halo of two rows in LDS, runs the first loop on the tile grown by one row
//                               (colours: in place in LDS, a 7-point star reads the other colour only; Jacobi: into a second
//                               tile), the second loop on the tile, and stores the tile.
//   k_small_residual_restrict   `Residual = RHS - A * Solution; RHS@coarser = R * Residual` without storing the residual
//                               (examg_residual_restrict): one coarse point per thread, the 27 residuals recomputed from L1 / L2.
// Seven launches per level and cycle instead of sixteen.  Every value is the loops' own expression in the loops' own order
// (k_stencil_generic, k_restrict, k_prolong_add: entries folded left to right, no contraction): bit-identical to the launch chain.
#include "examg_common.h"

namespace examg {

constexpr int SM_THREADS = 1024;    // 16 waves: a wave owns whole rows of the tile
constexpr int SM_SY = 4, SM_SZ = 4;      // output rows of a workgroup; staged: (SY + 4) x (SZ + 4) rows of n0 + 2 points
constexpr int SM_MAX_ENT = 7;

struct SmallStencil {
  int nent;
  int lo[SM_MAX_ENT];          // offsets in the LDS tile
  long long uo[SM_MAX_ENT];    // offsets in the u array
  double coef[SM_MAX_ENT];
};

// 7-point star with constant coefficients, entries in any order: what both kernels take (an in-place colour update in LDS is only
// the loop's result if no entry reaches a point of the same colour)
static bool small_star(const examg_stencil_t *st) {
  if (st->cfield || st->nent < 1 || st->nent > SM_MAX_ENT) return false;
  for (int k = 0; k < st->nent; ++k) {
    int nz = 0;
    for (int d = 0; d < 3; ++d) {
      if (st->off[k][d] < -1 || st->off[k][d] > 1) return false;
      nz += st->off[k][d] != 0;
    }
    if (nz > 1) return false;
  }
  return true;
}

// value of `Solution += P * Solution@coarser` at one fine point (k_prolong_add<3>)
__device__ __forceinline__ double small_prolong(const LayoutDev &lc, const double *uc, int i0, int i1, int i2) {
  const int ii[3] = {i0, i1, i2};
  int n[3], ci[3][2];
  double cw[3][2];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
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
          const double tv = ((cw[0][a] * cw[1][b]) * cw[2][c]) * uc[lidx_plain(lc, ci[0][a], ci[1][b], ci[2][c])];
          acc = first ? tv : acc + tv;
          first = false;
        }
      }
  return acc;
}

// COL: the two colour loops of a red-black sweep (else two Jacobi steps).  VAR 0: plain; 1: correction folded in; 2: zero input (COL)
// Launch-bound work: what counts is the length of the dependent chain, not the bytes.  16 waves per workgroup; a wave owns whole rows of
// the tile (lane = x, no index divisions) and issues all loads of its rows before the first use; the right-hand sides of both loops are
// in flight from the start.
template <bool COL, int VAR>
__global__ void __launch_bounds__(SM_THREADS)
k_small_two_stage(LayoutDev lu, const double *__restrict__ u_in, double *__restrict__ u_out, LayoutDev lf, const double *__restrict__ rhs,
                  SmallStencil st, double w, Box box, int first, int tiles_y, LayoutDev lc, const double *__restrict__ uc) {
  extern __shared__ double T[];
  const int n0 = box.n0(), LX = n0 + 2;
  constexpr int RY = SM_SY + 4, RZ = SM_SZ + 4, NW = SM_THREADS / 64;
  constexpr int ROWS0 = RY * RZ, ROWS1 = (SM_SY + 2) * (SM_SZ + 2), ROWS2 = SM_SY * SM_SZ;
  double *T2 = T + LX * ROWS0;                                        // Jacobi: the first step's values (same indexing as T)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ty = blockIdx.x % tiles_y, tz = blockIdx.x / tiles_y;
  const int oy = box.b1 + ty * SM_SY, oz = box.b2 + tz * SM_SZ;       // first output row of this workgroup
  const int x0 = box.b0 - 1, y0 = oy - 2, z0 = oz - 2;               // iterator coordinates of T[0]
  // ---- stage 0: the input tile (VAR 1: with the correction on the points of the box); rows r = wv, wv + NW, .. ---------------------
  {
    constexpr int NR = (ROWS0 + NW - 1) / NW;
    double v[NR][2];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int r = wv + j * NW;
      const int y = y0 + r % RY, z = z0 + r / RY;
      // rows beyond the box's one-point shell are never read (a row outside the box is not updated, so its neighbours are not needed)
      const bool need = VAR != 2 && r < ROWS0 && y >= box.b1 - 1 && y <= box.e1 && z >= box.b2 - 1 && z <= box.e2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int xl = lane + 64 * h;
        double val = 0.0;
        if (need && xl < LX) {
          const int x = x0 + xl;
          val = u_in[lidx_plain(lu, x, y, z)];
          if (VAR == 1 && x >= box.b0 && x < box.e0 && y >= box.b1 && y < box.e1 && z >= box.b2 && z < box.e2) val = val + small_prolong(lc, uc, x, y, z);
        }
        v[j][h] = val;
      }
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int r = wv + j * NW;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int xl = lane + 64 * h;
        if (r < ROWS0 && xl < LX) {
          T[r * LX + xl] = v[j][h];
          if (!COL) T2[r * LX + xl] = v[j][h];                         // points the first step does not update keep the input value
        }
      }
    }
  }
  __syncthreads();
  if (COL) {
    // ---- colour `first` on the tile grown by one row, then the other colour on the tile: in place (a star reads the other colour).
    //      Half a wave per row: lanes 0..31 / 32..63 take the points of the colour in two consecutive rows of the stage. ------------
    const int hl = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int stage = 0; stage < 2; ++stage) {
      const int colour = stage == 0 ? first : 1 - first;
      const int g = stage == 0 ? 1 : 0;                                // rows beyond the tile on each side
      const int ny = SM_SY + 2 * g, nrows = stage == 0 ? ROWS1 : ROWS2;
      for (int r = 2 * wv + hh; r < nrows; r += 2 * NW) {
        const int y = oy - g + r % ny, z = oz - g + r / ny;
        if (y < box.b1 || y >= box.e1 || z < box.b2 || z >= box.e2) continue;
        const int xf = box.b0 + (((box.b0 + y + z) & 1) != colour ? 1 : 0);
        for (int x = xf + 2 * hl; x < box.e0; x += 64) {
          const int it = (x - x0) + LX * ((y - y0) + RY * (z - z0));
          const double f = rhs[lidx_plain(lf, x, y, z)];
          double acc = st.coef[0] * T[it + st.lo[0]];
          for (int k = 1; k < st.nent; ++k) acc = acc + st.coef[k] * T[it + st.lo[k]];
          T[it] = T[it] + w * (f - acc);
        }
      }
      __syncthreads();
    }
  } else {
    // ---- first Jacobi step on the tile grown by one row: T -> T2; second step on the tile: T2 -> T ------------------------------------
    for (int r = wv; r < ROWS1; r += NW) {
      const int y = oy - 1 + r % (SM_SY + 2), z = oz - 1 + r / (SM_SY + 2);
      if (y < box.b1 || y >= box.e1 || z < box.b2 || z >= box.e2) continue;
      for (int x = box.b0 + lane; x < box.e0; x += 64) {
        const int it = (x - x0) + LX * ((y - y0) + RY * (z - z0));
        const double f = rhs[lidx_plain(lf, x, y, z)];
        double acc = st.coef[0] * T[it + st.lo[0]];
        for (int k = 1; k < st.nent; ++k) acc = acc + st.coef[k] * T[it + st.lo[k]];
        T2[it] = T[it] + w * (f - acc);
      }
    }
    __syncthreads();
    for (int r = wv; r < ROWS2; r += NW) {
      const int y = oy + r % SM_SY, z = oz + r / SM_SY;
      if (y >= box.e1 || z >= box.e2) continue;
      for (int x = box.b0 + lane; x < box.e0; x += 64) {
        const int it = (x - x0) + LX * ((y - y0) + RY * (z - z0));
        const double f = rhs[lidx_plain(lf, x, y, z)];
        double acc = st.coef[0] * T2[it + st.lo[0]];
        for (int k = 1; k < st.nent; ++k) acc = acc + st.coef[k] * T2[it + st.lo[k]];
        T[it] = T2[it] + w * (f - acc);
      }
    }
    __syncthreads();
  }
  // ---- the tile's rows, box points only ---------------------------------------------------------------------------------------------
  for (int r = wv; r < ROWS2; r += NW) {
    const int y = oy + r % SM_SY, z = oz + r / SM_SY;
    if (y >= box.e1 || z >= box.e2) continue;
    for (int x = box.b0 + lane; x < box.e0; x += 64) u_out[lidx_plain(lu, x, y, z)] = T[(x - x0) + LX * ((y - y0) + RY * (z - z0))];
  }
}

// RHS@coarser = scale * R * (RHS - A * Solution) (k_stencil_generic<EXAMG_RESIDUAL> + k_restrict<3>): 32 lanes per coarse point -- lane
// j < 27 forms the residual of fine point j (x offset outermost, then y, then z: k_restrict's order) and its weighted term, all 27 in
// flight at once; the terms are then added in that order (every lane of the group folds the same sequence, lane 0 stores).
__global__ void __launch_bounds__(256)
k_small_residual_restrict(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs, LayoutDev lc,
                          double *__restrict__ fc, SmallStencil st, double scale, Box cb) {
  const long long total = cb.count();
  const int n0 = cb.n0(), n1 = cb.n1();
  const int j = threadIdx.x & 31;
  const long long t = ((long long)blockIdx.x * 256 + threadIdx.x) >> 5;       // coarse point of this half wave
  const bool live = t < total;
  const long long tt = live ? t : 0;
  const int I0 = cb.b0 + (int)(tt % n0);
  const long long row = tt / n0;
  const int I1 = cb.b1 + (int)(row % n1);
  const int I2 = cb.b2 + (int)(row / n1);
  double tv = 0.0;
  if (j < 27) {
    const int a = j / 9 - 1, b = (j / 3) % 3 - 1, c = j % 3 - 1;
    const double w1[3] = {0.25, 0.5, 0.25};
    const int i0 = 2 * I0 + a, i1 = 2 * I1 + b, i2 = 2 * I2 + c;
    const long long iu = lidx_plain(lu, i0, i1, i2);
    double r = st.coef[0] * u[iu + st.uo[0]];
    for (int k = 1; k < st.nent; ++k) r = r + st.coef[k] * u[iu + st.uo[k]];
    r = rhs[lidx_plain(lf, i0, i1, i2)] - r;
    const double wgt = scale * ((w1[a + 1] * w1[b + 1]) * w1[c + 1]);
    tv = wgt * r;
  }
  const int base = threadIdx.x & 32;          // first lane of this group within its wave
  double acc = __shfl(tv, base);
#pragma unroll
  for (int k = 1; k < 27; ++k) acc = acc + __shfl(tv, base + k);
  if (live && j == 0) fc[lidx_plain(lc, I0, I1, I2)] = acc;
}

static thread_local int g_small_disable = 0;      // examg_debug_small(1): the plain loops instead (A/B and parity tests)

static SmallStencil small_stencil(const examg_stencil_t *st, const LayoutDev &lu, int LX, int RY) {
  SmallStencil s;
  s.nent = st->nent;
  for (int k = 0; k < st->nent; ++k) {
    s.lo[k] = st->off[k][0] + LX * (st->off[k][1] + RY * st->off[k][2]);
    s.uo[k] = st->off[k][0] + lu.s1 * st->off[k][1] + lu.s2 * st->off[k][2];
    s.coef[k] = st->coef[k];
  }
  return s;
}

// ---- dispatch hooks of the entry points in kernels_twostage.hip / kernels_transfer.hip ---------------------------------------------
// Does the small-level pass take these arguments?  (rows shorter than the two-stage kernel's 64 points)
bool small_two_stage_ok(const examg_layout_t *lu, const examg_layout_t *lf, const examg_stencil_t *st, const Box &box) {
  return !g_small_disable && !lay_split(lu) && !lay_split(lf) && lu->nd == 3 && lf->nd == 3 && small_star(st) && box.n0() >= 2 && box.n0() < 64 && box.n1() >= 1 && box.n2() >= 1 &&
         box_inside(lu, box, 1) && box_inside(lf, box, 0);
}

// col: the two colour loops of a red-black sweep (colour `first` first), else two Jacobi steps;
// var: 0 plain (u_in read), 1 correction from (lc, uc) folded in, 2 zero input (u_in not read; red-black only)
int launch_small_two_stage(bool col, int var, const examg_layout_t *lu_, const double *u_in, double *u_out, const examg_layout_t *lf_, const double *rhs,
                           const examg_stencil_t *st, double w, int first, const Box &box, const examg_layout_t *lc_, const double *uc, hipStream_t s) {
  const LayoutDev lu = make_layout(lu_), lf = make_layout(lf_);
  const int LX = box.n0() + 2, RY = SM_SY + 4, RZ = SM_SZ + 4;
  const SmallStencil ss = small_stencil(st, lu, LX, RY);
  const int tiles_y = (box.n1() + SM_SY - 1) / SM_SY, tiles_z = (box.n2() + SM_SZ - 1) / SM_SZ;
  const size_t lds = (size_t)LX * RY * RZ * sizeof(double) * (col ? 1 : 2);
  const LayoutDev lc = lc_ ? make_layout(lc_) : lu;
  dim3 grid((unsigned)(tiles_y * tiles_z)), block(SM_THREADS);
  // the Jacobi form stages two tiles: 66.6 KB at 63-point rows, above the 64 KB a kernel gets without asking (gfx950: 160 KB per CU)
#define EXAMG_SMALL(C, V)                                                                                                                   \
  do {                                                                                                                                      \
    if (lds > 48 * 1024 && check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_small_two_stage<C, V>),                          \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "k_small_two_stage: LDS size")) \
      return 1;                                                                                                                             \
    hipLaunchKernelGGL((k_small_two_stage<C, V>), grid, block, lds, s, lu, u_in, u_out, lf, rhs, ss, w, box, first, tiles_y, lc, uc);       \
  } while (0)
  if (col) {
    if (var == 0) EXAMG_SMALL(true, 0);
    else if (var == 1) EXAMG_SMALL(true, 1);
    else EXAMG_SMALL(true, 2);
  } else {
    if (var == 2) { set_error("small-level Jacobi pair: no zero-input form"); return 1; }
    if (var == 0) EXAMG_SMALL(false, 0);
    else EXAMG_SMALL(false, 1);
  }
#undef EXAMG_SMALL
  EXAMG_CHECK_LAUNCH("k_small_two_stage");
  return 0;
}

// residual + restriction of a small level: the conditions of the one-pass form (fine footprint inside the residual loop's box) with
// coarse rows shorter than the wide kernel's 32 points
bool small_residual_restrict_ok(const examg_layout_t *lu, const examg_layout_t *lf, const examg_stencil_t *st, const examg_layout_t *lc, const Box &fb,
                                const Box &cb) {
  if (g_small_disable || lay_split(lu) || lay_split(lf) || lay_split(lc) || lu->nd != 3 || lf->nd != 3 || lc->nd != 3 || !small_star(st) || cb.count() == 0 || cb.n0() >= 32) return false;
  const bool inside = 2 * cb.b0 - 1 >= fb.b0 && 2 * (cb.e0 - 1) + 1 < fb.e0 && 2 * cb.b1 - 1 >= fb.b1 && 2 * (cb.e1 - 1) + 1 < fb.e1 &&
                      2 * cb.b2 - 1 >= fb.b2 && 2 * (cb.e2 - 1) + 1 < fb.e2;
  return inside && box_inside(lu, fb, 1) && box_inside(lf, fb, 0) && box_inside(lc, cb, 0);
}

int launch_small_residual_restrict(const examg_layout_t *lu_, const double *u, const examg_layout_t *lf_, const double *rhs, const examg_layout_t *lc_,
                                   double *fc, const examg_stencil_t *st, double scale, const Box &cb, hipStream_t s) {
  const LayoutDev lu = make_layout(lu_), lf = make_layout(lf_), lc = make_layout(lc_);
  const SmallStencil ss = small_stencil(st, lu, 0, 0);
  const long long nb = (cb.count() * 32 + 255) / 256;       // 32 lanes per coarse point
  hipLaunchKernelGGL(k_small_residual_restrict, dim3((unsigned)nb), dim3(256), 0, s, lu, u, lf, rhs, lc, fc, ss, scale, cb);
  EXAMG_CHECK_LAUNCH("k_small_residual_restrict");
  return 0;
}

}  // namespace examg

#ifdef EXAMG_DEBUG_HOOKS
extern "C" int examg_debug_small(int disable) {
  examg::g_small_disable = disable;
  return 0;
}
#endif
