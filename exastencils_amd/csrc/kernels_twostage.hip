// Two dependent 7-point stencil applications in ONE pass over HBM (gfx950):
//   * fused red-black Gauss-Seidel sweep: stage 1 updates the points of the first colour, stage 2 the
//     points of the other colour from the stage-1 values -- exactly the two `color with` loops of
//     Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:204-213, which the reference runs as two launches
//     over the full box (48 B per update pair; Compiler/src/exastencils/baseExt/l4/L4_ColorLoops.scala:44-66);
//   * two Jacobi steps (Testing/Smoothers/Jac.exa4:125-131 applied twice): temporal blocking, the
//     reference's IR_ContractingLoop idea (baseExt/ir/IR_ContractingLoop.scala) for the
//     CommFullTempBlockable layouts of the generated-from-L3 programs.
// Both read u and rhs once and write the result once: 24 B per point per pass.
//
// Structure: out of place (u_out != u_in, the caller swaps pointers), so tiles are independent: every workgroup
// recomputes the stage-1 values of its halo (2-point halo in x, y and z of the input, 1-point halo of the
// stage-1 field).  A workgroup loads a 128-point x window (2 per lane) and produces the inner 124; its waves own two rows
// each, share their y-neighbour rows through LDS and march in z with register pipelines for the input and for the
// stage-1 field.  x-neighbours come from the adjacent lanes.  (Round 1 also had a variant without LDS, every wave
// recomputing its own y-halo: 0.97 ms against 0.72 ms at 512^3; removed.)
// Arithmetic per point is the same expression, in the same order, as the one-stage kernels: results are
// bit-identical to running the two loops one after the other.
#include "examg_common.h"

#include <type_traits>

namespace examg {

struct TSGeom {
  int ntx, nty, ntz, zc, nblocks, remap;
  int xs;                  // 1: windows start one point further left, which makes every 16-byte access aligned (padded layouts)
  int ys, zs;              // PROL: row groups / z chunks start one point earlier (canonical parities, see the kernel)
  int first;               // COL: colour updated in stage 1
  Box box1;                // stage-1 box (contains the output box); points outside keep the input value
  int ax0, ax1, ay0, ay1, az0, az1;  // allocation of u in iterator coordinates, half open
};

constexpr int TS_OUT = 124;  // outputs per 128-point window

// A wave-uniform 64-bit value back on the scalar unit (a 64-bit multiply is selected as a vector instruction even for uniform
// operands and would drag every address that contains it into vector registers): set-up code only.
__device__ __forceinline__ long long uniform64(long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// Prolongation + correction folded into the pass (PROL variants): the stages see u + P(uc) on `box` instead of u --
// `Solution += Prolongation@coarser * Solution@coarser` followed by the first post-smoothing sweep
// (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:240-247) in one pass, without the 16 B per point of the correction loop.
struct TSProl {
  LayoutDev lc;
  const double *uc;
};

// ---------------------------------------------------------------------------------------------------------------
// The NW waves of a workgroup share one 128-point x window and a stack of 2*NW stage-1 rows.  Every wave
// owns two consecutive rows: it keeps their z pipeline (input planes q-1, q, q+1 and stage-1 planes m-1, m, m+1) in
// registers and publishes plane q+1 of the input and plane q of the stage-1 field in LDS, from where the waves above
// and below read their y-neighbour rows.  Each input row is loaded from memory once per workgroup (2*NW + 2 rows for
// 2*NW - 2 output rows) instead of three times, the first stage is evaluated on 2*NW rows instead of 4*NW - 4, and a
// wave needs ~100 instead of 250 registers, so two workgroups share a CU.  One barrier per plane (double-buffered LDS).
// ---------------------------------------------------------------------------------------------------------------
#ifndef TS_SCALAR_WV
#define TS_SCALAR_WV 1
#endif
//
// PROL: the coarse values a workgroup interpolates from -- (NI/2 + 1) rows x 65 columns per coarse plane -- pass through LDS
// as well: two plane buffers CB[P & 1], filled from a one-plane register prefetch every second step.  The correction of
// input plane q+1 is added at the start of step q (own rows and the outer halo row), before the plane is used or published,
// with the terms and the summation order of k_prolong_add3_pairs (kernels_transfer.hip): bit-identical to the two loops.
//
// VAR: 0 = as described; 1 = PROL; 2 = the input field is zero everywhere (a coarse level's first pre-smoothing pass after
// `Solution = 0`): nothing is loaded for it, the arithmetic is the same expression on the constant 0.0.
template <int ORDER, bool COL, int NW, bool NT, int WPE, int VAR, int PF = 0, int RPW = 2>
__global__ void __launch_bounds__(64 * NW, WPE)
k_two_stage7_lds(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs,
                 double *__restrict__ out, Coef7 k, double w, Box box, TSGeom g, TSProl pr) {
  constexpr bool PROL = VAR == 1, ZIN = VAR == 2;
  static_assert(RPW == 2 || VAR == 0, "the folded forms are written for two rows per wave");
  constexpr int NS = RPW * NW;      // stage-1 rows of the workgroup: s = 0 .. NS-1, global row rw0 - 1 + s (RPW rows per wave)
  constexpr int NI = NS + 2;        // input rows: i = 0 .. NI-1, global row rw0 - 2 + i (centre of stage-1 row s is i = s+1)
  constexpr int NO = NS - 2;        // output rows: stage-1 rows 1 .. NS-2
  constexpr int NCR = NI / 2 + 1, CW = 66, CT = NCR * CW;   // coarse tile: rows, row pitch (65 columns used), doubles per plane
  __shared__ d2 SM[2 * NI * 64 + 2 * NS * 64 + (PROL ? CT : 0)];
  d2 *const UBp = SM;                   // input plane p, row i:    UB(p & 1, i)
  d2 *const VBp = SM + 2 * NI * 64;     // stage-1 plane p, row s:  VB(p & 1, s)
  double *const CBp = reinterpret_cast<double *>(SM + 2 * NI * 64 + 2 * NS * 64);   // coarse plane P: CBp + (P & 1) * CT
#define UB(p, i) UBp[((p) * NI + (i)) * 64 + lane]
#define VB(p, s) VBp[((p) * NS + (s)) * 64 + lane]
  // threadIdx.y is the same for all lanes of a wave: as a scalar it keeps row predicates and LDS row addresses on the SALU
  const int lane = threadIdx.x, wv = TS_SCALAR_WV ? __builtin_amdgcn_readfirstlane(threadIdx.y) : threadIdx.y;
  int t = blockIdx.x;
  if (g.remap == 1) {  // XCD-contiguous tile order (workgroups are dealt round-robin to the 8 XCDs): neighbouring tiles share an L2
    const int per = g.nblocks >> 3;
    if (t < (per << 3)) t = (t & 7) * per + (t >> 3);
  } else if (g.remap == 2) {  // XCD-contiguous within every z layer of tiles: all XCDs stay in the same planes, y-neighbours share an L2
    const int xy = g.ntx * g.nty;
    const int lz = t / xy, r = t - lz * xy;
    const int per = xy >> 3;
    t = lz * xy + (r < (per << 3) ? (r & 7) * per + (r >> 3) : r);
  }
  const int tx = t % g.ntx;
  t /= g.ntx;
  const int ty = t % g.nty;
  const int tz = t / g.nty;
  const int xw = box.b0 - 2 - g.xs + TS_OUT * tx;   // first point of the window
  const int xa = xw + 2 * lane;
  const int rw0 = box.b1 - g.ys + ty * NO;      // first output row of the workgroup
  const int mb = box.b2 - g.zs + tz * g.zc;     // first output plane (PROL: the first chunk may start one plane before the box)
  const int mlo = max(mb, box.b2);
  const int me = min(mb + g.zc, box.e2);
  const Box &box1 = g.box1;
  const bool inx_a = xa >= box.b0 && xa < box.e0, inx_b = xa + 1 >= box.b0 && xa + 1 < box.e0;
  const bool in1_a = xa >= box1.b0 && xa < box1.e0, in1_b = xa + 1 >= box1.b0 && xa + 1 < box1.e0;
  const bool out_lane = lane >= 1 && lane <= 62;
  const bool st_a = out_lane && inx_a, st_b = out_lane && inx_b;

  // this wave: stage-1 rows s0 = 2*wv, s0 + 1; their centre input rows i = s + 1; global rows
  const int s0 = RPW * wv;
  int grow[RPW];
  bool row_in1[RPW], row_out[RPW];      // output rows: stage-1 rows 1 .. NS-2 inside the box
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    grow[r] = rw0 - 1 + s0 + r;
    row_in1[r] = grow[r] >= box1.b1 && grow[r] < box1.e1;
    row_out[r] = s0 + r >= 1 && s0 + r <= NS - 2 && grow[r] >= box.b1 && grow[r] < box.e1;
  }
  // the two outermost input rows (i = 0 and i = NI-1) are nobody's centre row: wave 0 / wave NW-1 carry them along
  const bool has_outer = wv == 0 || wv == NW - 1;
  const int outer_i = wv == 0 ? 0 : NI - 1;

  // Addressing: a SCALAR base per (row, plane) and a per-lane byte offset per row (`global_load_dwordx4 v, v_off, s[base]`): no
  // vector arithmetic per load (32-bit window offsets clamped per lane cost five vector instructions each).  Loads are
  // unconditional.  Rows and planes are clamped into the allocation as scalars -- a window that sticks out reads some other row
  // or plane, which no valid output depends on -- and the lane offset once: a pair that lies entirely left of the allocation's first
  // column moves up to the pair that ends on it (the pair "last element of the row before, first column" is read as it is: its
  // second element can be a needed ghost value), a pair entirely right of the last column down to the pair that begins on it.  On
  // the FIRST / LAST row of the allocation, where such a straddling pair would reach before / past the array on the first / last
  // plane, the pairs stop one column further in: that column of that row is an edge of the allocation box, which no 7-point
  // update and no pass-through of a needed point reads.
  // The byte offset of the plane being loaded is a running scalar: it advances by one plane per step while the plane index stays
  // inside the allocation (no 64-bit multiply in the loop).
  struct Site {
    const char *row;     // (xw, row) on the first allocated plane
    unsigned voff;       // this lane's byte offset
  };
  auto site = [&](const double *base, const LayoutDev &l, int R) {
    const int rc = min(max(R + l.ref1, 0), l.tot1 - 1), i0 = xw + l.ref0;
    Site st;
    st.row = reinterpret_cast<const char *>(base + ((long long)i0 + (long long)((unsigned)l.s1 * (unsigned)rc)));   // planes hold less than 2^32 elements (launcher)
    st.voff = (unsigned)(min(max(2 * lane, -i0 - (rc == 0 ? 0 : 1)), l.tot0 - 1 - i0 - (rc == l.tot1 - 1 ? 1 : 0)) * 8);
    return st;
  };
  struct PlaneCursor {
    int p;                     // plane index + ref2 (unclamped)
    long long bytes, step;     // byte offset of plane clamp(p), bytes per plane
    int last;                  // tot2 - 1
  };
  auto cursor = [&](const LayoutDev &l, int P) {
    PlaneCursor c;
    c.p = P + l.ref2;
    c.last = l.tot2 - 1;
    c.step = uniform64(l.s2 * 8);
    c.bytes = uniform64(l.s2 * 8 * min(max(c.p, 0), c.last));
    return c;
  };
  auto advance = [&](PlaneCursor &c) {
    ++c.p;
    c.bytes += (c.p >= 1 && c.p <= c.last) ? c.step : 0LL;
  };
  auto load_at = [&](const Site &st, long long pb) {
    return load2(reinterpret_cast<const double *>(st.row + pb + st.voff));
  };
  Site urow[RPW], frow[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    urow[r] = site(u, lu, grow[r]);
    frow[r] = site(rhs, lf, grow[r]);
  }
  const Site uouter = site(u, lu, rw0 - 2 + outer_i);
  PlaneCursor cu = cursor(lu, mb - 2), cf = cursor(lf, mb - 1);    // the next plane to load
  auto load_u = [&](const Site &st) {
    if constexpr (ZIN) return d2{0.0, 0.0};
    return load_at(st, cu.bytes);
  };
  auto load_f = [&](int r) { return load_at(frow[r], cf.bytes); };
  // stores: scalar base of (xw, row) on plane 0 of the output + the running offset of plane m + 16 bytes per lane (rows and planes of
  // output points lie inside the allocation; before the first output plane the offset is not used)
  const unsigned vo = (unsigned)lane * 16u;
  char *obase[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
    obase[r] = reinterpret_cast<char *>(out + ((long long)(xw + lu.ref0) + (long long)((unsigned)lu.s1 * (unsigned)max(grow[r] + lu.ref1, 0))));
  long long obytes = uniform64(lu.s2 * 8 * (long long)(mb - 2 + lu.ref2));   // plane m of the first step
  const long long ostep = uniform64(lu.s2 * 8);

  // register pipelines of the two own rows: rings of four slots.  The plane loop is unrolled four times by hand (the barrier
  // keeps the compiler from doing it) and step j of a group uses slot (j + k) & 3 for "k planes ahead", so no value ever moves
  // between registers (the rolled loop spent 62 of its ~280 vector instructions per step on such copies).  In step j:
  //   U[j], U[j+1], U[j+2], U[j+3]   input planes q-1, q, q+1, q+2 (in flight); plane q+3 is loaded into U[j] at the end
  //   V[j], V[j+1], V[j+2]           stage-1 planes m-1, m (m = q-1) and q (computed in the step)
  //   F[j], F[j+1], F[j+2]           rhs on planes m, q, q+1 (in flight); plane q+2 is loaded into F[j+3]
  //   O[j+2], O[j+3]                 outer halo row on planes q+1, q+2; plane q+3 is loaded into O[j]
  d2 U[4][RPW], V[4][RPW], F[4][RPW], O[4];
  d2 Ocur = {0.0, 0.0};
  O[0] = O[1] = O[2] = O[3] = Ocur;
#pragma unroll
  for (int j = 0; j < 4; ++j) {      // input planes mb-2 .. mb+1; the outer row is not needed on the first of them
#pragma unroll
    for (int r = 0; r < RPW; ++r) U[j][r] = load_u(urow[r]);
    if (has_outer && j >= 1) {
      const d2 o = load_u(uouter);
      if (j == 1) Ocur = o;
      else O[j] = o;
    }
    advance(cu);
  }
#pragma unroll
  for (int j = 1; j < 3; ++j) {      // rhs planes mb-1, mb
#pragma unroll
    for (int r = 0; r < RPW; ++r) F[j][r] = load_f(r);
    advance(cf);
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    F[0][r] = F[1][r];
    F[3][r] = F[1][r];
  }

  // ---- PROL: coarse tile bookkeeping and the correction of one pair ----
  // The launcher shifts windows, row groups and z chunks so that every workgroup starts on the same parities: xw even,
  // rw0 odd, mb even.  Lane l then holds an even-x point (coarse column cx0 + l, weight 1) and an odd-x point (columns
  // cx0 + l + 1, then cx0 + l, weight 1/2); a wave's first row is even (coarse tile row wv + 1), its second row odd (tile rows
  // wv + 2, then wv + 1); the outer row of wave 0 is odd (tile rows 1, 0), that of wave NW-1 even (tile row NW + 1); and a
  // chunk's first plane is even.  Which terms a correction has is known at compile time but for the plane's parity.
  const int cy0 = (rw0 - 3) >> 1;                // first coarse row of the tile
  const int Pa = (mb - 2) >> 1;                  // coarse plane of the first input plane
  // this thread's (up to) two elements of the coarse tile: tid and tid + 64 * NW (index, row and column are recomputed where they
  // are used -- every second step -- instead of being carried in registers through the plane loop)
  double Cpf[2] = {0.0, 0.0};                    // coarse plane in flight (the next one to enter LDS)
  auto cel = [&](int kk) { return wv * 64 + lane + kk * 64 * NW; };
  auto cload = [&](int kk, int P) {
    const int e = cel(kk), row = e / CW, col = e - row * CW;
    long long i = pr.lc.origin + (long long)((xw >> 1) + col) + pr.lc.s1 * (long long)(cy0 + row) + pr.lc.s2 * (long long)P;
    i = min(max(i, 0LL), pr.lc.size - 1);      // elements outside the coarse allocation are read by nobody
    return pr.uc[i];
  };
  // v = input pair of a row on plane p, `on` = row and plane inside the box; Th / Tl = tiles of the coarse planes (p+1)/2 and
  // (p-1)/2 of an odd plane (OP) or, twice, of plane p/2; rl = tile row of the row's first entry ((y+1)/2 resp. y/2): an odd row
  // (OY) also reads tile row rl - 1.  Entry order and association of k_prolong_add (kernels_transfer.hip): an odd index i has
  // the entries (i+1)/2, (i-1)/2 with weight 1/2; loops over x entries, then y, then z; weight ((wx * wy) * wz).
  auto corr = [&](d2 v, bool on, const double *Th, const double *Tl, int rl, auto OYc, auto OPc) {
    constexpr bool OY = decltype(OYc)::value, OP = decltype(OPc)::value;
    constexpr double wr = OY ? 0.5 : 1.0, wq = OP ? 0.5 : 1.0;
    constexpr double w1 = (1.0 * wr) * wq, w2 = (0.5 * wr) * wq;
    const double *h0 = Th + rl * CW + lane, *l0 = Tl + rl * CW + lane;   // row slot 0 on plane slots 0 / 1
    // c<column: 0 = cx0 + l, 1 = cx0 + l + 1><row slot><plane slot>
    const double c000 = h0[0], c100 = h0[1];
    double c001 = 0.0, c101 = 0.0, c010 = 0.0, c110 = 0.0, c011 = 0.0, c111 = 0.0;
    if constexpr (OP) { c001 = l0[0]; c101 = l0[1]; }
    if constexpr (OY) { c010 = h0[-CW]; c110 = h0[1 - CW]; }
    if constexpr (OY && OP) { c011 = l0[-CW]; c111 = l0[1 - CW]; }
    double acc1 = w1 == 1.0 ? c000 : w1 * c000;      // even-x point: its one column
    if constexpr (OP) acc1 = acc1 + w1 * c001;
    if constexpr (OY) acc1 = acc1 + w1 * c010;
    if constexpr (OY && OP) acc1 = acc1 + w1 * c011;
    double acc2 = w2 * c100;                          // odd-x point: column cx0 + l + 1, then cx0 + l
    if constexpr (OP) acc2 = acc2 + w2 * c101;
    if constexpr (OY) acc2 = acc2 + w2 * c110;
    if constexpr (OY && OP) acc2 = acc2 + w2 * c111;
    acc2 = acc2 + w2 * c000;
    if constexpr (OP) acc2 = acc2 + w2 * c001;
    if constexpr (OY) acc2 = acc2 + w2 * c010;
    if constexpr (OY && OP) acc2 = acc2 + w2 * c011;
    v.x = (inx_a && on) ? v.x + acc1 : v.x;
    v.y = (inx_b && on) ? v.y + acc2 : v.y;
    return v;
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  const bool rowon[2] = {grow[0] >= box.b1 && grow[0] < box.e1, grow[1] >= box.b1 && grow[1] < box.e1};
  const int orow = rw0 - 2 + outer_i;
  const bool orowon = orow >= box.b1 && orow < box.e1;
  // corrections of one input plane: the wave's two rows and (waves 0, NW-1) its outer halo row
  auto corr_plane = [&](d2 *U2, d2 &O, int p, const double *Th, const double *Tl, auto OPc) {
    const bool pon = p >= box.b2 && p < box.e2;
    U2[0] = corr(U2[0], pon && rowon[0], Th, Tl, wv + 1, F_{}, OPc);
    U2[1] = corr(U2[1], pon && rowon[1], Th, Tl, wv + 2, T_{}, OPc);
    if (wv == 0) O = corr(O, pon && orowon, Th, Tl, 1, T_{}, OPc);
    else if (wv == NW - 1) O = corr(O, pon && orowon, Th, Tl, NW + 1, F_{}, OPc);
  };
  if constexpr (PROL) {
    // coarse planes Pa, Pa + 1 cover the input planes mb-2 .. mb; plane Pa + 2 is in flight
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (cel(kk) >= CT) continue;
      const double a0 = cload(kk, Pa), a1 = cload(kk, Pa + 1);
      Cpf[kk] = cload(kk, Pa + 2);
      CBp[(Pa & 1) * CT + cel(kk)] = a0;
      CBp[((Pa + 1) & 1) * CT + cel(kk)] = a1;
    }
    __syncthreads();
    const double *Ta = CBp + (Pa & 1) * CT, *Tb = CBp + ((Pa + 1) & 1) * CT;
    d2 dummy = {0.0, 0.0};
    corr_plane(U[0], dummy, mb - 2, Ta, Ta, F_{});     // even plane: coarse plane Pa
    corr_plane(U[1], Ocur, mb - 1, Tb, Ta, T_{});      // odd plane: coarse planes Pa + 1, then Pa
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    V[0][r] = U[0][r];
    V[1][r] = U[0][r];
    V[2][r] = U[0][r];
    V[3][r] = U[0][r];
  }
  // publish input plane q0 = mb-1.  LDS plane buffers are chosen by the step's position in its group of four: the input plane
  // and the stage-1 plane of step j live in buffer j & 1 (compile-time LDS offsets)
  {
#pragma unroll
    for (int r = 0; r < RPW; ++r) UB(0, s0 + 1 + r) = U[1][r];
    if (has_outer) UB(0, outer_i) = Ocur;
  }
  __syncthreads();

  using I0_ = std::integral_constant<int, 0>;
  using I1_ = std::integral_constant<int, 1>;
  using I2_ = std::integral_constant<int, 2>;
  using I3_ = std::integral_constant<int, 3>;
  // COL: which point of a pair a stage updates alternates with row and plane: with PF = (xw + first stage-1 row + first step +
  // first colour) & 1 -- the same in every workgroup: windows advance by 124 columns, row groups by 2 NW - 2 rows and chunks by an
  // even number of planes (launcher, which picks the instantiation) -- the first point of row r is updated in stage 1 of step j,
  // and in stage 2 (one plane behind, the other colour), iff ((r + j) & 1) == PF: a compile-time fact in the unrolled loop.
  // (With an odd number of rows per wave the parity of a wave's first row alternates from wave to wave: PFW = PF ^ (wave & 1), one copy
  // of the loop per parity, each wave runs its own.)
  auto step = [&](const int q, auto PHc, auto PFWc) {
    constexpr int PH = decltype(PHc)::value, PFW = decltype(PFWc)::value;
    d2 (&Um)[RPW] = U[PH & 3], (&Uc)[RPW] = U[(PH + 1) & 3], (&Up)[RPW] = U[(PH + 2) & 3], (&Upf)[RPW] = U[(PH + 3) & 3];
    d2 (&Vm)[RPW] = V[PH & 3], (&Vc)[RPW] = V[(PH + 1) & 3], (&Vn)[RPW] = V[(PH + 2) & 3];
    d2 (&Fm)[RPW] = F[PH & 3], (&Fq)[RPW] = F[(PH + 1) & 3], (&Fnew)[RPW] = F[(PH + 3) & 3];
    d2 &Oc = O[(PH + 2) & 3], &Onew = O[PH & 3];
    const int m = q - 1;
    constexpr int ub = PH & 1, vb = (PH + 1) & 1;
    if constexpr (PROL) {   // input plane q+1 enters the pipeline here: add its correction first (mb is even: q + 1 is odd in steps 1, 3)
      const int P = (q + 1) >> 1;
      if constexpr (PH & 1) corr_plane(Up, Oc, q + 1, CBp + ((P + 1) & 1) * CT, CBp + (P & 1) * CT, T_{});
      else corr_plane(Up, Oc, q + 1, CBp + (P & 1) * CT, CBp + (P & 1) * CT, F_{});
    }
    // y-neighbour rows of this step from LDS
    const d2 ulo = UB(ub, s0);          // input row below the first own row
    const d2 uhi = UB(ub, s0 + RPW + 1);      // input row above the last own row
    // stage-1 rows below / above the own ones, read without conditions: the first wave's lower and the last wave's upper row do not
    // exist (clamped to some row of the buffer) and are neighbours of halo rows only, which stage 2 does not evaluate; before the
    // first output plane the buffer holds nothing yet and stage 2 does not run
    const d2 vlo = VB(vb, s0 >= 1 ? s0 - 1 : 0);
    const d2 vhi = VB(vb, s0 + RPW <= NS - 1 ? s0 + RPW : NS - 1);
    // prefetch: input plane q+3 of the own rows is not needed yet; plane q+2 is in flight (Upf), rhs q+1 in flight (Fqn)
    // ---- stage 1 on plane q, own rows ----
    const bool pin = q >= box1.b2 && q < box1.e2;
    auto stage1 = [&](auto Rc) {
      constexpr int r = decltype(Rc)::value;
      const d2 c = Uc[r];
      const bool on = pin && row_in1[r];     // wave-uniform; off: the plane / row passes through (evaluated all the same: no branch,
      const d2 ym = r == 0 ? ulo : Uc[r == 0 ? 0 : r - 1];    // no copies at a join -- the result is dropped by the select that the x range needs anyway)
      const d2 yp = r == RPW - 1 ? uhi : Uc[r == RPW - 1 ? r : r + 1];
      const d2 f = Fq[r];
      if (COL) {
        if constexpr (((r + PH) & 1) == PFW) {
          const double xl = lane_below0(c.y);
          const double acc = conv7<ORDER>(k, c.x, xl, c.y, ym.x, yp.x, Um[r].x, Up[r].x);
          const double nv = c.x + w * (f.x - acc);
          Vn[r].x = (in1_a && on) ? nv : c.x;
          Vn[r].y = c.y;
        } else {
          const double xr = lane_above0(c.x);
          const double acc = conv7<ORDER>(k, c.y, c.x, xr, ym.y, yp.y, Um[r].y, Up[r].y);
          const double nv = c.y + w * (f.y - acc);
          Vn[r].x = c.x;
          Vn[r].y = (in1_b && on) ? nv : c.y;
        }
      } else {
        const double xl = lane_below0(c.y), xr = lane_above0(c.x);
        const double acc_a = conv7<ORDER>(k, c.x, xl, c.y, ym.x, yp.x, Um[r].x, Up[r].x);
        const double acc_b = conv7<ORDER>(k, c.y, c.x, xr, ym.y, yp.y, Um[r].y, Up[r].y);
        const double na = c.x + w * (f.x - acc_a);
        const double nb = c.y + w * (f.y - acc_b);
        Vn[r].x = (in1_a && on) ? na : c.x;
        Vn[r].y = (in1_b && on) ? nb : c.y;
      }
    };
    stage1(I0_{});
    stage1(I1_{});
    if constexpr (RPW >= 3) stage1(I2_{});
    // ---- stage 2 on plane m = q-1, own rows that are output rows ----
    auto stage2 = [&](auto Rc) {
      constexpr int r = decltype(Rc)::value;
      if (m >= mlo && m < me && row_out[r]) {   // wave-uniform
        const d2 c = Vc[r];
        const d2 ym = r == 0 ? vlo : Vc[r == 0 ? 0 : r - 1];
        const d2 yp = r == RPW - 1 ? vhi : Vc[r == RPW - 1 ? r : r + 1];
        d2 o = c;
        if (COL) {
          if constexpr (((r + PH) & 1) == PFW) {
            const double xl = lane_below0(c.y);
            const double acc = conv7<ORDER>(k, c.x, xl, c.y, ym.x, yp.x, Vm[r].x, Vn[r].x);
            o.x = c.x + w * (Fm[r].x - acc);      // no select: a lane outside the box does not store
          } else {
            const double xr = lane_above0(c.x);
            const double acc = conv7<ORDER>(k, c.y, c.x, xr, ym.y, yp.y, Vm[r].y, Vn[r].y);
            o.y = c.y + w * (Fm[r].y - acc);
          }
        } else {
          const double xl = lane_below0(c.y), xr = lane_above0(c.x);
          const double acc_a = conv7<ORDER>(k, c.x, xl, c.y, ym.x, yp.x, Vm[r].x, Vn[r].x);
          const double acc_b = conv7<ORDER>(k, c.y, c.x, xr, ym.y, yp.y, Vm[r].y, Vn[r].y);
          o.x = c.x + w * (Fm[r].x - acc_a);
          o.y = c.y + w * (Fm[r].y - acc_b);
        }
        // full pairs leave under ONE exec region (two adjacent 8-byte non-temporal stores back to back): splitting them
        // into two separately predicated stores costs 14 % of the kernel
        double *qp = reinterpret_cast<double *>(obase[r] + obytes + vo);
        if (st_a && st_b) {
          if (NT) {
            store2_nt(qp, o);
          } else {
            d2u sv;
            sv.a = o.x;
            sv.b = o.y;
            *reinterpret_cast<d2u *>(qp) = sv;
          }
        } else if (st_a) {
          qp[0] = o.x;
        } else if (st_b) {
          qp[1] = o.y;
        }
      }
    };
    stage2(I0_{});
    stage2(I1_{});
    if constexpr (RPW >= 3) stage2(I2_{});
    // ---- publish input plane q+1 and stage-1 plane q; issue the next loads into the slots this step has finished with ----
    if (q < me) {
      constexpr int nb = (PH + 1) & 1;
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        UB(nb, s0 + 1 + r) = Up[r];
        VB(ub, s0 + r) = Vn[r];
      }
      if (has_outer) UB(nb, outer_i) = Oc;
      if constexpr (PROL && !(PH & 1)) {
        // even input plane q+1: coarse plane (q+3)/2 is first needed by input plane q+2 (corrected after this step's
        // barrier); its buffer held plane (q-1)/2, last read for input plane q at the start of step q-1
        const int Pn = (q + 3) >> 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          if (cel(kk) < CT) {
            CBp[(Pn & 1) * CT + cel(kk)] = Cpf[kk];
            Cpf[kk] = cload(kk, Pn + 1);
          }
      }
    }
    // loads without a condition (past the chunk they fetch planes nobody uses; offsets are clamped into the array)
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      Um[r] = load_u(urow[r]);      // the slot of plane q-1 is free: it becomes plane q+3
      Fnew[r] = load_f(r);          // rhs plane q+2
    }
    if (has_outer) Onew = load_u(uouter);
    advance(cu);
    advance(cf);
    obytes += ostep;
    (void)Upf;
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);    // nothing of the next step is scheduled into this one (register pressure: 4 waves per SIMD)
  };
  auto march = [&](auto PFWc) {
    // whole groups of four steps: up to three steps past the last one (q = me) run along -- their loads are clamped into the arrays, they
    // publish nothing and stage 2 stores nothing past plane me - 1.  With `break`s between the steps the exits shared a block with the
    // loop's back edge, and the compiler, joining the load states of all four steps at the loop header, waited for EVERY load in flight
    // at the top of each group: the prefetch distance of two steps was lost in every fourth step.
    // (tools/ab_libs.py, both builds in one process, 512^3: Jacobi pair 0.6238 -> 0.6226 ms, red-black sweep 0.6275 -> 0.6241, 256^3 pair
    // 0.0929 -> 0.0912; the form with the correction folded in lost -- 0.700 -> 0.717 -- and keeps its exits.)
    if constexpr (PROL) {
      for (int q = mb - 1; q <= me; q += 4) {
        step(q, I0_{}, PFWc);
        if (q + 1 > me) break;
        step(q + 1, I1_{}, PFWc);
        if (q + 2 > me) break;
        step(q + 2, I2_{}, PFWc);
        if (q + 3 > me) break;
        step(q + 3, I3_{}, PFWc);
      }
    } else {
      for (int q = mb - 1; q <= me; q += 4) {
        step(q, I0_{}, PFWc);
        step(q + 1, I1_{}, PFWc);
        step(q + 2, I2_{}, PFWc);
        step(q + 3, I3_{}, PFWc);
      }
    }
  };
  if constexpr (COL && (RPW & 1)) {
    if (wv & 1) march(std::integral_constant<int, PF ^ 1>{});
    else march(std::integral_constant<int, PF>{});
  } else {
    march(std::integral_constant<int, PF>{});
  }
#undef UB
#undef VB
}

// kernels_sf27pair.hip: two Jacobi steps on a 27-entry stencil field (records) in one pass; 1 = launched, 0 = not applicable, -1 = error
int sf27_jacobi2_try(const examg_layout_t *lu, const double *u_in, double *u_out, const examg_layout_t *lf, const double *rhs,
                     const examg_stencil_t *st, double w, const Box &box1, const Box &box2, hipStream_t s);

// launch knobs (debug build: examg_debug_two_stage*; per host thread): workgroup count target, tile order
static thread_local int g_ts_blocks = -1;      // workgroup count target; -1: by size and variant (launch_two_stage_lds)
static thread_local int g_ts_disable = 0;
// tile order: 0 = plain (x fastest, then y, then z chunk); 1 = one contiguous run of tiles per XCD; 2 = XCD-contiguous within every
// z layer of tiles: workgroups are dealt round-robin to the 8 XCDs, so within a layer each XCD gets a band of y-adjacent tiles
// (halo rows shared in its L2) while all XCDs stay in the same few planes (the z-halo planes a layer re-reads are still in the
// Infinity Cache, DRAM sees one moving front); -1 = default (2).  tools/sweep_two_stage2.py on MI355X, red-black sweep, ms,
// plain / per-XCD / layered at ~3072 workgroups: 256^3 0.122 / 0.095 / 0.095; 384^3 0.392 / 0.305 / 0.291; 448^3 0.495 / 0.481 / 0.417;
// 512^3 0.714 / 0.737 / 0.698.  Plain (cached) instead of non-temporal stores with the layered order: 0.703 against 0.670 ms.  With the layered order more, shorter z chunks pay at 512^3 and above (16 planes per chunk):
// 3072 / 6144 / 8192 workgroups 0.697 / 0.668 / 0.667 (plain order: 0.733 / 0.780 / -); 256^3 prefers ~3072 (0.098 vs 0.106).
static thread_local int g_ts_remap = -1;
static thread_local int g_ts_minzc = -1;     // minimum planes per z chunk; -1: 16, 8 on small boxes
// Which implementation: 5 / 8 = that many waves per workgroup; -2 = by size.
// round-1 tuning runs on MI355X, ms for a Jacobi pair / a red-black sweep (all variants bit-identical):
//   512^3: registers 0.97 / 0.93, LDS-5 0.79 / 0.77, LDS-8 0.717 / 0.716 (3072 workgroups), LDS-9 0.87 / 0.85
//   256^3: LDS-5 0.138, LDS-8 0.119-0.124;  128^3: LDS-5 0.028, LDS-8 0.031
// ~120 VGPRs -> 4 waves per SIMD = 16 per CU: two 8-wave workgroups fill a CU (a 9-wave workgroup runs alone)
static thread_local int g_ts_lds = -2;
static thread_local int g_ts_nt = -1;          // stores of the plain passes: -1 by size, 1 non-temporal, 0 plain (examg_debug_two_stage_nt)
static thread_local int g_ts_wpe = 1;          // plain passes: 4 = capped at 128 VGPRs (examg_debug_two_stage_prol(wpe + 10) sets it)
static thread_local int g_ts_prol_wpe = 1;     // PROL variants: 1 = uncapped (151 / 176 VGPRs: 512^3 0.742 ms), 4 = capped at 128 VGPRs (spills in the unrolled loop: 1.15 ms)

template <bool COL, int NW, int WPE = 1, int RPW = 2>
static int launch_two_stage_lds(const examg_layout_t *lu_, const double *u, const examg_layout_t *lf_, const double *rhs,
                                double *out, const examg_stencil_t *st, double w, int first, const Box &box, const Box &box1,
                                hipStream_t s, const TSProl *prol = nullptr, bool zero_in = false) {
  constexpr int NO = RPW * NW - 2;
  const LayoutDev lu = make_layout(lu_), lf = make_layout(lf_);
  TSGeom g;
  // padded layouts (even strides): start the windows so that every 16-byte access is aligned (kernels_stencil.hip: launch_zmarch)
  const bool even = !(lu.s1 & 1) && !(lu.s2 & 1) && !(lf.s1 & 1) && !(lf.s2 & 1);
  g.xs = (even && ((lu.origin + box.b0) & 1) && ((lf.origin + box.b0) & 1)) ? 1 : 0;
  g.ys = g.zs = 0;
  if (prol) {   // canonical parities (see the kernel): window start even, first output row odd, first output plane even
    g.xs = box.b0 & 1;
    g.ys = (box.b1 & 1) ? 0 : 1;
    g.zs = box.b2 & 1;
  }
  g.ntx = (box.n0() + g.xs + TS_OUT - 1) / TS_OUT;
  g.nty = (box.n1() + g.ys + NO - 1) / NO;
  const int xy = g.ntx * g.nty;
  // Workgroup count = chunk length in z.  From 5*10^7 points the plain passes take the shortest chunks (16 planes: 512^3 0.667 ms; a
  // fixed count of 8192 workgroups made the chunks of larger blocks long again -- 768^3 2.37 -> 2.20 ms, 1024^3 5.60 -> 5.13 ms with
  // 16-plane chunks, tools/sweep_two_stage_big.py).  With the correction folded in a step costs more and longer chunks pay: about 48
  // planes (512^3: 16 / 32 / 48 planes 0.954 / 0.907 / 0.882 ms; 768^3 2.99 / 2.90 / 2.90; 1024^3 6.86 / 6.48 / 6.33), but at least
  // ~1536 workgroups (384^3: 21 planes 0.427 against 16 planes 0.451).
  const int n2 = box.n2() + g.zs;
  int blocks_target = 3072;
  if (prol) {
    blocks_target = xy * ((n2 + 47) / 48);
    if (blocks_target < 1536) blocks_target = 1536;
  } else if (box.count() >= 50000000LL) {
    blocks_target = 1 << 24;
  }
  // In between (some 10^6 .. 5*10^7 points: the 192^3 .. 320^3 levels of a cycle) what counts is how evenly the workgroups fill the
  // 256 CUs: the chunk count that minimises  rounds of 256 workgroups x (planes per chunk + 6 planes of halo and start-up).
  // MI355X, plain sweep / Jacobi pair / correction + sweep, ms, 16-plane chunks -> this rule: 192^3 0.042 / 0.050 / 0.067 ->
  // 0.033 / 0.036 / 0.049 (5 chunks of 40 planes: 240 workgroups); 256^3 0.093 / 0.100 / 0.120 -> 0.089 / 0.092 / 0.108 (4 x 64: 228);
  // 320^3 0.173 / 0.172 / 0.198 -> 0.172 / 0.159 / 0.177 (7 x 48: 483).
  const bool mid = box.count() < 50000000LL && (long long)xy * ((n2 + 15) / 16) >= 512;
  if (mid) {
    long long best = -1;
    int best_t = 1;
    for (int t = 1; t <= (n2 + 7) / 8; ++t) {
      int c = (n2 + t - 1) / t;
      if (prol || COL) c += c & 1;
      const int tt = (n2 + c - 1) / c;
      const long long cost = (((long long)xy * tt + 255) / 256) * (c + 6);
      if (best < 0 || cost < best || (cost == best && tt > best_t)) { best = cost; best_t = tt; }
    }
    blocks_target = xy * best_t;
  }
  if (g_ts_blocks > 0) blocks_target = g_ts_blocks;
  int ntz = (blocks_target + xy - 1) / xy;
  if (ntz < 1) ntz = 1;
  int zc = (n2 + ntz - 1) / ntz;
  // at least 16 planes per chunk (4 halo planes each); 8 on small boxes that would leave most of the chip idle otherwise
  // (tools/sweep_two_stage3.py, 5-wave workgroups, 16 / 8 / 4 planes: 128^3 0.0243 / 0.0186 / 0.0300 ms, 96^3 0.0235 / 0.0149 / 0.0129)
  int minzc = g_ts_minzc;
  // (14 and 10 since the plane loop runs groups of four steps, zc + 2 per chunk: 16 and 8 before.  tools/ab_chunks4.py, 512^3, 14 / 16 / 18 / 22
  // planes: Jacobi pair 0.622 / 0.644 / 0.625 / 0.644 ms, red-black sweep 0.624 / 0.647 / 0.638 / 0.647)
  if (minzc < 0) minzc = ((long long)xy * ((n2 + 15) / 16) < 512 || (mid && g_ts_blocks <= 0)) ? (prol ? 8 : 10) : (prol ? 16 : 14);
  if (zc < minzc) zc = minzc;
  // the plane loop runs whole groups of four steps (a chunk of zc planes takes zc + 2): chunk lengths of 4 k + 2 planes waste none --
  // 18 instead of 16 at 512^3.  Even in any case: every workgroup starts on the same plane parity (PROL: canonical parities; COL: one PF)
  if (g_ts_minzc < 0 && g_ts_blocks <= 0 && !prol) zc += (6 - (zc & 3)) & 3;
  if (prol || COL) zc += zc & 1;
  // the kernel forms the offset of a row within a plane as a 32-bit product
  if (lu.s2 >= (1LL << 32) || lf.s2 >= (1LL << 32)) { set_error("examg two-stage kernel: a plane must hold less than 2^32 elements"); return 1; }
  if (zc > n2) zc = n2 + ((prol || COL) ? (n2 & 1) : 0);
  g.zc = zc;
  g.ntz = (n2 + zc - 1) / zc;
  g.nblocks = xy * g.ntz;
  g.remap = g_ts_remap >= 0 ? g_ts_remap : 2;
  g.first = first;
  g.box1 = box1;
  g.ax0 = -lu.ref0; g.ax1 = lu.tot0 - lu.ref0;
  g.ay0 = -lu.ref1; g.ay1 = lu.tot1 - lu.ref1;
  g.az0 = -lu.ref2; g.az1 = lu.tot2 - lu.ref2;
  Coef7 k;
  for (int i = 0; i < 7; ++i) k.c[i] = st->coef[i];
  const int ord = canonical_order7(st);
  dim3 block(64, NW, 1), grid(g.nblocks, 1, 1);
  TSProl pr;
  pr.lc = lu;
  pr.uc = nullptr;
  if (prol) pr = *prol;
  if (zero_in && !(COL && NW != 6)) {
    set_error("examg two-stage kernel: no zero-input variant of this form");
    return 1;
  }
  if (RPW != 2 && (prol || zero_in)) { set_error("examg two-stage kernel: the folded forms have two rows per wave"); return 1; }
  // COL: the kernel instantiation by the parity that fixes which point of a pair each unrolled step updates (see the kernel)
  const int pf = COL ? ((box.b0 - 2 - g.xs) + (box.b1 - g.ys - 1) + (box.b2 - g.zs - 1) + first) & 1 : 0;
  // Stores: non-temporal where the pass streams (its output would only push the inputs of the next pass out of the caches), plain where
  // the arrays of a level -- input, output, right-hand side -- fit the 256 MB Infinity Cache together and the next pass of the level reads
  // what this one wrote (tools/ab_nt.py, ping-pong passes, non-temporal / plain, ms: 128^3 0.0158 / 0.0142, 192^3 0.0433 / 0.0339,
  // 256^3 0.0888 / 0.1086, 384^3 0.305 / 0.333, 512^3 0.652 / 0.708).  Plain passes only.
  const bool nt = g_ts_nt >= 0 ? g_ts_nt != 0 : box.count() * 24LL > 200000000LL;
#define EXAMG_TS_LAUNCH(ORD, W, V, PFV)                                                                                                               \
  do {                                                                                                                                                \
    if ((V) == 0 && (W) == 1 && !nt)                                                                                                                  \
      hipLaunchKernelGGL((k_two_stage7_lds<ORD, COL, NW, false, 1, 0, PFV, RPW>), grid, block, 0, s, lu, u, lf, rhs, out, k, w, box, g, pr);           \
    else                                                                                                                                              \
      hipLaunchKernelGGL((k_two_stage7_lds<ORD, COL, NW, true, W, V, PFV, ((V) == 0 ? RPW : 2)>), grid, block, 0, s, lu, u, lf, rhs, out, k, w, box, g, pr); \
  } while (0)
#define EXAMG_TS_LAUNCH_PF(ORD, W, V)              \
  do {                                             \
    if constexpr (COL) {                           \
      if (pf) EXAMG_TS_LAUNCH(ORD, W, V, 1);       \
      else EXAMG_TS_LAUNCH(ORD, W, V, 0);          \
    } else {                                       \
      EXAMG_TS_LAUNCH(ORD, W, V, 0);               \
    }                                              \
  } while (0)
#define EXAMG_TS_LAUNCH_ORD(W, V)                  \
  do {                                             \
    if (ord == 0) EXAMG_TS_LAUNCH_PF(0, W, V);     \
    else EXAMG_TS_LAUNCH_PF(1, W, V);              \
  } while (0)
  if (prol) {
    if (g_ts_prol_wpe == 4) EXAMG_TS_LAUNCH_ORD((NW == 6 ? 1 : 4), 1);
    else EXAMG_TS_LAUNCH_ORD(1, 1);
  } else if (zero_in) {
    if constexpr (COL && NW != 6) EXAMG_TS_LAUNCH_ORD(WPE, 2);
  } else {
    bool done = false;
    if constexpr (NW != 6 && RPW == 2) {     // the register cap exists for the two-row form only
      if (g_ts_wpe == 4) {
        EXAMG_TS_LAUNCH_ORD(4, 0);
        done = true;
      }
    }
    if (!done) EXAMG_TS_LAUNCH_ORD(1, 0);
  }
#undef EXAMG_TS_LAUNCH_ORD
#undef EXAMG_TS_LAUNCH_PF
#undef EXAMG_TS_LAUNCH
  EXAMG_CHECK_LAUNCH("k_two_stage7_lds");
  return 0;
}

template <bool COL>
static int launch_two_stage(const examg_layout_t *lu_, const double *u, const examg_layout_t *lf_, const double *rhs,
                            double *out, const examg_stencil_t *st, double w, int first, const Box &box, hipStream_t s,
                            const Box *box1 = nullptr, const TSProl *prol = nullptr, bool zero_in = false) {
  const Box &b1 = box1 ? *box1 : box;
  int impl = g_ts_lds == -2 ? (box.n1() >= 192 ? 8 : 5) : g_ts_lds;
  if (zero_in && impl == 6) impl = 5;
  // plain passes on the largest levels: three rows per wave -- 22 of 26 rows of a workgroup are outputs instead of 14 of 18 (100 KB of
  // LDS, 164-189 VGPRs: one workgroup per CU, which the two-row Jacobi pair has anyway).  512^3, same process: Jacobi pair 0.667 ->
  // 0.650 ms, red-black sweep 0.672 -> 0.657 (another box: 0.636 -> 0.628, 0.656 -> 0.628)
  if (g_ts_lds == -2 && impl == 8 && !prol && !zero_in && box.count() >= 50000000LL) impl = 83;
  if (impl == 83 && !prol && !zero_in)      // three rows per wave: plain passes only
    return launch_two_stage_lds<COL, 8, 1, 3>(lu_, u, lf_, rhs, out, st, w, first, box, b1, s, prol, zero_in);
  if (impl == 83) impl = 8;
  if (impl == 8) return launch_two_stage_lds<COL, 8>(lu_, u, lf_, rhs, out, st, w, first, box, b1, s, prol, zero_in);
  if (impl == 6) return launch_two_stage_lds<COL, 6>(lu_, u, lf_, rhs, out, st, w, first, box, b1, s, prol, zero_in);
  return launch_two_stage_lds<COL, 5>(lu_, u, lf_, rhs, out, st, w, first, box, b1, s, prol, zero_in);
}

// The coarse footprint of the correction loop on `box` (examg_prolong_add's checks)
static bool prolong_args_ok(const char *who, const examg_layout_t *lc, const Box &box) {
  if (box.b0 < 0 || box.b1 < 0 || box.b2 < 0) { set_error("%s: negative fine index", who); return false; }
  int32_t cb[3] = {box.b0 / 2, box.b1 / 2, box.b2 / 2}, ce[3] = {box.e0 / 2 + 1, box.e1 / 2 + 1, box.e2 / 2 + 1};
  if (lc->nd != 3 || !box_inside(lc, make_box(cb, ce), 0)) { set_error("%s: coarse footprint leaves the coarse allocation", who); return false; }
  return true;
}

static bool two_stage_ok(const examg_layout_t *lu, const examg_layout_t *lf, const examg_stencil_t *st, const Box &box) {
  return !g_ts_disable && !lay_split(lu) && !lay_split(lf) && lu->nd == 3 && canonical_order7(st) >= 0 && box.n0() >= 64 && box_inside(lu, box, 1) &&
         box_inside(lf, box, 0);
}

// Separate stage boxes: the 16-byte loads of the kernel read the first (last) element of the rhs array as the second (first) half of
// a pair whose other half lies before (past) the array, which it cannot do on the first (last) row of the first (last) plane.  That
// element is needed only if stage 1 reaches one column further than stage 2 there while stage 2 starts (ends) on the first (last)
// row and plane of the rhs allocation: a corner no block decomposition produces (box 2 leaves out the duplicate planes of ALL
// interior faces) -- such a call takes the fallback.
static bool stage_boxes_ok(const examg_layout_t *lf_, const Box &box1, const Box &box2) {
  const LayoutDev lf = make_layout(lf_);
  const bool lo = box1.b0 < box2.b0 && box2.b1 + lf.ref1 == 0 && box2.b2 + lf.ref2 == 0;
  const bool hi = box1.e0 > box2.e0 && box2.e1 + lf.ref1 == lf.tot1 && box2.e2 + lf.ref2 == lf.tot2;
  return !lo && !hi;
}

// ---------------------------------------------------------------------------------------------------------------
// THREE dependent Jacobi steps in one pass (temporal blocking of depth 3; `repeat 5 times with contraction` of
// Testing/PolyExpl/Jac3Dcc.exa4:27 runs as 3 + 2; baseExt/ir/IR_ContractingLoop.scala:45-196): 24 B per point for three updates.
// The structure of k_two_stage7_lds with one more stage behind it: a workgroup of NW waves owns a 128-point x window and
// NS = 3 NW stage-1 rows, every wave three consecutive rows; in step q a wave evaluates
//   stage 1 on plane q   (V(q)   from the input planes q-1, q, q+1),
//   stage 2 on plane q-1 (W(q-1) from V(q-2), V(q-1), V(q)),
//   stage 3 on plane q-2 (output from W(q-3), W(q-2), W(q-1)),
// all three pipelines in registers (rings of four slots, the plane loop unrolled four times by hand).  Halo: three points on
// every side -- 120 of 128 columns, NS - 4 of NS + 2 rows, zc of zc + 6 planes are outputs.  Only a wave's FIRST and LAST row are
// anybody's y neighbours: those two rows of every field pass through LDS (96 KB for eight waves), the outer input rows of the first
// and the last wave stay in the registers of the wave that loads them.  One barrier per plane.
// Points outside the box pass through stages 1 and 2 with their value (boundary and ghost values), stage 3 stores inside the box only.
// Every value is the expression of the one-step kernel in its order: bit-identical to three Jacobi loops one after the other.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TS3_OUT = 120;   // outputs per 128-point window: lanes 2 .. 61

// COL: three colour loops of a red-black smoother instead -- stage 1 updates the points of colour g.first, stage 2 the other colour, stage 3
// colour g.first again (three sweeps = six colour loops = two such passes, the second one starting with the other colour).  Which point of a
// pair a stage updates is the same for all three stages of a step (plane and colour both move by one from stage to stage) and wave-uniform
// per row: one convolution per pair and stage, the other point passes through.
template <int ORDER, int NW, bool NT, int RPW = 3, bool COL = false>
__global__ void __launch_bounds__(64 * NW, 1)
k_three_stage7_lds(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs,
                   double *__restrict__ out, Coef7 k, double w, Box box, TSGeom g) {
  constexpr int NS = RPW * NW;      // stage-1 rows s = 0 .. NS-1, global row rw0 - 2 + s
  constexpr int NO = NS - 4;        // output rows: s = 2 .. NS-3
  // edge rows of the waves, two plane buffers each: [wave][field][buffer][lo / hi][lane] -- a wave's twelve rows lie within 12 KB, so one
  // address register per wave addressed (own, below, above) and immediate offsets reach all of them
  __shared__ d2 SM[3 * 2 * 2 * NW * 64];
#define EB(f, p, h, wq) EBW(wave_rows(wq), f, p, h)
#define EBW(rows, f, p, h) (rows)[((((f) * 2 + (p)) * 2 + (h)) * 64)]
  // a wave's block of rows as scalar offset + this lane's 16 bytes, formed where it is used: kept loop-invariant in registers, the three
  // addresses (own block, the waves below and above) cost the registers that the plane loop does not have
  auto wave_rows = [&](int wq) {
    unsigned off = (unsigned)wq * (3 * 2 * 2 * 64 * 16);
    asm volatile("" : "+s"(off));
    return reinterpret_cast<d2 *>(reinterpret_cast<char *>(SM) + off + threadIdx.x * 16u);
  };
  const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
  int t = blockIdx.x;
  if (g.remap == 2) {  // XCD-contiguous within every z layer of tiles (see k_two_stage7_lds)
    const int xy = g.ntx * g.nty;
    const int lz = t / xy, r = t - lz * xy;
    const int per = xy >> 3;
    t = lz * xy + (r < (per << 3) ? (r & 7) * per + (r >> 3) : r);
  }
  const int tx = t % g.ntx;
  t /= g.ntx;
  const int ty = t % g.nty;
  const int tz = t / g.nty;
  const int xw = box.b0 - 4 + TS3_OUT * tx;     // first point of the window
  const int xa = xw + 2 * lane;
  const int rw0 = box.b1 + ty * NO;             // first output row of the workgroup
  const int mb = box.b2 + tz * g.zc;            // first output plane
  const int me = min(mb + g.zc, box.e2);
  const bool inx_a = xa >= box.b0 && xa < box.e0, inx_b = xa + 1 >= box.b0 && xa + 1 < box.e0;
  const bool out_lane = lane >= 2 && lane <= 61;
  const bool st_a = out_lane && inx_a, st_b = out_lane && inx_b;
  const int s0 = RPW * wv;
  int grow[RPW];
  bool row_in[RPW], row_out[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    grow[r] = rw0 - 2 + s0 + r;
    row_in[r] = grow[r] >= box.b1 && grow[r] < box.e1;
    row_out[r] = s0 + r >= 2 && s0 + r <= NS - 3 && row_in[r];
  }
  const bool has_outer = wv == 0 || wv == NW - 1;      // input rows rw0 - 3 (below the first wave) and rw0 - 2 + NS (above the last)
  const int outer_row = wv == 0 ? rw0 - 3 : rw0 - 2 + NS;
  const int wlo = wv >= 1 ? wv - 1 : 0, whi = wv + 1 <= NW - 1 ? wv + 1 : NW - 1;

  // addressing as in k_two_stage7_lds: scalar base per row, one clamped byte offset per lane, running scalar plane offsets
  struct Site {
    const char *row;
    unsigned voff;
  };
  auto site = [&](const double *base, const LayoutDev &l, int R) {
    const int rc = min(max(R + l.ref1, 0), l.tot1 - 1), i0 = xw + l.ref0;
    Site st;
    st.row = reinterpret_cast<const char *>(base + ((long long)i0 + (long long)((unsigned)l.s1 * (unsigned)rc)));
    st.voff = (unsigned)(min(max(2 * lane, -i0 - (rc == 0 ? 0 : 1)), l.tot0 - 1 - i0 - (rc == l.tot1 - 1 ? 1 : 0)) * 8);
    return st;
  };
  struct PlaneCursor {
    int p;
    long long bytes, step;
    int last;
  };
  auto cursor = [&](const LayoutDev &l, int P) {
    PlaneCursor c;
    c.p = P + l.ref2;
    c.last = l.tot2 - 1;
    c.step = uniform64(l.s2 * 8);
    c.bytes = uniform64(l.s2 * 8 * min(max(c.p, 0), c.last));
    return c;
  };
  auto advance = [&](PlaneCursor &c) {
    ++c.p;
    c.bytes += (c.p >= 1 && c.p <= c.last) ? c.step : 0LL;
  };
  auto load_at = [&](const Site &st, long long pb) { return load2(reinterpret_cast<const double *>(st.row + pb + st.voff)); };
  Site urow[RPW], frow[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    urow[r] = site(u, lu, grow[r]);
    frow[r] = site(rhs, lf, grow[r]);
  }
  const Site uouter = site(u, lu, outer_row);
  const int q0 = mb - 2;                                           // first step
  PlaneCursor cu = cursor(lu, q0 - 1), cf = cursor(lf, q0);        // the next plane to load
  const unsigned vo = (unsigned)lane * 16u;
  char *obase[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
    obase[r] = reinterpret_cast<char *>(out + ((long long)(xw + lu.ref0) + (long long)((unsigned)lu.s1 * (unsigned)max(grow[r] + lu.ref1, 0))));
  long long obytes = uniform64(lu.s2 * 8 * (long long)(q0 - 2 + lu.ref2));   // plane q-2 of the first step (not used before the first output plane)
  const long long ostep = uniform64(lu.s2 * 8);

  // rings: in step q = q0 + 4 n + PH plane p of every field lives in slot (p - q + PH + 1) & 3
  d2 U[4][RPW], V[4][RPW], W[4][RPW], F[4][RPW], O[4];
  O[0] = O[1] = O[2] = O[3] = d2{0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 4; ++j) {      // input planes q0-1 .. q0+2
#pragma unroll
    for (int r = 0; r < RPW; ++r) U[j][r] = load_at(urow[r], cu.bytes);
    if (has_outer && j >= 1) O[j] = load_at(uouter, cu.bytes);
    advance(cu);
  }
#pragma unroll
  for (int j = 1; j < 3; ++j) {      // rhs planes q0, q0+1
#pragma unroll
    for (int r = 0; r < RPW; ++r) F[j][r] = load_at(frow[r], cf.bytes);
    advance(cf);
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    F[0][r] = F[1][r];
    F[3][r] = F[1][r];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      V[j][r] = U[0][r];
      W[j][r] = U[0][r];
    }
  }
  // publish the edge rows of input plane q0
  EB(0, 0, 0, wv) = U[1][0];
  EB(0, 0, 1, wv) = U[1][RPW - 1];
  // every load of the prologue has arrived before the plane loop starts: at the loop header the compiler otherwise joins "the prologue's
  // loads may still be out" with the steady state and waits for ALL loads in flight at the top of every fourth step
  __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
  __syncthreads();

  using I0_ = std::integral_constant<int, 0>;
  using I1_ = std::integral_constant<int, 1>;
  using I2_ = std::integral_constant<int, 2>;
  using I3_ = std::integral_constant<int, 3>;
  // one Jacobi update of a pair: centre c, y neighbours ym / yp, z neighbours zm / zp, right-hand side f
  auto jac = [&](d2 c, d2 ym, d2 yp, d2 zm, d2 zp, d2 f) {
    const double xl = lane_below0(c.y), xr = lane_above0(c.x);
    const double acc_a = conv7<ORDER>(k, c.x, xl, c.y, ym.x, yp.x, zm.x, zp.x);
    const double acc_b = conv7<ORDER>(k, c.y, c.x, xr, ym.y, yp.y, zm.y, zp.y);
    d2 n;
    n.x = c.x + w * (f.x - acc_a);
    n.y = c.y + w * (f.y - acc_b);
    return n;
  };
  auto jac_x = [&](d2 c, d2 ym, d2 yp, d2 zm, d2 zp, d2 f) {      // the first point of the pair only
    const double xl = lane_below0(c.y);
    const double acc = conv7<ORDER>(k, c.x, xl, c.y, ym.x, yp.x, zm.x, zp.x);
    return c.x + w * (f.x - acc);
  };
  auto jac_y = [&](d2 c, d2 ym, d2 yp, d2 zm, d2 zp, d2 f) {      // the second point only
    const double xr = lane_above0(c.x);
    const double acc = conv7<ORDER>(k, c.y, c.x, xr, ym.y, yp.y, zm.y, zp.y);
    return c.y + w * (f.y - acc);
  };
  // COL: the update of a pair in the step of plane q, row r -- the first point if its colour (x + y + z) & 1 is the colour of the stage
  auto upd = [&](bool first_point, d2 c, d2 ym, d2 yp, d2 zm, d2 zp, d2 f) {
    if constexpr (!COL) return jac(c, ym, yp, zm, zp, f);
    d2 n = c;
    if (first_point) n.x = jac_x(c, ym, yp, zm, zp, f);      // wave-uniform
    else n.y = jac_y(c, ym, yp, zm, zp, f);
    return n;
  };
  const int cpar = xw + g.first;
  auto step = [&](const int q, auto PHc) {
    constexpr int PH = decltype(PHc)::value;
    d2 (&Um)[RPW] = U[PH & 3], (&Uc)[RPW] = U[(PH + 1) & 3], (&Up)[RPW] = U[(PH + 2) & 3];
    d2 (&Vmm)[RPW] = V[(PH + 3) & 3], (&Vm)[RPW] = V[PH & 3], (&Vn)[RPW] = V[(PH + 1) & 3];          // planes q-2, q-1, q (written here)
    d2 (&Wmm)[RPW] = W[(PH + 2) & 3], (&Wm)[RPW] = W[(PH + 3) & 3], (&Wn)[RPW] = W[PH & 3];          // planes q-3, q-2, q-1 (written here)
    d2 (&Fq)[RPW] = F[(PH + 1) & 3], (&Fm)[RPW] = F[PH & 3], (&Fmm)[RPW] = F[(PH + 3) & 3];          // rhs on planes q, q-1, q-2
    constexpr int ub = PH & 1, vb = (PH + 1) & 1;
    // y-neighbour rows from the waves below and above (input plane q, stage-1 plane q-1, stage-2 plane q-2); the first wave's lower and
    // the last wave's upper input row are their own outer rows; of the stage fields those rows do not exist (some row is read instead:
    // they are neighbours of rows that are nobody's input)
    const d2 *const below = wave_rows(wlo), *const above = wave_rows(whi);
    d2 ulo = EBW(below, 0, ub, 1), uhi = EBW(above, 0, ub, 0);
    const d2 vlo = EBW(below, 1, vb, 1), vhi = EBW(above, 1, vb, 0);
    const d2 zlo = EBW(below, 2, vb, 1), zhi = EBW(above, 2, vb, 0);
    if (wv == 0) ulo = O[(PH + 1) & 3];
    if (wv == NW - 1) uhi = O[(PH + 1) & 3];
    // The stages row by row, the MIDDLE row of every stage first: it needs no other wave's rows, so its arithmetic runs while the LDS
    // reads above are still on their way; the first and the last row, which do, follow.
    const bool pin1 = q >= box.b2 && q < box.e2, pin2 = q - 1 >= box.b2 && q - 1 < box.e2;
    const bool run2 = q >= mb;          // wave-uniform: before that nothing reads what stage 2 would produce
    const bool run3 = q - 2 >= mb && q - 2 < me;      // output planes of the chunk
    bool fp[RPW];       // COL: does this step update the first point of the pairs of row r (stage 1 on plane q with colour g.first, stage 2 on
#pragma unroll          // plane q-1 with the other colour, stage 3 on plane q-2 with g.first again: the same parity)
    for (int r = 0; r < RPW; ++r) fp[r] = ((cpar + grow[r] + q) & 1) == 0;
    auto stage1 = [&](auto Rc) {        // plane q
      constexpr int r = decltype(Rc)::value;
      const d2 c = Uc[r];
      const d2 n = upd(fp[r], c, r == 0 ? ulo : Uc[r == 0 ? 0 : r - 1], r == RPW - 1 ? uhi : Uc[r == RPW - 1 ? r : r + 1], Um[r], Up[r], Fq[r]);
      const bool on = pin1 && row_in[r];
      Vn[r].x = (inx_a && on) ? n.x : c.x;
      Vn[r].y = (inx_b && on) ? n.y : c.y;
    };
    auto stage2 = [&](auto Rc) {        // plane q-1
      constexpr int r = decltype(Rc)::value;
      if (run2) {
        const d2 c = Vm[r];
        const d2 n = upd(fp[r], c, r == 0 ? vlo : Vm[r == 0 ? 0 : r - 1], r == RPW - 1 ? vhi : Vm[r == RPW - 1 ? r : r + 1], Vmm[r], Vn[r], Fm[r]);
        const bool on = pin2 && row_in[r];
        Wn[r].x = (inx_a && on) ? n.x : c.x;
        Wn[r].y = (inx_b && on) ? n.y : c.y;
      }
    };
    auto stage3 = [&](auto Rc) {        // plane q-2: output rows inside the box
      constexpr int r = decltype(Rc)::value;
      if (run3 && row_out[r]) {         // wave-uniform
        const d2 c = Wm[r];
        const d2 o = upd(fp[r], c, r == 0 ? zlo : Wm[r == 0 ? 0 : r - 1], r == RPW - 1 ? zhi : Wm[r == RPW - 1 ? r : r + 1], Wmm[r], Wn[r], Fmm[r]);
        double *qp = reinterpret_cast<double *>(obase[r] + obytes + vo);
        if (st_a && st_b) {
          if (NT) store2_nt(qp, o);
          else store2(qp, o);
        } else if (st_a) {
          qp[0] = o.x;
        } else if (st_b) {
          qp[1] = o.y;
        }
      }
    };
    if constexpr (RPW == 3) {
      stage1(I1_{});
      stage2(I1_{});
      stage3(I1_{});
      stage1(I0_{});
      stage1(I2_{});
      stage2(I0_{});
      stage2(I2_{});
      stage3(I0_{});
      stage3(I2_{});
    } else {
      stage1(I0_{});
      stage1(I1_{});
      stage2(I0_{});
      stage2(I1_{});
      stage3(I0_{});
      stage3(I1_{});
    }
    // ---- publish the edge rows of input plane q+1, stage-1 plane q, stage-2 plane q-1 for the next step ----
    if (q <= me) {
      d2 *const own = wave_rows(wv);
      EBW(own, 0, vb, 0) = Up[0];
      EBW(own, 0, vb, 1) = Up[RPW - 1];
      EBW(own, 1, ub, 0) = Vn[0];
      EBW(own, 1, ub, 1) = Vn[RPW - 1];
      EBW(own, 2, ub, 0) = Wn[0];
      EBW(own, 2, ub, 1) = Wn[RPW - 1];
    }
    // loads without a condition: input plane q+3 into the slot of plane q-1, rhs plane q+2 into the slot of plane q-2
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      Um[r] = load_at(urow[r], cu.bytes);
      Fmm[r] = load_at(frow[r], cf.bytes);
    }
    if (has_outer) O[PH & 3] = load_at(uouter, cu.bytes);
    advance(cu);
    advance(cf);
    obytes += ostep;
    __syncthreads();
#ifndef TS3_NO_SCHED_BARRIER
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
  const int qe = me + 1;      // last step: stage 3 on plane me - 1
  // whole groups of four steps: up to three steps past the last one run along (their loads are clamped into the arrays, stage 3 stores
  // nothing past plane me - 1).  With `break`s inside the group the exits share a block with the loop's back edge: the compiler then joins
  // the load states of all four steps at the loop header and waits for EVERY load in flight at the top of each group.
  for (int q = q0; q <= qe; q += 4) {
    step(q, I0_{});
    step(q + 1, I1_{});
    step(q + 2, I2_{});
    step(q + 3, I3_{});
  }
#undef EB
#undef EBW
}

static thread_local int g_ts3_zc = -1;         // planes per z chunk of the three-step pass; -1: by size (examg_debug_three_stage)
static thread_local int g_ts3_disable = 0;
static thread_local long long g_ts3_minpts = 8000000LL;      // examg_debug_three_stage(2, ..): 2^20, so that the parity tests reach the kernel on small boxes

// the three-stage pass: 3-D 7-point constant stencils on rows of at least 64 points and at least 8 * 10^6 points -- below, a pass per pair of
// stages is faster (traced V-cycle, 128^3: two passes of three colour loops 43 + 41 us against three sweeps of 15-20 us; 256^3: 2 x 0.110 ms
// against 3 x 0.094)
static bool three_stage_ok(const examg_layout_t *lu, const examg_layout_t *lf, const examg_stencil_t *st, const Box &box) {
  if (g_ts3_disable || !two_stage_ok(lu, lf, st, box) || box.n1() < 20 || box.count() < g_ts3_minpts) return false;
  const LayoutDev u = make_layout(lu), f = make_layout(lf);
  return u.s2 < (1LL << 32) && f.s2 < (1LL << 32);
}

template <int NW, int RPW, bool COL>
static int launch_three_stage_shape(const examg_layout_t *lu_, const double *u, const examg_layout_t *lf_, const double *rhs, double *out,
                                    const examg_stencil_t *st, double w, const Box &box, hipStream_t s, int first) {
  constexpr int NO = RPW * NW - 4;
  const LayoutDev lu = make_layout(lu_), lf = make_layout(lf_);
  TSGeom g;
  g.xs = g.ys = g.zs = 0;
  g.ntx = (box.n0() + TS3_OUT - 1) / TS3_OUT;
  g.nty = (box.n1() + NO - 1) / NO;
  const int xy = g.ntx * g.nty, n2 = box.n2();
  // planes per chunk: six halo planes and two start-up steps each, one workgroup per CU -- the chunk count that minimises
  // rounds of 256 workgroups x (planes per chunk + 8), chunks of 16 .. 64 planes where the box has them.  tools/time_three_stage.py --dbg,
  // MI355X, ms per pass at 512^3, 16 / 24 / 40 / 48 / 56 / 64 / 128 planes: 0.887 / 0.834 / 0.825 / 0.834 / 0.815 / 0.899 / 1.023 (another box:
  // 0.861 / 0.810 / - / 0.801 / - / 0.867 / -); 256^3, 16 / 20 / 44 / 64: 0.154 / 0.129 / 0.130 / 0.173
  int zc = n2;
  {
    long long best = -1;
    for (int t = 1; t <= (n2 + 7) / 8; ++t) {
      const int c = (n2 + t - 1) / t, tt = (n2 + c - 1) / c;
      if ((c > 64 && t < (n2 + 7) / 8) || (c < 16 && t > 1)) continue;
      const long long cost = (((long long)xy * tt + 255) / 256) * (((c + 4 + 3) & ~3) + 4);      // whole groups of four steps
      if (best < 0 || cost < best) { best = cost; zc = c; }
    }
    if (zc + 3 < n2) zc = (zc + 3) & ~3;       // zc + 4 steps: a multiple of four wastes none
  }
  if (g_ts3_zc > 0) zc = g_ts3_zc;
  if (zc > n2) zc = n2;
  g.zc = zc;
  g.ntz = (n2 + zc - 1) / zc;
  g.nblocks = xy * g.ntz;
  g.remap = g_ts_remap >= 0 ? g_ts_remap : 2;
  g.first = first;
  g.box1 = box;
  g.ax0 = -lu.ref0; g.ax1 = lu.tot0 - lu.ref0;
  g.ay0 = -lu.ref1; g.ay1 = lu.tot1 - lu.ref1;
  g.az0 = -lu.ref2; g.az1 = lu.tot2 - lu.ref2;
  Coef7 k;
  for (int i = 0; i < 7; ++i) k.c[i] = st->coef[i];
  const int ord = canonical_order7(st);
  const bool nt = g_ts_nt >= 0 ? g_ts_nt != 0 : box.count() * 24LL > 200000000LL;    // the store policy of the two-step passes
  dim3 block(64, NW, 1), grid(g.nblocks, 1, 1);
#define EXAMG_TS3(ORD, NTV) hipLaunchKernelGGL((k_three_stage7_lds<ORD, NW, NTV, RPW, COL>), grid, block, 0, s, lu, u, lf, rhs, out, k, w, box, g)
  if (ord == 0) {
    if (nt) EXAMG_TS3(0, true);
    else EXAMG_TS3(0, false);
  } else {
    if (nt) EXAMG_TS3(1, true);
    else EXAMG_TS3(1, false);
  }
#undef EXAMG_TS3
  EXAMG_CHECK_LAUNCH("k_three_stage7_lds");
  return 0;
}

static int launch_three_stage(const examg_layout_t *lu, const double *u, const examg_layout_t *lf, const double *rhs, double *out,
                              const examg_stencil_t *st, double w, const Box &box, hipStream_t s) {
  // eight waves of three rows (256 VGPRs, two waves per SIMD).  Twelve waves of two rows -- the same 24-row tile at three waves per SIMD --
  // spill 44 registers at 168 and run at half the speed (512^3: 1.68 against 0.83 ms; written, measured, removed)
  return launch_three_stage_shape<8, 3, false>(lu, u, lf, rhs, out, st, w, box, s, 0);
}

static int launch_three_colours(const examg_layout_t *lu, const double *u, const examg_layout_t *lf, const double *rhs, double *out,
                                const examg_stencil_t *st, double w, int first, const Box &box, hipStream_t s) {
  return launch_three_stage_shape<8, 3, true>(lu, u, lf, rhs, out, st, w, box, s, first);
}

}  // namespace examg

using namespace examg;

#ifdef EXAMG_DEBUG_HOOKS
extern "C" int examg_debug_two_stage_lds(int nw) {
  g_ts_lds = nw < 0 ? -2 : nw;
  return 0;
}

extern "C" int examg_debug_two_stage_nt(int nt) {
  g_ts_nt = nt < 0 ? -1 : (nt ? 1 : 0);
  return 0;
}

extern "C" int examg_debug_two_stage_prol(int wpe) {
  if (wpe >= 10) g_ts_wpe = wpe - 10 == 4 ? 4 : 1;     // 11 / 14: the plain passes
  else g_ts_prol_wpe = wpe == 1 ? 1 : 4;
  return 0;
}

extern "C" int examg_debug_three_stage(int disable, int zc) {
  g_ts3_disable = disable == 1;
  g_ts3_minpts = disable == 2 ? (1LL << 20) : 8000000LL;
  g_ts3_zc = zc > 0 ? zc : -1;
  return 0;
}

extern "C" int examg_debug_two_stage(int disable, int blocks, int remap, int wy) {
  g_ts_disable = disable;
  if (blocks != 0) g_ts_blocks = blocks > 0 ? blocks : -1;     // 0: unchanged; negative: back to the rule
  if (remap >= 0) g_ts_remap = remap;
  g_ts_minzc = wy > 0 ? wy : -1;    // 4th argument: minimum planes per z chunk (<= 0: default rule)
  return 0;
}
#endif

// Will examg_jacobi2_boxes / examg_rbgs_sweep_fused_boxes take the one-pass kernel for these arguments (1) or their
// fallback through `tmp` (0)?  The ONE place this is decided: callers that overlap the pass with work on another stream ask
// here, because the fallback writes `tmp` on the launch stream.
extern "C" int examg_two_stage_eligible(const examg_layout_t *lu, const examg_layout_t *lf, const examg_stencil_t *st,
                                        const int32_t *begin1, const int32_t *end1, const int32_t *begin2, const int32_t *end2) {
  if (!lu || !lf || !st || !begin1 || !end1 || !begin2 || !end2) return 0;
  const Box box1 = make_box(begin1, end1), box2 = make_box(begin2, end2);
  if (box2.count() == 0) return 0;
  if (box2.b0 < box1.b0 || box2.b1 < box1.b1 || box2.b2 < box1.b2 || box2.e0 > box1.e0 || box2.e1 > box1.e1 || box2.e2 > box1.e2) return 0;
  // launch-bound levels (rows shorter than 64 points): the small-level sweep, one box only (kernels_small.hip)
  if (box1.b0 == box2.b0 && box1.b1 == box2.b1 && box1.b2 == box2.b2 && box1.e0 == box2.e0 && box1.e1 == box2.e1 && box1.e2 == box2.e2 &&
      small_two_stage_ok(lu, lf, st, box2))
    return 1;
  if (!(two_stage_ok(lu, lf, st, box2) && box_inside(lu, box1, 1) && box_inside(lf, box1, 0) && stage_boxes_ok(lf, box1, box2))) return 0;
  // launch_two_stage_lds: row offsets within a plane are 32-bit products
  const LayoutDev u = make_layout(lu), f = make_layout(lf);
  return (u.s2 < (1LL << 32) && f.s2 < (1LL << 32)) ? 1 : 0;
}

// One full red-black sweep, out of place.
extern "C" int examg_rbgs_sweep_fused(const examg_layout_t *lu, const double *u_in, double *u_out,
                                      const examg_layout_t *lf, const double *rhs, const examg_stencil_t *st, double w,
                                      int first, const int32_t *begin, const int32_t *end, examg_stream_t stream) {
  if (!lu || !u_in || !u_out || !lf || !rhs || !st || !begin || !end) { set_error("examg_rbgs_sweep_fused: null argument"); return 1; }
  if (u_in == u_out) { set_error("examg_rbgs_sweep_fused: out of place only"); return 1; }
  if (first != 0 && first != 1) { set_error("examg_rbgs_sweep_fused: first colour must be 0 or 1"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (two_stage_ok(lu, lf, st, box)) return launch_two_stage<true>(lu, u_in, lf, rhs, u_out, st, w, first, box, (hipStream_t)stream);
  if (small_two_stage_ok(lu, lf, st, box)) return launch_small_two_stage(true, 0, lu, u_in, u_out, lf, rhs, st, w, first, box, nullptr, nullptr, (hipStream_t)stream);
  // general stencils / small boxes: bring the box and its one-stencil-reach shell over, then the two half
  // sweeps in place on the copy (the shell of u_out receives u_in's shell values -- see the header)
  const int reach = stencil_reach(st);
  int32_t b2[3], e2[3];
  for (int d = 0; d < 3; ++d) {
    const bool on = d < lu->nd;
    b2[d] = begin[d] - (on ? reach : 0);
    e2[d] = end[d] + (on ? reach : 0);
  }
  int rc = examg_axpby(lu, u_in, lu, u_out, 1.0, 0.0, b2, e2, stream);
  if (rc) return rc;
  rc = examg_rbgs_colour(lu, u_out, lf, rhs, st, w, first, begin, end, stream);
  if (rc) return rc;
  return examg_rbgs_colour(lu, u_out, lf, rhs, st, w, 1 - first, begin, end, stream);
}

// One full red-black sweep of the ZERO field (every value of u, boundary planes included, is 0.0): u_out receives on the box
// what examg_rbgs_sweep_fused would write for such a u_in, which is never read.
extern "C" int examg_rbgs_sweep_fused_zero(const examg_layout_t *lu, double *u_out, const examg_layout_t *lf, const double *rhs,
                                           const examg_stencil_t *st, double w, int first, const int32_t *begin,
                                           const int32_t *end, examg_stream_t stream) {
  if (!lu || !u_out || !lf || !rhs || !st || !begin || !end) { set_error("examg_rbgs_sweep_fused_zero: null argument"); return 1; }
  if (first != 0 && first != 1) { set_error("examg_rbgs_sweep_fused_zero: first colour must be 0 or 1"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (two_stage_ok(lu, lf, st, box))
    return launch_two_stage<true>(lu, u_out, lf, rhs, u_out, st, w, first, box, (hipStream_t)stream, nullptr, nullptr, true);
  if (small_two_stage_ok(lu, lf, st, box)) return launch_small_two_stage(true, 2, lu, nullptr, u_out, lf, rhs, st, w, first, box, nullptr, nullptr, (hipStream_t)stream);
  // general stencils / small boxes: zero the box and its one-stencil-reach shell, then the two half sweeps in place
  const int reach = stencil_reach(st);
  int32_t b2[3], e2[3];
  for (int d = 0; d < 3; ++d) {
    const bool on = d < lu->nd;
    b2[d] = begin[d] - (on ? reach : 0);
    e2[d] = end[d] + (on ? reach : 0);
  }
  int rc = examg_set(lu, u_out, 0.0, b2, e2, stream);
  if (rc) return rc;
  rc = examg_rbgs_colour(lu, u_out, lf, rhs, st, w, first, begin, end, stream);
  if (rc) return rc;
  return examg_rbgs_colour(lu, u_out, lf, rhs, st, w, 1 - first, begin, end, stream);
}

// `u += Prolongation * uc` on [begin,end) followed by one full red-black sweep on the same box, out of place: u_out receives
// the swept values on the box (every point of which the sweep rewrites); u_in is not modified.
extern "C" int examg_rbgs_sweep_fused_prolong(const examg_layout_t *lu, const double *u_in, double *u_out,
                                              const examg_layout_t *lf, const double *rhs, const examg_stencil_t *st, double w,
                                              int first, const int32_t *begin, const int32_t *end, const examg_layout_t *lc,
                                              const double *uc, examg_stream_t stream) {
  if (!lu || !u_in || !u_out || !lf || !rhs || !st || !begin || !end || !lc || !uc) { set_error("examg_rbgs_sweep_fused_prolong: null argument"); return 1; }
  if (u_in == u_out) { set_error("examg_rbgs_sweep_fused_prolong: out of place only"); return 1; }
  if (first != 0 && first != 1) { set_error("examg_rbgs_sweep_fused_prolong: first colour must be 0 or 1"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (two_stage_ok(lu, lf, st, box)) {
    if (!prolong_args_ok("examg_rbgs_sweep_fused_prolong", lc, box)) return 1;
    TSProl pr;
    pr.lc = make_layout(lc);
    pr.uc = uc;
    return launch_two_stage<true>(lu, u_in, lf, rhs, u_out, st, w, first, box, (hipStream_t)stream, nullptr, &pr);
  }
  if (small_two_stage_ok(lu, lf, st, box)) {
    if (!prolong_args_ok("examg_rbgs_sweep_fused_prolong", lc, box)) return 1;
    return launch_small_two_stage(true, 1, lu, u_in, u_out, lf, rhs, st, w, first, box, lc, uc, (hipStream_t)stream);
  }
  // general stencils / small boxes: the three loops one after the other on a copy (box + one-stencil-reach shell)
  const int reach = stencil_reach(st);
  int32_t b2[3], e2[3];
  for (int d = 0; d < 3; ++d) {
    const bool on = d < lu->nd;
    b2[d] = begin[d] - (on ? reach : 0);
    e2[d] = end[d] + (on ? reach : 0);
  }
  int rc = examg_axpby(lu, u_in, lu, u_out, 1.0, 0.0, b2, e2, stream);
  if (rc) return rc;
  rc = examg_prolong_add(lc, uc, lu, u_out, begin, end, stream);
  if (rc) return rc;
  rc = examg_rbgs_colour(lu, u_out, lf, rhs, st, w, first, begin, end, stream);
  if (rc) return rc;
  return examg_rbgs_colour(lu, u_out, lf, rhs, st, w, 1 - first, begin, end, stream);
}

// One red-black sweep with separate boxes (blocks with neighbours): colour `first` on [begin1,end1) (points outside keep
// u_in's value), then the other colour of that field on [begin2,end2), inside box 1; u_out receives the result on box 2.
extern "C" int examg_rbgs_sweep_fused_boxes(const examg_layout_t *lu, const double *u_in, double *u_out, double *tmp,
                                            const examg_layout_t *lf, const double *rhs, const examg_stencil_t *st, double w,
                                            int first, const int32_t *begin1, const int32_t *end1, const int32_t *begin2,
                                            const int32_t *end2, examg_stream_t stream) {
  if (!lu || !u_in || !u_out || !lf || !rhs || !st || !begin1 || !end1 || !begin2 || !end2) { set_error("examg_rbgs_sweep_fused_boxes: null argument"); return 1; }
  if (u_in == u_out) { set_error("examg_rbgs_sweep_fused_boxes: out of place only"); return 1; }
  if (first != 0 && first != 1) { set_error("examg_rbgs_sweep_fused_boxes: first colour must be 0 or 1"); return 1; }
  const Box box1 = make_box(begin1, end1), box2 = make_box(begin2, end2);
  if (box2.count() == 0) return 0;
  if (box2.b0 < box1.b0 || box2.b1 < box1.b1 || box2.b2 < box1.b2 || box2.e0 > box1.e0 || box2.e1 > box1.e1 || box2.e2 > box1.e2) {
    set_error("examg_rbgs_sweep_fused_boxes: the second box must lie inside the first");
    return 1;
  }
  if (two_stage_ok(lu, lf, st, box2) && box_inside(lu, box1, 1) && box_inside(lf, box1, 0) && stage_boxes_ok(lf, box1, box2))
    return launch_two_stage<true>(lu, u_in, lf, rhs, u_out, st, w, first, box2, (hipStream_t)stream, &box1);
  if (box1.b0 == box2.b0 && box1.b1 == box2.b1 && box1.b2 == box2.b2 && box1.e0 == box2.e0 && box1.e1 == box2.e1 && box1.e2 == box2.e2 &&
      small_two_stage_ok(lu, lf, st, box2))
    return launch_small_two_stage(true, 0, lu, u_in, u_out, lf, rhs, st, w, first, box2, nullptr, nullptr, (hipStream_t)stream);
  if (!tmp || tmp == u_in || tmp == u_out) { set_error("examg_rbgs_sweep_fused_boxes: fallback needs a distinct tmp array"); return 1; }
  const int reach = stencil_reach(st);
  int32_t b2[3], e2[3];
  for (int d = 0; d < 3; ++d) {
    const bool on = d < lu->nd;
    b2[d] = begin1[d] - (on ? reach : 0);
    e2[d] = end1[d] + (on ? reach : 0);
  }
  int rc = examg_axpby(lu, u_in, lu, tmp, 1.0, 0.0, b2, e2, stream);
  if (rc) return rc;
  rc = examg_rbgs_colour(lu, tmp, lf, rhs, st, w, first, begin1, end1, stream);     // in place on the copy
  if (rc) return rc;
  rc = examg_axpby(lu, tmp, lu, u_out, 1.0, 0.0, begin2, end2, stream);
  if (rc) return rc;
  // other colour: reads the copy, writes that colour's points of u_out
  return examg_stencil_op(EXAMG_SMOOTH, lu, tmp, lf, rhs, lu, u_out, st, w, 1 - first, begin2, end2, stream);
}

// Two Jacobi steps with separate boxes: stage 1 = J on [begin1,end1) (points outside keep u_in's value), stage 2 = J of
// that field on [begin2,end2) (inside box 1), written to u_out.  A block with neighbours uses box 2 = box 1 minus the
// duplicate planes at interior faces: everything stage 2 needs there is local (exastencils_amd/solver.py: Smoothers).
extern "C" int examg_jacobi2_boxes(const examg_layout_t *lu, const double *u_in, double *u_out, double *tmp,
                                   const examg_layout_t *lf, const double *rhs, const examg_stencil_t *st, double w,
                                   const int32_t *begin1, const int32_t *end1, const int32_t *begin2, const int32_t *end2,
                                   examg_stream_t stream) {
  if (!lu || !u_in || !u_out || !lf || !rhs || !st || !begin1 || !end1 || !begin2 || !end2) { set_error("examg_jacobi2_boxes: null argument"); return 1; }
  if (u_in == u_out) { set_error("examg_jacobi2_boxes: out of place only"); return 1; }
  const Box box1 = make_box(begin1, end1), box2 = make_box(begin2, end2);
  if (box2.count() == 0) return 0;
  if (box2.b0 < box1.b0 || box2.b1 < box1.b1 || box2.b2 < box1.b2 || box2.e0 > box1.e0 || box2.e1 > box1.e1 || box2.e2 > box1.e2) {
    set_error("examg_jacobi2_boxes: the stage-2 box must lie inside the stage-1 box");
    return 1;
  }
  if (two_stage_ok(lu, lf, st, box2) && box_inside(lu, box1, 1) && box_inside(lf, box1, 0) && stage_boxes_ok(lf, box1, box2))
    return launch_two_stage<false>(lu, u_in, lf, rhs, u_out, st, w, 0, box2, (hipStream_t)stream, &box1);
  if (box1.b0 == box2.b0 && box1.b1 == box2.b1 && box1.b2 == box2.b2 && box1.e0 == box2.e0 && box1.e1 == box2.e1 && box1.e2 == box2.e2 &&
      small_two_stage_ok(lu, lf, st, box2))
    return launch_small_two_stage(false, 0, lu, u_in, u_out, lf, rhs, st, w, 0, box2, nullptr, nullptr, (hipStream_t)stream);
  if (const int r27 = sf27_jacobi2_try(lu, u_in, u_out, lf, rhs, st, w, box1, box2, (hipStream_t)stream)) return r27 < 0 ? 1 : 0;
  if (!tmp || tmp == u_in || tmp == u_out) { set_error("examg_jacobi2_boxes: fallback needs a distinct tmp array"); return 1; }
  const int reach = stencil_reach(st);
  int32_t b2[3], e2[3];
  for (int d = 0; d < 3; ++d) {
    const bool on = d < lu->nd;
    b2[d] = begin1[d] - (on ? reach : 0);
    e2[d] = end1[d] + (on ? reach : 0);
  }
  int rc = examg_axpby(lu, u_in, lu, tmp, 1.0, 0.0, b2, e2, stream);
  if (rc) return rc;
  rc = examg_jacobi(lu, u_in, tmp, lf, rhs, st, w, begin1, end1, stream);
  if (rc) return rc;
  return examg_jacobi(lu, tmp, u_out, lf, rhs, st, w, begin2, end2, stream);
}

// `u += Prolongation * uc` on [begin,end) followed by two Jacobi steps on the same box: u_out receives the result on the box,
// u_in is not modified; `tmp` (a distinct array) is only used by the fallback.
extern "C" int examg_jacobi2_prolong(const examg_layout_t *lu, const double *u_in, double *u_out, double *tmp,
                                     const examg_layout_t *lf, const double *rhs, const examg_stencil_t *st, double w,
                                     const int32_t *begin, const int32_t *end, const examg_layout_t *lc, const double *uc,
                                     examg_stream_t stream) {
  if (!lu || !u_in || !u_out || !lf || !rhs || !st || !begin || !end || !lc || !uc) { set_error("examg_jacobi2_prolong: null argument"); return 1; }
  if (u_in == u_out) { set_error("examg_jacobi2_prolong: out of place only"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (two_stage_ok(lu, lf, st, box)) {
    if (!prolong_args_ok("examg_jacobi2_prolong", lc, box)) return 1;
    TSProl pr;
    pr.lc = make_layout(lc);
    pr.uc = uc;
    return launch_two_stage<false>(lu, u_in, lf, rhs, u_out, st, w, 0, box, (hipStream_t)stream, nullptr, &pr);
  }
  if (small_two_stage_ok(lu, lf, st, box)) {
    if (!prolong_args_ok("examg_jacobi2_prolong", lc, box)) return 1;
    return launch_small_two_stage(false, 1, lu, u_in, u_out, lf, rhs, st, w, 0, box, lc, uc, (hipStream_t)stream);
  }
  if (!tmp || tmp == u_in || tmp == u_out) { set_error("examg_jacobi2_prolong: fallback needs a distinct tmp array"); return 1; }
  const int reach = stencil_reach(st);
  int32_t b2[3], e2[3];
  for (int d = 0; d < 3; ++d) {
    const bool on = d < lu->nd;
    b2[d] = begin[d] - (on ? reach : 0);
    e2[d] = end[d] + (on ? reach : 0);
  }
  // u_out holds the corrected field for the first step, tmp (with the box's shell) the intermediate one
  int rc = examg_axpby(lu, u_in, lu, u_out, 1.0, 0.0, b2, e2, stream);
  if (rc) return rc;
  rc = examg_axpby(lu, u_in, lu, tmp, 1.0, 0.0, b2, e2, stream);
  if (rc) return rc;
  rc = examg_prolong_add(lc, uc, lu, u_out, begin, end, stream);
  if (rc) return rc;
  rc = examg_jacobi(lu, u_out, tmp, lf, rhs, st, w, begin, end, stream);
  if (rc) return rc;
  return examg_jacobi(lu, tmp, u_out, lf, rhs, st, w, begin, end, stream);
}

// Two Jacobi steps, u_in -> (u_in's values after two sweeps) in u_out; `tmp` is only used by the fallback
// (general stencils / small boxes), where it receives the intermediate sweep.
extern "C" int examg_jacobi2(const examg_layout_t *lu, const double *u_in, double *u_out, double *tmp,
                             const examg_layout_t *lf, const double *rhs, const examg_stencil_t *st, double w,
                             const int32_t *begin, const int32_t *end, examg_stream_t stream) {
  if (!lu || !u_in || !u_out || !lf || !rhs || !st || !begin || !end) { set_error("examg_jacobi2: null argument"); return 1; }
  if (u_in == u_out) { set_error("examg_jacobi2: out of place only"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (two_stage_ok(lu, lf, st, box)) return launch_two_stage<false>(lu, u_in, lf, rhs, u_out, st, w, 0, box, (hipStream_t)stream);
  if (small_two_stage_ok(lu, lf, st, box)) return launch_small_two_stage(false, 0, lu, u_in, u_out, lf, rhs, st, w, 0, box, nullptr, nullptr, (hipStream_t)stream);
  if (const int r27 = sf27_jacobi2_try(lu, u_in, u_out, lf, rhs, st, w, box, box, (hipStream_t)stream)) return r27 < 0 ? 1 : 0;
  if (!tmp || tmp == u_in || tmp == u_out) { set_error("examg_jacobi2: fallback needs a distinct tmp array"); return 1; }
  // the intermediate sweep needs the box's shell (Dirichlet / halo values) in tmp
  const int reach = stencil_reach(st);
  int32_t b2[3], e2[3];
  for (int d = 0; d < 3; ++d) {
    const bool on = d < lu->nd;
    b2[d] = begin[d] - (on ? reach : 0);
    e2[d] = end[d] + (on ? reach : 0);
  }
  int rc = examg_axpby(lu, u_in, lu, tmp, 1.0, 0.0, b2, e2, stream);
  if (rc) return rc;
  rc = examg_jacobi(lu, u_in, tmp, lf, rhs, st, w, begin, end, stream);
  if (rc) return rc;
  return examg_jacobi(lu, tmp, u_out, lf, rhs, st, w, begin, end, stream);
}

// Three Jacobi steps, u_in -> u_out on the box (u_out's planes outside the box are not written: the caller keeps boundary and ghost
// values there, as for examg_jacobi2); one pass where k_three_stage7_lds applies, otherwise a step into `tmp` and a pair from there (or
// three steps), for which `tmp` must be a distinct array.
extern "C" int examg_jacobi3(const examg_layout_t *lu, const double *u_in, double *u_out, double *tmp, const examg_layout_t *lf,
                             const double *rhs, const examg_stencil_t *st, double w, const int32_t *begin, const int32_t *end,
                             examg_stream_t stream) {
  if (!lu || !u_in || !u_out || !lf || !rhs || !st || !begin || !end) { set_error("examg_jacobi3: null argument"); return 1; }
  if (u_in == u_out) { set_error("examg_jacobi3: out of place only"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (three_stage_ok(lu, lf, st, box)) return launch_three_stage(lu, u_in, lf, rhs, u_out, st, w, box, (hipStream_t)stream);
  if (!tmp || tmp == u_in || tmp == u_out) { set_error("examg_jacobi3: fallback needs a distinct tmp array"); return 1; }
  const int reach = stencil_reach(st);
  int32_t b2[3], e2[3];
  for (int d = 0; d < 3; ++d) {
    const bool on = d < lu->nd;
    b2[d] = begin[d] - (on ? reach : 0);
    e2[d] = end[d] + (on ? reach : 0);
  }
  // the first step into tmp (with the box's shell: Dirichlet / halo values for the steps behind it)
  int rc = examg_axpby(lu, u_in, lu, tmp, 1.0, 0.0, b2, e2, stream);
  if (rc) return rc;
  rc = examg_jacobi(lu, u_in, tmp, lf, rhs, st, w, begin, end, stream);
  if (rc) return rc;
  if (examg_two_stage_eligible(lu, lf, st, begin, end, begin, end))      // the pair in one pass (it does not touch its tmp then)
    return examg_jacobi2(lu, tmp, u_out, nullptr, lf, rhs, st, w, begin, end, stream);
  rc = examg_axpby(lu, u_in, lu, u_out, 1.0, 0.0, b2, e2, stream);
  if (rc) return rc;
  rc = examg_jacobi(lu, tmp, u_out, lf, rhs, st, w, begin, end, stream);       // second step: tmp -> u_out
  if (rc) return rc;
  rc = examg_jacobi(lu, u_out, tmp, lf, rhs, st, w, begin, end, stream);       // third step: u_out -> tmp
  if (rc) return rc;
  return examg_axpby(lu, tmp, lu, u_out, 1.0, 0.0, begin, end, stream);
}

// 1 if examg_jacobi3 / examg_rbgs_colours3 will run their one-pass kernel for this box, 0 if they will run their loops one after the other
extern "C" int examg_three_stage_eligible(const examg_layout_t *lu, const examg_layout_t *lf, const examg_stencil_t *st, const int32_t *begin,
                                          const int32_t *end) {
  if (!lu || !lf || !st || !begin || !end) return 0;
  return three_stage_ok(lu, lf, st, make_box(begin, end)) ? 1 : 0;
}

// Three colour loops of a red-black smoother in ONE pass, out of place: colour `first` on [begin,end), then the other colour, then `first`
// again (`repeat 3 times { color with { (i0 + i1 + i2) % 2 ... } }`, Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:204-213, is six colour
// loops: two such passes, the second with first = 1 - first).  u_out receives the result on the box; bit-identical to three
// examg_rbgs_colour calls in place.  Where the one-pass kernel does not apply: a copy of the box with its shell and the three loops on it.
extern "C" int examg_rbgs_colours3(const examg_layout_t *lu, const double *u_in, double *u_out, const examg_layout_t *lf, const double *rhs,
                                   const examg_stencil_t *st, double w, int first, const int32_t *begin, const int32_t *end,
                                   examg_stream_t stream) {
  if (!lu || !u_in || !u_out || !lf || !rhs || !st || !begin || !end) { set_error("examg_rbgs_colours3: null argument"); return 1; }
  if (u_in == u_out) { set_error("examg_rbgs_colours3: out of place only"); return 1; }
  if (first != 0 && first != 1) { set_error("examg_rbgs_colours3: first colour must be 0 or 1"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (three_stage_ok(lu, lf, st, box)) return launch_three_colours(lu, u_in, lf, rhs, u_out, st, w, first, box, (hipStream_t)stream);
  const int reach = stencil_reach(st);
  int32_t b2[3], e2[3];
  for (int d = 0; d < 3; ++d) {
    const bool on = d < lu->nd;
    b2[d] = begin[d] - (on ? reach : 0);
    e2[d] = end[d] + (on ? reach : 0);
  }
  int rc = examg_axpby(lu, u_in, lu, u_out, 1.0, 0.0, b2, e2, stream);
  for (int k = 0; k < 3 && !rc; ++k) rc = examg_rbgs_colour(lu, u_out, lf, rhs, st, w, (k & 1) ? 1 - first : first, begin, end, stream);
  return rc;
}
