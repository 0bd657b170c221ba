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
  long long cplane;  // stride between the entries of a point: doubles per coefficient plane, or 1 under the entry-fastest transformation
  long long cpt;     // stride between points: 1, or nent under the entry-fastest transformation
  int wdiv;          // smoother weight of a stencil field: 0 (1.0 / diag) * w, 1 w / diag (EXAMG_WEIGHT_*)
  signed char o[EXAMG_MAX_ENTRIES][3];   // entry offsets per dimension: u under a layout transformation has no linear entry offsets
  long long ro[EXAMG_MAX_ENTRIES];       // ... the y / z part of an entry's offset (row and plane strides of the half arrays)
};

// ---------------------------------------------------------------------------------------------
// generic path
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256)
k_stencil_generic(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs, LayoutDev ld,
                  double *dst, LayoutDev lc, StencilDev st, double w, int colour, Box box, int row_w, int passthru) {
  const long long rows = (long long)box.n1() * box.n2();
  const long long total = rows * row_w;
  const bool small = total < (1LL << 31);      // 32-bit index arithmetic (wave-uniform choice; same indices)
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int c0, i1, i2;
    if (small) {
      const unsigned tt = (unsigned)t, row = tt / (unsigned)row_w, q = row / (unsigned)box.n1();
      c0 = (int)(tt - row * (unsigned)row_w);
      i1 = box.b1 + (int)(row - q * (unsigned)box.n1());
      i2 = box.b2 + (int)q;
    } else {
      const long long row = t / row_w;
      c0 = (int)(t - row * row_w);
      i1 = box.b1 + (int)(row % box.n1());
      i2 = box.b2 + (int)(row / box.n1());
    }
    int i0;
    if (colour >= 0) {
      // points of this row with (i0+i1+i2) % 2 == colour, no idle lanes
      const int first = box.b0 + (((box.b0 + i1 + i2) & 1) != colour ? 1 : 0);
      i0 = first + 2 * c0;
      if (passthru) {      // out of place: the other point of the pair is carried over (dst = u there), one loop instead of copy + loop
        const int p = first == box.b0 ? i0 + 1 : i0 - 1;
        if (p >= box.b0 && p < box.e0) dst[lidx(ld, p, i1, i2)] = u[lidx(lu, p, i1, i2)];
      }
    } else {
      i0 = box.b0 + c0;
    }
    if (i0 >= box.e0) continue;
    const long long iu = lidx(lu, i0, i1, i2);
    double acc;
    if (lu.half) {
      // u under the colour split (EXAMG_LAYOUT_SPLIT_X): every neighbour through the transformed index -- the points of one colour of a
      // row, and their x neighbours in the other half array, are contiguous across the lanes.  Same products, same order.
      // index of entry k's point: x through the split (column ax + dx lives at (ax + dx) / 2 of half (ax + dx) % 2), y and z by strides
      const int ax = i0 + lu.ref0;
      const long long rowb = lu.s1 * (i1 + lu.ref1) + lu.s2 * (i2 + lu.ref2);
      auto un = [&](int k) {
        const int a = ax + st.o[k][0];
        return u[(a >> 1) + (long long)(a & 1) * lu.half + rowb + st.ro[k]];
      };
      if (st.cfield) {
        const long long ic = lidx(lc, i0, i1, i2) * st.cpt;
        acc = st.cfield[ic] * un(0);
        for (int k = 1; k < st.nent; ++k) acc = acc + st.cfield[ic + k * st.cplane] * un(k);
        if (MODE == EXAMG_SMOOTH) {
          const double dg = st.cfield[ic + st.diag * st.cplane];
          const double ww = st.wdiv ? w / dg : (1.0 / dg) * w;
          acc = u[iu] + ww * (rhs[lidx(lf, i0, i1, i2)] - acc);
        }
      } else {
        acc = st.coef[0] * un(0);
        for (int k = 1; k < st.nent; ++k) acc = acc + st.coef[k] * un(k);
        if (MODE == EXAMG_SMOOTH) acc = u[iu] + w * (rhs[lidx(lf, i0, i1, i2)] - acc);
      }
    } else if (st.cfield) {
      const long long ic = lidx(lc, i0, i1, i2) * st.cpt;
      acc = st.cfield[ic] * u[iu + st.uo[0]];
      for (int k = 1; k < st.nent; ++k) acc = acc + st.cfield[ic + k * st.cplane] * u[iu + st.uo[k]];
      if (MODE == EXAMG_SMOOTH) {
        const double dg = st.cfield[ic + st.diag * st.cplane];
        const double ww = st.wdiv ? w / dg : (1.0 / dg) * w;
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
    const long long iu = lidx_plain(lu, i0, i1, i2), ic = lidx_plain(lc, i0, i1, i2);
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
      acc = v[0] + ww * (rhs[lidx_plain(lf, i0, i1, i2)] - acc);
    }
    if (MODE == EXAMG_RESIDUAL) acc = rhs[lidx_plain(lf, i0, i1, i2)] - acc;
    dst[lidx_plain(ld, i0, i1, i2)] = acc;
  }
}

// 27-entry stencil fields under the layout transformation `[x, y, z, i] => [i, x, y, z]` (EXAMG_CLAYOUT_ENTRY_FASTEST): the 27
// coefficients of a point are one 216-byte record, a row of the box one contiguous run of records.  A wave takes 64 consecutive
// points: their 13.5 KiB of coefficients arrive as ONE stream of 16-byte loads (lane l takes doubles 2 (64 i + l), ..: whole
// cache lines per wave), pass through a wave-private LDS strip (no barrier: the LDS operations of one wave execute in order) and
// come back as the lane's own 27 values (stride 27 doubles: 54 dwords, gcd with the 64 banks is 2 -- an 8-byte read of 32 lanes
// covers all banks once).  u: the point's 27 neighbours through L1 / L2 as in the unrolled kernel; same products in the same order.
constexpr int SF27_WAVES = 4;
static thread_local int g_sf27_run = -1;   // examg_debug_sf27_run (debug build): tiles per wave

struct SF27Tile {
  int x, nv, i1, i2;
  long long rec0;
};

template <int MODE>
__global__ void __launch_bounds__(64 * SF27_WAVES)
k_stencilfield27_rec(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs, LayoutDev ld,
                     double *__restrict__ dst, LayoutDev lc, const double *__restrict__ cf, UOffsets uo, double w, Box box, int tiles_x,
                     long long ntiles, int run) {
  __shared__ __attribute__((aligned(16))) double strip[SF27_WAVES][64 * 27];
  const int lane = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
  // a wave takes `run` consecutive tiles; the coefficient loads of tile j + 1 are in flight while tile j is transposed and summed
  const long long t0 = ((long long)blockIdx.x * SF27_WAVES + wv) * run;
  if (t0 >= ntiles) return;
  const long long t1 = min(t0 + run, ntiles);
  const long long cend = lc.size * 27 - 2;                    // last position a 16-byte load may start at
  auto tile_of = [&](long long tile) {
    SF27Tile t;
    const long long row = tile / tiles_x;
    const int tx = (int)(tile - row * tiles_x);
    t.i1 = box.b1 + (int)(row % box.n1());
    t.i2 = box.b2 + (int)(row / box.n1());
    t.x = box.b0 + tx * 64;
    t.nv = min(64, box.e0 - t.x);
    t.rec0 = lidx_plain(lc, t.x, t.i1, t.i2) * 27;
    return t;
  };
  auto load_raw = [&](d2 (&raw)[14], const SF27Tile &t) {
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      long long off = t.rec0 + 2 * (64 * i + lane);
      off = off < cend ? off : cend;                          // past the tile's records: some in-bounds value nobody reads
      raw[i] = load2(cf + off);
    }
  };
  double *sb = strip[wv];
  auto work = [&](const d2 (&raw)[14], const SF27Tile &t) {
    const int x = t.x + (lane < t.nv ? lane : 0);
    const long long iu = lidx_plain(lu, x, t.i1, t.i2);
    double v[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) v[k] = u[iu + uo.o[k]];
    double f = 0.0;
    if (MODE != EXAMG_APPLY) f = rhs[lidx_plain(lf, x, t.i1, t.i2)];
#pragma unroll
    for (int i = 0; i < 14; ++i)
      if (i < 13 || lane < 32) *reinterpret_cast<d2 *>(sb + 2 * (64 * i + lane)) = raw[i];
    double c[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) c[k] = sb[27 * lane + k];
    double acc = c[0] * v[0];
#pragma unroll
    for (int k = 1; k < 27; ++k) acc = acc + c[k] * v[k];
    if (MODE == EXAMG_SMOOTH) {
      const double ww = (1.0 / c[0]) * w;      // the centre entry comes first (dispatch condition)
      acc = v[0] + ww * (f - acc);
    }
    if (MODE == EXAMG_RESIDUAL) acc = f - acc;
    if (lane < t.nv) dst[lidx_plain(ld, x, t.i1, t.i2)] = acc;
  };
  d2 ra[14], rb[14];
  SF27Tile ta = tile_of(t0), tb = ta;
  load_raw(ra, ta);
  for (long long t = t0; t < t1; t += 2) {
    if (t + 1 < t1) { tb = tile_of(t + 1); load_raw(rb, tb); }
    work(ra, ta);
    if (t + 1 < t1) {
      if (t + 2 < t1) { ta = tile_of(t + 2); load_raw(ra, ta); }
      work(rb, tb);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 3-D 7-point constant-coefficient z-marching kernel
// ---------------------------------------------------------------------------------------------
// kernel-internal mode: the residual's squares are summed instead of the residual being stored (examg_residual_norm2)
constexpr int ZM_RESNORM = 3;

template <int MODE>
__device__ __forceinline__ double finish(double u, double acc, double f, double w) {
  if (MODE == EXAMG_APPLY) return acc;
  if (MODE == EXAMG_RESIDUAL || MODE == ZM_RESNORM) return f - acc;
  return u + w * (f - acc);
}

__device__ __forceinline__ double zm_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = v + __shfl_down(v, o);
  return v;
}

struct ZMarchGeom {
  int colour;         // COL kernels: the colour to update
  int xo;             // x of lane 0's first point in tile 0: box.b0, or box.b0 - 1 where that makes every 16-byte access aligned
  int ntx, nty, ntz;  // tiles per dim
  int zc;             // planes per march chunk
  int remap;          // 2: XCD-contiguous within every z layer of tiles (kernels_twostage.hip); 0: plain order
  int store;          // 0: non-temporal stores; 1: lanes in the partial first / last cache line of the wave's window store cached
                      // (the L2 merges the two halves of a line that two x-adjacent waves share); 2: all stores cached
};

template <bool ALIAS> struct Ptr { typedef double *__restrict__ out; typedef const double *__restrict__ in; };
template <> struct Ptr<true> { typedef double *out; typedef const double *in; };

// Tile = 128 x-points (2 per lane) by RY rows per wave, WY waves per workgroup, marching in z with a register pipeline
// (u[z-1], u[z], u[z+1]); the loads of step z+1 -- own rows of plane z+2, rhs of plane z+1, the two y-halo rows and the two
// window-edge values of plane z+1 -- are in flight while step z is computed.
// COL: red-black half sweep in place (dst == u): of the two points a lane holds per row exactly one has
//   (i0+i1+i2) % 2 == g.colour; it is updated, the other is written back unchanged (whole 16-byte stores).
//   Race-free: only values of the other colour (never written in this sweep) and the lane's own centre are used.
// Configuration from tools/lab/stencil_lab.hip on MI355X at 512^3: 2 rows per wave, 4 waves per workgroup,
// ~512 workgroups (two per CU), prefetch depth 1, edge values prefetched with the stage (they were the one load whose
// latency every step waited for: -4 % at 512^3, -10 % at 256^3), non-temporal 16-byte stores.
template <int MODE, int ORDER, int RY, int WY, bool COL>
__global__ void __launch_bounds__(64 * WY)
k_stencil7_zmarch(LayoutDev lu, typename Ptr<COL>::in u, LayoutDev lf, const double *__restrict__ rhs, LayoutDev ld,
                  typename Ptr<COL>::out dst, Coef7 k, double w, Box box, ZMarchGeom g) {
  const int lane = threadIdx.x;   // 0..63
  const int wv = threadIdx.y;     // wave in block
  int t = blockIdx.x;
  if (g.remap == 2) {   // workgroups are dealt round-robin to the 8 XCDs: within a layer of tiles every XCD takes a band of y-adjacent tiles
    const int xy = g.ntx * g.nty;
    const int lz = t / xy, r = t - lz * xy;
    const int per = xy >> 3;
    t = lz * xy + (r < (per << 3) ? (r & 7) * per + (r >> 3) : r);
  }
  const int tx = t % g.ntx;
  t /= g.ntx;
  const int ty = t % g.nty;
  const int tz = t / g.nty;

  const int x = g.xo + tx * 128 + lane * 2;
  const int rw = box.b1 + (ty * WY + wv) * RY;  // first row of this wave
  const int mb = box.b2 + tz * g.zc;
  const int me = min(mb + g.zc, box.e2);
  const int rwu = __builtin_amdgcn_readfirstlane(rw);                       // the same value in a scalar register
  const unsigned long long dst_lo = (unsigned long long)(uintptr_t)dst;   // only its low 7 bits matter
  // ZM_RESNORM: `dst` is the array of partial sums, one per wave of the grid (fixed geometry: reproducible run to run)
  double ssum = 0.0;
  if (rw >= box.e1) {        // wave-uniform
    if (MODE == ZM_RESNORM && lane == 0) dst[blockIdx.x * WY + wv] = 0.0;
    return;
  }
  const bool va = x >= box.b0 && x < box.e0, vb = x + 1 < box.e0;   // x + 1 >= b0 always
  // right neighbour of b comes from lane+1 unless that lane is past the box
  const bool rload = vb && (lane == 63 || x + 2 >= box.e0);
  const bool lload = va && lane == 0;
  const int xs = (va || vb) ? x : g.xo;  // safe column for idle lanes (never stored)

  const double *ur[RY];
  const double *fr[RY];
  double *dr[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    const int row = min(rw + r, box.e1);  // clamped rows re-read the upper halo row
    ur[r] = u + lu.origin + xs + lu.s1 * row;
    fr[r] = rhs + lf.origin + xs + lf.s1 * row;
    dr[r] = MODE == ZM_RESNORM ? dst : dst + ld.origin + xs + ld.s1 * row;
  }
  const double *uhm = u + lu.origin + xs + lu.s1 * (rw - 1);
  const double *uhp = u + lu.origin + xs + lu.s1 * min(rw + RY, box.e1);

  d2 um[RY], uc[RY];                       // planes m - 1 and m of the own rows
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    um[r] = load2(ur[r] + lu.s2 * (mb - 1));
    uc[r] = load2(ur[r] + lu.s2 * mb);
  }
  // one pipeline stage = everything step m needs from memory
  struct Stage {
    d2 up[RY], f[RY], hm, hp;
    double el[RY], er[RY];
  };
  auto load_stage = [&](Stage &st, int m) {
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      st.up[r] = load2(ur[r] + lu.s2 * (m + 1));
      if (MODE != EXAMG_APPLY) st.f[r] = load2(fr[r] + lf.s2 * m);
      if (!COL) {   // half sweeps: the prefetched edge values cost more (registers) than they hide -- 0.70 against 0.64 ms at 512^3
        st.el[r] = 0.0;
        st.er[r] = 0.0;
        if (lload) st.el[r] = ur[r][lu.s2 * m - 1];
        if (rload) st.er[r] = ur[r][lu.s2 * m + 2];
      }
    }
    st.hm = load2(uhm + lu.s2 * m);
    st.hp = load2(uhp + lu.s2 * m);
  };
  auto compute = [&](const Stage &st, int m) {
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      // wavefront-level x-halo exchange
      double xl = lane_below(uc[r].y);
      double xr = lane_above(uc[r].x);
      if (COL) {
        if (lload) xl = ur[r][lu.s2 * m - 1];
        if (rload) xr = ur[r][lu.s2 * m + 2];
      } else {
        if (lload) xl = st.el[r];
        if (rload) xr = st.er[r];
      }
      const d2 tm_ = (r == 0) ? st.hm : uc[r == 0 ? 0 : r - 1];
      const d2 tp_ = (r == RY - 1) ? st.hp : uc[r == RY - 1 ? r : r + 1];
      d2 o;
      if (COL) {
        // the point of this lane's pair that carries the colour: a if (x + row + m) has that parity, else b; x = xo +
        // 128*tx + 2*lane, so the parity is the same in every lane: scalar branch, one convolution, one lane exchange
        o = uc[r];
        if (((g.xo + rw + r + m) & 1) == g.colour) {
          const double acc = conv7<ORDER>(k, uc[r].x, xl, uc[r].y, tm_.x, tp_.x, um[r].x, st.up[r].x);
          o.x = finish<MODE>(uc[r].x, acc, st.f[r].x, w);
        } else {
          const double acc = conv7<ORDER>(k, uc[r].y, uc[r].x, xr, tm_.y, tp_.y, um[r].y, st.up[r].y);
          o.y = finish<MODE>(uc[r].y, acc, st.f[r].y, w);
        }
      } else {
        const double acc_a = conv7<ORDER>(k, uc[r].x, xl, uc[r].y, tm_.x, tp_.x, um[r].x, st.up[r].x);
        const double acc_b = conv7<ORDER>(k, uc[r].y, uc[r].x, xr, tm_.y, tp_.y, um[r].y, st.up[r].y);
        o.x = finish<MODE>(uc[r].x, acc_a, st.f[r].x, w);
        o.y = finish<MODE>(uc[r].y, acc_b, st.f[r].y, w);
      }
      if (MODE == ZM_RESNORM) {
        if (rw + r < box.e1) {
          if (va) ssum = ssum + o.x * o.x;
          if (vb) ssum = ssum + o.y * o.y;
        }
      } else if (rw + r < box.e1) {
        double *q = dr[r] + ld.s2 * m;
        bool cached = g.store == 2;
        if (g.store == 1) {
          // byte offset of the window's first point within its 128-byte line: wave-uniform (row, plane, tile) -> scalar ALU
          const unsigned a0 = (unsigned)(((unsigned long long)(ld.origin + g.xo + tx * 128 + ld.s1 * (long long)(rwu + r) + ld.s2 * (long long)m) * 8ull + dst_lo) & 127ull);
          const int nh = (int)(((128u - a0) & 127u) + 15u) >> 4;   // lanes that start inside the partial head line
          const int nt = (int)(a0 + 15u) >> 4;                      // lanes that end inside the partial tail line
          cached = lane < nh || lane >= 64 - nt;
        }
        if (va && vb) {
          if (cached) store2(q, o);
          else store2_nt(q, o);
        }
        else if (va) q[0] = o.x;
        else if (vb) q[1] = o.y;
      }
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      um[r] = uc[r];
      uc[r] = st.up[r];
    }
  };
  // software pipeline of depth 1: the loads of step m+1 are in flight while step m is computed
  Stage st[2];
  load_stage(st[0], mb);
  int m = mb;
  while (m < me) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (m < me) {
        if (m + 1 < me) load_stage(st[j ^ 1], m + 1);
        compute(st[j], m);
        ++m;
      }
    }
  }
  if (MODE == ZM_RESNORM) {
    ssum = zm_wave_sum(ssum);
    if (lane == 0) dst[blockIdx.x * WY + wv] = ssum;
  }
}

// ---------------------------------------------------------------------------------------------
// 3-D 7-point constant-coefficient ROW-marching kernel: a wave owns WHOLE rows (up to 512 points: NSEG = 4 segments of 128, two
// points per lane and segment), RY rows of them, and marches in z like the kernel above.  What that buys on the verbatim layout
// (515-double rows, whose odd strides put every row and plane at another offset within its 128-byte line -- the 128-point
// windows of the kernel above start and end in the middle of a line that the neighbouring window also writes):
//   * every output row is in ONE wave, so it can be stored as whole, 128-byte-aligned lines: the outputs pass through a
//     wave-private LDS strip (ds_write_b128, then ds_read2_b64 at the row's offset to the next line boundary -- no barrier,
//     LDS operations of one wave execute in order) and leave as 16-byte-aligned non-temporal stores of full lines; only the
//     two ends of a row are partial lines.  Measured in tools/lab/stencil_lab.hip (stores moved to line boundaries, timing only):
//     0.618 -> 0.590 ms at 512^3, the difference between the verbatim and the padded layout;
//   * no window edges inside a row: x-neighbours across segments come from the neighbouring lane by a wave rotate (DPP
//     wave_rol / wave_ror), edge loads only at the two ends of the row; no partial lines are fetched twice.
// Loads per point are those of the kernel above (own rows once, two y-halo rows per wave through L2).  256 VGPRs (+ ~46 spilled to
// AGPRs) with two rows per wave: one wave per SIMD, the software pipeline (the `u` planes and halo rows step m+1 needs are in flight
// during step m; the right-hand side of plane m, which only the last instructions of step m read, is loaded at the start of that
// step, ahead of the next stage's loads -- loads return in order) hides the latency instead of occupancy.  Arithmetic: conv7 /
// finish as above -- bit-identical.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double lane_rotl(double v) {   // lane l receives lane l+1, lane 63 receives lane 0
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x134, 0xF, 0xF, false);   // wave_rol:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x134, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_rotr(double v) {   // lane l receives lane l-1, lane 0 receives lane 63
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x13C, 0xF, 0xF, false);   // wave_ror:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x13C, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

struct RowMarchGeom {
  int nty, ntz;   // tiles of WY * RY rows, chunks of zc planes
  int zc;
  int remap;      // 2: bands of y-adjacent tiles per XCD within every z layer
};

// NSEG segments of 128 points per row: 4 for rows of 400 .. 512 points (level 9), 2 for rows of 144 .. 256 points (level 8: the 256^3
// block of configs[1], and every block of a strong-scaled 512^3) -- there with RY = 4 rows per wave in the registers the shorter rows
// leave (two halo rows per four own rows instead of per two)
template <int MODE, int ORDER, int RY, int WY, int NSEG>
__global__ void __launch_bounds__(64 * WY)
k_stencil7_rowmarch(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs, LayoutDev ld,
                    double *__restrict__ dst, Coef7 k, double w, Box box, RowMarchGeom g) {
  constexpr int RM_NSEG = NSEG;
  constexpr int RM_STRIP = RM_NSEG * 128 + 16;   // doubles of LDS per wave and row
  __shared__ __attribute__((aligned(16))) double strip[WY][RY][RM_STRIP];
  const int lane = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
  int t = blockIdx.x;
  if (g.remap == 2) {
    const int lz = t / g.nty, r = t - lz * g.nty;
    const int per = g.nty >> 3;
    t = lz * g.nty + (r < (per << 3) ? (r & 7) * per + (r >> 3) : r);
  }
  const int ty = t % g.nty;
  const int tz = t / g.nty;
  const int rw = box.b1 + (ty * WY + wv) * RY;   // first row of this wave
  if (rw >= box.e1) return;                      // wave-uniform
  const int mb = box.b2 + tz * g.zc;
  const int me = min(mb + g.zc, box.e2);
  const int n0 = box.n0();
  const int x0 = box.b0 + 2 * lane;              // the lane's first point in segment 0; segment j: + 128 j

  bool va[RM_NSEG], vb[RM_NSEG];
  int xs[RM_NSEG];                               // column the lane loads from (idle lanes: a safe one, never stored)
#pragma unroll
  for (int j = 0; j < RM_NSEG; ++j) {
    const int x = x0 + 128 * j;
    va[j] = x < box.e0;
    vb[j] = x + 1 < box.e0;
    xs[j] = va[j] ? x : box.b0;
  }
  // the last pair of the row whose second point is inside the box needs u[x + 2] from memory when the next pair is outside
  const bool has_r = (n0 & 1) == 0;
  const int qr = (n0 - 2) >> 1, jr = qr >> 6, lr = qr & 63;

  const double *ur[RY];
  const double *fr[RY];
  double *dr[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    const int row = min(rw + r, box.e1);   // clamped rows re-read the upper halo row (never stored)
    ur[r] = u + lu.origin + lu.s1 * row;
    fr[r] = rhs + lf.origin + lf.s1 * row;
    dr[r] = dst + ld.origin + ld.s1 * row;
  }
  const double *uhm = u + lu.origin + lu.s1 * (rw - 1);
  const double *uhp = u + lu.origin + lu.s1 * min(rw + RY, box.e1);

  d2 um[RY][RM_NSEG], uc[RY][RM_NSEG];
#pragma unroll
  for (int r = 0; r < RY; ++r)
#pragma unroll
    for (int j = 0; j < RM_NSEG; ++j) {
      um[r][j] = load2(ur[r] + xs[j] + lu.s2 * (mb - 1));
      uc[r][j] = load2(ur[r] + xs[j] + lu.s2 * mb);
    }
  // one pipeline stage = what step m needs of u; the right-hand side of a plane is needed by the last instructions of its step only:
  // its loads are issued at the start of that step, BEFORE the next stage's loads (loads return in order), not a step ahead
  struct Stage {
    d2 up[RY][RM_NSEG], hm[RM_NSEG], hp[RM_NSEG];
    double el[RY], er[RY];
  };
  struct Rhs {
    d2 f[RY][RM_NSEG];
  };
  auto load_rhs = [&](Rhs &fv, int m) {
    if (MODE == EXAMG_APPLY) return;
#pragma unroll
    for (int r = 0; r < RY; ++r)
#pragma unroll
      for (int j = 0; j < RM_NSEG; ++j) fv.f[r][j] = load2(fr[r] + xs[j] + lf.s2 * m);
  };
  auto load_stage = [&](Stage &st, int m) {
#pragma unroll
    for (int r = 0; r < RY; ++r) {
#pragma unroll
      for (int j = 0; j < RM_NSEG; ++j) {
        st.up[r][j] = load2(ur[r] + xs[j] + lu.s2 * (m + 1));
      }
      st.el[r] = 0.0;
      st.er[r] = 0.0;
      if (lane == 0) st.el[r] = ur[r][box.b0 - 1 + lu.s2 * m];
      if (has_r && lane == lr) st.er[r] = ur[r][box.b0 + 2 * qr + 2 + lu.s2 * m];
    }
#pragma unroll
    for (int j = 0; j < RM_NSEG; ++j) {
      st.hm[j] = load2(uhm + xs[j] + lu.s2 * m);
      st.hp[j] = load2(uhp + xs[j] + lu.s2 * m);
    }
  };
  auto compute = [&](const Stage &st, const Rhs &fv, int m) {
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      d2 o[RM_NSEG];
#pragma unroll
      for (int j = 0; j < RM_NSEG; ++j) {
        // x-neighbours: the adjacent lane; across segments the wave rotates (lane 63 of segment j <-> lane 0 of segment j + 1)
        double xl = lane_below(uc[r][j].y);
        double xr = lane_above(uc[r][j].x);
        if (j > 0) {
          const double t0 = lane_rotr(uc[r][j - 1].y);
          if (lane == 0) xl = t0;
        } else if (lane == 0) {
          xl = st.el[r];
        }
        if (j < RM_NSEG - 1) {
          const double t1 = lane_rotl(uc[r][j + 1].x);
          if (lane == 63) xr = t1;
        }
        if (has_r && j == jr && lane == lr) xr = st.er[r];
        const d2 tm_ = (r == 0) ? st.hm[j] : uc[r == 0 ? 0 : r - 1][j];
        const d2 tp_ = (r == RY - 1) ? st.hp[j] : uc[r == RY - 1 ? r : r + 1][j];
        const double acc_a = conv7<ORDER>(k, uc[r][j].x, xl, uc[r][j].y, tm_.x, tp_.x, um[r][j].x, st.up[r][j].x);
        const double acc_b = conv7<ORDER>(k, uc[r][j].y, uc[r][j].x, xr, tm_.y, tp_.y, um[r][j].y, st.up[r][j].y);
        o[j].x = finish<MODE>(uc[r][j].x, acc_a, MODE == EXAMG_APPLY ? 0.0 : fv.f[r][j].x, w);
        o[j].y = finish<MODE>(uc[r][j].y, acc_b, MODE == EXAMG_APPLY ? 0.0 : fv.f[r][j].y, w);
      }
      if (rw + r < box.e1) {
        double *row = dr[r] + box.b0 + ld.s2 * m;                       // &dst[x = b0] of this row and plane
        // doubles from the row's first point to the next 128-byte boundary: wave-uniform
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)row);
        const int hb = (int)(((128u - (lo & 127u)) & 127u) >> 3);
        double *sb = strip[wv][r];
#pragma unroll
        for (int j = 0; j < RM_NSEG; ++j) *reinterpret_cast<d2 *>(sb + 128 * j + 2 * lane) = o[j];
        // head of the row, up to the first line boundary: straight from the registers of segment 0 (a partial line)
        if (2 * lane + 1 < hb) store2_nt(row + 2 * lane, o[0]);
        else if (2 * lane < hb) row[2 * lane] = o[0].x;
#pragma unroll
        for (int j = 0; j < RM_NSEG; ++j) {
          const int p = hb + 128 * j + 2 * lane;                        // first of the lane's two points, relative to b0
          d2 v;
          v.x = sb[p];
          v.y = sb[p + 1];
          // 16-byte aligned: whole lines per wave.  Rows have more than 128 (NSEG - 1) + 15 points (dispatch): only the last segment can end early
          if (j < RM_NSEG - 1 || p + 1 < n0) __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(row + p));
          else if (p < n0) row[p] = v.x;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RY; ++r)
#pragma unroll
      for (int j = 0; j < RM_NSEG; ++j) {
        um[r][j] = uc[r][j];
        uc[r][j] = st.up[r][j];
      }
  };
  Stage st[2];
  Rhs fv;
  load_stage(st[0], mb);
  int m = mb;
  while (m < me) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (m < me) {
        load_rhs(fv, m);
        if (m + 1 < me) load_stage(st[j ^ 1], m + 1);
        compute(st[j], fv, m);
        ++m;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Red-black half sweep on the COLOUR-SPLIT layout (EXAMG_LAYOUT_SPLIT_X: `transform Solution with [x, y, z] => [x / 2, y, z, x % 2]`,
// Testing/LayoutTrafo/rbgs.exa4:2), 3-D 7-point constant stencil, in place.
// The points of one colour of a row are the even OR the odd columns: one of the two half rows, contiguous -- and so are their x
// neighbours (the other half row) and the values written.  A lane owns one column pair h (columns 2h, 2h + 1) of a row and marches in z
// with both halves of its pair for the planes z-1, z, z+1 in registers; the half that is updated alternates from plane to plane.  Per
// update: the pair of plane z+1 (16 B, streamed once), the two y neighbours of the updated half (L2: they are the pairs of the rows
// above and below, loaded a plane earlier by their waves), the right-hand side and the store (8 B each): 32 B instead of the 48 B a
// half sweep moves in the untransformed layout.  The x neighbour outside the pair comes from the adjacent lane (DPP wave shift).
// Arithmetic: conv7 / finish<EXAMG_SMOOTH> -- the loop's own expression, bit-identical to the generic kernel on either layout.
// ---------------------------------------------------------------------------------------------
struct SplitGeom {
  int hb, nh;        // first column pair of the box, number of column pairs (both parities)
  int ntx, nty, zc;  // windows of 64 pairs, tiles of SP_WY rows, planes per chunk
  int ax0, ax1;      // array x range of the box
};
constexpr int SP_WY = 4;

template <int ORDER>
__global__ void __launch_bounds__(64 * SP_WY)
k_rbgs_half_split7(LayoutDev lu, double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs, Coef7 k, double w, Box box, int colour,
                   SplitGeom g) {
  const int lane = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
  // workgroups are dealt round-robin to the 8 XCDs: within a z layer every XCD takes a band of y-adjacent row groups (all x windows of
  // them), so that the y neighbours a wave reads -- the pairs other waves loaded two planes earlier -- meet in ONE L2
  const int per_layer = g.ntx * g.nty;
  const int tz = blockIdx.x / per_layer;
  int r = blockIdx.x - tz * per_layer;
  if ((g.nty & 7) == 0) {
    const int band = r & 7, q = r >> 3;          // XCD, position within the XCD's share of the layer
    r = (band * (g.nty >> 3) + q / g.ntx) * g.ntx + q % g.ntx;
  }
  const int tx = r % g.ntx, ty = r / g.ntx;
  const int y = box.b1 + ty * SP_WY + wv;
  if (y >= box.e1) return;                       // wave-uniform
  const int mb = box.b2 + tz * g.zc, me = min(mb + g.zc, box.e2);
  const int h = g.hb + 64 * tx + lane;           // this lane's column pair
  const int hc = min(h, (int)lu.s1 - 1);         // clamped for loads (lanes beyond the row are never stored)
  double *E = u, *O = u + lu.half;
  const long long rowy = lu.s1 * (y + lu.ref1);
  auto at = [&](int z) { return rowy + lu.s2 * (z + lu.ref2) + hc; };
  double Em = E[at(mb - 1)], Om = O[at(mb - 1)], Ec = E[at(mb)], Oc = O[at(mb)];
  double Ep = E[at(mb + 1)], Op = O[at(mb + 1)];
  const bool first_lane = lane == 0, last_lane = lane == 63;
  for (int z = mb; z < me; ++z) {
    // parity of the array column of this row's updated points: (i0 + y + z) % 2 == colour with i0 = ax - ref0
    const int p = (colour + y + z + lu.ref0) & 1;          // wave-uniform
    const long long a = at(z);
    double *P = p ? O : E;
    const double *Q = p ? E : O;
    const double c = p ? Oc : Ec, q = p ? Ec : Oc, zm = p ? Om : Em, zp = p ? Op : Ep;
    const double ym = P[a - lu.s1], yp = P[a + lu.s1];
    // next plane's pair (in flight during this step)
    const long long an = at(min(z + 2, box.e2));          // (the plane beyond the box's shell is never used)
    const double En = E[an], On = O[an];
    // x neighbours: even column 2h: columns 2h - 1 = O[h - 1], 2h + 1 = O[h]; odd column 2h + 1: columns 2h = E[h], 2h + 2 = E[h + 1]
    double xm, xp;
    if (p == 0) {
      xm = lane_below(q);
      if (first_lane) xm = Q[a - 1];
      xp = q;
    } else {
      xm = q;
      xp = lane_above(q);
      if (last_lane) xp = Q[min(a + 1, rowy + lu.s2 * (z + lu.ref2) + lu.s1 - 1)];
    }
    const int ax = 2 * h + p;
    if (ax >= g.ax0 && ax < g.ax1) {
      const double f = rhs[lidx(lf, ax - lu.ref0, y, z)];
      const double acc = conv7<ORDER>(k, c, xm, xp, ym, yp, zm, zp);
      const double v = finish<EXAMG_SMOOTH>(c, acc, f, w);
      P[a] = v;                                   // (no later step of this sweep reads a point of the colour it updates)
    }
    Em = Ec; Om = Oc;
    Ec = Ep; Oc = Op;
    Ep = En; Op = On;
  }
}

static thread_local int g_force_generic = 0;  // test hook (debug build only): examg_debug_force_generic
static thread_local int g_passthru = 0;       // stencil_colour_passthrough (below): coloured loop out of place, the other colour carried over
// workgroup cap of the unrolled stencil-field kernel: none.  One short-lived workgroup per 256 points, dispatched in order, keeps
// the front that sweeps the 30 streams (27 coefficient planes, u, rhs, dst) narrow: 512^3, 27 entries: 7.8 ms with 16384
// grid-striding workgroups, 6.85 ms uncapped (round-2 sweep through examg_debug_sf27_blocks); a tiled form with XCD-contiguous order: 7.7 ms.
static thread_local int g_sf27_blocks = 1 << 30;
static thread_local int g_sf27_unrolled = 1;  // examg_debug_sf27(0): 27-entry stencil fields on the generic kernel

// kernels_stencilfield.hip
bool stencilfield7_ok(const examg_layout_t *lu, const examg_stencil_t *st, const Box &box, int colour);
int launch_stencilfield7(int mode, const LayoutDev &lu, const double *u, const LayoutDev &lf, const double *rhs, const LayoutDev &ld,
                         double *dst, const LayoutDev &lc, const double *cf, double w, const Box &box, hipStream_t s);

static thread_local int g_rm_on = -1, g_rm_zc = -1, g_rm_remap = -1;   // examg_debug_rowmarch (debug build): row-marching kernel off / on, planes per chunk, order
static thread_local int g_rm_ry2 = -1;                                  // ... rows per wave of the two-segment form (on >= 10: on - 10 rows)
#ifndef EXAMG_RM_RY
#define EXAMG_RM_RY 2
#define EXAMG_RM_WY 4
#endif
constexpr int RM_RY = EXAMG_RM_RY, RM_WY = EXAMG_RM_WY;

// the row-marching kernel takes boxes whose rows fit one wave -- 400 .. 512 points (level 9: four segments) or 144 .. 256 points (level 8:
// two segments) -- and that are long enough in y and z; returns the segments per row, 0: not taken
template <int MODE>
static int rowmarch_wanted(const Box &box, int colour) {
  if (MODE == ZM_RESNORM || colour >= 0) return 0;
  if (g_rm_on == 0) return 0;
  const int nseg = (box.n0() >= 400 && box.n0() <= 512) ? 4 : ((box.n0() >= 144 && box.n0() <= 256) ? 2 : 0);
  if (g_rm_on == 1) return nseg;
  // rows of 144 .. 256 points (256^3 blocks): measured, no gain -- 0.0715 ms (two rows per wave, 16-plane chunks) / 0.0731 (four rows)
  // against 0.0713 for the window kernel (tools/sweep_rowmarch.py 256, profiles/NOTES.md): the two-segment form stays a debug variant
  return (nseg == 4 && box.n1() >= 64 && box.n2() >= 16) ? nseg : 0;
}
// padded layouts (even strides) keep the window kernel: it starts its windows on a 16-byte boundary there and every access is
// aligned already (512^3, 544-double rows: 0.570 ms against 0.589 for the row-marching kernel)
static bool rowmarch_layout(const LayoutDev &lu, const LayoutDev &lf, const LayoutDev &ld) {
  const bool even = !(lu.s1 & 1) && !(lu.s2 & 1) && !(lf.s1 & 1) && !(lf.s2 & 1) && !(ld.s1 & 1) && !(ld.s2 & 1);
  return g_rm_on == 1 || !even;
}

template <int MODE, int ORDER>
static int launch_rowmarch(int nseg, const LayoutDev &lu, const double *u, const LayoutDev &lf, const double *rhs, const LayoutDev &ld, double *dst,
                           const Coef7 &k, double w, const Box &box, hipStream_t s) {
  RowMarchGeom g;
  const int ry = nseg == 4 ? RM_RY : (g_rm_ry2 > 0 ? g_rm_ry2 : 4);
  g.nty = (box.n1() + ry * RM_WY - 1) / (ry * RM_WY);
  // 512^3 (64 tiles of 8 rows per layer): 64-plane chunks = 512 workgroups, two even rounds over the 256 CUs (one workgroup per CU at 256
  // VGPRs): 0.565 / 0.578 ms on a fast / slow box; 32 or 128 planes +0.5 %, chunk lengths that leave a ragged last round (48, 86) +4..15 %.
  // 256^3 (16 tiles of 16 rows per layer): 16-plane chunks = 256 workgroups, one round
  int zc = g_rm_zc > 0 ? g_rm_zc : (nseg == 4 ? 64 : 16);
  if (zc > box.n2()) zc = box.n2();
  g.zc = zc;
  g.ntz = (box.n2() + zc - 1) / zc;
  g.remap = g_rm_remap >= 0 ? g_rm_remap : 2;
  dim3 block(64, RM_WY, 1), grid(g.nty * g.ntz, 1, 1);
  if (nseg == 4) hipLaunchKernelGGL((k_stencil7_rowmarch<MODE, ORDER, RM_RY, RM_WY, 4>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, k, w, box, g);
#ifdef EXAMG_DEBUG_HOOKS
  else if (ry == 2) hipLaunchKernelGGL((k_stencil7_rowmarch<MODE, ORDER, 2, RM_WY, 2>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, k, w, box, g);
  else hipLaunchKernelGGL((k_stencil7_rowmarch<MODE, ORDER, 4, RM_WY, 2>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, k, w, box, g);
#else
  else { (void)ry; return -1; }
#endif
  return (int)grid.x * RM_WY;
}

constexpr int ZM_RY = 2, ZM_WY = 4, ZM_BLOCKS = 512, ZM_BLOCKS_COL = 1024, ZM_MINCHUNK = 16;   // half sweeps: 0.68 ms at 512 workgroups, 0.64 at 1024
static thread_local int g_zm_blocks = -1, g_zm_minchunk = -1, g_zm_remap = -1, g_zm_store = -1;    // examg_debug_zmarch (debug build): workgroup target, planes per chunk, tile order

// returns the number of waves of the grid (ZM_RESNORM: the number of partial sums written to dst)
template <int MODE, int ORDER>
static int launch_zmarch(const LayoutDev &lu, const double *u, const LayoutDev &lf, const double *rhs, const LayoutDev &ld,
                         double *dst, const Coef7 &k, double w, const Box &box, hipStream_t s, int colour = -1, int max_waves = 0) {
  if (const int nseg = rowmarch_wanted<MODE>(box, colour); nseg && rowmarch_layout(lu, lf, ld)) {
    if (MODE != ZM_RESNORM) return launch_rowmarch<MODE == ZM_RESNORM ? EXAMG_RESIDUAL : MODE, ORDER>(nseg, lu, u, lf, rhs, ld, dst, k, w, box, s);
  }
  ZMarchGeom g;
  // Padded layouts (`align`, field/ir/IR_AddPaddingToFieldLayouts.scala:36-41) have even row lengths and put the lower duplicate
  // point on an even index: starting the windows one point to the left of an odd box makes every 16-byte load and store of
  // the sweep 16-byte aligned (the extra point is loaded, never stored).  Not possible in the verbatim layout, whose odd
  // strides flip the parity from row to row.  512^3, rows of 544 doubles: 0.62 -> 0.57 ms (tools/lab/stencil_lab.hip).
  const bool even = !(lu.s1 & 1) && !(lu.s2 & 1) && !(lf.s1 & 1) && !(lf.s2 & 1) && !(ld.s1 & 1) && !(ld.s2 & 1);
  const int pu = (int)((lu.origin + box.b0) & 1), pf = (int)((lf.origin + box.b0) & 1), pd = (int)((ld.origin + box.b0) & 1);
  const bool shift = even && pu == 1 && pd == 1 && (MODE == EXAMG_APPLY || pf == 1) && lu.ref0 + box.b0 >= 1 &&
                     ld.ref0 + box.b0 >= 1 && (MODE == EXAMG_APPLY || lf.ref0 + box.b0 >= 1);
  g.xo = box.b0 - (shift ? 1 : 0);
  g.ntx = (box.e0 - g.xo + 127) / 128;
  g.nty = (box.n1() + ZM_RY * ZM_WY - 1) / (ZM_RY * ZM_WY);
  const int xy = g.ntx * g.nty;
  // Workgroup count = chunk length in z, and the order of the tiles (tools/sweep_zmarch_order.py, sweep_big_others.py; all
  // bit-identical).  Below 5*10^7 points: ~512 long-lived workgroups (1024 for the half sweeps), plain order.  From there: short
  // chunks of 8 planes, as many workgroups as that takes, dealt to the XCDs in bands of y-adjacent tiles within every z layer
  // (order 2 of kernels_twostage.hip: halo rows meet in one L2, all XCDs stay in the same planes) -- with ~512 workgroups every
  // chunk of a large block is a whole column and the front that sweeps memory as wide as the block:
  //   Jacobi step   384^3 0.299 -> 0.261 ms, 512^3 0.586 -> 0.580, 576^3 1.04 -> 0.90, 768^3 2.59 -> 2.10, 1024^3 5.86 -> 4.83 (0.665)
  //   half sweep    384^3 0.335 -> 0.260,    512^3 0.645 -> 0.596, 576^3 1.06 -> 0.87, 768^3 2.74 -> 1.98, 1024^3 5.48 -> 4.70
  // The residual + norm mode writes one partial sum per wave into a fixed work buffer and keeps few workgroups (4096 from 2*10^8
  // points: 768^3 1.69 -> 1.53 ms).
  int target = colour >= 0 ? ZM_BLOCKS_COL : ZM_BLOCKS, minchunk = ZM_MINCHUNK;
  g.remap = 0;
  if (MODE == ZM_RESNORM) {
    if (box.count() >= 200000000LL) target = 4096;
  } else if (box.count() >= (even ? 150000000LL : 50000000LL)) {   // padded rows at 512^3: 0.563 ms with the long chunks, 0.585 with short ones
    target = 1 << 24;
    minchunk = box.count() >= 200000000LL ? 4 : 8;      // 4 .. 8 planes are within 2 % of each other; 768^3: 2.11 -> 2.02 ms with 4
    g.remap = 2;
  }
  if (g_zm_blocks > 0) target = g_zm_blocks;
  if (g_zm_minchunk > 0) minchunk = g_zm_minchunk;
  if (g_zm_remap >= 0) g.remap = g_zm_remap;
  int ntz = (target + xy - 1) / xy;
  if (ntz < 1) ntz = 1;
  int zc = (box.n2() + ntz - 1) / ntz;
  if (zc < minchunk) zc = minchunk;
  if (zc > box.n2()) zc = box.n2();
  g.zc = zc;
  g.ntz = (box.n2() + zc - 1) / zc;
  g.colour = colour;
  // stores: non-temporal where the loop streams, plain where input, output and right-hand side fit the 256 MB Infinity Cache together and
  // the next loop over the level reads what this one wrote (kernels_twostage.hip has the measurements for the two-stage passes)
  g.store = g_zm_store >= 0 ? g_zm_store : (box.count() * 24LL > 200000000LL ? 0 : 2);
  dim3 block(64, ZM_WY, 1), grid(g.ntx * g.nty * g.ntz, 1, 1);
  if (max_waves > 0 && (long long)grid.x * ZM_WY > max_waves) return -1;     // nothing launched
  if (colour >= 0) {
    if (MODE == EXAMG_SMOOTH)
      hipLaunchKernelGGL((k_stencil7_zmarch<EXAMG_SMOOTH, ORDER, ZM_RY, ZM_WY, true>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, k, w, box, g);
    return (int)grid.x * ZM_WY;
  }
  hipLaunchKernelGGL((k_stencil7_zmarch<MODE, ORDER, ZM_RY, ZM_WY, false>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, k, w, box, g);
  return (int)grid.x * ZM_WY;
}

// kernels_blas.hip
int launch_reduce_sum(const double *part, int n, double *result, hipStream_t s);
size_t reduce_work_doubles();

}  // namespace examg

using namespace examg;

#ifdef EXAMG_DEBUG_HOOKS
// Variant selection for the parity tests (debug build libexamg_dbg.so only; per host thread)
extern "C" int examg_debug_zmarch(int blocks, int minchunk, int remap) {
  g_zm_blocks = blocks;
  g_zm_minchunk = minchunk;
  g_zm_remap = remap;
  return 0;
}
extern "C" int examg_debug_rowmarch(int on, int zc, int remap) {
  g_rm_ry2 = -1;
  if (on >= 10) { g_rm_ry2 = on - 10; on = 1; }      // 12 / 14: the two-segment form with 2 / 4 rows per wave
  g_rm_on = on;
  g_rm_zc = zc;
  g_rm_remap = remap;
  return 0;
}
extern "C" int examg_debug_zmarch_store(int mode) {
  g_zm_store = mode;
  return 0;
}
extern "C" int examg_debug_sf27(int unrolled) {
  g_sf27_unrolled = unrolled;
  return 0;
}
extern "C" int examg_debug_sf27_run(int run) {
  g_sf27_run = run;
  return 0;
}
extern "C" int examg_debug_sf27_blocks(int blocks) {
  g_sf27_blocks = blocks > 0 ? blocks : (1 << 30);
  return 0;
}

extern "C" int examg_debug_force_generic(int on) {
  const int old = g_force_generic;
  g_force_generic = on;
  return old;
}
#endif

namespace examg {
// `dst = u` on the box followed by the coloured loop from u into dst (the shell passes of examg_rbgs_sweep_blocks), as ONE launch: the
// generic kernel writes the colour's points and carries the others over.  Same values as the two launches.
int stencil_colour_passthrough(const examg_layout_t *lu, const double *u, const examg_layout_t *lf, const double *rhs, const examg_layout_t *ld,
                               double *dst, const examg_stencil_t *st, double w, int colour, const int32_t *begin, const int32_t *end, hipStream_t s) {
  g_passthru = 1;
  const int rc = examg_stencil_op(EXAMG_SMOOTH, lu, u, lf, rhs, ld, dst, st, w, colour, begin, end, (examg_stream_t)s);
  g_passthru = 0;
  return rc;
}
}  // namespace examg

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

  // a field under a layout transformation (colour split): the generic kernel, which forms every address through the transformed
  // index; the coefficient field of a stencil field has its own transformation (ctransform) and a plain clayout
  const bool transformed = lay_split(lu_) || lay_split(ld_) || (lf_ && lay_split(lf_));
  if (st->cfield && lay_split(&st->clayout)) { set_error("examg_stencil_op: the coefficient layout of a stencil field cannot be colour-split"); return 1; }
  const int passthru = (g_passthru && colour >= 0 && u != dst) ? 1 : 0;
  const bool force_generic = g_force_generic || transformed || passthru;
  const int ord = canonical_order7(st);
  if (transformed && !g_force_generic && mode == EXAMG_SMOOTH && colour >= 0 && u == dst && lay_split(lu_) && memcmp(lu_, ld_, sizeof(*lu_)) == 0 &&
      lu_->nd == 3 && ord >= 0 && box.n0() >= 32 && box_inside(lu_, box, 1)) {
    // red-black half sweep on the colour-split layout: 32 B per update (k_rbgs_half_split7); the right-hand side on either layout
    Coef7 k;
    for (int i = 0; i < 7; ++i) k.c[i] = st->coef[i];
    SplitGeom g;
    g.ax0 = box.b0 + lu.ref0;
    g.ax1 = box.e0 + lu.ref0;
    g.hb = g.ax0 >> 1;
    g.nh = ((g.ax1 - 1) >> 1) - g.hb + 1;
    g.ntx = (g.nh + 63) / 64;
    g.nty = (box.n1() + SP_WY - 1) / SP_WY;
    // planes per chunk: enough workgroups for the chip (>= 2048), chunks of at least 16 planes (two planes of start-up each)
    int zc = box.n2();
    while (zc > 16 && (long long)g.ntx * g.nty * ((box.n2() + zc - 1) / zc) < 2048) zc = (zc + 1) / 2;
    g.zc = zc;
    const int ntz = (box.n2() + zc - 1) / zc;
    dim3 block(64, SP_WY, 1), grid((unsigned)(g.ntx * g.nty * ntz), 1, 1);
    if (ord == 0) hipLaunchKernelGGL((k_rbgs_half_split7<0>), grid, block, 0, s, lu, dst, lf, rhs, k, w, box, colour, g);
    else hipLaunchKernelGGL((k_rbgs_half_split7<1>), grid, block, 0, s, lu, dst, lf, rhs, k, w, box, colour, g);
    EXAMG_CHECK_LAUNCH("k_rbgs_half_split7");
    return 0;
  }
  const bool colour_ok = colour < 0 || (mode == EXAMG_SMOOTH && u == dst && memcmp(lu_, ld_, sizeof(*lu_)) == 0);
  if (!force_generic && lu_->nd == 3 && ord >= 0 && colour_ok && box.n0() >= 64) {
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

  if (!force_generic && stencilfield7_ok(lu_, st, box, colour)) {
    const LayoutDev lcf = make_layout(&st->clayout);
    return launch_stencilfield7(mode, lu, u, lf, rhs, ld, dst, lcf, st->cfield, w, box, s);
  }

  if (!force_generic && g_sf27_unrolled && st->cfield && st->nent == 27 && colour < 0 && st->diag == 0 && st->off[0][0] == 0 &&
      st->off[0][1] == 0 && st->off[0][2] == 0 && st->ctransform == EXAMG_CLAYOUT_ENTRY_FASTEST && lu_->nd == 3 &&
      (mode != EXAMG_SMOOTH || st->wform == EXAMG_WEIGHT_INV_TIMES) &&
      // the 16-byte loads of the record stream are clamped to the allocation; a pair that starts on the very last double of the
      // coefficient array would be shifted -- a box that holds the last allocated point (a coefficient layout without ghost or
      // pad layers) takes the generic kernel instead
      lidx(make_layout(&st->clayout), box.e0 - 1, box.e1 - 1, box.e2 - 1) != make_layout(&st->clayout).size - 1) {
    // transformed coefficient layout: ONE coefficient stream, transposed through LDS (k_stencilfield27_rec)
    const LayoutDev lc27 = make_layout(&st->clayout);
    UOffsets uo;
    for (int k = 0; k < 27; ++k) uo.o[k] = st->off[k][0] + lu.s1 * st->off[k][1] + lu.s2 * st->off[k][2];
    const int tiles_x = (box.n0() + 63) / 64;
    const long long ntiles = (long long)tiles_x * box.n1() * box.n2();
    const int run = g_sf27_run > 0 ? g_sf27_run : 2;   // 512^3: 1 tile 6.06 ms, 2 tiles 5.71, 4 tiles 5.93, 8 tiles 5.95 (tools/sf27_layouts.py)
    const long long nwaves = (ntiles + run - 1) / run;
    dim3 gridr((unsigned)((nwaves + SF27_WAVES - 1) / SF27_WAVES)), blockr(64, SF27_WAVES);
#define EXAMG_SF27R(M) hipLaunchKernelGGL((k_stencilfield27_rec<M>), gridr, blockr, 0, s, lu, u, lf, rhs, ld, dst, lc27, st->cfield, uo, w, box, tiles_x, ntiles, run)
    if (mode == EXAMG_APPLY) EXAMG_SF27R(EXAMG_APPLY);
    else if (mode == EXAMG_RESIDUAL) EXAMG_SF27R(EXAMG_RESIDUAL);
    else EXAMG_SF27R(EXAMG_SMOOTH);
#undef EXAMG_SF27R
    EXAMG_CHECK_LAUNCH("k_stencilfield27_rec");
    return 0;
  }

  if (!force_generic && g_sf27_unrolled && st->cfield && st->nent == 27 && colour < 0 && st->diag == 0 && st->off[0][0] == 0 &&
      st->off[0][1] == 0 && st->off[0][2] == 0 && st->ctransform == EXAMG_CLAYOUT_PLANES &&
      (mode != EXAMG_SMOOTH || st->wform == EXAMG_WEIGHT_INV_TIMES)) {
    const LayoutDev lc27 = make_layout(&st->clayout);
    UOffsets uo;
    for (int k = 0; k < 27; ++k) uo.o[k] = st->off[k][0] + lu.s1 * st->off[k][1] + lu.s2 * st->off[k][2];
    long long nb27 = (box.count() + 255) / 256;
    if (nb27 > g_sf27_blocks) nb27 = g_sf27_blocks;
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
    for (int d = 0; d < 3; ++d) sd.o[k][d] = (signed char)st->off[k][d];
    sd.ro[k] = lu.s1 * st->off[k][1] + lu.s2 * st->off[k][2];
  }
  sd.cfield = st->cfield;
  LayoutDev lc = lu;
  sd.cplane = 0;
  sd.cpt = 1;
  if (st->cfield) {
    lc = make_layout(&st->clayout);
    sd.cplane = lc.size;
    if (st->ctransform == EXAMG_CLAYOUT_ENTRY_FASTEST) { sd.cplane = 1; sd.cpt = st->nent; }
  }
  sd.wdiv = st->wform == EXAMG_WEIGHT_DIVIDE ? 1 : 0;
  const int row_w = colour >= 0 ? (box.n0() + 1) / 2 : box.n0();
  const long long total = (long long)row_w * box.n1() * box.n2();
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  dim3 grid((unsigned)nb), block(256);
  if (mode == EXAMG_APPLY)
    hipLaunchKernelGGL((k_stencil_generic<EXAMG_APPLY>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, lc, sd, w, colour, box, row_w, passthru);
  else if (mode == EXAMG_RESIDUAL)
    hipLaunchKernelGGL((k_stencil_generic<EXAMG_RESIDUAL>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, lc, sd, w, colour, box, row_w, passthru);
  else
    hipLaunchKernelGGL((k_stencil_generic<EXAMG_SMOOTH>), grid, block, 0, s, lu, u, lf, rhs, ld, dst, lc, sd, w, colour, box, row_w, passthru);
  EXAMG_CHECK_LAUNCH("k_stencil_generic");
  return 0;
}

// sum over [begin,end) of (rhs - A u)^2: the residual loop and the reduction loop of the norm as one pass, residual not stored
extern "C" int examg_residual_norm2(const examg_layout_t *lu_, const double *u, const examg_layout_t *lf_, const double *rhs,
                                    const examg_stencil_t *st, const int32_t *begin, const int32_t *end, const examg_layout_t *lr_,
                                    double *res, double *result, void *work, examg_stream_t stream) {
  if (!lu_ || !u || !lf_ || !rhs || !st || !begin || !end || !result || !work) { set_error("examg_residual_norm2: null argument"); return 1; }
  const Box box = make_box(begin, end);
  hipStream_t s = (hipStream_t)stream;
  if (box.count() == 0) return check_hip(hipMemsetAsync(result, 0, sizeof(double), s), "examg_residual_norm2 memset");
  const int ord = canonical_order7(st);
  if (!g_force_generic && !lay_split(lu_) && !lay_split(lf_) && lu_->nd == 3 && ord >= 0 && box.n0() >= 64 && box_inside(lu_, box, 1) && box_inside(lf_, box, 0)) {
    const LayoutDev lu = make_layout(lu_), lf = make_layout(lf_);
    Coef7 k;
    for (int i = 0; i < 7; ++i) k.c[i] = st->coef[i];
    int n;      // one partial sum per wave of the grid; a box with more waves than the work buffer holds takes the two-kernel path
    const int cap = (int)reduce_work_doubles();
    if (ord == 0) n = launch_zmarch<ZM_RESNORM, 0>(lu, u, lf, rhs, lu, (double *)work, k, 0.0, box, s, -1, cap);
    else n = launch_zmarch<ZM_RESNORM, 1>(lu, u, lf, rhs, lu, (double *)work, k, 0.0, box, s, -1, cap);
    if (n > 0) {
      EXAMG_CHECK_LAUNCH("k_stencil7_zmarch (residual norm)");
      return launch_reduce_sum((const double *)work, n, result, s);
    }
  }
  if (!lr_ || !res) { set_error("examg_residual_norm2: this stencil / box needs the residual array"); return 1; }
  int rc = examg_stencil_op(EXAMG_RESIDUAL, lu_, u, lf_, rhs, lr_, res, st, 0.0, -1, begin, end, stream);
  if (rc) return rc;
  return examg_dot(lr_, res, lr_, res, begin, end, result, work, stream);
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
