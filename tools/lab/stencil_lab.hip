// Kernel laboratory for the single-step 7-point sweep (Jacobi / residual) on MI355X: variants and ablations of the z-march
// kernel and an LDS-staged plane kernel, each timed with HIP events and compared bit for bit with a one-thread-per-point
// kernel.  Stand-alone (HIP runtime only):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o gpurun_out/stencil_lab tools/lab/stencil_lab.hip
//   gpurun_out/stencil_lab [n=512] [align=0]
// What wins here is ported into exastencils_amd/csrc/kernels_stencil.hip; results are summarised in DESIGN.md section 4.1.
#include "../exastencils_amd/csrc/examg_common.h"

#include <stdlib.h>
#include <string>
#include <vector>

using namespace examg;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef d2 d2_a8 __attribute__((aligned(8)));

__global__ void k_fill(double *x, long long n, unsigned long long seed) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    x[i] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  }
}

// reference: one thread per point, entries folded in ORDER 0
__global__ void __launch_bounds__(256) k_ref(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs,
                                             double *__restrict__ dst, Coef7 k, double w, Box box) {
  const long long total = box.count();
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int i0 = box.b0 + (int)(t % box.n0());
    const long long r = t / box.n0();
    const int i1 = box.b1 + (int)(r % box.n1()), i2 = box.b2 + (int)(r / box.n1());
    const long long iu = lidx(lu, i0, i1, i2);
    const double acc = conv7<0>(k, u[iu], u[iu - 1], u[iu + 1], u[iu - lu.s1], u[iu + lu.s1], u[iu - lu.s2], u[iu + lu.s2]);
    dst[iu] = u[iu] + w * (rhs[lidx(lf, i0, i1, i2)] - acc);
  }
}

struct Geo {
  int ntx, nty, ntz, zc, nblocks;
  int amask;   // ABL & 16: stores moved down to a multiple of amask + 1 bytes (timing only)
};

// --------------------------------------------------------------------------------------------------------------------
// z-march kernel as shipped in round 1 (kernels_stencil.hip), plus: EPF = edge values prefetched with the stage;
// ABL bits: 1 no y-halo row loads, 2 no rhs loads, 4 no stores, 8 no edge loads
// --------------------------------------------------------------------------------------------------------------------
template <int RY, int WY, int PF, bool EPF, int ABL, bool NTF = false, bool REMAP = false, int WX = 1>
__global__ void __launch_bounds__(64 * WY * WX)
k_zm(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs, double *__restrict__ dst, Coef7 k,
     double w, Box box, Geo g) {
  const int lane = threadIdx.x, wv = threadIdx.y / WX, wx = threadIdx.y % WX;
  int t = blockIdx.x;
  if (REMAP) {   // XCD-contiguous within every z layer of tiles
    const int xy = g.ntx * g.nty;
    const int lz = t / xy, r = t - lz * xy;
    const int per = xy >> 3;
    t = lz * xy + (r < (per << 3) ? (r & 7) * per + (r >> 3) : r);
  }
  const int tx = t % g.ntx;
  t /= g.ntx;
  const int tt = t % g.nty;
  const int tm = t / g.nty;
  const int x = box.b0 + (tx * WX + wx) * 128 + lane * 2;
  const int rw = box.b1 + (tt * WY + wv) * RY;
  const int mb = box.b2 + tm * g.zc;
  const int me = min(mb + g.zc, box.e2);
  if (rw >= box.e1 || x - 2 * lane >= box.e0) return;
  const bool va = x < box.e0, vb = x + 1 < box.e0;
  const bool rload = vb && (lane == 63 || x + 2 >= box.e0);
  const bool lload = va && lane == 0;
  const int xs = va ? x : box.b0;
  const double *ur[RY];
  const double *fr[RY];
  double *dr[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    const int row = min(rw + r, box.e1);
    ur[r] = u + lu.origin + xs + lu.s1 * row;
    fr[r] = rhs + lf.origin + xs + lf.s1 * row;
    dr[r] = dst + lu.origin + xs + lu.s1 * row;
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
    d2 up[RY], f[RY], hm, hp;
    double el[RY], er[RY];
  };
  auto load_stage = [&](Stage &st, int q) {
    const int m = mb + q;
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      st.up[r] = load2(ur[r] + lu.s2 * (m + 1));
      if (!(ABL & 2)) st.f[r] = NTF ? (d2)__builtin_nontemporal_load((const d2_a8 *)(fr[r] + lf.s2 * m)) : load2(fr[r] + lf.s2 * m);
      else st.f[r] = d2{1.0, 2.0};
      if (EPF && !(ABL & 8)) {
        st.el[r] = 0.0;
        st.er[r] = 0.0;
        if (lload) st.el[r] = ur[r][lu.s2 * m - 1];
        if (rload) st.er[r] = ur[r][lu.s2 * m + 2];
      }
    }
    if (!(ABL & 1)) {
      st.hm = load2(uhm + lu.s2 * m);
      st.hp = load2(uhp + lu.s2 * m);
    }
  };
  auto compute = [&](const Stage &st, int q) {
    const int m = mb + q;
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      double xl = lane_below(uc[r].y);
      double xr = lane_above(uc[r].x);
      if (!(ABL & 8)) {
        if (EPF) {
          if (lload) xl = st.el[r];
          if (rload) xr = st.er[r];
        } else {
          if (lload) xl = ur[r][lu.s2 * m - 1];
          if (rload) xr = ur[r][lu.s2 * m + 2];
        }
      }
      const d2 tm_ = (r == 0) ? ((ABL & 1) ? uc[0] : st.hm) : uc[r == 0 ? 0 : r - 1];
      const d2 tp_ = (r == RY - 1) ? ((ABL & 1) ? uc[RY - 1] : st.hp) : uc[r == RY - 1 ? r : r + 1];
      const double acc_a = conv7<0>(k, uc[r].x, xl, uc[r].y, tm_.x, tp_.x, um[r].x, st.up[r].x);
      const double acc_b = conv7<0>(k, uc[r].y, uc[r].x, xr, tm_.y, tp_.y, um[r].y, st.up[r].y);
      d2 o;
      o.x = uc[r].x + w * (st.f[r].x - acc_a);
      o.y = uc[r].y + w * (st.f[r].y - acc_b);
      if (rw + r < box.e1) {
        double *q_ = dr[r] + lu.s2 * m;
        if (ABL & 4) {
          if (o.x == 1.2345e300) q_[0] = o.y;   // keeps the arithmetic alive, never true
        } else if (ABL & 16) {
          // timing only: the wave's 1 KiB of output at the 128-byte boundary below its window -- what stores of whole, aligned
          // lines would cost (wrong addresses: neighbouring windows overlap by up to 15 doubles)
          double *qa = (double *)(((unsigned long long)(q_ - 2 * lane) & ~(unsigned long long)(g.amask))) + 2 * lane;
          if (vb) __builtin_nontemporal_store(o, (d2 *)qa);
        } else if (vb) {
          __builtin_nontemporal_store(o.x, q_);
          __builtin_nontemporal_store(o.y, q_ + 1);
        } else if (va) {
          q_[0] = o.x;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      um[r] = uc[r];
      uc[r] = st.up[r];
    }
  };
  const int cnt = me - mb;
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

// --------------------------------------------------------------------------------------------------------------------
// LDS-staged plane kernel: the NW waves of a workgroup own RY consecutive rows each of a 128-point x window and march in z.
// Every input row is loaded from memory once per workgroup: a wave publishes its rows of plane m+1 in LDS, from where the waves
// above and below read their y-neighbour rows in step m+1; only the two rows next to the workgroup's tile come from memory
// (first and last wave).  All loads are unconditional (32-bit offsets clamped into the array), x-neighbours come from the
// adjacent lanes, the two window-edge values from one 8-byte load per row (lanes < 32 fetch the left one, the others the right).
// Pipeline: plane m+2 of u and plane m+1 of rhs are in flight while plane m is computed.  One barrier per plane.
// --------------------------------------------------------------------------------------------------------------------
template <int SP>
__device__ __forceinline__ void store_pol(double *q, d2 o) {
  if (SP == 1) { *(d2_a8 *)q = o; }
  else if (SP == 2) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(o) : "memory"); }
  else if (SP == 3) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q), "v"(o) : "memory"); }
  else { __builtin_nontemporal_store(o, (d2_a8 *)q); }
}

template <int NW, int RY, int ABL, int SP = 0, bool NTF = false>
__global__ void __launch_bounds__(64 * NW)
k_lds(LayoutDev lu, const double *__restrict__ u, LayoutDev lf, const double *__restrict__ rhs, double *__restrict__ dst, Coef7 k,
      double w, Box box, Geo g) {
  constexpr int NR = NW * RY;
  __shared__ d2 UB[2][NR + 2][64];   // row i <-> global row rw0 - 1 + i
  const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
  int t = blockIdx.x;
  const int tx = t % g.ntx;
  t /= g.ntx;
  const int ty = t % g.nty;
  const int tz = t / g.nty;
  const int x0 = box.b0 + tx * 128;
  const int x = x0 + 2 * lane;
  const int rw0 = box.b1 + ty * NR;
  const int rw = rw0 + wv * RY;
  const int mb = box.b2 + tz * g.zc;
  const int me = min(mb + g.zc, box.e2);
  const bool va = x < box.e0, vb = x + 1 < box.e0;

  // offsets relative to the array start, clamped into [0, size - 2]
  const long long hi_u = lu.size - 2, hi_f = lf.size - 2;
  auto ldu = [&](int xx, int row, int p) {
    long long o = lu.origin + xx + lu.s1 * row + lu.s2 * p;
    o = min(max(o, 0LL), hi_u);
    return load2(u + o);
  };
  auto ldf = [&](int row, int p) {
    long long o = lf.origin + x + lf.s1 * row + lf.s2 * p;
    o = min(max(o, 0LL), hi_f);
    if (NTF) return (d2)__builtin_nontemporal_load((const d2_a8 *)(rhs + o));
    return load2(rhs + o);
  };
  auto lde = [&](int row, int p) {   // window-edge value: lanes < 32 -> u[x0 - 1], others -> u[x0 + 128]
    long long o = lu.origin + (lane < 32 ? x0 - 1 : x0 + 128) + lu.s1 * row + lu.s2 * p;
    o = min(max(o, 0LL), hi_u + 1);
    return u[o];
  };
  const bool edge_wave = wv == 0 || wv == NW - 1;
  const int hrow = wv == 0 ? rw0 - 1 : min(rw0 + NR, box.e1);   // the tile's halo row this wave carries along
  const int hidx = wv == 0 ? 0 : NR + 1;
  int rowc[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) rowc[r] = min(rw + r, box.e1);

  d2 um[RY], uc[RY], up[RY], upf[RY], f[RY], fpf[RY];
  double ec[RY], epf[RY];
  d2 hup = {0, 0}, hupf = {0, 0};
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    um[r] = ldu(x, rowc[r], mb - 1);
    uc[r] = ldu(x, rowc[r], mb);
    up[r] = ldu(x, rowc[r], mb + 1);
    upf[r] = ldu(x, rowc[r], min(mb + 2, me));
    f[r] = (ABL & 2) ? d2{1.0, 2.0} : ldf(rowc[r], mb);
    fpf[r] = (ABL & 2) ? d2{1.0, 2.0} : ldf(rowc[r], min(mb + 1, me - 1));
    ec[r] = lde(rowc[r], mb);
    epf[r] = lde(rowc[r], min(mb + 1, me - 1));
    UB[mb & 1][1 + wv * RY + r][lane] = uc[r];
  }
  if (edge_wave) {
    UB[mb & 1][hidx][lane] = ldu(x, hrow, mb);
    hup = ldu(x, hrow, mb + 1);
    hupf = ldu(x, hrow, min(mb + 2, me));
  }
  __syncthreads();
  for (int m = mb; m < me; ++m) {
    const int b = m & 1;
    const d2 ylo = UB[b][wv * RY][lane];
    const d2 yhi = UB[b][wv * RY + RY + 1][lane];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      double xl = lane_below(uc[r].y);
      double xr = lane_above(uc[r].x);
      if (lane == 0) xl = ec[r];
      if (lane == 63) xr = ec[r];
      const d2 tm_ = (r == 0) ? ylo : uc[r == 0 ? 0 : r - 1];
      const d2 tp_ = (r == RY - 1) ? yhi : uc[r == RY - 1 ? r : r + 1];
      const double acc_a = conv7<0>(k, uc[r].x, xl, uc[r].y, tm_.x, tp_.x, um[r].x, up[r].x);
      const double acc_b = conv7<0>(k, uc[r].y, uc[r].x, xr, tm_.y, tp_.y, um[r].y, up[r].y);
      d2 o;
      o.x = uc[r].x + w * (f[r].x - acc_a);
      o.y = uc[r].y + w * (f[r].y - acc_b);
      if (rw + r < box.e1) {
        double *q_ = dst + lu.origin + x + lu.s1 * (rw + r) + lu.s2 * m;
        if (ABL & 4) {
          if (o.x == 1.2345e300) q_[0] = o.y;
        } else if (vb) {
          store_pol<SP>(q_, o);
        } else if (va) {
          q_[0] = o.x;
        }
      }
    }
    if (m + 1 < me) {
#pragma unroll
      for (int r = 0; r < RY; ++r) UB[b ^ 1][1 + wv * RY + r][lane] = up[r];
      if (edge_wave) UB[b ^ 1][hidx][lane] = hup;
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      um[r] = uc[r];
      uc[r] = up[r];
      up[r] = upf[r];
      f[r] = fpf[r];
      ec[r] = epf[r];
    }
    hup = hupf;
    if (m + 1 < me) {
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        upf[r] = ldu(x, rowc[r], min(m + 3, me));
        if (!(ABL & 2)) fpf[r] = ldf(rowc[r], min(m + 2, me - 1));
        epf[r] = lde(rowc[r], min(m + 2, me - 1));
      }
      if (edge_wave) hupf = ldu(x, hrow, min(m + 3, me));
    }
    __syncthreads();
  }
}

// --------------------------------------------------------------------------------------------------------------------
struct Ctx {
  LayoutDev lu, lf;
  double *u, *un, *f, *ref;
  Coef7 k;
  double w;
  Box box;
  long long updates;
};

static examg_layout_t node_layout(int n, int ghost, int align) {
  examg_layout_t l;
  memset(&l, 0, sizeof(l));
  l.nd = 3;
  for (int d = 0; d < 3; ++d) {
    l.ghost_l[d] = l.ghost_r[d] = ghost;
    l.dup_l[d] = l.dup_r[d] = 1;
    l.inner[d] = n - 1;
  }
  if (align) {
    l.pad_l[0] = (align - ghost % align) % align;
    const int tot = l.pad_l[0] + ghost + 1 + (n - 1) + 1 + ghost;
    l.pad_r[0] = (align - tot % align) % align;
  }
  return l;
}

template <typename F>
static float time_it(F &&launch, int reps = 20) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) launch();
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return ms / reps;
}

static bool same(const Ctx &c, std::vector<double> &a, std::vector<double> &b) {
  CHECK(hipMemcpy(a.data(), c.un, c.lu.size * 8, hipMemcpyDeviceToHost));
  long long bad = 0;
  for (long long i = 0; i < c.lu.size; ++i)
    if (memcmp(&a[i], &b[i], 8) != 0) ++bad;
  return bad == 0;
}

static std::vector<double> g_ref, g_tmp;

static void report(const Ctx &c, const char *name, float ms, bool check) {
  const char *ok = "-";
  if (check) ok = same(c, g_tmp, g_ref) ? "bit-exact" : "MISMATCH";
  printf("%-44s %8.4f ms  %6.0f GB/s  frac %.3f  %s\n", name, ms, 24.0 * c.updates / ms / 1e6, 24.0 * c.updates / ms / 1e6 / 8000.0, ok);
  fflush(stdout);
}

static int g_minchunk = 16;
static int g_amask = 127;

template <int RY, int WY, int PF, bool EPF, int ABL, bool NTF = false, bool REMAP = false, int WX = 1>
static void run_zm(const Ctx &c, int blocks, const char *tag) {
  Geo g;
  g.ntx = (c.box.n0() + 128 * WX - 1) / (128 * WX);
  g.nty = (c.box.n1() + RY * WY - 1) / (RY * WY);
  const int xy = g.ntx * g.nty;
  int ntz = (blocks + xy - 1) / xy;
  if (ntz < 1) ntz = 1;
  int zc = (c.box.n2() + ntz - 1) / ntz;
  if (zc < g_minchunk) zc = g_minchunk;
  if (zc > c.box.n2()) zc = c.box.n2();
  g.zc = zc;
  g.ntz = (c.box.n2() + zc - 1) / zc;
  g.nblocks = xy * g.ntz;
  g.amask = g_amask;
  CHECK(hipMemset(c.un, 0, c.lu.size * 8));
  auto launch = [&]() {
    hipLaunchKernelGGL((k_zm<RY, WY, PF, EPF, ABL, NTF, REMAP, WX>), dim3(g.nblocks), dim3(64, WY * WX), 0, 0, c.lu, c.u, c.lf, c.f, c.un, c.k, c.w, c.box, g);
  };
  const float ms = time_it(launch);
  char name[160];
  snprintf(name, sizeof name, "zm ry%d wy%d wx%d pf%d epf%d abl%d ntf%d remap%d wgs%d zc%d %s", RY, WY, WX, PF, (int)EPF, ABL, (int)NTF, (int)REMAP, g.nblocks, zc, tag);
  report(c, name, ms, ABL == 0);
}

template <int NW, int RY, int ABL, int SP = 0, bool NTF = false>
static void run_lds(const Ctx &c, int blocks, const char *tag) {
  Geo g;
  g.ntx = (c.box.n0() + 127) / 128;
  g.nty = (c.box.n1() + NW * RY - 1) / (NW * RY);
  const int xy = g.ntx * g.nty;
  int ntz = (blocks + xy - 1) / xy;
  if (ntz < 1) ntz = 1;
  int zc = (c.box.n2() + ntz - 1) / ntz;
  if (zc < 8) zc = 8;
  if (zc > c.box.n2()) zc = c.box.n2();
  g.zc = zc;
  g.ntz = (c.box.n2() + zc - 1) / zc;
  g.nblocks = xy * g.ntz;
  CHECK(hipMemset(c.un, 0, c.lu.size * 8));
  auto launch = [&]() {
    hipLaunchKernelGGL((k_lds<NW, RY, ABL, SP, NTF>), dim3(g.nblocks), dim3(64, NW), 0, 0, c.lu, c.u, c.lf, c.f, c.un, c.k, c.w, c.box, g);
  };
  const float ms = time_it(launch);
  char name[160];
  snprintf(name, sizeof name, "lds nw%d ry%d abl%d sp%d ntf%d wgs%d zc%d %s", NW, RY, ABL, SP, (int)NTF, g.nblocks, zc, tag);
  report(c, name, ms, ABL == 0);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512;
  const int align = argc > 2 ? atoi(argv[2]) : 0;
  const char *only = argc > 3 ? argv[3] : "";
  examg_layout_t lu_ = node_layout(n, 1, align), lf_ = node_layout(n, 0, align);
  if (argc > 5) {   // extra right padding of the rows (doubles), both layouts
    lu_.pad_r[0] += atoi(argv[5]);
    lf_.pad_r[0] += atoi(argv[5]);
  }
  Ctx c;
  c.lu = make_layout(&lu_);
  c.lf = make_layout(&lf_);
  CHECK(hipMalloc(&c.u, c.lu.size * 8));
  CHECK(hipMalloc(&c.un, c.lu.size * 8));
  CHECK(hipMalloc(&c.ref, c.lu.size * 8));
  CHECK(hipMalloc(&c.f, c.lf.size * 8));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c.u, c.lu.size, 12345ull);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c.f, c.lf.size, 777ull);
  const double h2 = (double)n * n;
  c.k.c[0] = 6.0 * h2;
  for (int i = 1; i < 7; ++i) c.k.c[i] = -1.0 * h2;
  c.w = 0.8 / c.k.c[0];
  const int b0 = argc > 4 ? atoi(argv[4]) : 1;   // 0: windows start at the duplicate point (timing of aligned access only)
  c.box = Box{b0, 1, 1, n, n, n};
  c.updates = c.box.count();
  CHECK(hipMemset(c.ref, 0, c.lu.size * 8));
  hipLaunchKernelGGL(k_ref, dim3(8192), dim3(256), 0, 0, c.lu, c.u, c.lf, c.f, c.ref, c.k, c.w, c.box);
  CHECK(hipDeviceSynchronize());
  g_ref.resize(c.lu.size);
  g_tmp.resize(c.lu.size);
  CHECK(hipMemcpy(g_ref.data(), c.ref, c.lu.size * 8, hipMemcpyDeviceToHost));
  printf("n=%d align=%d row=%d doubles, %lld updates\n", n, align, c.lu.tot0, c.updates);
  const std::string sel(only);
  auto want = [&](const char *grp) { return sel.empty() || sel.find(grp) != std::string::npos; };

  if (want("zm")) {
    run_zm<2, 4, 1, false, 0>(c, 1024, "(round-1 shipped)");
    run_zm<2, 4, 1, true, 0>(c, 1024, "edge prefetch");
    run_zm<2, 4, 2, true, 0>(c, 1024, "edge prefetch");
    run_zm<2, 4, 1, true, 0>(c, 512, "");
    run_zm<2, 4, 1, true, 0>(c, 256, "");
    run_zm<2, 4, 1, true, 0>(c, 2048, "");
    run_zm<1, 4, 1, true, 0>(c, 512, "");
    run_zm<1, 4, 2, true, 0>(c, 512, "");
    run_zm<1, 8, 1, true, 0>(c, 256, "");
    run_zm<4, 4, 1, true, 0>(c, 1024, "");
  }
  if (want("r3")) {   // round 3: tile shapes in the short-chunk regime (XCD-banded order, as many workgroups as the chunks need)
    for (int rnd = 0; rnd < 2; ++rnd) {
      for (int mc : {8, 16, 32}) {
        g_minchunk = mc;
        run_zm<2, 4, 1, true, 0, false, true, 1>(c, 1 << 24, "shipped shape");
        run_zm<2, 8, 1, true, 0, false, true, 1>(c, 1 << 24, "");
        run_zm<4, 4, 1, true, 0, false, true, 1>(c, 1 << 24, "");
        run_zm<2, 4, 1, true, 0, false, true, 2>(c, 1 << 24, "");
        run_zm<2, 2, 1, true, 0, false, true, 2>(c, 1 << 24, "");
        run_zm<2, 2, 1, true, 0, false, true, 4>(c, 1 << 24, "");
        run_zm<4, 2, 1, true, 0, false, true, 2>(c, 1 << 24, "");
        run_zm<2, 4, 2, true, 0, false, true, 1>(c, 1 << 24, "pf2");
        run_zm<2, 8, 2, true, 0, false, true, 1>(c, 1 << 24, "pf2");
      }
    }
    g_minchunk = 16;
  }
  if (want("front")) {   // round 3: ONE thin front -- as many workgroups as xy tiles, whole columns, deep software pipeline instead of occupancy
    g_minchunk = 16;
    for (int rnd = 0; rnd < 2; ++rnd) {
      run_zm<2, 4, 1, true, 0, false, true, 1>(c, 256, "");
      run_zm<2, 4, 2, true, 0, false, true, 1>(c, 256, "");
      run_zm<2, 4, 3, true, 0, false, true, 1>(c, 256, "");
      run_zm<2, 4, 4, true, 0, false, true, 1>(c, 256, "");
      run_zm<2, 4, 6, true, 0, false, true, 1>(c, 256, "");
      run_zm<1, 8, 2, true, 0, false, true, 1>(c, 256, "");
      run_zm<1, 8, 4, true, 0, false, true, 1>(c, 256, "");
      run_zm<1, 8, 6, true, 0, false, true, 1>(c, 256, "");
      run_zm<2, 4, 2, true, 0, false, true, 1>(c, 512, "");
      run_zm<2, 4, 3, true, 0, false, true, 1>(c, 512, "");
      run_zm<2, 4, 4, true, 0, false, true, 1>(c, 512, "");
      run_zm<1, 4, 4, true, 0, false, true, 1>(c, 512, "");
      run_zm<1, 4, 6, true, 0, false, true, 1>(c, 512, "");
      run_zm<2, 4, 3, true, 0, false, false, 1>(c, 256, "plain order");
    }
  }
  if (want("sabl")) {   // round 3: ablations in the shipped short-chunk regime (run for align 0 and for align 16 with b0 = 0)
    g_minchunk = 8;
    for (int rnd = 0; rnd < 2; ++rnd) {
      run_zm<2, 4, 1, true, 0, false, true, 1>(c, 1 << 24, "full");
      run_zm<2, 4, 1, true, 1, false, true, 1>(c, 1 << 24, "no halo rows");
      run_zm<2, 4, 1, true, 2, false, true, 1>(c, 1 << 24, "no rhs");
      run_zm<2, 4, 1, true, 4, false, true, 1>(c, 1 << 24, "no stores");
      run_zm<2, 4, 1, true, 6, false, true, 1>(c, 1 << 24, "no rhs, no stores");
      run_zm<2, 4, 1, true, 7, false, true, 1>(c, 1 << 24, "u rows only");
      run_zm<2, 4, 1, true, 3, false, true, 1>(c, 1 << 24, "u rows + stores");
      g_amask = 127;
      run_zm<2, 4, 1, true, 16, false, true, 1>(c, 1 << 24, "full, stores moved to line boundaries (timing only)");
      g_amask = 63;
      run_zm<2, 4, 1, true, 16, false, true, 1>(c, 1 << 24, "full, stores moved to 64-byte boundaries (timing only)");
      g_amask = 31;
      run_zm<2, 4, 1, true, 16, false, true, 1>(c, 1 << 24, "full, stores moved to 32-byte boundaries (timing only)");
      g_amask = 15;
      run_zm<2, 4, 1, true, 16, false, true, 1>(c, 1 << 24, "full, stores moved to 16-byte boundaries (timing only)");
      g_amask = 127;
      run_zm<2, 4, 1, true, 19, false, true, 1>(c, 1 << 24, "u rows + stores at line boundaries (timing only)");
    }
    g_minchunk = 16;
  }
  if (want("abl")) {
    run_zm<2, 4, 1, false, 1>(c, 1024, "no halo rows");
    run_zm<2, 4, 1, false, 2>(c, 1024, "no rhs");
    run_zm<2, 4, 1, false, 4>(c, 1024, "no stores");
    run_zm<2, 4, 1, false, 8>(c, 1024, "no edge loads");
    run_zm<2, 4, 1, false, 9>(c, 1024, "no halo, no edge");
    run_zm<2, 4, 1, false, 11>(c, 1024, "u rows only + store");
    run_zm<2, 4, 1, false, 15>(c, 1024, "u rows only, no store");
  }
  if (want("lds")) {
    run_lds<8, 2, 0>(c, 512, "");
    run_lds<8, 2, 0>(c, 1024, "");
    run_lds<8, 2, 0>(c, 128, "");
    run_lds<8, 2, 0>(c, 256, "");
    run_lds<8, 2, 0>(c, 2048, "");
    run_lds<4, 2, 0>(c, 512, "");
    run_lds<4, 2, 0>(c, 1024, "");
    run_lds<4, 4, 0>(c, 512, "");
    run_lds<4, 4, 0>(c, 1024, "");
    run_lds<8, 4, 0>(c, 256, "");
    run_lds<8, 4, 0>(c, 512, "");
    run_lds<16, 1, 0>(c, 512, "");
    run_lds<16, 2, 0>(c, 256, "");
    run_lds<8, 1, 0>(c, 1024, "");
    run_lds<8, 2, 2>(c, 512, "no rhs");
    run_lds<8, 2, 4>(c, 512, "no stores");
  }
  if (want("zz")) {
    for (int rep = 0; rep < 2; ++rep) {
      run_zm<2, 4, 1, false, 0>(c, 1024, "(round-1 shipped)");
      run_zm<2, 4, 1, true, 0>(c, 512, "");
      run_zm<2, 4, 2, true, 0>(c, 512, "");
      run_zm<2, 4, 1, true, 0, true>(c, 512, "nt rhs");
      run_zm<2, 4, 1, true, 0, false, true>(c, 512, "xcd remap");
      run_zm<2, 4, 1, true, 0, true, true>(c, 512, "nt rhs, xcd remap");
      run_zm<4, 4, 1, true, 0>(c, 512, "");
      run_zm<4, 4, 1, true, 0>(c, 256, "");
      run_zm<2, 4, 1, true, 0>(c, 448, "");
      run_zm<2, 4, 1, true, 0>(c, 576, "");
      run_zm<2, 4, 1, true, 0>(c, 16384, "");
    }
  }
  if (want("row")) {
    run_zm<2, 4, 1, false, 0>(c, 1024, "(round-1 shipped)");
    run_zm<2, 4, 1, true, 0>(c, 512, "");
  }
  if (want("lay")) {
    for (int rep = 0; rep < 2; ++rep) {
      run_zm<2, 4, 1, true, 0>(c, 512, "");
      run_zm<2, 4, 1, true, 0, false, true>(c, 512, "layered");
      run_zm<2, 4, 1, true, 0, false, true>(c, 1024, "layered");
      run_zm<2, 4, 1, true, 0, false, true>(c, 2048, "layered");
      run_zm<2, 4, 1, true, 0, false, true>(c, 4096, "layered");
      run_zm<2, 4, 1, true, 0, false, true>(c, 8192, "layered");
      run_zm<2, 4, 1, true, 0, false, false>(c, 8192, "");
      run_zm<2, 2, 1, true, 0, false, true>(c, 8192, "layered");
      run_zm<4, 4, 1, true, 0, false, true>(c, 4096, "layered");
    }
  }
  if (want("wx")) {
    for (int rep = 0; rep < 2; ++rep) {
      run_zm<2, 4, 1, false, 0>(c, 1024, "(round-1 shipped)");
      run_zm<2, 4, 1, true, 0>(c, 512, "");
      run_zm<2, 1, 1, true, 0, false, false, 4>(c, 512, "");
      run_zm<2, 1, 1, true, 0, false, false, 4>(c, 1024, "");
      run_zm<2, 2, 1, true, 0, false, false, 4>(c, 512, "");
      run_zm<2, 2, 1, true, 0, false, false, 4>(c, 256, "");
      run_zm<4, 1, 1, true, 0, false, false, 4>(c, 512, "");
      run_zm<4, 2, 1, true, 0, false, false, 4>(c, 256, "");
      run_zm<2, 4, 1, true, 0, false, false, 4>(c, 256, "");
      run_zm<2, 2, 1, true, 0, false, false, 2>(c, 512, "");
      run_zm<2, 4, 1, true, 0, false, false, 2>(c, 512, "");
      run_zm<2, 4, 1, true, 0, false, false, 2>(c, 256, "");
    }
  }
  if (want("al")) {
    for (int rep = 0; rep < 2; ++rep) {
      run_zm<2, 4, 1, false, 0>(c, 1024, "(round-1 shipped)");
      run_zm<2, 4, 1, true, 0>(c, 512, "");
      run_zm<2, 4, 1, true, 0>(c, 1024, "");
      run_lds<8, 4, 0, 0>(c, 512, "nt");
    }
  }
  if (want("pol")) {
    run_lds<8, 4, 0, 0>(c, 512, "nt");
    run_lds<8, 4, 0, 1>(c, 512, "plain");
    run_lds<8, 4, 0, 2>(c, 512, "sc1");
    run_lds<8, 4, 0, 3>(c, 512, "sc0 sc1");
    run_lds<8, 4, 0, 0, true>(c, 512, "nt rhs loads");
    run_lds<8, 4, 0, 1, true>(c, 512, "plain st, nt rhs loads");
    run_lds<8, 2, 0, 0, true>(c, 1024, "nt rhs loads");
    run_lds<8, 4, 0, 0>(c, 4096, "");
    run_lds<8, 4, 0, 0>(c, 16384, "");
    run_lds<8, 2, 0, 0>(c, 8192, "");
    run_lds<8, 2, 0, 0>(c, 16384, "");
    run_lds<4, 4, 0, 0>(c, 2048, "");
    run_lds<4, 4, 0, 0>(c, 8192, "");
    run_zm<2, 4, 1, true, 0>(c, 384, "");
    run_zm<2, 4, 1, true, 0>(c, 640, "");
    run_zm<2, 4, 1, true, 0>(c, 768, "");
    run_zm<2, 4, 1, true, 0>(c, 4096, "");
    run_zm<2, 4, 1, true, 0>(c, 8192, "");
    run_zm<2, 2, 1, true, 0>(c, 512, "");
    run_zm<2, 2, 1, true, 0>(c, 1024, "");
    run_zm<2, 8, 1, true, 0>(c, 512, "");
    run_zm<4, 2, 1, true, 0>(c, 512, "");
  }
  return 0;
}
