// BLAS-1 loops, reductions, Dirichlet boundary handling, halo pack/unpack, library plumbing (gfx950).
//
// Reference loops: Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:113-119 (ResNorm), :160-198
// (CG vector updates), :226-229 (Solution@coarser = 0); apply bc:
// Compiler/src/exastencils/boundary/ir/IR_ApplyBCFunction.scala:53-104 (the reference launches
// one kernel per face -- here one launch covers all faces); pack/unpack:
// communication/ir/IR_NoInterpPacking.scala:53-83; reductions: the reference writes one term per
// point to a scratch array and launches log2(n) halving kernels
// (parallelization/api/cuda/CUDA_Reduction.scala:84-131, CUDA_KernelFunctions.scala:112-238) -- here a
// two-stage fixed-tree reduction (wave shuffle -> LDS -> one partial per workgroup -> one workgroup).
#include <stdarg.h>

#include "examg_common.h"

namespace examg {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_hip(hipError_t e, const char *what) {
  if (e == hipSuccess) return 0;
  set_error("%s: %s", what, hipGetErrorString(e));
  return 2;
}

static inline dim3 grid_for(long long total, int cap = 8192) {
  long long nb = (total + 255) / 256;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  return dim3((unsigned)nb);
}

__device__ __forceinline__ void unflatten(const Box &box, long long t, int &i0, int &i1, int &i2) {
  const int n0 = box.n0(), n1 = box.n1();
  i0 = box.b0 + (int)(t % n0);
  const long long row = t / n0;
  i1 = box.b1 + (int)(row % n1);
  i2 = box.b2 + (int)(row / n1);
}

__global__ void __launch_bounds__(256) k_set(LayoutDev l, double *x, double v, Box box) {
  const long long total = box.count();
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int i0, i1, i2;
    unflatten(box, t, i0, i1, i2);
    x[lidx(l, i0, i1, i2)] = v;
  }
}

// form: 0 => a*x   1 => x (copy)   2 => y + a*x   3 => x + b*y   4 => a*x + b*y
// DEV: 0 none, 1 => a = sign * num/den, 2 => b = num/den (read from device memory)
template <int DEV>
__global__ void __launch_bounds__(256)
k_axpby(LayoutDev lx, const double *x, LayoutDev ly, double *y, double a, double b, int form, double sign,
        const double *num, const double *den, Box box) {
  if (DEV == 1) a = sign * (*num / *den);
  if (DEV == 2) b = *num / *den;
  const long long total = box.count();
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int i0, i1, i2;
    unflatten(box, t, i0, i1, i2);
    const double xv = x[lidx(lx, i0, i1, i2)];
    const long long k = lidx(ly, i0, i1, i2);
    double r;
    switch (form) {
      case 0: r = a * xv; break;
      case 1: r = xv; break;
      case 2: r = y[k] + a * xv; break;
      case 3: r = xv + b * y[k]; break;
      default: r = a * xv + b * y[k]; break;
    }
    y[k] = r;
  }
}

// ---- reductions -------------------------------------------------------------------------------
constexpr int RED_BLOCK = 256;
constexpr int RED_MAX_BLOCKS = 2048;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = v + __shfl_down(v, o);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o));
  return v;
}

template <bool IS_MAX>
__device__ __forceinline__ double block_reduce(double v) {
  __shared__ double sm[RED_BLOCK / 64];
  v = IS_MAX ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) sm[wv] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    r = sm[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = IS_MAX ? fmax(r, sm[i]) : r + sm[i];
  }
  __syncthreads();
  return r;  // valid in thread 0
}

__global__ void __launch_bounds__(RED_BLOCK)
k_dot_partial(LayoutDev lx, const double *__restrict__ x, LayoutDev ly, const double *__restrict__ y, Box box, double *part) {
  const long long total = box.count();
  double s = 0.0;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int i0, i1, i2;
    unflatten(box, t, i0, i1, i2);
    s = s + x[lidx(lx, i0, i1, i2)] * y[lidx(ly, i0, i1, i2)];
  }
  const double r = block_reduce<false>(s);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

// Long rows: a wave streams whole rows, two points per lane with 16-byte loads, no index divisions per element; the rows a
// wave takes and the order it adds them in depend only on the launch geometry, so the sum is reproducible run to run.
__global__ void __launch_bounds__(RED_BLOCK)
k_dot_rows(LayoutDev lx, const double *__restrict__ x, LayoutDev ly, const double *__restrict__ y, Box box, double *part) {
  const int lane = threadIdx.x & 63;
  const int n1 = box.n1();
  const long long nrows = (long long)n1 * box.n2();
  const long long wave = (long long)blockIdx.x * (RED_BLOCK / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (RED_BLOCK / 64);
  double s = 0.0;
  for (long long r = wave; r < nrows; r += nwaves) {
    const int i1 = box.b1 + (int)(r % n1), i2 = box.b2 + (int)(r / n1);
    const double *px = x + lidx(lx, 0, i1, i2), *py = y + lidx(ly, 0, i1, i2);
    for (int i0 = box.b0 + 2 * lane; i0 < box.e0; i0 += 128) {
      if (i0 + 1 < box.e0) {
        const d2 a = load2(px + i0), b = load2(py + i0);
        s = s + a.x * b.x;
        s = s + a.y * b.y;
      } else {
        s = s + px[i0] * py[i0];
      }
    }
  }
  const double r = block_reduce<false>(s);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

// A point function: a postfix program (include/examg.h), evaluated in the order of the expression tree.
struct ExprEval {
  examg_expr_t e;
  __device__ double operator()(double x, double y, double z) const {
    double st[24];
    int sp = 0;
    for (int i = 0; i < e.n; ++i) {
      switch (e.op[i]) {
        case EXAMG_OP_CONST: st[sp++] = e.c[i]; break;
        case EXAMG_OP_X: st[sp++] = x; break;
        case EXAMG_OP_Y: st[sp++] = y; break;
        case EXAMG_OP_Z: st[sp++] = z; break;
        case EXAMG_OP_ADD: --sp; st[sp - 1] = st[sp - 1] + st[sp]; break;
        case EXAMG_OP_SUB: --sp; st[sp - 1] = st[sp - 1] - st[sp]; break;
        case EXAMG_OP_MUL: --sp; st[sp - 1] = st[sp - 1] * st[sp]; break;
        case EXAMG_OP_DIV: --sp; st[sp - 1] = st[sp - 1] / st[sp]; break;
        case EXAMG_OP_NEG: st[sp - 1] = -st[sp - 1]; break;
        case EXAMG_OP_SIN: st[sp - 1] = sin(st[sp - 1]); break;
        case EXAMG_OP_COS: st[sp - 1] = cos(st[sp - 1]); break;
        case EXAMG_OP_EXP: st[sp - 1] = exp(st[sp - 1]); break;
        case EXAMG_OP_SINH: st[sp - 1] = sinh(st[sp - 1]); break;
        case EXAMG_OP_COSH: st[sp - 1] = cosh(st[sp - 1]); break;
        case EXAMG_OP_SQRT: st[sp - 1] = sqrt(st[sp - 1]); break;
        case EXAMG_OP_POW: --sp; st[sp - 1] = pow(st[sp - 1], st[sp]); break;
        case EXAMG_OP_TAN: st[sp - 1] = tan(st[sp - 1]); break;
        case EXAMG_OP_LOG: st[sp - 1] = log(st[sp - 1]); break;
        case EXAMG_OP_FABS: st[sp - 1] = fabs(st[sp - 1]); break;
        case EXAMG_OP_MAX: --sp; st[sp - 1] = fmax(st[sp - 1], st[sp]); break;
        case EXAMG_OP_MIN: --sp; st[sp - 1] = fmin(st[sp - 1], st[sp]); break;
        case EXAMG_OP_TANH: st[sp - 1] = tanh(st[sp - 1]); break;
        default: st[sp++] = __builtin_nan(""); break;
      }
    }
    return st[0];
  }
};

template <class F>
__global__ void __launch_bounds__(RED_BLOCK)
k_maxerr_partial(LayoutDev l, const double *__restrict__ x, Geom g, F fn, Box box, double *part) {
  const long long total = box.count();
  double m = 0.0;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int i0, i1, i2;
    unflatten(box, t, i0, i1, i2);
    const double px = i0 * g.h0 + g.pb0, py = i1 * g.h1 + g.pb1, pz = i2 * g.h2 + g.pb2;
    m = fmax(m, fabs(x[lidx(l, i0, i1, i2)] - fn(px, py, pz)));
  }
  const double r = block_reduce<true>(m);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

template <bool IS_MAX>
__global__ void __launch_bounds__(RED_BLOCK) k_reduce_final(const double *part, int n, double *result) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s = IS_MAX ? fmax(s, part[i]) : s + part[i];
  const double r = block_reduce<IS_MAX>(s);
  if (threadIdx.x == 0) *result = r;
}

// ---- analytic fills ---------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) k_fill_fn(LayoutDev l, double *x, Geom g, F fn, Box box) {
  const long long total = box.count();
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int i0, i1, i2;
    unflatten(box, t, i0, i1, i2);
    const double px = i0 * g.h0 + g.pb0, py = i1 * g.h1 + g.pb1, pz = i2 * g.h2 + g.pb2;
    x[lidx(l, i0, i1, i2)] = fn(px, py, pz);
  }
}

struct FaceBoxes {
  Box box[6];
  long long start[7];  // prefix sums of counts
  int n;
};

template <class F>
__global__ void __launch_bounds__(256) k_apply_dirichlet(LayoutDev l, double *x, Geom g, F fn, FaceBoxes fb) {
  const long long total = fb.start[fb.n];
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int f = 0;
    while (f + 1 < fb.n && t >= fb.start[f + 1]) ++f;
    int i0, i1, i2;
    unflatten(fb.box[f], t - fb.start[f], i0, i1, i2);
    const double px = i0 * g.h0 + g.pb0, py = i1 * g.h1 + g.pb1, pz = i2 * g.h2 + g.pb2;
    x[lidx(l, i0, i1, i2)] = fn(px, py, pz);
  }
}

__global__ void __launch_bounds__(256)
k_init_varcoeff(LayoutDev lc, double *cf, Geom g, ExprEval a, Box box, int nd) {
  const long long total = box.count();
  const long long plane = lc.size;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int i0, i1, i2;
    unflatten(box, t, i0, i1, i2);
    const double x = i0 * g.h0 + g.pb0, y = i1 * g.h1 + g.pb1, z = i2 * g.h2 + g.pb2;
    const double hx = g.h0, hy = g.h1, hz = g.h2;
    const double axp = a(x + (0.5 * hx), y, z), axm = a(x - (0.5 * hx), y, z);
    const double ayp = a(x, y + (0.5 * hy), z), aym = a(x, y - (0.5 * hy), z);
    const long long k = lidx(lc, i0, i1, i2);
    if (nd == 3) {
      const double azp = a(x, y, z + (0.5 * hz)), azm = a(x, y, z - (0.5 * hz));
      cf[k + 0 * plane] = (((axp + axm) / (hx * hx)) + ((ayp + aym) / (hy * hy))) + ((azp + azm) / (hz * hz));
      cf[k + 1 * plane] = (-1.0 * axp) / (hx * hx);
      cf[k + 2 * plane] = (-1.0 * axm) / (hx * hx);
      cf[k + 3 * plane] = (-1.0 * ayp) / (hy * hy);
      cf[k + 4 * plane] = (-1.0 * aym) / (hy * hy);
      cf[k + 5 * plane] = (-1.0 * azp) / (hz * hz);
      cf[k + 6 * plane] = (-1.0 * azm) / (hz * hz);
    } else {
      cf[k + 0 * plane] = ((axp + axm) / (hx * hx)) + ((ayp + aym) / (hy * hy));
      cf[k + 1 * plane] = (-1.0 * axp) / (hx * hx);
      cf[k + 2 * plane] = (-1.0 * axm) / (hx * hx);
      cf[k + 3 * plane] = (-1.0 * ayp) / (hy * hy);
      cf[k + 4 * plane] = (-1.0 * aym) / (hy * hy);
    }
  }
}

// ---- layout transformation of a stencil field's coefficients: entry-slowest planes <-> entry-fastest records ---------------
// one workgroup moves a tile of 64 points x K entries through LDS so that both sides are accessed in runs of consecutive doubles
template <bool TO_AOS>
__global__ void __launch_bounds__(256) k_transform_sf(const double *__restrict__ src, double *__restrict__ dst, long long size, int K) {
  __shared__ double tile[64 * EXAMG_MAX_ENTRIES];
  const long long p0 = (long long)blockIdx.x * 64;
  const int np = (int)(size - p0 < 64 ? size - p0 : 64);
  const int n = np * K;
  for (int t = threadIdx.x; t < n; t += 256) {
    if (TO_AOS) { const int k = t / np, i = t - k * np; tile[i * K + k] = src[(long long)k * size + p0 + i]; }
    else tile[t] = src[p0 * K + t];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < n; t += 256) {
    if (TO_AOS) dst[p0 * K + t] = tile[t];
    else { const int k = t / np, i = t - k * np; dst[(long long)k * size + p0 + i] = tile[i * K + k]; }
  }
}

// ---- halo pack / unpack ---------------------------------------------------------------------
template <bool PACK>
__global__ void __launch_bounds__(256) k_pack(LayoutDev l, double *x, double *buf, Box box) {
  const long long total = box.count();
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int i0, i1, i2;
    unflatten(box, t, i0, i1, i2);
    if (PACK) buf[t] = x[lidx(l, i0, i1, i2)];
    else x[lidx(l, i0, i1, i2)] = buf[t];
  }
}

__global__ void __launch_bounds__(256) k_fill_random(double *x, long long n, unsigned long long seed) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned long long zz = seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)(i + 1);
    zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ULL;
    zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBULL;
    zz = zz ^ (zz >> 31);
    x[i] = (double)(zz >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  }
}

}  // namespace examg

using namespace examg;

extern "C" int examg_version(void) { return 100; }
extern "C" const char *examg_last_error(void) { return g_err; }
extern "C" int examg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int examg_set(const examg_layout_t *l_, double *x, double v, const int32_t *begin, const int32_t *end,
                         examg_stream_t stream) {
  if (!l_ || !x || !begin || !end) { set_error("examg_set: null argument"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (!box_inside(l_, box, 0)) { set_error("examg_set: box leaves the allocation"); return 1; }
  hipLaunchKernelGGL(k_set, grid_for(box.count()), dim3(256), 0, (hipStream_t)stream, make_layout(l_), x, v, box);
  EXAMG_CHECK_LAUNCH("k_set");
  return 0;
}

static int axpby_impl(const examg_layout_t *lx_, const double *x, const examg_layout_t *ly_, double *y, double a, double b,
                      int dev, double sign, const double *num, const double *den, const int32_t *begin,
                      const int32_t *end, examg_stream_t stream) {
  if (!lx_ || !x || !ly_ || !y || !begin || !end) { set_error("examg_axpby: null argument"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (!box_inside(lx_, box, 0) || !box_inside(ly_, box, 0)) { set_error("examg_axpby: box leaves an allocation"); return 1; }
  int form;
  if (dev == 1) form = (b == 1.0) ? 2 : ((b == 0.0) ? 0 : 4);
  else if (dev == 2) form = (a == 1.0) ? 3 : 4;
  else if (b == 0.0) form = (a == 1.0) ? 1 : 0;
  else if (b == 1.0) form = 2;
  else if (a == 1.0) form = 3;
  else form = 4;
  const LayoutDev lx = make_layout(lx_), ly = make_layout(ly_);
  hipStream_t s = (hipStream_t)stream;
  if (dev == 0) hipLaunchKernelGGL((k_axpby<0>), grid_for(box.count()), dim3(256), 0, s, lx, x, ly, y, a, b, form, sign, num, den, box);
  else if (dev == 1) hipLaunchKernelGGL((k_axpby<1>), grid_for(box.count()), dim3(256), 0, s, lx, x, ly, y, a, b, form, sign, num, den, box);
  else hipLaunchKernelGGL((k_axpby<2>), grid_for(box.count()), dim3(256), 0, s, lx, x, ly, y, a, b, form, sign, num, den, box);
  EXAMG_CHECK_LAUNCH("k_axpby");
  return 0;
}

extern "C" int examg_axpby(const examg_layout_t *lx, const double *x, const examg_layout_t *ly, double *y, double a,
                           double b, const int32_t *begin, const int32_t *end, examg_stream_t stream) {
  return axpby_impl(lx, x, ly, y, a, b, 0, 1.0, nullptr, nullptr, begin, end, stream);
}

extern "C" int examg_axpby_dev(const examg_layout_t *lx, const double *x, const examg_layout_t *ly, double *y, double a,
                               double b, int which, double sign, const double *num, const double *den,
                               const int32_t *begin, const int32_t *end, examg_stream_t stream) {
  if (!num || !den) { set_error("examg_axpby_dev: null scalar pointer"); return 1; }
  if (which != 0 && which != 1) { set_error("examg_axpby_dev: which must be 0 or 1"); return 1; }
  return axpby_impl(lx, x, ly, y, a, b, which + 1, sign, num, den, begin, end, stream);
}

constexpr int RED_WORK_DOUBLES = 32768;   // >= RED_MAX_BLOCKS; kernels that write one partial sum per wave need more than the dot kernels
extern "C" size_t examg_reduce_work_bytes(void) { return (size_t)RED_WORK_DOUBLES * sizeof(double); }

namespace examg {
// for kernels that write their own partial sums (kernels_stencil.hip: examg_residual_norm2)
size_t reduce_work_doubles() { return (size_t)RED_WORK_DOUBLES; }
int launch_reduce_sum(const double *part, int n, double *result, hipStream_t s) {
  hipLaunchKernelGGL((k_reduce_final<false>), dim3(1), dim3(RED_BLOCK), 0, s, part, n, result);
  EXAMG_CHECK_LAUNCH("k_reduce_final");
  return 0;
}
}  // namespace examg

static int red_blocks(long long total) {
  long long nb = (total + RED_BLOCK * 4 - 1) / (RED_BLOCK * 4);
  if (nb > RED_MAX_BLOCKS) nb = RED_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  return (int)nb;
}

extern "C" int examg_dot(const examg_layout_t *lx_, const double *x, const examg_layout_t *ly_, const double *y,
                         const int32_t *begin, const int32_t *end, double *result, void *work, examg_stream_t stream) {
  if (!lx_ || !x || !ly_ || !y || !begin || !end || !result || !work) { set_error("examg_dot: null argument"); return 1; }
  const Box box = make_box(begin, end);
  hipStream_t s = (hipStream_t)stream;
  if (box.count() == 0) return check_hip(hipMemsetAsync(result, 0, sizeof(double), s), "examg_dot memset");
  if (!box_inside(lx_, box, 0) || !box_inside(ly_, box, 0)) { set_error("examg_dot: box leaves an allocation"); return 1; }
  const int nb = red_blocks(box.count());
  if (box.n0() >= 128 && !lay_split(lx_) && !lay_split(ly_))
    hipLaunchKernelGGL(k_dot_rows, dim3(nb), dim3(RED_BLOCK), 0, s, make_layout(lx_), x, make_layout(ly_), y, box, (double *)work);
  else
    hipLaunchKernelGGL(k_dot_partial, dim3(nb), dim3(RED_BLOCK), 0, s, make_layout(lx_), x, make_layout(ly_), y, box, (double *)work);
  hipLaunchKernelGGL((k_reduce_final<false>), dim3(1), dim3(RED_BLOCK), 0, s, (const double *)work, nb, result);
  EXAMG_CHECK_LAUNCH("k_dot");
  return 0;
}

template <class F>
static int max_err_impl(const char *who, const examg_layout_t *l_, const double *x, const examg_geom_t *g, const F &fn,
                        const int32_t *begin, const int32_t *end, double *result, void *work, examg_stream_t stream) {
  if (!l_ || !x || !g || !begin || !end || !result || !work) { set_error("examg_max_err: null argument"); return 1; }
  const Box box = make_box(begin, end);
  hipStream_t s = (hipStream_t)stream;
  if (box.count() == 0) return check_hip(hipMemsetAsync(result, 0, sizeof(double), s), "examg_max_err memset");
  if (!box_inside(l_, box, 0)) { set_error("examg_max_err: box leaves the allocation"); return 1; }
  const int nb = red_blocks(box.count());
  hipLaunchKernelGGL((k_maxerr_partial<F>), dim3(nb), dim3(RED_BLOCK), 0, s, make_layout(l_), x, make_geom(g), fn, box, (double *)work);
  hipLaunchKernelGGL((k_reduce_final<true>), dim3(1), dim3(RED_BLOCK), 0, s, (const double *)work, nb, result);
  EXAMG_CHECK_LAUNCH(who);
  return 0;
}

template <class F>
static int fill_impl(const char *who, const examg_layout_t *l_, double *x, const examg_geom_t *g, const F &fn, const int32_t *begin,
                     const int32_t *end, examg_stream_t stream) {
  if (!l_ || !x || !g || !begin || !end) { set_error("examg_fill: null argument"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (!box_inside(l_, box, 0)) { set_error("examg_fill: box leaves the allocation"); return 1; }
  hipLaunchKernelGGL((k_fill_fn<F>), grid_for(box.count()), dim3(256), 0, (hipStream_t)stream, make_layout(l_), x, make_geom(g), fn, box);
  EXAMG_CHECK_LAUNCH(who);
  return 0;
}

template <class F>
static int dirichlet_impl(const char *who, const examg_layout_t *l, double *x, const examg_geom_t *g, const F &fn, uint32_t face_mask,
                          examg_stream_t stream, bool tangential_ghost = true) {
  if (!l || !x || !g) { set_error("examg_apply_dirichlet: null argument"); return 1; }
  FaceBoxes fb;
  fb.n = 0;
  fb.start[0] = 0;
  for (int d = 0; d < l->nd; ++d)
    for (int side = 0; side < 2; ++side) {
      if (!(face_mask & (1u << (2 * d + side)))) continue;
      int b[3] = {0, 0, 0}, e[3] = {1, 1, 1};
      for (int t = 0; t < l->nd; ++t) {
        if (t == d) {
          if (side == 0) { b[t] = 0; e[t] = l->dup_l[t]; }                       // DLB..DLE
          else { b[t] = l->dup_l[t] + l->inner[t]; e[t] = b[t] + l->dup_r[t]; }  // DRB..DRE
        } else {
          b[t] = tangential_ghost ? -l->ghost_l[t] : 0;                          // GLB..GRE (apply bc) or DLB..DRE (`only dup on boundary`)
          e[t] = l->dup_l[t] + l->inner[t] + l->dup_r[t] + (tangential_ghost ? l->ghost_r[t] : 0);
        }
      }
      Box bx{b[0], b[1], b[2], e[0], e[1], e[2]};
      if (bx.count() == 0) continue;
      fb.box[fb.n] = bx;
      fb.start[fb.n + 1] = fb.start[fb.n] + bx.count();
      ++fb.n;
    }
  if (fb.n == 0) return 0;
  hipLaunchKernelGGL((k_apply_dirichlet<F>), grid_for(fb.start[fb.n], 2048), dim3(256), 0, (hipStream_t)stream, make_layout(l), x, make_geom(g), fn, fb);
  EXAMG_CHECK_LAUNCH(who);
  return 0;
}

static bool expr_ok(const examg_expr_t *e) {
  if (!e || e->n < 1 || e->n > EXAMG_MAX_EXPR) { set_error("examg expression: null or too long"); return false; }
  int sp = 0;      // the stack discipline is checked on the host, the kernel trusts it
  for (int i = 0; i < e->n; ++i) {
    switch (e->op[i]) {
      case EXAMG_OP_CONST: case EXAMG_OP_X: case EXAMG_OP_Y: case EXAMG_OP_Z: ++sp; break;
      case EXAMG_OP_ADD: case EXAMG_OP_SUB: case EXAMG_OP_MUL: case EXAMG_OP_DIV: case EXAMG_OP_POW: case EXAMG_OP_MAX: case EXAMG_OP_MIN:
        if (sp < 2) { set_error("examg expression: stack underflow"); return false; }
        --sp;
        break;
      case EXAMG_OP_NEG: case EXAMG_OP_SIN: case EXAMG_OP_COS: case EXAMG_OP_EXP: case EXAMG_OP_SINH: case EXAMG_OP_COSH: case EXAMG_OP_SQRT:
      case EXAMG_OP_TAN: case EXAMG_OP_LOG: case EXAMG_OP_FABS: case EXAMG_OP_TANH:
        if (sp < 1) { set_error("examg expression: stack underflow"); return false; }
        break;
      default: set_error("examg expression: unknown opcode"); return false;
    }
    if (sp > 24) { set_error("examg expression: stack deeper than 24"); return false; }
  }
  if (sp != 1) { set_error("examg expression: must leave exactly one value"); return false; }
  return true;
}

extern "C" int examg_max_err_expr(const examg_layout_t *l_, const double *x, const examg_geom_t *g, const examg_expr_t *e,
                                  const int32_t *begin, const int32_t *end, double *result, void *work, examg_stream_t stream) {
  if (!expr_ok(e)) return 1;
  return max_err_impl("k_maxerr_expr", l_, x, g, ExprEval{*e}, begin, end, result, work, stream);
}

extern "C" int examg_fill_expr(const examg_layout_t *l_, double *x, const examg_geom_t *g, const examg_expr_t *e,
                               const int32_t *begin, const int32_t *end, examg_stream_t stream) {
  if (!expr_ok(e)) return 1;
  return fill_impl("k_fill_expr", l_, x, g, ExprEval{*e}, begin, end, stream);
}

extern "C" int examg_apply_dirichlet_expr(const examg_layout_t *l, double *x, const examg_geom_t *g, const examg_expr_t *e,
                                          uint32_t face_mask, examg_stream_t stream) {
  if (!expr_ok(e)) return 1;
  return dirichlet_impl("k_apply_dirichlet_expr", l, x, g, ExprEval{*e}, face_mask, stream);
}

extern "C" int examg_fill_dup_faces_expr(const examg_layout_t *l, double *x, const examg_geom_t *g, const examg_expr_t *e,
                                         uint32_t face_mask, examg_stream_t stream) {
  if (!expr_ok(e)) return 1;
  return dirichlet_impl("k_fill_dup_faces_expr", l, x, g, ExprEval{*e}, face_mask, stream, false);
}

extern "C" int examg_init_varcoeff7(const examg_layout_t *lc, double *cfield, const examg_geom_t *g, const examg_expr_t *a,
                                    const int32_t *begin, const int32_t *end, examg_stream_t stream) {
  if (!lc || !cfield || !g || !begin || !end) { set_error("examg_init_varcoeff7: null argument"); return 1; }
  if (!expr_ok(a)) return 1;
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (!box_inside(lc, box, 0)) { set_error("examg_init_varcoeff7: box leaves the allocation"); return 1; }
  hipLaunchKernelGGL(k_init_varcoeff, grid_for(box.count()), dim3(256), 0, (hipStream_t)stream, make_layout(lc), cfield, make_geom(g), ExprEval{*a}, box, lc->nd);
  EXAMG_CHECK_LAUNCH("k_init_varcoeff");
  return 0;
}

extern "C" int examg_transform_stencilfield(const examg_layout_t *lc, int nent, const double *src, double *dst, int to_entry_fastest,
                                            examg_stream_t stream) {
  if (!lc || !src || !dst || src == dst) { set_error("examg_transform_stencilfield: null argument, or src == dst"); return 1; }
  if (nent < 1 || nent > EXAMG_MAX_ENTRIES) { set_error("examg_transform_stencilfield: nent %d out of range", nent); return 1; }
  const long long size = make_layout(lc).size;
  if (size <= 0) return 0;
  const dim3 grid((unsigned)((size + 63) / 64));
  if (to_entry_fastest) hipLaunchKernelGGL((k_transform_sf<true>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, size, nent);
  else hipLaunchKernelGGL((k_transform_sf<false>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, size, nent);
  EXAMG_CHECK_LAUNCH("k_transform_sf");
  return 0;
}

extern "C" int examg_pack(const examg_layout_t *l, const double *x, double *buf, const int32_t *begin,
                          const int32_t *end, examg_stream_t stream) {
  if (!l || !x || !buf || !begin || !end) { set_error("examg_pack: null argument"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (!box_inside(l, box, 0)) { set_error("examg_pack: box leaves the allocation"); return 1; }
  hipLaunchKernelGGL((k_pack<true>), grid_for(box.count()), dim3(256), 0, (hipStream_t)stream, make_layout(l), const_cast<double *>(x), buf, box);
  EXAMG_CHECK_LAUNCH("k_pack");
  return 0;
}

extern "C" int examg_unpack(const examg_layout_t *l, double *x, const double *buf, const int32_t *begin,
                            const int32_t *end, examg_stream_t stream) {
  if (!l || !x || !buf || !begin || !end) { set_error("examg_unpack: null argument"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (!box_inside(l, box, 0)) { set_error("examg_unpack: box leaves the allocation"); return 1; }
  hipLaunchKernelGGL((k_pack<false>), grid_for(box.count()), dim3(256), 0, (hipStream_t)stream, make_layout(l), x, const_cast<double *>(buf), box);
  EXAMG_CHECK_LAUNCH("k_unpack");
  return 0;
}

extern "C" int examg_fill_random(double *x, int64_t n, uint64_t seed, examg_stream_t stream) {
  if (!x) { set_error("examg_fill_random: null argument"); return 1; }
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_fill_random, grid_for(n), dim3(256), 0, (hipStream_t)stream, x, (long long)n, (unsigned long long)seed);
  EXAMG_CHECK_LAUNCH("k_fill_random");
  return 0;
}

// ---- external fields (interfacing/ir/IR_CopyToExternalField.scala:31-90, IR_CopyFromExternalField.scala) ----------
// get<Name>(dest, slot) / set<Name>(src, slot): copy between a caller-owned array in its own layout and the internal
// field over [DLB - min(ghost_int, ghost_ext), DRE + min(ghost_int, ghost_ext)) per dimension; iterator coordinates
// coincide (0 = lower duplicate node in both layouts).
static int external_box(const examg_layout_t *li, const examg_layout_t *le, int32_t *b, int32_t *e) {
  if (li->nd != le->nd) { set_error("external field: dimensionality mismatch"); return 1; }
  for (int d = 0; d < 3; ++d) {
    if (d >= li->nd) { b[d] = 0; e[d] = 1; continue; }
    if (li->dup_l[d] != le->dup_l[d] || li->dup_r[d] != le->dup_r[d] || li->inner[d] != le->inner[d]) {
      set_error("external field: duplicate/inner extents differ in dimension %d", d);
      return 1;
    }
    const int gl = li->ghost_l[d] < le->ghost_l[d] ? li->ghost_l[d] : le->ghost_l[d];
    const int gr = li->ghost_r[d] < le->ghost_r[d] ? li->ghost_r[d] : le->ghost_r[d];
    b[d] = -gl;
    e[d] = li->dup_l[d] + li->inner[d] + li->dup_r[d] + gr;
  }
  return 0;
}

// The same field under another layout transformation: a copy over the whole allocation, every point through both index maps.
extern "C" int examg_transform_field(const examg_layout_t *lsrc, const double *src, const examg_layout_t *ldst, double *dst, examg_stream_t stream) {
  if (!lsrc || !src || !ldst || !dst) { set_error("examg_transform_field: null argument"); return 1; }
  if (src == dst) { set_error("examg_transform_field: out of place only"); return 1; }
  examg_layout_t a = *lsrc, b = *ldst;
  a.transform = b.transform = EXAMG_LAYOUT_PLAIN;
  if (memcmp(&a, &b, sizeof(a)) != 0) { set_error("examg_transform_field: the two layouts must differ in their transformation only"); return 1; }
  int32_t begin[3], end[3];
  for (int d = 0; d < 3; ++d) {
    begin[d] = -(lsrc->pad_l[d] + lsrc->ghost_l[d]);
    end[d] = begin[d] + lay_tot(lsrc, d);
  }
  return examg_axpby(lsrc, src, ldst, dst, 1.0, 0.0, begin, end, stream);
}

extern "C" int64_t examg_layout_size(const examg_layout_t *l) { return l ? (int64_t)make_layout(l).size : 0; }

extern "C" int examg_copy_to_external(const examg_layout_t *l_int, const double *x_int, const examg_layout_t *l_ext,
                                      double *dest, examg_stream_t stream) {
  if (!l_int || !x_int || !l_ext || !dest) { set_error("examg_copy_to_external: null argument"); return 1; }
  int32_t b[3], e[3];
  if (external_box(l_int, l_ext, b, e)) return 1;
  return examg_axpby(l_int, x_int, l_ext, dest, 1.0, 0.0, b, e, stream);
}

extern "C" int examg_copy_from_external(const examg_layout_t *l_ext, const double *src, const examg_layout_t *l_int,
                                        double *x_int, examg_stream_t stream) {
  if (!l_int || !x_int || !l_ext || !src) { set_error("examg_copy_from_external: null argument"); return 1; }
  int32_t b[3], e[3];
  if (external_box(l_int, l_ext, b, e)) return 1;
  return examg_axpby(l_ext, src, l_int, x_int, 1.0, 0.0, b, e, stream);
}

// ---- config 4: 27-entry stencil field of -div(a grad u) - k^2 u (trilinear elements, lumped mass); same expression
// order as oracle/examg_oracle.c:orc_init_helmholtz27 ----------------------------------------------------------
namespace examg {
__global__ void __launch_bounds__(256) k_init_helmholtz27(LayoutDev lc, double *cf, Geom g, ExprEval a, double ksq, Box box) {
  const long long total = box.count();
  const long long plane = lc.size;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int i0, i1, i2;
    unflatten(box, t, i0, i1, i2);
    const double h = g.h0;
    const double x = i0 * g.h0 + g.pb0, y = i1 * g.h1 + g.pb1, z = i2 * g.h2 + g.pb2;
    double ae[2][2][2];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx)
#pragma unroll
      for (int sy = 0; sy < 2; ++sy)
#pragma unroll
        for (int sz = 0; sz < 2; ++sz)
          ae[sx][sy][sz] = a(x + (sx ? 0.5 : -0.5) * h, y + (sy ? 0.5 : -0.5) * h, z + (sz ? 0.5 : -0.5) * h);
    const long long k = lidx(lc, i0, i1, i2);
    int ent = 1;
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int nnz = (dx != 0) + (dy != 0) + (dz != 0);
          double s = 0.0;
          bool first = true;
#pragma unroll
          for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int sy = 0; sy < 2; ++sy)
#pragma unroll
              for (int sz = 0; sz < 2; ++sz) {
                const bool okx = dx == 0 || (dx > 0) == (sx == 1), oky = dy == 0 || (dy > 0) == (sy == 1),
                           okz = dz == 0 || (dz > 0) == (sz == 1);
                if (okx && oky && okz) { s = first ? ae[sx][sy][sz] : s + ae[sx][sy][sz]; first = false; }
              }
          const double kf = nnz == 0 ? (1.0 / 3.0) : (nnz == 1 ? 0.0 : (-1.0 / 12.0));
          double c = (s * kf) / (h * h);
          if (nnz == 0) { c = c - ksq; cf[k] = c; }
          else { cf[k + (long long)ent * plane] = c; ++ent; }
        }
  }
}
}  // namespace examg

extern "C" int examg_init_helmholtz27(const examg_layout_t *lc, double *cfield, const examg_geom_t *g, const examg_expr_t *a,
                                      double ksq, const int32_t *begin, const int32_t *end, examg_stream_t stream) {
  if (!lc || !cfield || !g || !begin || !end) { set_error("examg_init_helmholtz27: null argument"); return 1; }
  if (!expr_ok(a)) return 1;
  if (lc->nd != 3) { set_error("examg_init_helmholtz27: 3-D only"); return 1; }
  const Box box = make_box(begin, end);
  if (box.count() == 0) return 0;
  if (!box_inside(lc, box, 0)) { set_error("examg_init_helmholtz27: box leaves the allocation"); return 1; }
  hipLaunchKernelGGL(examg::k_init_helmholtz27, grid_for(box.count()), dim3(256), 0, (hipStream_t)stream, make_layout(lc), cfield, make_geom(g), examg::ExprEval{*a}, ksq, box);
  EXAMG_CHECK_LAUNCH("k_init_helmholtz27");
  return 0;
}
