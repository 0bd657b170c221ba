// Lab (timing only): one Jacobi step on a 27-entry stencil field at 512^3 with the coefficients (a) as 216-byte records per point,
// transposed through a wave-private LDS strip (the shipped k_stencilfield27_rec, simplified), (b) BLOCKED: [x / 64][entry][x % 64] --
// the transformation `[x, y, z, i] => [x % 64, i, x / 64, y, z]` -- where a lane reads its 27 entries with coalesced 8-byte loads and
// nothing passes through LDS.  Same bytes, same u accesses; prints ms and GB/s of 240 B per point.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/sf27_blocked_lab.bin tools/lab/sf27_blocked_lab.hip && tools/lab/sf27_blocked_lab.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int N = 512, P0 = N + 3, P1 = N + 3;            // u: ghost 1, rows of 515
constexpr int C0 = N + 1, CB = (C0 + 63) / 64;            // coefficient rows of 513 points = 9 blocks
struct Off { long long o[27]; };

__global__ void __launch_bounds__(256) k_records(const double *__restrict__ u, const double *__restrict__ f, double *__restrict__ out,
                                                 const double *__restrict__ cf, Off uo, double w, long long ntiles) {
  __shared__ double strip[4][64 * 27];
  const int lane = threadIdx.x, wv = threadIdx.y;
  const long long tile = (long long)blockIdx.x * 4 + wv;
  if (tile >= ntiles) return;
  const long long row = tile / 8;
  const int tx = (int)(tile % 8);
  const int y = 1 + (int)(row % (N - 1)), z = 1 + (int)(row / (N - 1));
  const int x = 1 + tx * 64 + lane;
  const bool ok = x < N;
  const int xc = ok ? x : N - 1;
  const long long rec0 = (((long long)z * C0 + y) * C0 + (1 + tx * 64)) * 27;
  double raw[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) raw[i] = cf[rec0 + 64 * i + lane];
  const long long iu = ((long long)(z + 1) * P1 + (y + 1)) * P0 + (xc + 1);
  double v[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) v[k] = u[iu + uo.o[k]];
  const double fv = f[((long long)z * C0 + y) * C0 + xc];
  double *sb = strip[wv];
#pragma unroll
  for (int i = 0; i < 27; ++i) sb[64 * i + lane] = raw[i];
  double c[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) c[k] = sb[27 * lane + k];
  double acc = c[0] * v[0];
#pragma unroll
  for (int k = 1; k < 27; ++k) acc = acc + c[k] * v[k];
  const double r = v[0] + ((1.0 / c[0]) * w) * (fv - acc);
  if (ok) __builtin_nontemporal_store(r, out + iu);
}

template <int RUN>
__global__ void __launch_bounds__(256) k_blocked(const double *__restrict__ u, const double *__restrict__ f, double *__restrict__ out,
                                                 const double *__restrict__ cf, Off uo, double w, long long ntiles) {
  const int lane = threadIdx.x, wv = threadIdx.y;
  const long long t0 = ((long long)blockIdx.x * 4 + wv) * RUN;
#pragma unroll
  for (int q = 0; q < RUN; ++q) {
    const long long tile = t0 + q;
    if (tile >= ntiles) return;
    const long long row = tile / 8;
    const int tx = (int)(tile % 8);
    const int y = 1 + (int)(row % (N - 1)), z = 1 + (int)(row / (N - 1));
    const int x = tx * 64 + lane;                 // block-aligned window: columns 64 tx .. 64 tx + 63 of the allocation
    const bool ok = x >= 1 && x < N;
    const int xc = ok ? x : 1;
    const long long blk = ((((long long)z * C0 + y) * CB + tx) * 27) * 64 + lane;
    double c[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) c[k] = cf[blk + 64 * k];
    const long long iu = ((long long)(z + 1) * P1 + (y + 1)) * P0 + (xc + 1);
    double v[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) v[k] = u[iu + uo.o[k]];
    const double fv = f[((long long)z * C0 + y) * C0 + xc];
    double acc = c[0] * v[0];
#pragma unroll
    for (int k = 1; k < 27; ++k) acc = acc + c[k] * v[k];
    const double r = v[0] + ((1.0 / c[0]) * w) * (fv - acc);
    if (ok) __builtin_nontemporal_store(r, out + iu);
  }
}

int main() {
  const long long usz = (long long)P0 * P1 * (N + 3), fsz = (long long)C0 * C0 * C0;
  const long long csz_rec = fsz * 27, csz_blk = (long long)C0 * C0 * CB * 27 * 64;
  double *u, *f, *out, *cr, *cb;
  CHECK(hipMalloc(&u, usz * 8)); CHECK(hipMalloc(&out, usz * 8)); CHECK(hipMalloc(&f, fsz * 8));
  CHECK(hipMalloc(&cr, csz_rec * 8)); CHECK(hipMalloc(&cb, csz_blk * 8));
  CHECK(hipMemset(u, 0, usz * 8)); CHECK(hipMemset(f, 0, fsz * 8)); CHECK(hipMemset(cr, 0x3f, csz_rec * 8)); CHECK(hipMemset(cb, 0x3f, csz_blk * 8));
  Off uo;
  int k = 0;
  uo.o[k++] = 0;
  for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx)
    if (dx || dy || dz) uo.o[k++] = dx + (long long)P0 * dy + (long long)P0 * P1 * dz;
  const long long ntiles = 8LL * (N - 1) * (N - 1);
  const double pts = (double)(N - 1) * (N - 1) * (N - 1);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto launch, const char *name) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-34s %.3f ms  %.0f GB/s of 240 B per point\n", name, ms, 240.0 * pts / ms / 1e6);
  };
  dim3 blk(64, 4);
  for (int rep = 0; rep < 2; ++rep) {
    time([&] { hipLaunchKernelGGL(k_records, dim3((unsigned)((ntiles + 3) / 4)), blk, 0, 0, u, f, out, cr, uo, 0.8, ntiles); }, "records + LDS strip, 1 tile");
    time([&] { hipLaunchKernelGGL(k_blocked<1>, dim3((unsigned)((ntiles + 3) / 4)), blk, 0, 0, u, f, out, cb, uo, 0.8, ntiles); }, "blocked, 1 tile per wave");
    time([&] { hipLaunchKernelGGL(k_blocked<2>, dim3((unsigned)((ntiles / 2 + 3) / 4 + 1)), blk, 0, 0, u, f, out, cb, uo, 0.8, ntiles); }, "blocked, 2 tiles per wave");
    time([&] { hipLaunchKernelGGL(k_blocked<4>, dim3((unsigned)((ntiles / 4 + 3) / 4 + 1)), blk, 0, 0, u, f, out, cb, uo, 0.8, ntiles); }, "blocked, 4 tiles per wave");
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
