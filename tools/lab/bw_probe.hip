// Bandwidth probe for MI355X: what does a plain 2-read + 1-write stream reach, by access shape?
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/bw_probe tools/lab/bw_probe.hip ; run on the GPU box.
// Used to place the Jacobi kernel against the practical streaming ceiling of this device.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef double dv2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0 triad o = a + w*b ; 1 copy o = a ; 2 read a+b ; 3 write o = w
// CHUNK: each block owns one contiguous range (else grid-stride); UNR: independent 16-byte accesses per thread per trip
template <int MODE, bool NT, bool CHUNK, int UNR>
__global__ void __launch_bounds__(256) k(double *__restrict__ o, const double *__restrict__ a, const double *__restrict__ b,
                                         long long n2, double w) {
  const dv2 *a2 = (const dv2 *)a;
  const dv2 *b2 = (const dv2 *)b;
  dv2 *o2 = (dv2 *)o;
  long long start, end, stride;
  if (CHUNK) {
    const long long per = (n2 + gridDim.x - 1) / gridDim.x;
    start = per * blockIdx.x + threadIdx.x;
    end = per * (blockIdx.x + 1);
    if (end > n2) end = n2;
    stride = blockDim.x;
  } else {
    start = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    end = n2;
    stride = (long long)gridDim.x * blockDim.x;
  }
  dv2 acc = {0, 0};
  long long i = start;
  for (; i + (UNR - 1) * stride < end; i += UNR * stride) {
    dv2 va[UNR], vb[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (MODE != 3) va[u] = a2[i + u * stride];
      if (MODE == 0 || MODE == 2) vb[u] = b2[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      dv2 r;
      if (MODE == 0) r = va[u] + w * vb[u];
      else if (MODE == 1) r = va[u];
      else if (MODE == 3) r = dv2{w, w};
      else { acc += va[u] + vb[u]; continue; }
      if (NT) { __builtin_nontemporal_store(r.x, &o[2 * (i + u * stride)]); __builtin_nontemporal_store(r.y, &o[2 * (i + u * stride) + 1]); }
      else o2[i + u * stride] = r;
    }
  }
  for (; i < end; i += stride) {
    dv2 r;
    if (MODE == 0) r = a2[i] + w * b2[i];
    else if (MODE == 1) r = a2[i];
    else if (MODE == 3) r = dv2{w, w};
    else { acc += a2[i] + b2[i]; continue; }
    o2[i] = r;
  }
  if (MODE == 2 && acc.x + acc.y == 123.456) o[0] = acc.x;
}

template <int MODE, bool NT, bool CHUNK, int UNR>
static void run(const char *name, double *o, double *a, double *b, long long n, int blocks, double bytes_per_elem) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int reps = 20;
  hipLaunchKernelGGL((k<MODE, NT, CHUNK, UNR>), dim3(blocks), dim3(256), 0, 0, o, a, b, n / 2, 0.5);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<MODE, NT, CHUNK, UNR>), dim3(blocks), dim3(256), 0, 0, o, a, b, n / 2, 0.5);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("%-10s nt=%d chunk=%d unr=%d blocks=%6d  %.4f ms  %7.0f GB/s\n", name, (int)NT, (int)CHUNK, UNR, blocks, ms,
         bytes_per_elem * n / ms / 1e6);
  fflush(stdout);
}

// calibration kernels for the FETCH_SIZE / WRITE_SIZE counters: known byte counts in the stencil kernels' access shape
struct __attribute__((packed, aligned(8))) pk2 { double a, b; };
__global__ void __launch_bounds__(256) k_read_unaligned(double *o, const double *a, long long n2) {
  // 16-byte loads at 8-byte alignment (a is offset by one double), like rows of the 515-wide reference layout
  dv2 acc = {0, 0};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
    const pk2 v = *reinterpret_cast<const pk2 *>(a + 2 * i);
    acc.x += v.a;
    acc.y += v.b;
  }
  if (acc.x + acc.y == 123.456) o[0] = acc.x;
}
__global__ void __launch_bounds__(256) k_write_unaligned(double *o, long long n2, double w) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
    __builtin_nontemporal_store(w, &o[2 * i]);
    __builtin_nontemporal_store(w, &o[2 * i + 1]);
  }
}

int main(int argc, char **argv) {
  if (argc > 1) {  // calibration mode: one launch each, known traffic (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
    const long long n = 515LL * 515 * 515;
    double *a, *o;
    CHECK(hipMalloc(&a, (n + 2) * 8));
    CHECK(hipMalloc(&o, (n + 2) * 8));
    CHECK(hipMemset(a, 0, (n + 2) * 8));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<2, false, false, 1>), dim3(4096), dim3(256), 0, 0, o, a, a, n / 2, 0.5);   // reads a twice: 2 * n * 8 B
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_read_unaligned, dim3(4096), dim3(256), 0, 0, o, a + 1, n / 2);           // n * 8 B at 8-byte alignment
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<3, true, false, 1>), dim3(4096), dim3(256), 0, 0, o, a, a, n / 2, 0.5);    // writes n * 8 B
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_write_unaligned, dim3(4096), dim3(256), 0, 0, o + 1, n / 2, 0.5);        // writes n * 8 B, 8-byte aligned
    CHECK(hipDeviceSynchronize());
    printf("calibration: n*8 = %lld bytes\n", n * 8);
    return 0;
  }
  const long long n = 515LL * 515 * 515;  // one 512^3 field slot in the reference layout
  double *a, *b, *o;
  CHECK(hipMalloc(&a, n * 8));
  CHECK(hipMalloc(&b, n * 8));
  CHECK(hipMalloc(&o, n * 8));
  CHECK(hipMemset(a, 0, n * 8));
  CHECK(hipMemset(b, 0, n * 8));
  hipLaunchKernelGGL((k<3, false, false, 1>), dim3(4096), dim3(256), 0, 0, a, a, b, n / 2, 1.25);
  hipLaunchKernelGGL((k<3, false, false, 1>), dim3(4096), dim3(256), 0, 0, b, a, b, n / 2, 0.75);
  const int bl[] = {512, 1024, 2048, 4096, 8192, 16384, 65536};
  for (int blocks : bl) {
    run<0, false, false, 1>("triad", o, a, b, n, blocks, 24);
    run<0, true, false, 1>("triad", o, a, b, n, blocks, 24);
    run<0, false, false, 4>("triad", o, a, b, n, blocks, 24);
    run<0, true, false, 4>("triad", o, a, b, n, blocks, 24);
    run<0, false, true, 1>("triad", o, a, b, n, blocks, 24);
    run<0, true, true, 4>("triad", o, a, b, n, blocks, 24);
  }
  for (int blocks : bl) {
    run<1, false, false, 4>("copy", o, a, b, n, blocks, 16);
    run<1, true, false, 4>("copy", o, a, b, n, blocks, 16);
    run<2, false, false, 4>("read2", o, a, b, n, blocks, 16);
    run<3, false, false, 4>("write", o, a, b, n, blocks, 8);
    run<3, true, false, 4>("write", o, a, b, n, blocks, 8);
  }
  return 0;
}
