// Cost of a device-wide barrier inside a persistent kernel on gfx950, against back-to-back launches: what one kernel for the launch-bound
// coarse levels of the V-cycle (<= 64^3) can gain.  Every phase is a 7-point sweep over an n^3 array (ping-pong), G workgroups of 256
// threads, separated by (a) a grid barrier with agent-scope release / acquire or (b) a kernel boundary (same stream; also from a hipGraph).
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/barrier_probe.bin tools/lab/barrier_probe.hip && tools/lab/barrier_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned *count, unsigned *gen, unsigned nwg) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gen, g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__device__ __forceinline__ void sweep(const double *__restrict__ x, double *__restrict__ y, int n, int wg, int nwg) {
  const int s1 = n + 2, s2 = (n + 2) * (n + 2);
  const long long total = (long long)n * n * n;
  for (long long t = (long long)wg * 256 + threadIdx.x; t < total; t += (long long)nwg * 256) {
    const int i = (int)(t % n), j = (int)((t / n) % n), k = (int)(t / ((long long)n * n));
    const long long c = (i + 1) + (long long)s1 * (j + 1) + (long long)s2 * (k + 1);
    y[c] = 0.4 * x[c] + 0.1 * (x[c - 1] + x[c + 1] + x[c - s1] + x[c + s1] + x[c - s2] + x[c + s2]);
  }
}

__global__ void __launch_bounds__(256) k_persistent(double *a, double *b, int n, int phases, unsigned *bar) {
  for (int p = 0; p < phases; ++p) {
    sweep((p & 1) ? b : a, (p & 1) ? a : b, n, blockIdx.x, gridDim.x);
    grid_barrier(bar, bar + 32, gridDim.x);
  }
}

__global__ void __launch_bounds__(256) k_one(const double *x, double *y, int n) { sweep(x, y, n, blockIdx.x, gridDim.x); }

int main() {
  const int phases = 64;
  unsigned *bar;
  CK(hipMalloc(&bar, 256));
  CK(hipMemset(bar, 0, 256));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int n : {16, 32, 64, 128}) {
    const size_t sz = (size_t)(n + 2) * (n + 2) * (n + 2);
    double *a, *b;
    CK(hipMalloc(&a, sz * 8));
    CK(hipMalloc(&b, sz * 8));
    CK(hipMemset(a, 0, sz * 8));
    CK(hipMemset(b, 0, sz * 8));
    for (int G : {32, 64, 128, 256, 512}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k_persistent, dim3(G), dim3(256), 0, s, a, b, n, phases, bar);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("n %3d  G %3d  persistent: %6.2f us per phase\n", n, G, best * 1e3f / phases);
    }
    // the same phases as separate launches, replayed from a graph
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(k_one, dim3(256), dim3(256), 0, s, (p & 1) ? b : a, (p & 1) ? a : b, n);
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(exec, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("n %3d  G 256  graph of %d launches: %6.2f us per phase\n", n, phases, best * 1e3f / phases);
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
    CK(hipFree(a));
    CK(hipFree(b));
  }
  return 0;
}
