// Feasibility probe of the peer-write transport (round 3): N processes on ONE device share fine-grained regions through HIP IPC,
// write into each other's regions and signal / wait with device-side sequence flags -- no host involvement between kernels,
// also inside a hipGraph.  Usage: ipc_probe <dir> <rank> <nranks> [alloc: 0 finegrained | 1 uncached | 2 plain]
// (start one process per rank; handles travel through files in <dir>).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <string>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      fprintf(stderr, "rank %d: %s -> %s (line %d)\n", g_rank, #x, hipGetErrorString(e_), __LINE__); \
      exit(2);                                                                                  \
    }                                                                                           \
  } while (0)

static int g_rank = -1;

struct Region {          // lives at the start of every rank's shared allocation
  unsigned long long ready[8];   // written by the peer that sends to me
  unsigned long long ack[8];     // written by the peer that received from me
  double slab[2][1 << 16];       // two receive buffers
};

struct Local {           // ordinary device memory of the rank
  unsigned long long seq_out, seq_in, err;
  unsigned int done_out, done_in;
};

__device__ __forceinline__ bool wait_ge(const unsigned long long *flag, unsigned long long want, long long timeout_ticks) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
    if (wall_clock64() - t0 > timeout_ticks) return false;
    __builtin_amdgcn_s_sleep(8);
  }
  return true;
}

// send n doubles into the peer's slab[seq & 1], then publish seq + 1 in the peer's ready flag
__global__ void k_send(const double *src, int n, Region *peer, Region *mine, Local *loc, long long timeout) {
  __shared__ unsigned long long s_seq;
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    const unsigned long long seq = loc->seq_out;
    s_seq = seq;
    // buffer seq & 1 is free once message seq - 2 was consumed: ack >= seq - 1
    s_ok = (seq < 2) ? 1 : (wait_ge(&mine->ack[0], seq - 1, timeout) ? 1 : 0);
    if (!s_ok) atomicExch(&loc->err, 1ull);
  }
  __syncthreads();
  const unsigned long long seq = s_seq;
  if (s_ok) {
    double *dst = peer->slab[seq & 1];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = __hip_atomic_fetch_add(&loc->done_out, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      loc->done_out = 0;
      loc->seq_out = seq + 1;
      if (s_ok) __hip_atomic_store(&peer->ready[0], seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void k_recv(double *dst, int n, Region *peer, Region *mine, Local *loc, long long timeout) {
  __shared__ unsigned long long s_seq;
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    const unsigned long long seq = loc->seq_in;
    s_seq = seq;
    s_ok = wait_ge(&mine->ready[0], seq + 1, timeout) ? 1 : 0;
    if (!s_ok) atomicExch(&loc->err, 2ull);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  const unsigned long long seq = s_seq;
  if (s_ok) {
    const double *src = mine->slab[seq & 1];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
      dst[i] = __builtin_nontemporal_load(src + i);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = __hip_atomic_fetch_add(&loc->done_in, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      loc->done_in = 0;
      loc->seq_in = seq + 1;
      if (s_ok) __hip_atomic_store(&peer->ack[0], seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void k_fill(double *x, int n, double base) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = base + i;
}
__global__ void k_add1(double *x, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] += 1.0;
}

static void write_file(const std::string &p, const void *d, size_t n) {
  std::string tmp = p + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb");
  fwrite(d, 1, n, f);
  fclose(f);
  rename(tmp.c_str(), p.c_str());
}
static void read_file(const std::string &p, void *d, size_t n) {
  for (int t = 0; t < 3000; ++t) {
    FILE *f = fopen(p.c_str(), "rb");
    if (f) {
      size_t got = fread(d, 1, n, f);
      fclose(f);
      if (got == n) return;
    }
    usleep(10000);
  }
  fprintf(stderr, "rank %d: timed out waiting for %s\n", g_rank, p.c_str());
  exit(3);
}
static void barrier(const std::string &dir, int rank, int n, int &epoch) {
  char c = 1;
  write_file(dir + "/bar" + std::to_string(epoch) + "_" + std::to_string(rank), &c, 1);
  for (int r = 0; r < n; ++r) read_file(dir + "/bar" + std::to_string(epoch) + "_" + std::to_string(r), &c, 1);
  ++epoch;
}

int main(int argc, char **argv) {
  if (argc < 4) return 1;
  const std::string dir = argv[1];
  const int rank = atoi(argv[2]), nranks = atoi(argv[3]);
  const int alloc = argc > 4 ? atoi(argv[4]) : 0;
  g_rank = rank;
  int epoch = 0;
  CK(hipSetDevice(0));
  Region *mine = nullptr;
  if (alloc == 0) CK(hipExtMallocWithFlags((void **)&mine, sizeof(Region), hipDeviceMallocFinegrained));
  else if (alloc == 1) CK(hipExtMallocWithFlags((void **)&mine, sizeof(Region), hipDeviceMallocUncached));
  else CK(hipMalloc((void **)&mine, sizeof(Region)));
  CK(hipMemset(mine, 0, sizeof(Region)));
  CK(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  CK(hipIpcGetMemHandle(&h, mine));
  write_file(dir + "/h" + std::to_string(rank), &h, sizeof(h));
  const int peer_rank = (rank + 1) % nranks;      // ring: send to rank+1, receive from rank-1
  hipIpcMemHandle_t hp;
  read_file(dir + "/h" + std::to_string(peer_rank), &hp, sizeof(hp));
  Region *peer = nullptr;
  CK(hipIpcOpenMemHandle((void **)&peer, hp, hipIpcMemLazyEnablePeerAccess));
  // the rank that sends to me acks into... my ack flag is written by my receiver = peer_rank; my ready flag by rank-1: with a ring
  // of 2 both are the same process; with more ranks the receiver of my messages is peer_rank (acks come from there) -- but k_recv
  // acks into `peer` = the rank it sends to, which is only right for 2 ranks.  For n > 2 open the sender's region too.
  const int from_rank = (rank + nranks - 1) % nranks;
  Region *from = peer;
  if (from_rank != peer_rank) {
    hipIpcMemHandle_t hf;
    read_file(dir + "/h" + std::to_string(from_rank), &hf, sizeof(hf));
    CK(hipIpcOpenMemHandle((void **)&from, hf, hipIpcMemLazyEnablePeerAccess));
  }
  printf("rank %d: region %p, peer(%d) %p, from(%d) %p, alloc kind %d\n", rank, (void *)mine, peer_rank, (void *)peer, from_rank, (void *)from, alloc);

  Local *loc;
  CK(hipMalloc((void **)&loc, sizeof(Local)));
  CK(hipMemset(loc, 0, sizeof(Local)));
  const int n = 1 << 16;
  double *a, *b;
  CK(hipMalloc((void **)&a, n * 8));
  CK(hipMalloc((void **)&b, n * 8));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int clk_khz = 100000;   // wall_clock64: 100 MHz
  const long long timeout = 5ll * clk_khz * 1000;   // 5 s
  CK(hipDeviceSynchronize());
  barrier(dir, rank, nranks, epoch);

  // ---- test 1: token ring, 200 rounds; every round each rank sends a (base + i) and receives the previous rank's --------
  const int rounds = 200;
  hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, s, a, n, 1000.0 * rank);
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < rounds; ++r) {
    hipLaunchKernelGGL(k_send, dim3(8), dim3(256), 0, s, a, n, peer, mine, loc, timeout);
    hipLaunchKernelGGL(k_recv, dim3(8), dim3(256), 0, s, b, n, from, mine, loc, timeout);
    hipLaunchKernelGGL(k_add1, dim3(64), dim3(256), 0, s, a, n);
  }
  CK(hipStreamSynchronize(s));
  auto t1 = std::chrono::steady_clock::now();
  Local hl;
  CK(hipMemcpy(&hl, loc, sizeof(hl), hipMemcpyDeviceToHost));
  double *hb = (double *)malloc(n * 8);
  CK(hipMemcpy(hb, b, n * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < n; ++i) bad += hb[i] != 1000.0 * from_rank + i + (rounds - 1);
  printf("rank %d: eager ring %d rounds: %.1f us/round, err %llu, seq %llu/%llu, mismatches %d\n", rank, rounds,
         std::chrono::duration<double, std::micro>(t1 - t0).count() / rounds, hl.err, hl.seq_out, hl.seq_in, bad);
  barrier(dir, rank, nranks, epoch);

  // ---- test 2: the same three launches x 10 captured into a graph, replayed 50 times ------------------------------------
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int r = 0; r < 10; ++r) {
    hipLaunchKernelGGL(k_send, dim3(8), dim3(256), 0, s, a, n, peer, mine, loc, timeout);
    hipLaunchKernelGGL(k_recv, dim3(8), dim3(256), 0, s, b, n, from, mine, loc, timeout);
    hipLaunchKernelGGL(k_add1, dim3(64), dim3(256), 0, s, a, n);
  }
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < 50; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  t1 = std::chrono::steady_clock::now();
  CK(hipMemcpy(&hl, loc, sizeof(hl), hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb, b, n * 8, hipMemcpyDeviceToHost));
  bad = 0;
  for (int i = 0; i < n; ++i) bad += hb[i] != 1000.0 * from_rank + i + (rounds + 500 - 1);
  printf("rank %d: graph ring 500 rounds: %.1f us/round, err %llu, seq %llu/%llu, mismatches %d\n", rank,
         std::chrono::duration<double, std::micro>(t1 - t0).count() / 500, hl.err, hl.seq_out, hl.seq_in, bad);
  barrier(dir, rank, nranks, epoch);

  // ---- test 3: a receive with nobody sending must time out (short timeout), not hang -------------------------------------
  if (rank == 0) {
    hipLaunchKernelGGL(k_recv, dim3(8), dim3(256), 0, s, b, n, from, mine, loc, (long long)clk_khz * 200);   // 0.2 s
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(&hl, loc, sizeof(hl), hipMemcpyDeviceToHost));
    printf("rank %d: lonely receive returned with err %llu (expected 2)\n", rank, hl.err);
  }
  barrier(dir, rank, nranks, epoch);
  CK(hipIpcCloseMemHandle(peer));
  if (from != peer) CK(hipIpcCloseMemHandle(from));
  CK(hipFree(mine));
  printf("rank %d: done\n", rank);
  return 0;
}
