// Peer-write transport of libexamg: `communicate <field>`, the scalar all-reduce and the coarse-level all-gather between the
// blocks of ONE node without a communication library -- every rank owns a region of uncached device memory, exported through
// HIP IPC and mapped by its neighbours (xGMI peer mapping across GPUs, a plain mapping when several ranks share one device);
// a send kernel packs the field box STRAIGHT INTO the neighbour's receive slab (posted writes over the pair's own xGMI link)
// and publishes a sequence number there; the receive kernel waits for that number and unpacks.  All ordering is done by
// device-side flags with counters that live in device memory, so every call is stream-ordered, needs no host round trip and
// replays from a hipGraph (the RCCL group calls of examg_comm.hip hang inside a stream capture on this stack).
//
// Replaces the MPI branch of IR_CommunicateFunction.compileBody (Compiler/src/exastencils/communication/ir/
// IR_CommunicateFunction.scala:412-471: pack -> MPI_Isend / MPI_Irecv -> wait -> unpack, IR_RemoteSend.scala:48-58,
// IR_RemoteRecv.scala:50-65) the way SURVEY.md section 5.8 names as the alternative transport -- direct peer writes with
// unpack-from-own-slab -- and MPI_Allreduce / the gather of the agglomerated coarse levels
// (parallelization/api/mpi/MPI_Reduction.scala:100-126).
//
// Protocol, per directed channel (axis d, side s of the sender = axis d, side 1-s of the receiver); message number q = 0, 1, ..:
//   sender    wait until ack[channel] >= q - 1      (slab q & 1 of the receiver is free: message q - 2 was unpacked)
//             pack box -> receiver.slab[channel'][q & 1];  fence(system);  receiver.ready[channel'] = q + 1
//   receiver  wait until ready[channel'] >= q + 1;  unpack slab[channel'][q & 1] -> box;  sender.ack[channel] = q + 1
// Both sides execute the same sequence of exchanges (one program, many blocks), as MPI requires of them too.  A wait that sees
// no progress for EXAMG_PEER_TIMEOUT_MS (default 120 s; the tests use a short one) sets the communicator's error word and every later wait returns at
// once: a lost peer ends in an error at the next examg_comm_status(), never in a kernel that spins forever.
//
// Why a staging slab and not a write into the neighbour's ghost planes: field arrays are ordinary (coarse-grained) device memory,
// which the owning GPU caches in its L2 without probing -- a peer's write is only safe into memory mapped uncached on both sides.
#include <stdlib.h>
#include <unistd.h>

#include <map>
#include <mutex>
#include <string>

#include "examg_comm_internal.h"

namespace examg {

enum { PEER_MAX_RANKS = 64, PEER_RED_MAX = 8, PEER_HEADER_BYTES = 16384 };

struct PeerHeader {
  unsigned long long ready[6];   // [2 d + side]: messages that arrived through my face (d, side); written by that neighbour
  unsigned long long ack[6];     // [2 d + side]: messages I sent through my face (d, side) that the neighbour has unpacked
  unsigned long long red_ready[2][PEER_MAX_RANKS];          // all-reduce: contribution of rank r for epoch parity p is in
  double red_val[2][PEER_MAX_RANKS][PEER_RED_MAX];
  unsigned long long gat_ready[PEER_MAX_RANKS];             // all-gather: piece of rank r has arrived
  unsigned long long gat_ack[PEER_MAX_RANKS];               // all-gather: rank r has consumed my piece
};
static_assert(sizeof(PeerHeader) <= PEER_HEADER_BYTES, "PeerHeader must fit its slot");

struct PeerLocal {               // ordinary device memory of the rank: the counters that make the calls replayable
  unsigned long long seq_out[6], seq_in[6];
  unsigned int done_out[6], done_in[6];
  unsigned long long red_epoch;
  unsigned long long gat_seq;
  unsigned int gat_done_send[PEER_MAX_RANKS], gat_done_recv[PEER_MAX_RANKS], gat_done_all;
  unsigned long long err;        // 0, or (code << 8 | channel / rank) of the first wait that gave up
};

struct PeerWire {                // what travels between the ranks, inside EXAMG_PEER_HANDLE_BYTES
  hipIpcMemHandle_t mem;
  unsigned long long region_bytes, slab_bytes, gather_bytes;
  unsigned long long self_ptr;   // the region's address in the owner's process: what a rank of the SAME process uses (no IPC mapping of one's own memory)
  int rank, pid;
};
static_assert(sizeof(PeerWire) <= EXAMG_PEER_HANDLE_BYTES, "EXAMG_PEER_HANDLE_BYTES too small");

struct PeerState {
  char *mine = nullptr;                 // own region: header | 6 channels x 2 slabs | nranks gather pieces
  char *remote[PEER_MAX_RANKS] = {};    // mapped regions (remote[rank] == mine)
  bool remote_ipc[PEER_MAX_RANKS] = {}; // remote[r] came from hipIpcOpenMemHandle (not the own region, not a region of this process)
  char **remote_dev = nullptr;          // the same table in device memory
  PeerLocal *loc = nullptr;
  size_t region_bytes = 0, slab_bytes = 0, gather_bytes = 0;
  long long timeout_ticks = 0;
  bool connected = false;
};

namespace {

__device__ __forceinline__ bool peer_wait_ge(const unsigned long long *flag, unsigned long long want, PeerLocal *loc, long long timeout) {
  if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= want) return true;
  if (__hip_atomic_load(&loc->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;   // an earlier wait gave up: do not wait again
  const long long t0 = wall_clock64();   // 100 MHz
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
    if (wall_clock64() - t0 > timeout) return false;
    __builtin_amdgcn_s_sleep(4);
  }
  return true;
}

__device__ __forceinline__ void unflat(const Box &box, long long t, int &i0, int &i1, int &i2) {
  const int n0 = box.n0(), n1 = box.n1();
  const long long r = t / n0;
  i0 = box.b0 + (int)(t - r * n0);
  i2 = box.b2 + (int)(r / n1);
  i1 = box.b1 + (int)(r - (long long)(r / n1) * n1);
}

struct SendDesc {
  Box box;
  char *remote;   // receiver's region
  int ch;         // my channel 2 d + side
  int rch;        // the receiver's channel 2 d + (1 - side)
};
struct SendList {
  SendDesc m[6];
};
struct RecvDesc {
  Box box;
  char *remote;   // sender's region (for the ack)
  int ch;         // my channel 2 d + side
  int rch;        // the sender's channel 2 d + (1 - side)
};
struct RecvList {
  RecvDesc m[6];
};

__device__ __forceinline__ double *slab_of(char *region, int ch, unsigned long long q, size_t slab_bytes) {
  return reinterpret_cast<double *>(region + PEER_HEADER_BYTES + (size_t)(2 * ch + (int)(q & 1)) * slab_bytes);
}

__global__ void __launch_bounds__(256) k_peer_send(LayoutDev l, const double *x, SendList list, char *mine, PeerLocal *loc, size_t slab_bytes,
                                                   long long timeout) {
  __shared__ unsigned long long s_q;
  __shared__ int s_ok;
  const SendDesc &m = list.m[blockIdx.y];
  if (threadIdx.x == 0) {
    const unsigned long long q = loc->seq_out[m.ch];
    s_q = q;
    int ok = 1;
    if (q >= 2) ok = peer_wait_ge(&reinterpret_cast<PeerHeader *>(mine)->ack[m.ch], q - 1, loc, timeout) ? 1 : 0;
    if (!ok) atomicCAS(&loc->err, 0ull, (1ull << 8) | (unsigned)m.ch);
    s_ok = ok;
  }
  __syncthreads();
  const unsigned long long q = s_q;
  if (s_ok) {
    double *dst = slab_of(m.remote, m.rch, q, slab_bytes);
    const long long total = m.box.count();
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
      int i0, i1, i2;
      unflat(m.box, t, i0, i1, i2);
      dst[t] = x[lidx(l, i0, i1, i2)];
    }
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = __hip_atomic_fetch_add(&loc->done_out[m.ch], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {     // last workgroup of this message: every other one has fenced its writes
      loc->done_out[m.ch] = 0;
      loc->seq_out[m.ch] = q + 1;
      if (s_ok) __hip_atomic_store(&reinterpret_cast<PeerHeader *>(m.remote)->ready[m.rch], q + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void __launch_bounds__(256) k_peer_recv(LayoutDev l, double *x, RecvList list, char *mine, PeerLocal *loc, size_t slab_bytes,
                                                   long long timeout) {
  __shared__ unsigned long long s_q;
  __shared__ int s_ok;
  const RecvDesc &m = list.m[blockIdx.y];
  if (threadIdx.x == 0) {
    const unsigned long long q = loc->seq_in[m.ch];
    s_q = q;
    const int ok = peer_wait_ge(&reinterpret_cast<PeerHeader *>(mine)->ready[m.ch], q + 1, loc, timeout) ? 1 : 0;
    if (!ok) atomicCAS(&loc->err, 0ull, (2ull << 8) | (unsigned)m.ch);
    s_ok = ok;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");    // every wave: what the sender wrote before its release is visible now
  const unsigned long long q = s_q;
  if (s_ok) {
    const double *src = slab_of(mine, m.ch, q, slab_bytes);
    const long long total = m.box.count();
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
      int i0, i1, i2;
      unflat(m.box, t, i0, i1, i2);
      x[lidx(l, i0, i1, i2)] = src[t];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = __hip_atomic_fetch_add(&loc->done_in[m.ch], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      loc->done_in[m.ch] = 0;
      loc->seq_in[m.ch] = q + 1;
      if (s_ok) __hip_atomic_store(&reinterpret_cast<PeerHeader *>(m.remote)->ack[m.rch], q + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// all-reduce of n <= PEER_RED_MAX doubles: every rank writes its values into every rank's header (all pairs have a direct link),
// then sums what arrived in rank order -- the same order, hence the same bits, on every rank.  Epoch parity selects one of two
// value sets: a rank can only be one epoch ahead of any other (it needs everybody's contribution to finish an epoch).
__global__ void __launch_bounds__(64) k_peer_allreduce(double *x, int n, int op, int me, int nranks, char **remote, PeerLocal *loc, long long timeout) {
  __shared__ int s_bad;
  const int t = threadIdx.x;
  const unsigned long long e = loc->red_epoch;
  const int par = (int)(e & 1);
  if (t == 0) s_bad = 0;
  __syncthreads();
  if (t < nranks) {
    PeerHeader *h = reinterpret_cast<PeerHeader *>(remote[t]);
    for (int j = 0; j < n; ++j) h->red_val[par][me][j] = x[j];
    __threadfence_system();
    __hip_atomic_store(&h->red_ready[par][me], e + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    PeerHeader *mine = reinterpret_cast<PeerHeader *>(remote[me]);
    if (!peer_wait_ge(&mine->red_ready[par][t], e + 1, loc, timeout)) {
      atomicCAS(&loc->err, 0ull, (3ull << 8) | (unsigned)t);
      s_bad = 1;
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  if (t < n && !s_bad) {
    const PeerHeader *mine = reinterpret_cast<const PeerHeader *>(remote[me]);
    double acc = mine->red_val[par][0][t];
    for (int r = 1; r < nranks; ++r) {
      const double v = mine->red_val[par][r][t];
      acc = op == 0 ? acc + v : (op == 1 ? (v > acc ? v : acc) : (v < acc ? v : acc));
    }
    x[t] = acc;
  }
  __syncthreads();
  if (t == 0) loc->red_epoch = e + 1;
}

// all-gather, send half: my piece into every rank's gather area (slot = my rank); the own piece goes straight to `recv`
__global__ void __launch_bounds__(256) k_peer_gather_send(const double *send, double *recv, long long n, int me, char **remote, PeerLocal *loc,
                                                          size_t gather_off, size_t gather_bytes, long long timeout) {
  __shared__ int s_ok;
  const int p = blockIdx.y;
  const unsigned long long g = loc->gat_seq;
  if (p == me) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) recv[(long long)me * n + t] = send[t];
    return;
  }
  if (threadIdx.x == 0) {
    int ok = 1;
    if (g >= 1) ok = peer_wait_ge(&reinterpret_cast<PeerHeader *>(remote[me])->gat_ack[p], g, loc, timeout) ? 1 : 0;
    if (!ok) atomicCAS(&loc->err, 0ull, (4ull << 8) | (unsigned)p);
    s_ok = ok;
  }
  __syncthreads();
  if (s_ok) {
    double *dst = reinterpret_cast<double *>(remote[p] + gather_off + (size_t)me * gather_bytes);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) dst[t] = send[t];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = __hip_atomic_fetch_add(&loc->gat_done_send[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      loc->gat_done_send[p] = 0;
      if (s_ok) __hip_atomic_store(&reinterpret_cast<PeerHeader *>(remote[p])->gat_ready[me], g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// all-gather, receive half: the pieces that arrived in my gather area -> recv, acknowledged to their senders
__global__ void __launch_bounds__(256) k_peer_gather_recv(double *recv, long long n, int me, int nranks, char **remote, PeerLocal *loc, size_t gather_off,
                                                          size_t gather_bytes, long long timeout) {
  __shared__ int s_ok;
  const int r = blockIdx.y;
  const unsigned long long g = loc->gat_seq;
  if (r != me) {
    if (threadIdx.x == 0) {
      const int ok = peer_wait_ge(&reinterpret_cast<PeerHeader *>(remote[me])->gat_ready[r], g + 1, loc, timeout) ? 1 : 0;
      if (!ok) atomicCAS(&loc->err, 0ull, (5ull << 8) | (unsigned)r);
      s_ok = ok;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (s_ok) {
      const double *src = reinterpret_cast<const double *>(remote[me] + gather_off + (size_t)r * gather_bytes);
      for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) recv[(long long)r * n + t] = src[t];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned int prev = __hip_atomic_fetch_add(&loc->gat_done_recv[r], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == gridDim.x - 1) {
        loc->gat_done_recv[r] = 0;
        if (s_ok) __hip_atomic_store(&reinterpret_cast<PeerHeader *>(remote[r])->gat_ack[me], g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = __hip_atomic_fetch_add(&loc->gat_done_all, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x * gridDim.y - 1) {   // every workgroup has read gat_seq by now
      loc->gat_done_all = 0;
      loc->gat_seq = g + 1;
    }
  }
}

unsigned copy_blocks(long long count) {      // few, grid-striding workgroups: a waiting workgroup holds its slot
  long long nb = (count + 2047) / 2048;
  if (nb > 64) nb = 64;
  if (nb < 1) nb = 1;
  return (unsigned)nb;
}

// HIP IPC maps an exported region ONCE per process (a second hipIpcOpenMemHandle of the same handle fails with "invalid device
// context"): the blocks of one process that share a neighbour in another process share its mapping, counted here.
struct IpcMapping {
  void *ptr;
  int refs;
};
std::mutex g_ipc_mutex;
std::map<std::string, IpcMapping> g_ipc_open;

hipError_t ipc_open_shared(void **ptr, const hipIpcMemHandle_t &h) {
  const std::string key(reinterpret_cast<const char *>(&h), sizeof(h));
  std::lock_guard<std::mutex> lock(g_ipc_mutex);
  auto it = g_ipc_open.find(key);
  if (it != g_ipc_open.end()) {
    ++it->second.refs;
    *ptr = it->second.ptr;
    return hipSuccess;
  }
  const hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e == hipSuccess) g_ipc_open[key] = IpcMapping{*ptr, 1};
  return e;
}

void ipc_close_shared(void *ptr) {
  std::lock_guard<std::mutex> lock(g_ipc_mutex);
  for (auto it = g_ipc_open.begin(); it != g_ipc_open.end(); ++it) {
    if (it->second.ptr != ptr) continue;
    if (--it->second.refs == 0) {
      (void)hipIpcCloseMemHandle(ptr);
      g_ipc_open.erase(it);
    }
    return;
  }
}

void release_mappings(PeerState *p, int me, int nranks) {
  for (int r = 0; r < nranks; ++r) {
    if (r != me && p->remote[r] && p->remote_ipc[r]) ipc_close_shared(p->remote[r]);
    p->remote[r] = nullptr;
    p->remote_ipc[r] = false;
  }
  p->connected = false;
}

}  // namespace

int peer_phase(examg_comm *c, const examg_layout_t *l_, double *x, const PeerMsg *sends, int ns, const PeerMsg *recvs, int nr, hipStream_t s) {
  PeerState *p = c->peer;
  if (!p->connected) { set_error("examg_exchange: the peer-write communicator is not connected (examg_comm_peer_alloc / _connect)"); return 1; }
  if (ns > 6 || nr > 6) { set_error("examg_exchange: more than six messages in one phase"); return 1; }
  const LayoutDev l = make_layout(l_);
  if (ns > 0) {
    SendList list;
    unsigned g = 1;
    for (int i = 0; i < ns; ++i) {
      const Box box{sends[i].b[0], sends[i].b[1], sends[i].b[2], sends[i].e[0], sends[i].e[1], sends[i].e[2]};
      if (!box_inside(l_, box, 0)) { set_error("examg_exchange: send box leaves the allocation"); return 1; }
      if ((size_t)box.count() * 8 > p->slab_bytes) {
        set_error("examg_exchange: a message of %lld bytes exceeds the slabs of the peer-write communicator (%zu; examg_comm_peer_alloc)", box.count() * 8, p->slab_bytes);
        return 1;
      }
      if (sends[i].peer < 0 || sends[i].peer >= c->size || !p->remote[sends[i].peer]) { set_error("examg_exchange: no mapping for rank %d", sends[i].peer); return 1; }
      list.m[i] = SendDesc{box, p->remote[sends[i].peer], 2 * sends[i].d + sends[i].side, 2 * sends[i].d + (1 - sends[i].side)};
      const unsigned gi = copy_blocks(box.count());
      if (gi > g) g = gi;
    }
    hipLaunchKernelGGL(k_peer_send, dim3(g, ns), dim3(256), 0, s, l, x, list, p->mine, p->loc, p->slab_bytes, p->timeout_ticks);
    EXAMG_CHECK_LAUNCH("k_peer_send");
  }
  if (nr > 0) {
    RecvList list;
    unsigned g = 1;
    for (int i = 0; i < nr; ++i) {
      const Box box{recvs[i].b[0], recvs[i].b[1], recvs[i].b[2], recvs[i].e[0], recvs[i].e[1], recvs[i].e[2]};
      if (!box_inside(l_, box, 0)) { set_error("examg_exchange: receive box leaves the allocation"); return 1; }
      if ((size_t)box.count() * 8 > p->slab_bytes) { set_error("examg_exchange: a message exceeds the slabs of the peer-write communicator"); return 1; }
      if (recvs[i].peer < 0 || recvs[i].peer >= c->size || !p->remote[recvs[i].peer]) { set_error("examg_exchange: no mapping for rank %d", recvs[i].peer); return 1; }
      list.m[i] = RecvDesc{box, p->remote[recvs[i].peer], 2 * recvs[i].d + recvs[i].side, 2 * recvs[i].d + (1 - recvs[i].side)};
      const unsigned gi = copy_blocks(box.count());
      if (gi > g) g = gi;
    }
    hipLaunchKernelGGL(k_peer_recv, dim3(g, nr), dim3(256), 0, s, l, x, list, p->mine, p->loc, p->slab_bytes, p->timeout_ticks);
    EXAMG_CHECK_LAUNCH("k_peer_recv");
  }
  return 0;
}

int peer_allreduce(examg_comm *c, double *x, int n, int op, hipStream_t s) {
  PeerState *p = c->peer;
  if (!p->connected) { set_error("examg_allreduce: the peer-write communicator is not connected"); return 1; }
  for (int off = 0; off < n; off += PEER_RED_MAX) {
    const int m = n - off < PEER_RED_MAX ? n - off : PEER_RED_MAX;
    hipLaunchKernelGGL(k_peer_allreduce, dim3(1), dim3(64), 0, s, x + off, m, op, c->rank, c->size, p->remote_dev, p->loc, p->timeout_ticks);
    EXAMG_CHECK_LAUNCH("k_peer_allreduce");
  }
  return 0;
}

int peer_allgather(examg_comm *c, const double *send, double *recv, long long n, hipStream_t s) {
  PeerState *p = c->peer;
  if (!p->connected) { set_error("examg_allgather: the peer-write communicator is not connected"); return 1; }
  if (n == 0) return 0;
  if ((size_t)n * 8 > p->gather_bytes) {
    set_error("examg_allgather: a piece of %lld bytes exceeds the gather area of the peer-write communicator (%zu; examg_comm_peer_alloc)", n * 8, p->gather_bytes);
    return 1;
  }
  const size_t off = PEER_HEADER_BYTES + 12 * p->slab_bytes;
  const unsigned g = copy_blocks(n);
  hipLaunchKernelGGL(k_peer_gather_send, dim3(g, c->size), dim3(256), 0, s, send, recv, n, c->rank, p->remote_dev, p->loc, off, p->gather_bytes, p->timeout_ticks);
  EXAMG_CHECK_LAUNCH("k_peer_gather_send");
  hipLaunchKernelGGL(k_peer_gather_recv, dim3(g, c->size), dim3(256), 0, s, recv, n, c->rank, c->size, p->remote_dev, p->loc, off, p->gather_bytes, p->timeout_ticks);
  EXAMG_CHECK_LAUNCH("k_peer_gather_recv");
  return 0;
}

void peer_destroy(examg_comm *c) {
  PeerState *p = c->peer;
  if (!p) return;
  release_mappings(p, c->rank, c->size);
  if (p->mine) (void)hipFree(p->mine);
  if (p->loc) (void)hipFree(p->loc);
  if (p->remote_dev) (void)hipFree(p->remote_dev);
  delete p;
  c->peer = nullptr;
}

}  // namespace examg

using namespace examg;

extern "C" int examg_comm_create_peer(examg_comm_t **comm, int nranks, int rank) {
  if (!comm) { set_error("examg_comm_create_peer: null argument"); return 1; }
  if (const char *ff = getenv("EXAMG_PEER_FORCE_FAIL")) {   // fault injection for the fallback tests of the callers
    if (*ff == '1') { set_error("examg_comm_create_peer: refused (EXAMG_PEER_FORCE_FAIL=1)"); return 1; }
  }
  if (nranks < 1 || nranks > PEER_MAX_RANKS || rank < 0 || rank >= nranks) { set_error("examg_comm_create_peer: rank %d of %d (at most %d ranks)", rank, nranks, (int)PEER_MAX_RANKS); return 1; }
  examg_comm *c = new examg_comm;
  c->rank = rank;
  c->size = nranks;
  c->peer = new PeerState;
  long long ms = 120000;     // long enough for a neighbour that writes a field to disk or captures a graph; callers check examg_comm_status at every host sync
  if (const char *e = getenv("EXAMG_PEER_TIMEOUT_MS")) { const long long v = atoll(e); if (v > 0) ms = v; }
  c->peer->timeout_ticks = ms * 100000;   // wall_clock64 counts at 100 MHz
  if (check_hip(hipMalloc((void **)&c->peer->loc, sizeof(PeerLocal)), "examg_comm_create_peer") ||
      check_hip(hipMalloc((void **)&c->peer->remote_dev, sizeof(char *) * PEER_MAX_RANKS), "examg_comm_create_peer")) {
    peer_destroy(c);
    delete c;
    return 1;
  }
  *comm = c;
  return 0;
}

// (Re)allocate the own region: collective in the sense that every rank must call it, exchange the handles by its own means
// (MPI_Allgather in a generated program, torch.distributed, files) and then call examg_comm_peer_connect.  The caller makes
// sure that no exchange is in flight on ANY rank (device synchronised, then a host barrier) before a region is replaced.
extern "C" int examg_comm_peer_alloc(examg_comm_t *comm, size_t slab_bytes, size_t gather_bytes, void *handle_out) {
  if (!comm || !comm->peer || !handle_out) { set_error("examg_comm_peer_alloc: not a peer-write communicator / null argument"); return 1; }
  PeerState *p = comm->peer;
  if (check_hip(hipDeviceSynchronize(), "examg_comm_peer_alloc")) return 1;
  release_mappings(p, comm->rank, comm->size);
  if (p->mine) { (void)hipFree(p->mine); p->mine = nullptr; }
  slab_bytes = (slab_bytes + 255) & ~(size_t)255;
  gather_bytes = (gather_bytes + 255) & ~(size_t)255;
  const size_t total = PEER_HEADER_BYTES + 12 * slab_bytes + (size_t)comm->size * gather_bytes;
  // uncached on both sides (MTYPE_UC): a peer's posted writes and this GPU's reads meet in memory, not in an L2
  hipError_t e = hipExtMallocWithFlags((void **)&p->mine, total, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags((void **)&p->mine, total, hipDeviceMallocFinegrained);
  }
  if (check_hip(e, "examg_comm_peer_alloc: hipExtMallocWithFlags")) return 1;
  if (check_hip(hipMemset(p->mine, 0, total), "examg_comm_peer_alloc") || check_hip(hipMemset(p->loc, 0, sizeof(PeerLocal)), "examg_comm_peer_alloc") ||
      check_hip(hipDeviceSynchronize(), "examg_comm_peer_alloc"))
    return 1;
  p->region_bytes = total;
  p->slab_bytes = slab_bytes;
  p->gather_bytes = gather_bytes;
  PeerWire w;
  memset(&w, 0, sizeof(w));
  if (comm->size > 1 && check_hip(hipIpcGetMemHandle(&w.mem, p->mine), "examg_comm_peer_alloc: hipIpcGetMemHandle (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)")) return 1;
  w.region_bytes = total;
  w.slab_bytes = slab_bytes;
  w.gather_bytes = gather_bytes;
  w.self_ptr = (unsigned long long)(uintptr_t)p->mine;
  w.rank = comm->rank;
  w.pid = (int)getpid();
  memset(handle_out, 0, EXAMG_PEER_HANDLE_BYTES);
  memcpy(handle_out, &w, sizeof(w));
  return 0;
}

// First step of growing the regions: give up the mappings of the OTHER ranks' regions.  Every rank calls it, then a host barrier, then
// examg_comm_peer_alloc -- so that no region is freed by its owner while another process still has it mapped.
extern "C" int examg_comm_peer_release(examg_comm_t *comm) {
  if (!comm || !comm->peer) { set_error("examg_comm_peer_release: not a peer-write communicator"); return 1; }
  if (check_hip(hipDeviceSynchronize(), "examg_comm_peer_release")) return 1;
  release_mappings(comm->peer, comm->rank, comm->size);
  return 0;
}

extern "C" int examg_comm_peer_connect(examg_comm_t *comm, const void *all_handles) {
  if (!comm || !comm->peer || !all_handles) { set_error("examg_comm_peer_connect: not a peer-write communicator / null argument"); return 1; }
  PeerState *p = comm->peer;
  if (!p->mine) { set_error("examg_comm_peer_connect: examg_comm_peer_alloc first"); return 1; }
  release_mappings(p, comm->rank, comm->size);
  for (int r = 0; r < comm->size; ++r) {
    PeerWire w;
    memcpy(&w, (const char *)all_handles + (size_t)r * EXAMG_PEER_HANDLE_BYTES, sizeof(w));
    if (w.rank != r) { set_error("examg_comm_peer_connect: handle %d belongs to rank %d", r, w.rank); return 1; }
    if (w.slab_bytes != p->slab_bytes || w.gather_bytes != p->gather_bytes) {
      set_error("examg_comm_peer_connect: rank %d allocated slabs of %llu / %llu bytes, this rank %zu / %zu", r, w.slab_bytes, w.gather_bytes, p->slab_bytes, p->gather_bytes);
      return 1;
    }
    if (r == comm->rank) { p->remote[r] = p->mine; continue; }
    if (w.pid == (int)getpid()) {
      // several blocks in ONE process (one host thread and stream per block, as the reference's fragments of a block): the
      // neighbour's region is in this address space already -- HIP IPC cannot map memory into the process that exported it
      p->remote[r] = (char *)(uintptr_t)w.self_ptr;
      continue;
    }
    void *ptr = nullptr;
    if (check_hip(ipc_open_shared(&ptr, w.mem), "examg_comm_peer_connect: hipIpcOpenMemHandle")) return 1;
    p->remote[r] = (char *)ptr;
    p->remote_ipc[r] = true;
  }
  if (check_hip(hipMemcpy(p->remote_dev, p->remote, sizeof(char *) * PEER_MAX_RANKS, hipMemcpyHostToDevice), "examg_comm_peer_connect")) return 1;
  p->connected = true;
  return 0;
}

extern "C" size_t examg_comm_peer_slab_bytes(const examg_comm_t *comm) { return (comm && comm->peer && comm->peer->mine) ? comm->peer->slab_bytes : 0; }
extern "C" size_t examg_comm_peer_gather_bytes(const examg_comm_t *comm) { return (comm && comm->peer && comm->peer->mine) ? comm->peer->gather_bytes : 0; }

// 0 = no wait of the peer-write transport has given up so far (synchronises `stream` first); otherwise 1 and the error text
// names the wait (kind, channel / rank).  RCCL communicators always report 0.
extern "C" int examg_comm_status(examg_comm_t *comm, examg_stream_t stream) {
  if (!comm) { set_error("examg_comm_status: null argument"); return 1; }
  if (!comm->peer) return 0;
  unsigned long long err = 0;
  // on `stream`, not on the legacy stream: a host with several blocks per process may be recording a graph on another thread's stream
  if (check_hip(hipMemcpyAsync(&err, &comm->peer->loc->err, sizeof(err), hipMemcpyDeviceToHost, (hipStream_t)stream), "examg_comm_status") ||
      check_hip(hipStreamSynchronize((hipStream_t)stream), "examg_comm_status"))
    return 1;
  if (err == 0) return 0;
  static const char *kind[] = {"?", "send waited for the acknowledgement of its slab", "receive waited for a message", "all-reduce waited for rank",
                               "all-gather send waited for the acknowledgement of rank", "all-gather waited for the piece of rank"};
  const unsigned k = (unsigned)(err >> 8), ch = (unsigned)(err & 0xff);
  set_error("peer-write transport, rank %d: %s %u -- no progress within the timeout (EXAMG_PEER_TIMEOUT_MS); the communicator is unusable",
            comm->rank, k < 6 ? kind[k] : "?", ch);
  return 1;
}
