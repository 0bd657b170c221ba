// Block-to-block transport of libexamg over RCCL: `communicate <field>` (exch<Field>_<level>(slot)) and the scalar
// all-reduce that follows a reduction loop, as C entry points a generated C++ host (or the Python mirror, through ctypes)
// calls -- one process per GPU, ncclSend / ncclRecv groups between the axis neighbours (each GPU pair has its own xGMI link),
// everything stream-ordered: no host synchronisation.  (RCCL point-to-point groups hang inside a hipGraph capture on ROCm 7.2;
// the second transport, examg_peer.hip -- peer writes through HIP IPC -- is the capturable one.  The entry points below serve
// both: a communicator made by examg_comm_create_peer routes its phases there.)
//
// Replaces, in the generated program: IR_CommunicateFunction.compileBody
// (Compiler/src/exastencils/communication/ir/IR_CommunicateFunction.scala:194-219,412-471: duplicate layers first -- per
// axis the own UPPER duplicate plane to the '+' neighbour, received into the LOWER plane from the '-' neighbour -- then ghost
// layers per axis in both directions with tangential extent GLB..GRE, so that edge/corner ghosts become valid with six
// neighbours only; index ranges IR_PackInfoDuplicate.scala:15-39 / IR_PackInfoGhost.scala:13-60; pack / unpack
// IR_NoInterpPacking.scala:53-83; MPI_Isend / MPI_Irecv + waits IR_RemoteSend.scala:48-58, IR_RemoteRecv.scala:50-65;
// local exchange between fragments of one process IR_NoInterpPacking.scala:101-131) and MPI_Allreduce(MPI_IN_PLACE, ..)
// (parallelization/api/mpi/MPI_Reduction.scala:100-126).
//
// RCCL is bound at run time (dlopen of librccl.so.1, or $EXAMG_RCCL_LIB): a single-GPU host never needs it, and inside a
// PyTorch process the copy PyTorch has already loaded is the one that is used (same soname).
#include <dlfcn.h>
#include <stdlib.h>

#include "examg_comm_internal.h"

using namespace examg;

namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
  if (g_rccl.handle) return 0;
  const char *env = getenv("EXAMG_RCCL_LIB");
  const char *names[] = {env, "librccl.so.1", "librccl.so"};
  void *h = nullptr;
  for (const char *n : names) {
    if (!n || !*n) continue;
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    set_error("examg_comm: cannot load RCCL (librccl.so.1; set EXAMG_RCCL_LIB): %s", dlerror());
    return 1;
  }
#define EXAMG_SYM(field, name)                                             \
  *(void **)(&g_rccl.field) = dlsym(h, name);                              \
  if (!g_rccl.field) { set_error("examg_comm: RCCL lacks %s", name); return 1; }
  EXAMG_SYM(GetUniqueId, "ncclGetUniqueId")
  EXAMG_SYM(CommInitRank, "ncclCommInitRank")
  EXAMG_SYM(CommDestroy, "ncclCommDestroy")
  EXAMG_SYM(Send, "ncclSend")
  EXAMG_SYM(Recv, "ncclRecv")
  EXAMG_SYM(GroupStart, "ncclGroupStart")
  EXAMG_SYM(GroupEnd, "ncclGroupEnd")
  EXAMG_SYM(AllReduce, "ncclAllReduce")
  EXAMG_SYM(AllGather, "ncclAllGather")
  EXAMG_SYM(GetErrorString, "ncclGetErrorString")
#undef EXAMG_SYM
  g_rccl.handle = h;
  return 0;
}

int check_nccl(ncclResult_t r, const char *what) {
  if (r == ncclSuccess) return 0;
  set_error("%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  return 1;
}

}  // namespace

static_assert(sizeof(ncclUniqueId) == EXAMG_COMM_ID_BYTES, "EXAMG_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

extern "C" int examg_comm_unique_id(void *id) {
  if (!id) { set_error("examg_comm_unique_id: null argument"); return 1; }
  if (load_rccl()) return 1;
  ncclUniqueId u;
  if (check_nccl(g_rccl.GetUniqueId(&u), "ncclGetUniqueId")) return 1;
  memcpy(id, &u, sizeof(u));
  return 0;
}

extern "C" int examg_comm_create(examg_comm_t **comm, const void *id, int nranks, int rank) {
  if (!comm) { set_error("examg_comm_create: null argument"); return 1; }
  if (nranks < 1 || rank < 0 || rank >= nranks) { set_error("examg_comm_create: rank %d of %d", rank, nranks); return 1; }
  examg_comm *c = new examg_comm;
  c->rank = rank;
  c->size = nranks;
  const char *sv = getenv("EXAMG_COMM_SELF_RCCL");
  c->self_via_rccl = sv && *sv == '1';
  if (nranks > 1 || c->self_via_rccl) {
    if (!id) { set_error("examg_comm_create: a communicator of %d ranks needs the unique id of rank 0", nranks); delete c; return 1; }
    if (load_rccl()) { delete c; return 1; }
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    if (check_nccl(g_rccl.CommInitRank(&c->nccl, nranks, u, rank), "ncclCommInitRank")) { delete c; return 1; }
  }
  *comm = c;
  return 0;
}

extern "C" int examg_comm_destroy(examg_comm_t *comm) {
  if (!comm) return 0;
  int rc = 0;
  if (comm->nccl) rc = check_nccl(g_rccl.CommDestroy(comm->nccl), "ncclCommDestroy");
  if (comm->peer) peer_destroy(comm);
  if (comm->fork) (void)hipEventDestroy(comm->fork);
  if (comm->join) (void)hipEventDestroy(comm->join);
  if (comm->side) (void)hipStreamDestroy(comm->side);
  delete comm;
  return rc;
}

extern "C" int examg_comm_rank(const examg_comm_t *comm) { return comm ? comm->rank : -1; }
extern "C" int examg_comm_size(const examg_comm_t *comm) { return comm ? comm->size : -1; }

namespace {

struct Msg {
  int peer;
  Range box;
  double *buf;
  int d, side;   // axis and side (0 = minus, 1 = plus) of this block the message leaves through / arrives at
};

// one phase: pack -> group(recv.., send..) -> unpack  (IR_CommunicateFunction.scala:194-219)
int phase(examg_comm_t *c, const examg_layout_t *l, double *x, Msg *sends, int ns, Msg *recvs, int nr, hipStream_t s) {
  if (ns == 0 && nr == 0) return 0;
  if (c->peer) {   // peer-write transport: the send kernel packs straight into the neighbour's receive slab
    PeerMsg ps[6], pr[6];
    for (int i = 0; i < ns; ++i) {
      ps[i].peer = sends[i].peer; ps[i].d = sends[i].d; ps[i].side = sends[i].side;
      for (int t = 0; t < 3; ++t) { ps[i].b[t] = sends[i].box.b[t]; ps[i].e[t] = sends[i].box.e[t]; }
    }
    for (int i = 0; i < nr; ++i) {
      pr[i].peer = recvs[i].peer; pr[i].d = recvs[i].d; pr[i].side = recvs[i].side;
      for (int t = 0; t < 3; ++t) { pr[i].b[t] = recvs[i].box.b[t]; pr[i].e[t] = recvs[i].box.e[t]; }
    }
    return peer_phase(c, l, x, ps, ns, pr, nr, s);
  }
  for (int i = 0; i < ns; ++i)
    if (examg_pack(l, x, sends[i].buf, sends[i].box.b, sends[i].box.e, s)) return 1;
  bool any_remote = false;
  for (int i = 0; i < ns; ++i) any_remote = any_remote || sends[i].peer != c->rank || c->self_via_rccl;
  for (int i = 0; i < nr; ++i) any_remote = any_remote || recvs[i].peer != c->rank || c->self_via_rccl;
  if (any_remote) {
    if (!c->nccl) { set_error("examg_exchange: neighbour on another rank but the communicator has no RCCL handle"); return 1; }
    if (check_nccl(g_rccl.GroupStart(), "ncclGroupStart")) return 1;
  }
  // a block that is its own neighbour (periodic dimension with one block): what leaves on one side arrives on the other --
  // the message sent towards `side` pairs with the receive from `-side`; without RCCL the receive slab is the send slab
  for (int i = 0; i < nr; ++i) {
    if (recvs[i].peer == c->rank && !c->self_via_rccl) continue;
    if (check_nccl(g_rccl.Recv(recvs[i].buf, (size_t)recvs[i].box.count(), ncclDouble, recvs[i].peer, c->nccl, s), "ncclRecv")) {
      (void)g_rccl.GroupEnd();
      return 1;
    }
  }
  for (int i = 0; i < ns; ++i) {
    if (sends[i].peer == c->rank && !c->self_via_rccl) continue;
    if (check_nccl(g_rccl.Send(sends[i].buf, (size_t)sends[i].box.count(), ncclDouble, sends[i].peer, c->nccl, s), "ncclSend")) {
      (void)g_rccl.GroupEnd();
      return 1;
    }
  }
  if (any_remote && check_nccl(g_rccl.GroupEnd(), "ncclGroupEnd")) return 1;
  for (int i = 0; i < nr; ++i) {
    const double *src = recvs[i].buf;
    if (recvs[i].peer == c->rank && !c->self_via_rccl) {
      // pairing by position: sends are listed towards (-, +), receives from (+, -) -- what went out towards - is what
      // comes in from + (see examg_exchange); a block is its own neighbour on both sides of an axis or on neither
      if (ns != nr) { set_error("examg_exchange: a self-neighbour needs both sides of the axis"); return 1; }
      src = sends[i].buf;
    }
    if (examg_unpack(l, x, src, recvs[i].box.b, recvs[i].box.e, s)) return 1;
  }
  return 0;
}

}  // namespace

extern "C" size_t examg_exchange_workspace_bytes(const examg_layout_t *l) {
  if (!l) return 0;
  long long n = 0;
  for (int d = 0; d < l->nd; ++d) n += 4 * face_count(l, d);
  return (size_t)n * sizeof(double);
}

extern "C" int examg_exchange(examg_comm_t *comm, const examg_layout_t *l, double *x, const examg_neighbors_t *nb, int what,
                              void *workspace, size_t workspace_bytes, examg_stream_t stream) {
  if (!comm || !l || !x || !nb) { set_error("examg_exchange: null argument"); return 1; }
  if (!(what & (EXAMG_EXCH_DUP | EXAMG_EXCH_GHOST))) return 0;
  bool any = false;
  for (int d = 0; d < l->nd; ++d) any = any || nb->rank[d][0] >= 0 || nb->rank[d][1] >= 0;
  if (!any) return 0;   // no neighbours: the generated exch function is empty
  if (!comm->peer && (!workspace || workspace_bytes < examg_exchange_workspace_bytes(l))) { set_error("examg_exchange: workspace too small"); return 1; }
  for (int d = 0; d < l->nd; ++d)
    for (int s = 0; s < 2; ++s)
      if (nb->rank[d][s] >= comm->size) { set_error("examg_exchange: neighbour rank %d outside the communicator", nb->rank[d][s]); return 1; }
  hipStream_t s = (hipStream_t)stream;
  double *ws = comm->peer ? nullptr : (double *)workspace;   // the peer-write transport owns its slabs
  const int nd = l->nd;
  if (what & EXAMG_EXCH_DUP) {
    for (int d = 0; d < nd; ++d) {
      if (l->dup_l[d] == 0 && l->dup_r[d] == 0) continue;
      Msg snd[1], rcv[1];
      int ns = 0, nr = 0;
      Range sb, rb;
      dup_ranges(l, d, sb, rb);
      if (nb->rank[d][1] >= 0) snd[ns++] = Msg{nb->rank[d][1], sb, ws ? ws + slot_offset(l, d, 1, 0) : nullptr, d, 1};
      if (nb->rank[d][0] >= 0) rcv[nr++] = Msg{nb->rank[d][0], rb, ws ? ws + slot_offset(l, d, 0, 1) : nullptr, d, 0};
      if (phase(comm, l, x, snd, ns, rcv, nr, s)) return 1;
    }
  }
  if (what & EXAMG_EXCH_GHOST) {
    Msg snd[6], rcv[6];
    int ns = 0, nr = 0;
    for (int d = 0; d < nd; ++d) {
      if (l->ghost_l[d] == 0 && l->ghost_r[d] == 0) continue;
      // sends towards minus, then plus; receives from plus, then minus: when both neighbours of an axis are the same rank
      // (two blocks, periodic) the first message sent (towards -) is the first one the peer expects (from its +)
      const int d_ns = ns, d_nr = nr;
      for (int side = 0; side < 2; ++side) {
        if (nb->rank[d][side] < 0) continue;
        Range sb, rb;
        ghost_ranges(l, d, side ? +1 : -1, sb, rb);
        snd[ns++] = Msg{nb->rank[d][side], sb, ws ? ws + slot_offset(l, d, side, 0) : nullptr, d, side};
      }
      for (int side = 1; side >= 0; --side) {
        if (nb->rank[d][side] < 0) continue;
        Range sb, rb;
        ghost_ranges(l, d, side ? +1 : -1, sb, rb);
        rcv[nr++] = Msg{nb->rank[d][side], rb, ws ? ws + slot_offset(l, d, side, 1) : nullptr, d, side};
      }
      if (!(what & EXAMG_EXCH_CONCURRENT_AXES)) {
        if (phase(comm, l, x, snd + d_ns, ns - d_ns, rcv + d_nr, nr - d_nr, s)) return 1;
        ns = d_ns;
        nr = d_nr;
      }
    }
    if (what & EXAMG_EXCH_CONCURRENT_AXES) {
      // face ghosts only (5/7-point loops): all axes in one group; per axis the receive order still mirrors the send order.
      // phase() pairs self-messages by position within an axis, so the axes go through it one at a time when a block is its
      // own neighbour, and as ONE group otherwise
      bool self = false;
      for (int i = 0; i < ns; ++i) self = self || (snd[i].peer == comm->rank && !comm->peer);
      if (!self) {
        if (phase(comm, l, x, snd, ns, rcv, nr, s)) return 1;
      } else {
        int is = 0, ir = 0;
        for (int d = 0; d < nd; ++d) {
          int cs = 0, cr = 0;
          for (int side = 0; side < 2; ++side)
            if (nb->rank[d][side] >= 0 && (l->ghost_l[d] || l->ghost_r[d])) { ++cs; ++cr; }
          if (phase(comm, l, x, snd + is, cs, rcv + ir, cr, s)) return 1;
          is += cs;
          ir += cr;
        }
      }
    }
  }
  return 0;
}

extern "C" int examg_allreduce(examg_comm_t *comm, double *x, int n, int op, examg_stream_t stream) {
  if (!comm || !x || n < 0) { set_error("examg_allreduce: bad argument"); return 1; }
  if (op < 0 || op > 2) { set_error("examg_allreduce: op must be 0 (sum), 1 (max) or 2 (min)"); return 1; }
  if (comm->size == 1 && !comm->self_via_rccl) return 0;
  if (comm->peer) return peer_allreduce(comm, x, n, op, (hipStream_t)stream);
  if (!comm->nccl) { set_error("examg_allreduce: communicator has no RCCL handle"); return 1; }
  const ncclRedOp_t rop = op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin);
  return check_nccl(g_rccl.AllReduce(x, x, (size_t)n, ncclDouble, rop, comm->nccl, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int examg_allgather(examg_comm_t *comm, const double *send, double *recv, int64_t n, examg_stream_t stream) {
  if (!comm || !send || !recv || n < 0) { set_error("examg_allgather: bad argument"); return 1; }
  if (comm->size == 1 && !comm->self_via_rccl) {
    if (send != recv) return check_hip(hipMemcpyAsync(recv, send, (size_t)n * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream), "examg_allgather");
    return 0;
  }
  if (comm->peer) return peer_allgather(comm, send, recv, (long long)n, (hipStream_t)stream);
  if (!comm->nccl) { set_error("examg_allgather: communicator has no RCCL handle"); return 1; }
  return check_nccl(g_rccl.AllGather(send, recv, (size_t)n, ncclDouble, comm->nccl, (hipStream_t)stream), "ncclAllGather");
}

// ---------------------------------------------------------------------------------------------------------------------------
// Smoother passes on a block WITH neighbours, halo traffic overlapped with the interior kernel (the reference's core / boundary
// split, baseExt/ir/IR_LoopOverPointsInOneFragment.scala:143-222, applied to a pair of dependent sweeps): one C call per pass.
//
//   launch stream   fused two-stage kernel on the loop's box shrunk by 1 (first stage) / 2 (second stage) at interior faces: reads
//                   no ghost value, starts at once
//   side stream     1. exchange ghost layers of u_in                            (as the smoother's `communicate` does)
//                   2. first stage on the three planes next to every interior face, into tmp          (thin launches)
//                   3. exchange ghost layers of tmp                             (the neighbours' first-stage values)
//                   4. second stage on the two planes next to every interior face, into u_out         (thin launches)
//   join            events; the launch stream continues when both are done
//
// Two exchanges per pass -- as many as the two plain loops have -- and results bit-identical to them.  tmp: scratch array of
// u's layout.  Its duplicate planes on PHYSICAL faces are read tangentially by step 4 and written by nobody: they are copied
// from u_in each time.  When the one-pass kernel is not eligible (examg_two_stage_eligible: other stencils or entry orders,
// short rows) everything runs in sequence on the launch stream (the fallback uses tmp as its own scratch).
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

// side stream and events live in the communicator (created on first use; a communicator belongs to one host thread and one device)
typedef examg_comm Overlap;
Overlap *overlap_of(examg_comm_t *c) {
  if (!c->side) {
    // the shell work -- exchanges and thin launches -- is a chain of short kernels beside a pass that fills the chip: at the highest
    // priority its workgroups are placed ahead of the interior pass' (EXAMG_SIDE_PRIORITY=0: default priority, for A/B runs)
    int least = 0, greatest = 0;
    const char *pe = getenv("EXAMG_SIDE_PRIORITY");
    const bool high = !(pe && pe[0] == '0') && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least;
    if (check_hip(high ? hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, greatest) : hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking),
                  "hipStreamCreate"))
      return nullptr;
    if (check_hip(hipEventCreateWithFlags(&c->fork, hipEventDisableTiming), "hipEventCreate")) return nullptr;
    if (check_hip(hipEventCreateWithFlags(&c->join, hipEventDisableTiming), "hipEventCreate")) return nullptr;
  }
  return c;
}

struct Faces {
  int n;
  int d[6], side[6];
};

Faces interior_faces(const examg_layout_t *l, const examg_neighbors_t *nb) {
  Faces f;
  f.n = 0;
  for (int d = 0; d < l->nd; ++d)
    for (int s = 0; s < 2; ++s)
      if (nb->rank[d][s] >= 0) { f.d[f.n] = d; f.side[f.n] = s ? +1 : -1; ++f.n; }
  return f;
}

void shrunk(const Faces &f, const int32_t *b, const int32_t *e, int k, int32_t *bb, int32_t *ee) {
  for (int d = 0; d < 3; ++d) { bb[d] = b[d]; ee[d] = e[d]; }
  for (int i = 0; i < f.n; ++i) {
    if (f.side[i] < 0) bb[f.d[i]] = b[f.d[i]] + k;
    else ee[f.d[i]] = e[f.d[i]] - k;
  }
}

void slab(const int32_t *b, const int32_t *e, int d, int side, int k, int32_t *sb, int32_t *se) {
  for (int t = 0; t < 3; ++t) { sb[t] = b[t]; se[t] = e[t]; }
  if (side < 0) se[d] = b[d] + k < e[d] ? b[d] + k : e[d];
  else sb[d] = e[d] - k > b[d] ? e[d] - k : b[d];
}

// tmp's duplicate planes on physical faces <- u_in's
int copy_physical_planes(const examg_layout_t *l, const examg_neighbors_t *nb, const double *src, double *tmp, hipStream_t s) {
  for (int d = 0; d < l->nd; ++d)
    for (int sd = 0; sd < 2; ++sd) {
      if (nb->rank[d][sd] >= 0) continue;
      int32_t pb[3] = {0, 0, 0}, pe[3] = {1, 1, 1};
      for (int t = 0; t < l->nd; ++t) {
        const Marks m = marks(l, t);
        pb[t] = m.DLB;
        pe[t] = m.DRE;
      }
      const Marks m = marks(l, d);
      if (sd == 0) { pb[d] = m.DLB; pe[d] = m.DLE; }
      else { pb[d] = m.DRB; pe[d] = m.DRE; }
      if (examg_axpby(l, src, l, tmp, 1.0, 0.0, pb, pe, s)) return 1;
    }
  return 0;
}

template <bool COL>
int pass_blocks(const char *who, examg_comm_t *comm, const examg_neighbors_t *nb, const examg_layout_t *lu, const double *u_in, double *u_out,
                double *tmp, const examg_layout_t *lf, const double *rhs, const examg_stencil_t *st, double w, int first,
                const int32_t *begin, const int32_t *end, int exchange_flags, void *workspace, size_t workspace_bytes, int overlap,
                examg_stream_t stream) {
  if (!comm || !nb || !lu || !u_in || !u_out || !tmp || !lf || !rhs || !st || !begin || !end) { set_error("%s: null argument", who); return 1; }
  if (u_in == u_out || tmp == u_in || tmp == u_out) { set_error("%s: u_in, u_out and tmp must be three arrays", who); return 1; }
  hipStream_t main = (hipStream_t)stream;
  const Faces f = interior_faces(lu, nb);
  const int what = EXAMG_EXCH_GHOST | (exchange_flags & EXAMG_EXCH_CONCURRENT_AXES);
  if (f.n == 0) {   // no neighbours: the plain one-pass forms
    if (COL) return examg_rbgs_sweep_fused(lu, u_in, u_out, lf, rhs, st, w, first, begin, end, stream);
    return examg_jacobi2_boxes(lu, u_in, u_out, tmp, lf, rhs, st, w, begin, end, begin, end, stream);
  }
  int32_t b1[3], e1[3], b2[3], e2[3];
  shrunk(f, begin, end, 1, b1, e1);
  shrunk(f, begin, end, 2, b2, e2);
  bool has_interior = true;
  for (int d = 0; d < lu->nd; ++d) has_interior = has_interior && e2[d] > b2[d];
  const bool fused = has_interior && examg_two_stage_eligible(lu, lf, st, b1, e1, b2, e2) == 1;

  auto interior = [&](hipStream_t s) -> int {
    if (!has_interior) return 0;
    if (COL) return examg_rbgs_sweep_fused_boxes(lu, u_in, u_out, fused ? nullptr : tmp, lf, rhs, st, w, first, b1, e1, b2, e2, s);
    return examg_jacobi2_boxes(lu, u_in, u_out, tmp, lf, rhs, st, w, b1, e1, b2, e2, s);
  };
  auto shell = [&](hipStream_t s) -> int {
    if (examg_exchange(comm, lu, const_cast<double *>(u_in), nb, what, workspace, workspace_bytes, s)) return 1;
    if (!(exchange_flags & EXAMG_PASS_TMP_PLANES_VALID) && copy_physical_planes(lu, nb, u_in, tmp, s)) return 1;
    int32_t sb[3], se[3];
    for (int i = 0; i < f.n; ++i) {        // first stage on three planes
      slab(begin, end, f.d[i], f.side[i], 3, sb, se);
      if (COL) {                           // tmp = u_in on the slab with the colour's points updated (reads u_in only): one launch
        if (stencil_colour_passthrough(lu, u_in, lf, rhs, lu, tmp, st, w, first, sb, se, s)) return 1;
      } else if (examg_stencil_op(EXAMG_SMOOTH, lu, u_in, lf, rhs, lu, tmp, st, w, -1, sb, se, s)) {
        return 1;
      }
    }
    if (examg_exchange(comm, lu, tmp, nb, what, workspace, workspace_bytes, s)) return 1;
    for (int i = 0; i < f.n; ++i) {        // second stage on two planes
      slab(begin, end, f.d[i], f.side[i], 2, sb, se);
      if (COL) {
        if (stencil_colour_passthrough(lu, tmp, lf, rhs, lu, u_out, st, w, 1 - first, sb, se, s)) return 1;
      } else if (examg_stencil_op(EXAMG_SMOOTH, lu, tmp, lf, rhs, lu, u_out, st, w, -1, sb, se, s)) {
        return 1;
      }
    }
    return 0;
  };

  if (!(overlap && fused)) {   // sequence on the launch stream: the fallback of the interior pass needs tmp first
    if (interior(main)) return 1;
    return shell(main);
  }
  Overlap *o = overlap_of(comm);
  if (!o) return 1;
  if (check_hip(hipEventRecord(o->fork, main), who)) return 1;              // what was issued so far (u_in, rhs) is visible to the side stream
  if (check_hip(hipStreamWaitEvent(o->side, o->fork, 0), who)) return 1;
  if (shell(o->side)) return 1;
  if (interior(main)) return 1;
  if (check_hip(hipEventRecord(o->join, o->side), who)) return 1;
  return check_hip(hipStreamWaitEvent(main, o->join, 0), who);
}

}  // namespace

extern "C" int examg_jacobi2_blocks(examg_comm_t *comm, const examg_neighbors_t *nb, const examg_layout_t *lu, const double *u_in,
                                    double *u_out, double *tmp, const examg_layout_t *lf, const double *rhs, const examg_stencil_t *st,
                                    double w, const int32_t *begin, const int32_t *end, int exchange_flags, void *workspace,
                                    size_t workspace_bytes, int overlap, examg_stream_t stream) {
  return pass_blocks<false>("examg_jacobi2_blocks", comm, nb, lu, u_in, u_out, tmp, lf, rhs, st, w, 0, begin, end, exchange_flags, workspace,
                            workspace_bytes, overlap, stream);
}

extern "C" int examg_rbgs_sweep_blocks(examg_comm_t *comm, const examg_neighbors_t *nb, const examg_layout_t *lu, const double *u_in,
                                       double *u_out, double *tmp, const examg_layout_t *lf, const double *rhs, const examg_stencil_t *st,
                                       double w, int first, const int32_t *begin, const int32_t *end, int exchange_flags, void *workspace,
                                       size_t workspace_bytes, int overlap, examg_stream_t stream) {
  if (first != 0 && first != 1) { set_error("examg_rbgs_sweep_blocks: first colour must be 0 or 1"); return 1; }
  return pass_blocks<true>("examg_rbgs_sweep_blocks", comm, nb, lu, u_in, u_out, tmp, lf, rhs, st, w, first, begin, end, exchange_flags, workspace,
                           workspace_bytes, overlap, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Transfer operators on a block WITH neighbours, one C call each (the reference's core / boundary split,
// baseExt/ir/IR_LoopOverPointsInOneFragment.scala:143-222, around the two `communicate` statements of mgCycle,
// Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:215-237).
//
// examg_residual_restrict_blocks:  communicate Solution; Residual = RHS - A * Solution; communicate Residual;
//                                  RHS@coarser = scale * Restriction * Residual
//   launch stream   the one-pass residual + restriction kernel on the coarse box shrunk by one point at interior faces: its fine
//                   footprint ends on the duplicate planes, so it reads no ghost value of Solution and never stores a residual
//   side stream     1. exchange the ghost layers of Solution
//                   2. residual on the two fine planes next to every interior face, into `res`        (thin launches)
//                   3. exchange `res` (duplicate + ghost layers, axis by axis: the restriction reads edge / corner ghosts)
//                   4. restriction of the one coarse plane on every interior face, from `res`          (thin launches, disjoint)
//   The duplicate part of `communicate Solution` (EXAMG_EXCH_DUP in exchange_flags; the loops read those planes) runs first, in
//   sequence.  `res` afterwards holds the residual on the two-plane shell only -- nothing reads the rest (mgCycle overwrites it).
// ---------------------------------------------------------------------------------------------------------------------------
extern "C" int examg_residual_restrict_blocks(examg_comm_t *comm, const examg_neighbors_t *nb, const examg_layout_t *lu, double *u,
                                              const examg_layout_t *lf, const double *rhs, const examg_layout_t *lr, double *res,
                                              const examg_stencil_t *st, const examg_layout_t *lc, double *fc, double scale,
                                              const int32_t *fbegin, const int32_t *fend, const int32_t *cbegin, const int32_t *cend,
                                              int exchange_flags, void *workspace, size_t workspace_bytes, int overlap, examg_stream_t stream) {
  const char *who = "examg_residual_restrict_blocks";
  if (!comm || !nb || !lu || !u || !lf || !rhs || !lr || !res || !st || !lc || !fc || !fbegin || !fend || !cbegin || !cend) { set_error("%s: null argument", who); return 1; }
  hipStream_t main = (hipStream_t)stream;
  const Faces f = interior_faces(lu, nb);
  if (f.n == 0) return examg_residual_restrict(lu, u, lf, rhs, lr, res, st, lc, fc, scale, fbegin, fend, cbegin, cend, stream);
  const int ghost_what = EXAMG_EXCH_GHOST | (exchange_flags & EXAMG_EXCH_CONCURRENT_AXES);
  if ((exchange_flags & EXAMG_EXCH_DUP) && examg_exchange(comm, lu, u, nb, EXAMG_EXCH_DUP, workspace, workspace_bytes, main)) return 1;
  int32_t cb1[3], ce1[3];
  shrunk(f, cbegin, cend, 1, cb1, ce1);
  bool has_interior = true;
  for (int d = 0; d < lu->nd; ++d) has_interior = has_interior && ce1[d] > cb1[d];

  auto shell = [&](hipStream_t s) -> int {
    if (examg_exchange(comm, lu, u, nb, ghost_what, workspace, workspace_bytes, s)) return 1;
    int32_t sb[3], se[3];
    for (int i = 0; i < f.n; ++i) {
      slab(fbegin, fend, f.d[i], f.side[i], 2, sb, se);
      if (examg_residual(lu, u, lf, rhs, lr, res, st, sb, se, s)) return 1;
    }
    if (examg_exchange(comm, lr, res, nb, EXAMG_EXCH_GHOST | (exchange_flags & EXAMG_EXCH_DUP), workspace, workspace_bytes, s)) return 1;
    int32_t lo[3], hi[3];
    for (int t = 0; t < 3; ++t) { lo[t] = cbegin[t]; hi[t] = cend[t]; }
    for (int d = 0; d < lu->nd; ++d) {           // disjoint coarse slabs: later axes exclude the planes earlier ones have done
      for (int sd = 0; sd < 2; ++sd) {
        if (nb->rank[d][sd] < 0) continue;
        for (int t = 0; t < 3; ++t) { sb[t] = lo[t]; se[t] = hi[t]; }
        if (sd == 0) se[d] = cbegin[d] + 1;
        else sb[d] = cend[d] - 1;
        if (se[d] <= sb[d]) continue;
        if (examg_restrict(lr, res, lc, fc, scale, sb, se, s)) return 1;
      }
      lo[d] = cb1[d];
      hi[d] = ce1[d];
    }
    return 0;
  };
  auto interior = [&](hipStream_t s) -> int {
    if (!has_interior) return 0;
    return examg_residual_restrict(lu, u, lf, rhs, lr, res, st, lc, fc, scale, fbegin, fend, cb1, ce1, s);
  };
  // the interior pass may fall back to residual + restriction THROUGH `res` (other stencils, short rows): then it must not run
  // beside the shell work, which writes `res` too -- ask the same question the entry point asks
  const bool one_pass = has_interior && examg_residual_restrict_one_pass(lu, lf, st, lc, fbegin, fend, cb1, ce1) == 1;
  if (!(overlap && one_pass)) {
    // in sequence: the whole residual first (the fallback stores it), then the exchange of `res`, then one restriction
    if (examg_exchange(comm, lu, u, nb, ghost_what, workspace, workspace_bytes, main)) return 1;
    if (examg_residual(lu, u, lf, rhs, lr, res, st, fbegin, fend, main)) return 1;
    if (examg_exchange(comm, lr, res, nb, EXAMG_EXCH_GHOST | (exchange_flags & EXAMG_EXCH_DUP), workspace, workspace_bytes, main)) return 1;
    return examg_restrict(lr, res, lc, fc, scale, cbegin, cend, main);
  }
  Overlap *o = overlap_of(comm);
  if (!o) return 1;
  if (check_hip(hipEventRecord(o->fork, main), who)) return 1;
  if (check_hip(hipStreamWaitEvent(o->side, o->fork, 0), who)) return 1;
  if (shell(o->side)) return 1;
  if (interior(main)) return 1;
  if (check_hip(hipEventRecord(o->join, o->side), who)) return 1;
  return check_hip(hipStreamWaitEvent(main, o->join, 0), who);
}

// examg_prolong_add_blocks:  communicate Solution@coarser; Solution += Prolongation@coarser * Solution@coarser
//   The interpolation of node values reads duplicate and inner coarse points only (fine node i lies between coarse nodes i / 2
//   and (i + 1) / 2, both inside the block): the duplicate part of the exchange (EXAMG_EXCH_DUP in exchange_flags) runs first,
//   the ghost part beside the kernel on the side stream.
extern "C" int examg_prolong_add_blocks(examg_comm_t *comm, const examg_neighbors_t *nb, const examg_layout_t *lc, double *uc,
                                        const examg_layout_t *lfine, double *uf, const int32_t *begin, const int32_t *end, int exchange_flags,
                                        void *workspace, size_t workspace_bytes, int overlap, examg_stream_t stream) {
  const char *who = "examg_prolong_add_blocks";
  if (!comm || !nb || !lc || !uc || !lfine || !uf || !begin || !end) { set_error("%s: null argument", who); return 1; }
  hipStream_t main = (hipStream_t)stream;
  const Faces f = interior_faces(lc, nb);
  if (f.n == 0) return examg_prolong_add(lc, uc, lfine, uf, begin, end, stream);
  if ((exchange_flags & EXAMG_EXCH_DUP) && examg_exchange(comm, lc, uc, nb, EXAMG_EXCH_DUP, workspace, workspace_bytes, main)) return 1;
  if (!overlap) {
    if (examg_exchange(comm, lc, uc, nb, EXAMG_EXCH_GHOST, workspace, workspace_bytes, main)) return 1;
    return examg_prolong_add(lc, uc, lfine, uf, begin, end, stream);
  }
  Overlap *o = overlap_of(comm);
  if (!o) return 1;
  if (check_hip(hipEventRecord(o->fork, main), who)) return 1;
  if (check_hip(hipStreamWaitEvent(o->side, o->fork, 0), who)) return 1;
  if (examg_exchange(comm, lc, uc, nb, EXAMG_EXCH_GHOST, workspace, workspace_bytes, o->side)) return 1;
  if (examg_prolong_add(lc, uc, lfine, uf, begin, end, stream)) return 1;
  if (check_hip(hipEventRecord(o->join, o->side), who)) return 1;
  return check_hip(hipStreamWaitEvent(main, o->join, 0), who);
}
