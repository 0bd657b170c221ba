// Internals shared by the two block-to-block transports of libexamg: examg_comm.hip (RCCL send / recv groups) and
// examg_peer.hip (peer writes into HIP-IPC regions).  Not part of the ABI.
#pragma once
#include <rccl/rccl.h>

#include "examg_common.h"

namespace examg {
struct PeerState;   // examg_peer.hip
}

struct examg_comm {
  ncclComm_t nccl = nullptr;   // null for a one-rank communicator created without RCCL and for a peer-write communicator
  int rank = 0, size = 1;
  bool self_via_rccl = false;  // periodic self-exchange through ncclSend/ncclRecv to the own rank (one-GPU test of the transport)
  hipStream_t side = nullptr;  // second stream and fork / join events of the overlapped passes (created on first use)
  hipEvent_t fork = nullptr, join = nullptr;
  examg::PeerState *peer = nullptr;   // peer-write transport (examg_comm_create_peer); then nccl stays null
};

namespace examg {

// ---- index ranges (iterator coordinates: 0 = lower duplicate node) -----------------------------------------------------------
struct Marks {
  int GLB, DLB, DLE, IB, IE, DRB, DRE, GRB, GRE;
};
static inline Marks marks(const examg_layout_t *l, int d) {
  Marks m;
  m.GLB = -l->ghost_l[d];
  m.DLB = 0;
  m.DLE = l->dup_l[d];
  m.IB = m.DLE;
  m.IE = m.IB + l->inner[d];
  m.DRB = m.IE;
  m.DRE = m.DRB + l->dup_r[d];
  m.GRB = m.DRE;
  m.GRE = m.GRB + l->ghost_r[d];
  return m;
}

struct Range {
  int32_t b[3], e[3];
  long long count() const {
    long long n = 1;
    for (int d = 0; d < 3; ++d) n *= (e[d] > b[d] ? e[d] - b[d] : 0);
    return n;
  }
};

// duplicate layers along axis d (IR_PackInfoDuplicate.scala:15-39): send DRB..DRE, receive into DLB..DLE, tangentially DLB..DRE
static inline void dup_ranges(const examg_layout_t *l, int d, Range &snd, Range &rcv) {
  for (int t = 0; t < 3; ++t) {
    snd.b[t] = rcv.b[t] = 0;
    snd.e[t] = rcv.e[t] = 1;
  }
  for (int t = 0; t < l->nd; ++t) {
    const Marks m = marks(l, t);
    if (t == d) {
      snd.b[t] = m.DRB; snd.e[t] = m.DRE;
      rcv.b[t] = m.DLB; rcv.e[t] = m.DLE;
    } else {
      snd.b[t] = rcv.b[t] = m.DLB;
      snd.e[t] = rcv.e[t] = m.DRE;
    }
  }
}

// ghost layers along axis d towards `side` (IR_PackInfoGhost.scala:13-60): send the first / last inner planes, receive into the
// ghost planes of that side; tangentially GLB..GRE (ghosts of earlier axes travel along)
static inline void ghost_ranges(const examg_layout_t *l, int d, int side, Range &snd, Range &rcv) {
  for (int t = 0; t < 3; ++t) {
    snd.b[t] = rcv.b[t] = 0;
    snd.e[t] = rcv.e[t] = 1;
  }
  for (int t = 0; t < l->nd; ++t) {
    const Marks m = marks(l, t);
    if (t == d) {
      // what goes towards - fills the neighbour's + ghost layers (all blocks share the layout) and vice versa
      if (side < 0) {
        snd.b[t] = m.IB; snd.e[t] = m.IB + l->ghost_r[t];
        rcv.b[t] = m.DLB - l->ghost_l[t]; rcv.e[t] = m.DLB;
      } else {
        snd.b[t] = m.IE - l->ghost_l[t]; snd.e[t] = m.IE;
        rcv.b[t] = m.GRB; rcv.e[t] = m.GRB + l->ghost_r[t];
      }
    } else {
      snd.b[t] = rcv.b[t] = m.GLB;
      snd.e[t] = rcv.e[t] = m.GRE;
    }
  }
}

static inline long long face_count(const examg_layout_t *l, int d) {   // points of the largest message of axis d (one ghost or duplicate slab)
  long long n = 1;
  for (int t = 0; t < l->nd; ++t) {
    const Marks m = marks(l, t);
    if (t == d) {
      int w = l->ghost_l[t] > l->ghost_r[t] ? l->ghost_l[t] : l->ghost_r[t];
      if (l->dup_r[t] > w) w = l->dup_r[t];
      n *= w;
    } else {
      n *= (m.GRE - m.GLB);
    }
  }
  return n;
}

// workspace: per axis d and side s (0 = minus, 1 = plus) one send and one receive slab
static inline long long slot_offset(const examg_layout_t *l, int d, int s, int recv) {
  long long off = 0;
  for (int t = 0; t < d; ++t) off += 4 * face_count(l, t);
  return off + (2 * s + recv) * face_count(l, d);
}


// ---- peer-write backend (examg_peer.hip), called by the dispatching entry points of examg_comm.hip ---------------------------
struct PeerMsg {
  int peer;          // rank of the block on the other end (the own rank: periodic dimension with one block)
  int d, side;       // channel: axis and side (0 = minus, 1 = plus) of THIS block the message leaves through / arrives at
  int32_t b[3], e[3];
};
int peer_phase(examg_comm *c, const examg_layout_t *l, double *x, const PeerMsg *sends, int ns, const PeerMsg *recvs, int nr,
               hipStream_t s);
int peer_allreduce(examg_comm *c, double *x, int n, int op, hipStream_t s);
int peer_allgather(examg_comm *c, const double *send, double *recv, long long n, hipStream_t s);
void peer_destroy(examg_comm *c);

}  // namespace examg
