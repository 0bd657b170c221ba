"""`communicate <field>`: duplicate- and ghost-layer exchange between the blocks of the decomposition,
and the scalar all-reduce that follows every reduction loop.

Reference: IR_CommunicateFunction.compileBody (Compiler/src/exastencils/communication/ir/
IR_CommunicateFunction.scala:412-471): duplicate layers first -- per axis, own UPPER duplicate plane to
the '+' neighbour, received into the LOWER plane from the '-' neighbour (comm_batchCommunication) --
then ghost layers per axis in both directions, tangential extent GLB..GRE so that edge/corner ghosts
become valid with 6 neighbours only (comm_syncGhostData; ranges IR_PackInfoDuplicate.scala:15-39,
IR_PackInfoGhost.scala:13-60); pack -> send / recv -> unpack (:194-219); MPI_Allreduce after reduction
loops (parallelization/api/mpi/MPI_Reduction.scala:100-126).

Transport, two forms behind one interface:
  * "c": libexamg's own transport (include/examg.h: examg_exchange / examg_allreduce / examg_allgather,
    exastencils_amd/csrc/examg_comm.hip) -- ncclSend / ncclRecv groups of RCCL between the axis neighbours (one process per
    GPU; each pair of GPUs has its own xGMI link), pack / unpack kernels, all stream-ordered and capturable into a hipGraph.
    This is the product path on GPUs and the one a generated C++ host shares; the 128-byte RCCL id of rank 0 travels through
    torch.distributed (whatever backend the launcher initialised), which otherwise carries no data.
  * "torch": torch.distributed point-to-point batches around `ops.pack/unpack` -- backend "gloo" with the CPU oracle ops in the
    multi-process CPU tests, or "gloo" with device arrays staged through the host (several ranks rehearsing on ONE GPU, where
    RCCL refuses two ranks on one device).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

from .domain import RectDomain
from .field import Field


_SHARED_C_COMMS: Dict[Tuple, object] = {}     # (library, world size, rank) -> examg_comm_t*: one RCCL communicator per process


class Communicator:
    def __init__(self, domain: RectDomain, ops, group=None, concurrent_ghost_axes: bool = False,
                 consistent_duplicates: bool = False, transport: str = "auto"):
        """concurrent_ghost_axes: where the caller declares that only face ghosts will be read (`axis_only=True`:
        5/7-point stencil loops), send the ghost planes of all axes in ONE point-to-point batch instead of axis by axis.
        Face ghosts are identical; edge/corner ghosts -- which only the axis-by-axis order makes valid and which
        restriction, prolongation and 27-point stencils read -- stay stale.  Cuts the latency of such an exchange by the
        number of dimensions."""
        self.domain, self.ops, self.group = domain, ops, group
        self.concurrent_ghost_axes = concurrent_ghost_axes
        # consistent_duplicates: every loop of the multigrid programs computes the duplicate (shared) planes on BOTH blocks
        # from the same inputs (iteration offsets 0 at interior faces) with the same kernel, i.e. to the same bits; the
        # upstream duplicate exchange of `communicate` then rewrites values with themselves and can be left out.  Opt-in:
        # a caller that writes one side only (external data, different arithmetic per block) must keep it.
        self.consistent_duplicates = consistent_duplicates
        self.dist = None
        if domain.world_size > 1:
            import torch.distributed as dist

            if not dist.is_initialized():
                raise RuntimeError("a decomposition with %d blocks needs torch.distributed to be initialised" % domain.world_size)
            if dist.get_world_size(group) != domain.world_size:
                raise RuntimeError("world size %d != number of blocks %d" % (dist.get_world_size(group), domain.world_size))
            if dist.get_rank(group) != domain.rank:
                raise RuntimeError("rank mismatch between torch.distributed and the domain")
            self.dist = dist
        # rehearsal mode: backend "gloo" with device arrays (several ranks sharing one GPU) -- messages are staged through
        # the host.  Product runs use "nccl" (RCCL), where device buffers go on the wire directly.
        self._stage = bool(self.dist is not None and self.dist.get_backend(group) == "gloo" and getattr(ops, "device", None) is not None
                           and getattr(ops.device, "type", "cpu") != "cpu")
        self._bufs: Dict[Tuple, object] = {}
        self.stats = {"messages": 0, "bytes": 0}
        # transport selection: the C transport whenever the kernel layer is libexamg on a GPU and messages need no host staging
        on_gpu = hasattr(ops, "L") and hasattr(ops.L, "examg_exchange") and getattr(getattr(ops, "device", None), "type", "cpu") != "cpu"
        if transport == "auto":
            import os

            forced = os.environ.get("EXAMG_TRANSPORT", "")       # "torch": keep every message on torch.distributed (diagnosis)
            transport = forced if forced in ("c", "torch") else ("c" if (on_gpu and not self._stage) else "torch")
        if transport == "c" and not on_gpu:
            raise RuntimeError("the C transport needs the HIP kernel layer (HipOps)")
        self.transport = transport
        self._c = None
        self._ws: Dict[Tuple, object] = {}
        self._nb = None
        if transport == "c" and (self.dist is not None or any(domain.periodic)):
            try:
                self._c_create()
            except Exception as ex:      # RCCL not loadable, id exchange failed: the torch.distributed path still works
                import warnings

                warnings.warn("libexamg transport unavailable (%s); using torch.distributed point-to-point" % (ex,))
                self._c, self.transport = None, "torch"

    # -- C transport (libexamg / RCCL) -----------------------------------------------------------------
    def _c_create(self):
        import ctypes as C

        from . import lib as _lib

        L, dom = self.ops.L, self.domain
        nb = _lib.NeighborsC()
        for d in range(3):
            for s_, side in enumerate((-1, +1)):
                r = dom.neighbor(d, side) if d < dom.nd else None
                nb.rank[d][s_] = -1 if r is None else int(r)
        self._nb = nb
        key = (id(L), dom.world_size, dom.rank)
        if self.dist is not None and self.group is None and key in _SHARED_C_COMMS:
            self._c, self._c_shared = _SHARED_C_COMMS[key], True
            return
        self._c_shared = False
        idbuf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
        if self.dist is not None:
            torch = self.ops.torch
            if dom.rank == 0:
                _lib.check(L.examg_comm_unique_id(idbuf), "examg_comm_unique_id")
            dev = self.ops.device if self.dist.get_backend(self.group) == "nccl" else "cpu"
            t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=dev)
            self.dist.broadcast(t, 0, group=self.group)
            idbuf = (C.c_ubyte * _lib.COMM_ID_BYTES)(*t.cpu().tolist())
            idp = C.cast(idbuf, C.c_void_p)
        else:
            idp = None
            import os

            if os.environ.get("EXAMG_COMM_SELF_RCCL") == "1":     # one-GPU test of the RCCL path: self-messages through RCCL
                _lib.check(L.examg_comm_unique_id(idbuf), "examg_comm_unique_id")
                idp = C.cast(idbuf, C.c_void_p)
        h = C.c_void_p()
        _lib.check(L.examg_comm_create(C.byref(h), idp, dom.world_size, dom.rank), "examg_comm_create")
        self._c = h
        if self.dist is not None and self.group is None:
            _SHARED_C_COMMS[key] = h
            self._c_shared = True

    def close(self):
        if self._c is not None and not getattr(self, "_c_shared", False):
            self.ops.L.examg_comm_destroy(self._c)
        self._c = None

    def _c_exchange(self, f: Field, x, what: int):
        import ctypes as C

        from . import lib as _lib

        L = self.ops.L
        key = (f.layout,)
        ws = self._ws.get(key)
        nbytes = int(L.examg_exchange_workspace_bytes(C.byref(f.lc)))
        if ws is None:
            ws = self._ws[key] = self.ops.new_array(max(1, nbytes // 8))
        _lib.check(L.examg_exchange(self._c, C.byref(f.lc), self.ops.ptr(x), C.byref(self._nb), int(what), self.ops.ptr(ws), nbytes,
                                    self.ops._stream()), "examg_exchange")

    # -- buffers -----------------------------------------------------------------------------------
    def _buf(self, key, n: int):
        b = self._bufs.get(key)
        if b is None or b.numel() != n:
            b = self.ops.new_array(n)
            self._bufs[key] = b
        return b

    # -- index ranges ------------------------------------------------------------------------------
    @staticmethod
    def dup_ranges(layout, nd: int, d: int):
        """(send box, recv box): send DRB..DRE (own upper dup) / recv DLB..DLE, tangential DLB..DRE."""
        sb, se, rb, re_ = [0, 0, 0], [1, 1, 1], [0, 0, 0], [1, 1, 1]
        for t in range(nd):
            if t == d:
                sb[t], se[t] = layout.idx("DRB", t), layout.idx("DRE", t)
                rb[t], re_[t] = layout.idx("DLB", t), layout.idx("DLE", t)
            else:
                sb[t], se[t] = layout.idx("DLB", t), layout.idx("DRE", t)
                rb[t], re_[t] = sb[t], se[t]
        return (sb, se), (rb, re_)

    @staticmethod
    def ghost_ranges(layout, nd: int, d: int, side: int):
        """Boxes for the exchange with the neighbour on `side` of axis d: what I send there (my first/last
        inner planes) and where I receive what it sends me (my ghost planes on that side)."""
        g = layout.ghost[d]
        sb, se, rb, re_ = [0, 0, 0], [1, 1, 1], [0, 0, 0], [1, 1, 1]
        for t in range(nd):
            if t == d:
                if side < 0:
                    sb[t], se[t] = layout.idx("IB", t), layout.idx("IB", t) + g
                    rb[t], re_[t] = layout.idx("GLE", t) - g, layout.idx("GLE", t)
                else:
                    sb[t], se[t] = layout.idx("IE", t) - g, layout.idx("IE", t)
                    rb[t], re_[t] = layout.idx("GRB", t), layout.idx("GRB", t) + g
            else:
                sb[t], se[t] = layout.idx("GLB", t), layout.idx("GRE", t)
                rb[t], re_[t] = sb[t], se[t]
        return (sb, se), (rb, re_)

    @staticmethod
    def _count(box) -> int:
        n = 1
        for t in range(3):
            n *= max(0, box[1][t] - box[0][t])
        return n

    # -- exch<Field>_<level>(slot) ----------------------------------------------------------------
    def exchange(self, f: Field, slot: Optional[int] = None, what: str = "all", axis_only: bool = False):
        if self.dist is None and not any(self.domain.periodic):
            return   # single block, non-periodic: no neighbours, the generated exch function is empty
        lay, dom, nd = f.layout, self.domain, self.domain.nd
        x = f.data(slot)
        if self._c is not None:
            from .lib import EXCH_CONCURRENT_AXES, EXCH_DUP, EXCH_GHOST

            w = 0
            if what in ("all", "dup") and lay.communicates_dup and max(lay.dup) > 0 and not self.consistent_duplicates:
                w |= EXCH_DUP
            if what in ("all", "ghost") and lay.communicates_ghost and max(lay.ghost) > 0:
                w |= EXCH_GHOST
                if self.concurrent_ghost_axes and axis_only:
                    w |= EXCH_CONCURRENT_AXES
            if w:
                self._c_exchange(f, x, w)
            return
        if what in ("all", "dup") and lay.communicates_dup and max(lay.dup) > 0 and not self.consistent_duplicates:
            for d in range(nd):
                plus, minus = dom.neighbor(d, +1), dom.neighbor(d, -1)
                sbox, rbox = self.dup_ranges(lay, nd, d)
                sends, recvs = [], []
                if plus is not None:
                    sends.append((plus, sbox, ("dup", f.name, f.level, d, "s")))
                if minus is not None:
                    recvs.append((minus, rbox, ("dup", f.name, f.level, d, "r")))
                self._phase(f, x, sends, recvs)
        if what in ("all", "ghost") and lay.communicates_ghost and max(lay.ghost) > 0:
            all_sends, all_recvs = [], []
            for d in range(nd):
                sends, recvs = [], []
                for side in (-1, +1):
                    peer = dom.neighbor(d, side)
                    if peer is None:
                        continue
                    sbox, rbox = self.ghost_ranges(lay, nd, d, side)
                    sends.append((peer, sbox, ("ghost", f.name, f.level, d, side, "s")))
                    recvs.append((peer, rbox, ("ghost", f.name, f.level, d, side, "r")))
                if self.concurrent_ghost_axes and axis_only:
                    all_sends += sends
                    all_recvs += recvs
                else:
                    self._phase(f, x, sends, recvs)
            if self.concurrent_ghost_axes and axis_only:
                self._phase(f, x, all_sends, all_recvs)

    def _phase(self, f: Field, x, sends: List, recvs: List):
        """pack -> isend ; irecv -> wait -> unpack for one axis (IR_CommunicateFunction.scala:194-219)."""
        if not sends and not recvs:
            return
        dist, ops = self.dist, self.ops
        me = self.domain.rank
        # a periodic dimension with one block: this block is its own neighbour -- what goes out on one side comes in on the
        # other (send towards `side` pairs with the receive from `-side`), copied through the pack buffer
        for peer, sbox, skey in [x_ for x_ in sends if x_[0] == me]:
            sside = skey[4] if skey[0] == "ghost" else +1
            for rpeer, rbox, rkey in recvs:
                rside = rkey[4] if rkey[0] == "ghost" else -1
                if rpeer == me and rkey[:4] == skey[:4] and rside == -sside:
                    buf = self._buf(skey, self._count(sbox))
                    ops.pack(f.lc, x, buf, sbox[0], sbox[1])
                    ops.unpack(f.lc, x, buf, rbox[0], rbox[1])
        sends = [x_ for x_ in sends if x_[0] != me]
        recvs = [x_ for x_ in recvs if x_[0] != me]
        if not sends and not recvs:
            return
        p2p, rbufs = [], []
        # receives are posted in the opposite side order of the sends: when both neighbours of an axis are the same rank
        # (two blocks, periodic) the first message sent (towards -) is the first one the peer expects (from +)
        for peer, box, key in reversed(recvs):
            buf = self._buf(key, self._count(box))
            wire = buf.cpu() if self._stage else buf
            rbufs.append((buf, wire, box))
            p2p.append(dist.P2POp(dist.irecv, wire, peer, self.group))
        for peer, box, key in sends:
            buf = self._buf(key, self._count(box))
            ops.pack(f.lc, x, buf, box[0], box[1])
            p2p.append(dist.P2POp(dist.isend, buf.cpu() if self._stage else buf, peer, self.group))
            self.stats["messages"] += 1
            self.stats["bytes"] += 8 * buf.numel()
        for w in dist.batch_isend_irecv(p2p):
            w.wait()
        for buf, wire, box in rbufs:
            if self._stage:
                buf.copy_(wire)
            ops.unpack(f.lc, x, buf, box[0], box[1])

    def c_pass(self, kind: str, S: Field, u_in, u_out, tmp, F: Field, A, w: float, first: int, begin, end, axis_only: bool, overlap: bool):
        """One overlapped smoother pass on a block with neighbours as ONE library call (examg_jacobi2_blocks /
        examg_rbgs_sweep_blocks: interior two-stage kernel on the launch stream, exchanges and shell launches on the
        communicator's side stream).  Only with the C transport; True if the call was made."""
        if self._c is None or not (S.layout.communicates_ghost and max(S.layout.ghost) > 0):
            return False
        import ctypes as C

        from . import lib as _lib

        ops, L = self.ops, self.ops.L
        key = (S.layout,)
        ws = self._ws.get(key)
        nbytes = int(L.examg_exchange_workspace_bytes(C.byref(S.lc)))
        if ws is None:
            ws = self._ws[key] = ops.new_array(max(1, nbytes // 8))
        flags = _lib.EXCH_CONCURRENT_AXES if (self.concurrent_ghost_axes and axis_only) else 0
        sc = A.c_struct(ops.ptr)
        if kind == "jacobi2":
            rc = L.examg_jacobi2_blocks(self._c, C.byref(self._nb), C.byref(S.lc), ops.ptr(u_in), ops.ptr(u_out), ops.ptr(tmp), C.byref(F.lc),
                                        ops.ptr(F.data()), C.byref(sc), float(w), _lib.ivec(begin), _lib.ivec(end), flags, ops.ptr(ws), nbytes,
                                        1 if overlap else 0, ops._stream())
        else:
            rc = L.examg_rbgs_sweep_blocks(self._c, C.byref(self._nb), C.byref(S.lc), ops.ptr(u_in), ops.ptr(u_out), ops.ptr(tmp), C.byref(F.lc),
                                           ops.ptr(F.data()), C.byref(sc), float(w), int(first), _lib.ivec(begin), _lib.ivec(end), flags,
                                           ops.ptr(ws), nbytes, 1 if overlap else 0, ops._stream())
        _lib.check(rc, "examg_%s_blocks" % ("jacobi2" if kind == "jacobi2" else "rbgs_sweep"))
        return True

    # -- reductions across blocks -------------------------------------------------------------------
    def allreduce(self, t, op: str = "sum"):
        """MPI_Allreduce(MPI_IN_PLACE, &x, 1, MPI_DOUBLE, op) on a device scalar."""
        if self.dist is None:
            return t
        if self._c is not None:
            from .lib import check

            check(self.ops.L.examg_allreduce(self._c, self.ops.ptr(t), int(t.numel()), 0 if op == "sum" else 1, self.ops._stream()),
                  "examg_allreduce")
            return t
        rop = self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX
        if self._stage:
            c = t.cpu()
            self.dist.all_reduce(c, op=rop, group=self.group)
            t.copy_(c)
            return t
        self.dist.all_reduce(t, op=rop, group=self.group)
        return t

    def all_gather(self, outs: List, t):
        """Every rank's `t` into `outs[rank]` (coarse-level agglomeration, exastencils_amd/solver.py); `outs` are consecutive
        pieces of ONE array (solver.py allocates them that way), which is what the C transport writes into."""
        if self._c is not None:
            from .lib import check

            n = int(t.numel())
            base = outs[0]
            for r, o in enumerate(outs):
                if o.data_ptr() != base.data_ptr() + 8 * n * r:
                    raise RuntimeError("all_gather: the receive pieces must be consecutive parts of one array")
            check(self.ops.L.examg_allgather(self._c, self.ops.ptr(t), self.ops.ptr(base), n, self.ops._stream()), "examg_allgather")
            return
        if self._stage:
            couts = [o.cpu() for o in outs]
            self.dist.all_gather(couts, t.cpu(), group=self.group)
            for o, c in zip(outs, couts):
                o.copy_(c)
            return
        self.dist.all_gather(outs, t, group=self.group)

    # -- assertion mode for `consistent_duplicates` ---------------------------------------------------------
    def check_duplicates(self, f: Field, slot: Optional[int] = None) -> bool:
        """Exchange the duplicate planes of `f` into scratch buffers (the field is not touched) and compare them bit for bit
        with the planes this block holds: True on every rank iff all shared planes agree.  Run once before relying on
        `consistent_duplicates=True` (which leaves the upstream duplicate exchange out because both owners of a shared
        plane compute it from the same inputs with the same kernel)."""
        if self.dist is None:
            return True
        lay, dom, nd, ops = f.layout, self.domain, self.domain.nd, self.ops
        x = f.data(slot)
        bad = 0
        # this one-time check travels through torch.distributed: device buffers on the "nccl" backend, host copies on "gloo"
        host_wire = self.dist.get_backend(self.group) != "nccl" and getattr(getattr(ops, "device", None), "type", "cpu") != "cpu"
        for d in range(nd):
            plus, minus = dom.neighbor(d, +1), dom.neighbor(d, -1)
            sbox, rbox = self.dup_ranges(lay, nd, d)
            n = self._count(sbox)
            p2p = []
            rbuf = sbuf = None
            if minus is not None and minus != dom.rank:
                rbuf = self._buf(("dupcheck", d, "r"), n)
                wire_r = rbuf.cpu() if host_wire else rbuf
                p2p.append(self.dist.P2POp(self.dist.irecv, wire_r, minus, self.group))
            if plus is not None and plus != dom.rank:
                sbuf = self._buf(("dupcheck", d, "s"), n)
                ops.pack(f.lc, x, sbuf, sbox[0], sbox[1])
                ops.synchronize()
                p2p.append(self.dist.P2POp(self.dist.isend, sbuf.cpu() if host_wire else sbuf, plus, self.group))
            if p2p:
                for w in self.dist.batch_isend_irecv(p2p):
                    w.wait()
            if rbuf is not None:
                mine = self._buf(("dupcheck", d, "m"), n)
                ops.pack(f.lc, x, mine, rbox[0], rbox[1])
                ops.synchronize()
                import numpy as np

                got = wire_r.detach().cpu().numpy()
                if not np.array_equal(got.view(np.uint64), np.asarray(ops.to_host(mine)).view(np.uint64)):
                    bad += 1
        t = self.ops.torch.tensor([float(bad)], dtype=self.ops.torch.float64)
        if self.dist.get_backend(self.group) == "nccl":
            t = t.to(ops.device)
        self.dist.all_reduce(t, group=self.group)
        return float(t.item()) == 0.0
