"""A stand-in for the few torch.distributed calls the product's bootstrap uses (exastencils_amd/comm.py, bench.py), carried by files
in one directory.  Test infrastructure: with it a test process can host SEVERAL ranks (one thread and one HIP stream per rank;
torch.distributed knows one rank per process), which is how the 8-rank decompositions (2 x 2 x 2, 1 x 2 x 4) are rehearsed on a
one-GPU box whose process guard allows fewer than 8 processes on the card (tests/test_gpu_ranks8.py).  Only the bootstrap
(128-byte handles, verdicts, timing scalars, the one-time duplicate check) travels this way; halo traffic, all-reduce and all-gather
of the run itself go through the peer-write transport on the device.

One object per rank.  Every collective is an all-gather of byte strings: rank r publishes `<dir>/c<seq>.<r>` (written to a
temporary name, then renamed) and reads the files of the others; a rank removes its file of collective seq - 2 when it enters
collective seq (every rank has left seq - 2 by then: leaving seq - 1 needs everybody's file of seq - 1, which is written after
seq - 2 was read).  Point-to-point messages are `<dir>/p<src>.<dst>.<n>` with one counter per ordered pair, removed by the receiver."""
import os
import time

import numpy as np
import torch


class _Work:
    def __init__(self, fn=None):
        self._fn = fn

    def wait(self):
        if self._fn is not None:
            self._fn()
            self._fn = None
        return True


class P2POp:
    def __init__(self, op, tensor, peer, group=None, tag=0):
        self.op, self.tensor, self.peer = op, tensor, peer


class FileDist:
    class ReduceOp:
        SUM, MAX, MIN = "sum", "max", "min"

    P2POp = P2POp

    def __init__(self, directory: str, rank: int, world: int, timeout: float = 150.0):
        self.dir, self.rank, self.world, self.timeout = directory, int(rank), int(world), timeout
        self._seq = 0
        self._sent, self._rcvd = {}, {}
        os.makedirs(directory, exist_ok=True)

    # -- what comm.py / bench.py ask the module ---------------------------------------------------------------------------------
    def is_initialized(self):
        return True

    def get_world_size(self, group=None):
        return self.world

    def get_rank(self, group=None):
        return self.rank

    def get_backend(self, group=None):
        return "file"            # not "nccl": tensors on this wire are host tensors

    # -- files ------------------------------------------------------------------------------------------------------------------
    def _put(self, name: str, data: bytes):
        path = os.path.join(self.dir, name)
        tmp = path + ".tmp%d" % self.rank
        with open(tmp, "wb") as f:
            f.write(data)
        os.rename(tmp, path)

    def _get(self, name: str, remove: bool = False) -> bytes:
        path = os.path.join(self.dir, name)
        t0 = time.monotonic()
        nap = 0.0002
        while True:
            try:
                with open(path, "rb") as f:
                    data = f.read()
                if remove:
                    os.remove(path)
                return data
            except FileNotFoundError:
                if time.monotonic() - t0 > self.timeout:
                    raise TimeoutError("FileDist rank %d: %s did not appear within %g s" % (self.rank, name, self.timeout)) from None
                time.sleep(nap)
                nap = min(nap * 1.5, 0.01)

    def _allgather_bytes(self, data: bytes):
        q = self._seq
        self._seq += 1
        if q >= 2:
            try:
                os.remove(os.path.join(self.dir, "c%d.%d" % (q - 2, self.rank)))
            except FileNotFoundError:
                pass
        self._put("c%d.%d" % (q, self.rank), data)
        return [data if r == self.rank else self._get("c%d.%d" % (q, r)) for r in range(self.world)]

    # -- collectives on host tensors ------------------------------------------------------------------------------------------------
    @staticmethod
    def _bytes(t) -> bytes:
        return t.detach().cpu().contiguous().numpy().tobytes()

    @staticmethod
    def _fill(t, data: bytes):
        a = np.frombuffer(data, dtype=t.detach().cpu().numpy().dtype).reshape(tuple(t.shape))
        t.copy_(torch.from_numpy(a.copy()))

    def barrier(self, group=None):
        self._allgather_bytes(b"b")

    def all_gather(self, outs, t, group=None):
        for o, d in zip(outs, self._allgather_bytes(self._bytes(t))):
            self._fill(o, d)

    def all_reduce(self, t, op="sum", group=None):
        parts = [np.frombuffer(d, dtype=t.detach().cpu().numpy().dtype) for d in self._allgather_bytes(self._bytes(t))]
        acc = parts[0].copy()
        for p in parts[1:]:                      # rank order: the same bits on every rank
            acc = acc + p if op == "sum" else (np.maximum(acc, p) if op == "max" else np.minimum(acc, p))
        self._fill(t, acc.tobytes())

    def broadcast(self, t, src, group=None):
        self._fill(t, self._allgather_bytes(self._bytes(t) if self.rank == src else b"")[src])

    # -- point to point -----------------------------------------------------------------------------------------------------------
    def isend(self, t, dst, group=None, tag=0):
        n = self._sent.get(dst, 0)
        self._sent[dst] = n + 1
        self._put("p%d.%d.%d" % (self.rank, dst, n), self._bytes(t))
        return _Work()

    def irecv(self, t, src, group=None, tag=0):
        n = self._rcvd.get(src, 0)
        self._rcvd[src] = n + 1
        return _Work(lambda: self._fill(t, self._get("p%d.%d.%d" % (src, self.rank, n), remove=True)))

    def send(self, t, dst, group=None, tag=0):
        self.isend(t, dst).wait()

    def recv(self, t, src, group=None, tag=0):
        self.irecv(t, src).wait()

    def batch_isend_irecv(self, ops):
        return [o.op(o.tensor, o.peer) for o in ops]
