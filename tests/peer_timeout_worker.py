"""Two ranks on one device; after one good exchange rank 0 exchanges AGAIN while rank 1 does not: rank 0's receive kernel must give up
after EXAMG_PEER_TIMEOUT_MS, the communicator must report it (examg_comm_status), and nothing may hang.  Test infrastructure
(tests/test_gpu_peer.py).  usage: peer_timeout_worker.py <rank> <port> <out.json>"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    from exastencils_amd.comm import Communicator
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.field import Field
    from exastencils_amd.layout import FieldLayout
    from exastencils_amd.lib import ExamgError
    from exastencils_amd.ops import HipOps

    ops = HipOps(0)
    dom = RectDomain(3, (1, 1, 2), rank, (2, 2, 1))
    comm = Communicator(dom, ops)
    S = Field("Solution", 4, FieldLayout.node(3, dom.ncells(4), 1, True, True, 0), ops, 1, None)
    ops.fill_random(S.data(), 5 + rank)
    comm.exchange(S, None, "all")
    comm.check()                      # a matched exchange: no error
    dist.barrier()
    res = {"rank": rank, "transport": comm.transport}
    if rank == 0:
        t0 = time.time()
        comm.exchange(S, None, "ghost")          # nobody answers
        try:
            comm.check()
            res["error"] = None
        except ExamgError as ex:
            res["error"] = str(ex)
        res["seconds"] = time.time() - t0
        comm.exchange(S, None, "ghost")          # a communicator that gave up does not wait again
        t1 = time.time()
        ops.synchronize()
        res["second_attempt_seconds"] = time.time() - t1
    json.dump(res, open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
