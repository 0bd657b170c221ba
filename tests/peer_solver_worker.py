"""One rank of the decomposed solver runs of tests/test_gpu_peer.py: BASELINE configs[3] (27-entry Helmholtz, Jacobi) and configs[4]
(FMG start + red-black cycles) in their multi-process form, fresh child processes sharing ONE device, every halo exchange and
reduction device-resident through the peer-write transport.  Test infrastructure.

usage: peer_solver_worker.py <rank> <world> <bx,by,bz> <port> <dir> <case>      (case: a key of <dir>/cases.json)
writes <dir>/<case>_<rank>.npy (owned box of the finest solution) and <dir>/<case>_<rank>.json."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    blocks = tuple(int(x) for x in sys.argv[3].split(","))
    port, out, case = int(sys.argv[4]), sys.argv[5], sys.argv[6]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    from exastencils_amd.comm import Communicator
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.ops import HipOps
    from exastencils_amd.solver import ConfigL3, SolverFromL3

    kw = json.load(open(os.path.join(out, "cases.json")))[case]
    ops = HipOps(0)
    flen = tuple(2 // blocks[d] for d in range(3))
    dom = RectDomain(3, blocks, rank, flen)
    comm = Communicator(dom, ops, concurrent_ghost_axes=True)
    cfg = ConfigL3(frag_len=flen, **kw)
    P = SolverFromL3(cfg, ops, dom, comm)
    P.setup()
    its = P.Solve()
    comm.check()
    S = P.Solution[cfg.max_level]
    lay, nc = S.layout, dom.ncells(cfg.max_level)
    a = ops.to_host(S.data()).reshape(lay.shape_zyx)
    own = a[tuple(slice(lay.ref(d), lay.ref(d) + nc[d] + 1) for d in (2, 1, 0))].copy()
    np.save(os.path.join(out, "%s_%d.npy" % (case, rank)), own)
    json.dump({"transport": comm.transport, "it": its, "res": list(P.res_history), "err": list(getattr(P, "err_history", [])),
               "exchanges": comm.stats["c_exchanges"], "dup_consistent": bool(comm.check_duplicates(S))},
              open(os.path.join(out, "%s_%d.json" % (case, rank)), "w"))
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
