"""Cost of one ghost-layer exchange per axis on a 512^3-cell block (515^3-double field, reference layout), peer-write transport with the
block as its own neighbour (periodic in ONE dimension: pack into the own receive slab, flag, unpack -- the kernels and the flag protocol of
a real neighbour, minus the link).  x faces are 515 x 515 single doubles at a stride of one row (IR_PackInfoGhost.scala:13-60: one
64-byte segment touched per value), y faces 515 rows of 515 doubles at a stride of one plane, z faces one contiguous plane.
usage: python tools/exchange_faces.py [level]        (tools/gpu_check.sh runs it on the GPU box)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from exastencils_amd.comm import Communicator
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.field import Field
    from exastencils_amd.layout import FieldLayout
    from exastencils_amd.ops import HipOps

    L = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    ops = HipOps(0)
    out = {"level": L}
    for d, name in enumerate("xyz"):
        per = [False] * 3
        per[d] = True
        dom = RectDomain(3, (1, 1, 1), 0, periodic=tuple(per))
        comm = Communicator(dom, ops, consistent_duplicates=True, transport="peer")
        S = Field("Solution", L, FieldLayout.node(3, dom.ncells(L), 1, True, True, 0), ops, 1, None)
        ops.fill_random(S.data(), 5)
        for _ in range(5):
            comm.exchange(S, None, "ghost")
        comm.check()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            comm.exchange(S, None, "ghost")
        e1.record()
        torch.cuda.synchronize()
        comm.check()
        nface = (dom.ncells(L)[0] + 3) ** 2
        out[name] = {"us_per_exchange_both_sides": e0.elapsed_time(e1) / n * 1e3, "bytes_per_face": 8 * nface}
        comm.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
