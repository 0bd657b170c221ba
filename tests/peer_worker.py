"""One rank of the multi-process GPU tests of the peer-write transport (tests/test_gpu_peer.py starts N of these as fresh child
processes that share ONE device; torch.distributed/gloo carries only the 128-byte IPC handles).  Test infrastructure.

usage: peer_worker.py <rank> <world> <bx,by,bz> <port> <dir>
  <dir>/u.npy, f.npy, uc.npy   global node arrays written by the parent (ghost-1 layout / no-ghost layout / coarse ghost-1 layout)
  <dir>/spec.json              {"level": L, "pairs": k, "sweeps": k, "cycle": {...ConfigL4 overrides...}}
writes <dir>/out_<rank>.npz (owned boxes of every case) and <dir>/out_<rank>.json (histories, checks)."""
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
    port, out = int(sys.argv[4]), sys.argv[5]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    run_rank(rank, world, blocks, out, dist, None)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


def run_rank(rank, world, blocks, out, dist, dist_module):
    """Everything one rank does.  dist_module None: torch.distributed (one rank per process, main above); otherwise the bootstrap object
    every Communicator is given (tests/filedist.py: several ranks per process, tests/ranks_host.py)."""
    import torch

    from exastencils_amd.comm import Communicator as _Communicator
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.field import Field, laplace_fd
    from exastencils_amd.layout import FieldLayout
    from exastencils_amd.ops import HipOps
    from exastencils_amd.smoothers import jacobi_pair, rbgs_sweep

    def Communicator(dom, ops, **kw):
        return _Communicator(dom, ops, dist_module=dist_module, **kw)

    spec = json.load(open(os.path.join(out, "spec.json")))
    L = spec["level"]
    ops = HipOps(0)
    total = spec.get("total_frag", 2)          # fragment lengths of all blocks of a dimension add up to this
    flen = tuple(total // blocks[d] for d in range(3))
    dom = RectDomain(3, blocks, rank, flen)
    result = {"transport": None, "checks": {}}
    arrays = {}

    nc, ncc = dom.ncells(L), dom.ncells(L - 1)
    lay_u = FieldLayout.node(3, nc, 1, True, True, 0)
    lay_f = FieldLayout.node(3, nc, 0, True, False, 0)
    lay_c = FieldLayout.node(3, ncc, 1, True, True, 0)
    lay_cf = FieldLayout.node(3, ncc, 0, True, False, 0)

    def local(gname, n_cells, ghost):
        g = np.load(os.path.join(out, gname), mmap_mode="r")
        sl = tuple(slice(dom.pos[d] * n_cells[d], dom.pos[d] * n_cells[d] + n_cells[d] + 1 + 2 * ghost) for d in (2, 1, 0))
        return np.ascontiguousarray(g[sl]).reshape(-1)

    def owned(arr, lay, n_cells):
        a = ops.to_host(arr).reshape(lay.shape_zyx)
        sl = tuple(slice(lay.ref(d), lay.ref(d) + n_cells[d] + 1) for d in (2, 1, 0))
        return a[sl].copy()

    u0, f0, uc0 = local("u.npy", nc, 1), local("f.npy", nc, 0), local("uc.npy", ncc, 1)
    A = laplace_fd(3, dom.h(L), "mp")
    w = 0.8 / A.diag

    # ---- 1. communicate: scramble every layer an exchange writes, exchange, compare with the consistent global field ----------
    for name, kw in (("axis_by_axis", {}), ("one_batch", {"concurrent_ghost_axes": True})):
        comm = Communicator(dom, ops, **kw)
        result["transport"] = comm.transport
        S = Field("Solution", L, lay_u, ops, 1, None)
        a = u0.copy().reshape(lay_u.shape_zyx)
        for d in range(3):
            ax = 2 - d
            if dom.neighbor(d, -1) is not None:      # lower ghost + lower duplicate plane come from the neighbour
                idx = [slice(None)] * 3
                idx[ax] = slice(0, 2)
                a[tuple(idx)] = -7.0
            if dom.neighbor(d, +1) is not None:
                idx = [slice(None)] * 3
                idx[ax] = slice(a.shape[ax] - 1, a.shape[ax])
                a[tuple(idx)] = -9.0
        S.slots[0].copy_(ops.from_host(a.reshape(-1)))
        comm.exchange(S, None, "all", axis_only=bool(kw))
        comm.check()
        got = ops.to_host(S.data()).reshape(lay_u.shape_zyx)
        want = u0.reshape(lay_u.shape_zyx)
        if kw:      # one batch: face ghosts only -- compare the ghost planes without their edges
            ok = True
            inner = slice(1, -1)
            for d in range(3):
                ax = 2 - d
                for side, pos in ((-1, 0), (+1, got.shape[ax] - 1)):
                    if dom.neighbor(d, side) is None:
                        continue
                    idx = [inner] * 3
                    idx[ax] = pos
                    ok = ok and np.array_equal(got[tuple(idx)], want[tuple(idx)])
                if dom.neighbor(d, -1) is not None:
                    idx = [inner] * 3
                    idx[ax] = 1
                    ok = ok and np.array_equal(got[tuple(idx)], want[tuple(idx)])
        else:
            # everything that has a source: all but the ghost planes on physical faces (scrambled entries there stay as they are)
            idx = [slice(None)] * 3
            for d in range(3):
                ax = 2 - d
                lo = 1 if dom.neighbor(d, -1) is None else 0
                hi = got.shape[ax] - 1 if dom.neighbor(d, +1) is None else got.shape[ax]
                idx[ax] = slice(lo, hi)
            ok = np.array_equal(got[tuple(idx)], want[tuple(idx)])
        result["checks"]["exchange_" + name] = bool(ok)

    comm = Communicator(dom, ops, concurrent_ghost_axes=True, consistent_duplicates=True)

    # ---- 2. k pairs of Jacobi steps, interior kernel overlapped with the shell + exchanges (examg_jacobi2_blocks) ---------------
    S = Field("Solution", L, lay_u, ops, 2, None)
    F = Field("RHS", L, lay_f, ops, 1, None)
    T = Field("Tmp", L, lay_u, ops, 1, None)
    for t in S.slots + T.slots:
        t.copy_(ops.from_host(u0))
    F.slots[0].copy_(ops.from_host(f0))
    for _ in range(spec["pairs"]):
        jacobi_pair(ops, comm, dom, S, F, A, w, T)
    comm.check()
    arrays["jacobi"] = owned(S.data(), lay_u, nc)

    # ---- 3. k red-black sweeps (examg_rbgs_sweep_blocks) ---------------------------------------------------------------------------
    S = Field("Solution", L, lay_u, ops, 1, None)
    S.slots[0].copy_(ops.from_host(u0))
    alt = ops.from_host(u0)
    T.slots[0].copy_(ops.from_host(u0))
    for _ in range(spec["sweeps"]):
        alt = rbgs_sweep(ops, comm, dom, S, F, A, w, alt, T, 0)
    comm.check()
    arrays["rbgs"] = owned(S.data(), lay_u, nc)

    # ---- 4. transfer operators as one call each ------------------------------------------------------------------------------------
    R = Field("Residual", L, lay_u, ops, 1, None)
    Fc = Field("RHSc", L - 1, lay_cf, ops, 1, None)
    fb, fe = dom.loop_bounds(lay_u)
    cb, ce = dom.loop_bounds(lay_cf)
    made = comm.c_residual_restrict(S, F, R, A, Fc, 1.0, fb, fe, cb, ce, axis_only=True, overlap=True)
    comm.check()
    result["checks"]["residual_restrict_blocks_called"] = bool(made)
    arrays["coarse_rhs"] = owned(Fc.data(), lay_cf, ncc)
    Sc = Field("SolutionC", L - 1, lay_c, ops, 1, None)
    Sc.slots[0].copy_(ops.from_host(uc0))
    made = comm.c_prolong_add(Sc, S, fb, fe, overlap=True)
    comm.check()
    result["checks"]["prolong_add_blocks_called"] = bool(made)
    arrays["prolong"] = owned(S.data(), lay_u, nc)

    # ---- 5. all-reduce / all-gather --------------------------------------------------------------------------------------------------
    t = ops.from_host(np.array([1.0 + rank, -2.0 * rank, 0.5], dtype=np.float64))
    comm.allreduce(t[:1], "sum")
    comm.allreduce(t[1:2], "max")
    piece = ops.from_host(np.arange(1000, dtype=np.float64) + 1000.0 * rank)
    allp = ops.new_array(1000 * world)
    comm.all_gather([allp[r * 1000:(r + 1) * 1000] for r in range(world)], piece)
    comm.check()
    th = ops.to_host(t)
    want = np.concatenate([np.arange(1000, dtype=np.float64) + 1000.0 * r for r in range(world)])
    result["checks"]["allreduce"] = bool(th[0] == sum(1.0 + r for r in range(world)) and th[1] == 0.0)
    result["checks"]["allgather"] = bool(np.array_equal(ops.to_host(allp), want))

    # ---- 6. the V-cycle leg of bench.py at N > 1: eager, then replayed from a hipGraph --------------------------------------------
    from exastencils_amd.solver import ConfigL4, SolverFromL4

    cy = spec["cycle"]
    cfg = ConfigL4(nd=3, min_level=cy["min_level"], max_level=L, frag_len=flen, tol=cy["tol"], fused_rbgs=True, fused_residual_restrict=True,
                   agglomerate_level=cy["agglomerate_level"], fused_residual_norm=True, agglomerate_extra_levels=cy.get("extra", 0))
    P = SolverFromL4(cfg, ops, dom, comm)
    dist.barrier()        # (hosted ranks: the constructor records the agglomerated levels' graph; no rank of the process goes on before all have)
    P.setup()
    its = P.Solve()
    comm.check()
    result["eager"] = {"it": its, "res": P.res_history}
    arrays["cycle_eager"] = owned(P.Solution[L].data(), P.Solution[L].layout, nc)
    P.reset()
    P.capture_cycle()
    dist.barrier()
    P.reset()
    its = P.Solve(use_graph=True)
    comm.check()
    result["graph"] = {"it": its, "res": P.res_history}
    arrays["cycle_graph"] = owned(P.Solution[L].data(), P.Solution[L].layout, nc)
    result["dup_consistent"] = bool(comm.check_duplicates(P.Solution[L]))
    # ... and with deep halos (ConfigL4.deep_halo: sweeps and residual + restriction without shells): the same history, the same bits
    import dataclasses

    # (one rank per process only: with several ranks hosted by one process the regions grow here, in the middle of a Solve, while the
    #  other thread of the process is still launching -- the rehearsal harness's coupling, see tests/ranks_host.py)
    if dist_module is None:
        D = SolverFromL4(dataclasses.replace(cfg, deep_halo=True), ops, dom, comm)
        dist.barrier()
        D.setup()
        its_d = D.Solve()
        comm.check()
        result["deep"] = {"it": its_d, "res": D.res_history}
        result["checks"]["cycle_deep_halo_equals_shell"] = bool(
            its_d == result["eager"]["it"] and D.res_history == result["eager"]["res"] and
            np.array_equal(owned(D.Solution[L].data(), D.Solution[L].layout, nc), arrays["cycle_eager"]))

    # ---- 7. the same passes by DEEPER HALOS (two ghost layers of Solution, one of RHS: one exchange and one kernel per pass, no shell;
    #          exastencils_amd/smoothers.py: deep_halo_boxes): the same bits ---------------------------------------------------------
    if dist_module is None:      # (hosted ranks: see above)
        lay_u2 = FieldLayout.node(3, nc, 2, True, True, 0)
        lay_f1 = FieldLayout.node(3, nc, 1, True, True, 0)
        full = Communicator(dom, ops)

        def widened(flat, lay_from, lay_to, k):
            a = np.zeros(lay_to.shape_zyx)
            a[k:-k, k:-k, k:-k] = flat.reshape(lay_from.shape_zyx)
            return a.reshape(-1)

        S2 = Field("Solution", L, lay_u2, ops, 2, None)
        F1 = Field("RHS", L, lay_f1, ops, 1, None)
        T2 = Field("Tmp", L, lay_u2, ops, 1, None)
        for t in S2.slots + T2.slots:
            t.copy_(ops.from_host(widened(u0, lay_u, lay_u2, 1)))
        F1.slots[0].copy_(ops.from_host(widened(f0, lay_f, lay_f1, 1)))
        for s_ in (0, 1):
            full.exchange(S2, s_, "all")
        full.exchange(F1, None, "all")
        for _ in range(spec["pairs"]):
            jacobi_pair(ops, comm, dom, S2, F1, A, w, T2)
        comm.check()
        result["checks"]["jacobi_deep_halo_equals_shell"] = bool(np.array_equal(owned(S2.data(), lay_u2, nc), arrays["jacobi"]))
        S2 = Field("Solution", L, lay_u2, ops, 1, None)
        S2.slots[0].copy_(ops.from_host(widened(u0, lay_u, lay_u2, 1)))
        full.exchange(S2, None, "all")
        alt2 = S2.data().clone()
        for _ in range(spec["sweeps"]):
            alt2 = rbgs_sweep(ops, comm, dom, S2, F1, A, w, alt2, T2, 0)
        comm.check()
        result["checks"]["rbgs_deep_halo_equals_shell"] = bool(np.array_equal(owned(S2.data(), lay_u2, nc), arrays["rbgs"]))

    np.savez(os.path.join(out, "out_%d.npz" % rank), **arrays)
    json.dump(result, open(os.path.join(out, "out_%d.json" % rank), "w"))


if __name__ == "__main__":
    main()
