"""N > 1 path on the CPU: world_size-2 (and 4) `gloo` process groups run the product's decomposition, halo
exchange (exastencils_amd.comm) and cycle drivers (exastencils_amd.solver) with the oracle's loops standing
in for the HIP kernels, and must reproduce the single-fragment run of the same global problem.

What this pins: rank -> block position, neighbour ranks, iteration offsets at interior faces, duplicate-layer
exchange direction, ghost exchange ranges incl. the GLB..GRE tangential extent that makes edge/corner ghosts
valid with 6 neighbours (needed by 27-point stencils), and the all-reduce after reduction loops."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


# Testing/FMG/3D_VarCoeff.exa4 at a reduced size: stencil field, FMG start, error print
_FMG_VAR = dict(nd=3, min_level=1, max_level=4, smoother="jacobi", omega=0.85, stencil="varcoeff", restrict_scale=1.0, tol=1e-5,
                cg_max=1024, bc_fn=6, rhs_fn=5, sol_fn=6, coef_fn=7, kappa=10.0, fmg=True)
# BASELINE configs[4]'s algorithm: FMG start (Testing/FMG/3D_Trigonometric.exa4:189-242) + red-black cycles
_FMG_RBGS = dict(nd=3, min_level=1, max_level=4, smoother="rbgs", omega=1.0, stencil="scaled", restrict_scale=1.0, tol=1e-8, cg_max=512,
                 bc_fn=1, fmg=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import mg

    mg.lib().orc_set_num_threads(2)


def _global_index(dom, lay, level):
    """Slices of this block's owned box [DLB, DRE) in a global node array of the same level (z, y, x)."""
    nc = dom.ncells(level)
    sl_g, sl_l = [], []
    for d in (2, 1, 0):
        if d >= dom.nd:
            sl_g.append(slice(0, 1))
            sl_l.append(slice(0, 1))
            continue
        o = dom.pos[d] * nc[d]
        sl_g.append(slice(o, o + nc[d] + 1))
        sl_l.append(slice(lay.ref(d), lay.ref(d) + nc[d] + 1))
    return tuple(sl_g), tuple(sl_l)


# -------------------------------------------------------------------------------------------------
def _worker_solver(rank, world, port, blocks, case, out_dir):
    _init(rank, world, port)
    from oracle_ops import OracleOps

    from exastencils_amd.comm import Communicator
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.solver import ConfigL3, ConfigL4, SolverFromL3, SolverFromL4

    ops = OracleOps()
    flen = tuple(2 // blocks[d] if d < 3 else 1 for d in range(3))
    dom = RectDomain(3, blocks, rank, flen)
    comm = Communicator(dom, ops, concurrent_ghost_axes=case.endswith("_cg"), consistent_duplicates="_nodup" in case)
    if case in ("jacobi_l3", "jacobi_l3_tb", "jacobi_l3_tb_cg"):
        P = SolverFromL3(ConfigL3(nd=3, min_level=1, max_level=4, frag_len=flen, smoother="jacobi", omega=0.8, stencil="unit",
                                  restrict_scale=4.0, tol=1e-5, cg_max=512, temporal_blocking="_tb" in case), ops, dom, comm)
    elif case == "rbgs_l3":
        P = SolverFromL3(ConfigL3(nd=3, min_level=1, max_level=4, frag_len=flen, smoother="rbgs", omega=1.0, stencil="unit",
                                  restrict_scale=4.0, tol=1e-5, cg_max=512), ops, dom, comm)
    elif case == "rbgs_l3_fused":
        P = SolverFromL3(ConfigL3(nd=3, min_level=1, max_level=4, frag_len=flen, smoother="rbgs", omega=1.0, stencil="unit",
                                  restrict_scale=4.0, tol=1e-5, cg_max=512, fused_rbgs=True), ops, dom, comm)
    elif case == "fmg_varcoeff":
        P = SolverFromL3(ConfigL3(**_FMG_VAR, frag_len=flen), ops, dom, comm)
    elif case == "fmg_rbgs_fused":
        # the second array of the out-of-place sweeps follows the boundary planes SetFuncDir / ResetBC rewrite, physical faces only
        P = SolverFromL3(ConfigL3(**_FMG_RBGS, frag_len=flen, fused_rbgs=True), ops, dom, comm)
    elif case == "helmholtz27":
        from test_host_logic import HELMHOLTZ27

        P = SolverFromL3(ConfigL3(**HELMHOLTZ27, frag_len=flen), ops, dom, comm)
    else:
        P = SolverFromL4(ConfigL4(nd=3, min_level=1, max_level=4, frag_len=flen, tol=1e-6, fused_coarse=False,
                                  fused_rbgs="_fused" in case, agglomerate_level=2 if "_agg" in case else None,
                                  fused_residual_norm="_rnorm" in case,      # Solve's residual + norm as one pass, across blocks
                                  agglomerate_extra_levels=1 if "_aggx" in case else 0), ops, dom, comm)
    P.setup()
    P.Solve()
    S = P.Solution[4]
    arr = S.data().numpy().reshape(S.layout.shape_zyx)
    sg, sl = _global_index(dom, S.layout, 4)
    np.save(os.path.join(out_dir, "sol_%d.npy" % rank), arr[sl])
    json.dump({"res": P.res_history, "it": P.iterations, "slices": [[s.start, s.stop] for s in sg], "log": P.log,
               "messages": comm.stats["messages"]}, open(os.path.join(out_dir, "res_%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


def _reference_single(case):
    from oracle_ops import OracleOps

    from exastencils_amd.solver import ConfigL3, ConfigL4, SolverFromL3, SolverFromL4

    ops = OracleOps()
    flen = (2, 2, 2)
    if case in ("jacobi_l3", "jacobi_l3_tb", "jacobi_l3_tb_cg"):
        P = SolverFromL3(ConfigL3(nd=3, min_level=1, max_level=4, frag_len=flen, smoother="jacobi", omega=0.8, stencil="unit",
                                  restrict_scale=4.0, tol=1e-5, cg_max=512), ops)
    elif case in ("rbgs_l3", "rbgs_l3_fused"):
        P = SolverFromL3(ConfigL3(nd=3, min_level=1, max_level=4, frag_len=flen, smoother="rbgs", omega=1.0, stencil="unit",
                                  restrict_scale=4.0, tol=1e-5, cg_max=512), ops)
    elif case == "fmg_varcoeff":
        P = SolverFromL3(ConfigL3(**_FMG_VAR, frag_len=flen), ops)
    elif case == "fmg_rbgs_fused":
        P = SolverFromL3(ConfigL3(**_FMG_RBGS, frag_len=flen), ops)
    elif case == "helmholtz27":
        from test_host_logic import HELMHOLTZ27

        P = SolverFromL3(ConfigL3(**HELMHOLTZ27, frag_len=flen), ops)
    else:
        # "_aggx": the gathered hierarchy of the decomposed run coarsens one level below min_level = the single block with min_level 0
        P = SolverFromL4(ConfigL4(nd=3, min_level=0 if "_aggx" in case else 1, max_level=4, frag_len=flen, tol=1e-6, fused_coarse=False), ops)
    P.setup()
    P.Solve()
    S = P.Solution[4]
    lay = S.layout
    arr = S.data().numpy().reshape(lay.shape_zyx)
    own = arr[lay.ref(2):lay.ref(2) + 33, lay.ref(1):lay.ref(1) + 33, lay.ref(0):lay.ref(0) + 33]
    return P, own


@pytest.mark.parametrize("blocks,case", [((2, 1, 1), "jacobi_l3"), ((2, 1, 1), "rbgs_l4"), ((1, 2, 1), "rbgs_l3"), ((1, 2, 2), "rbgs_l3_fused"),
                                         ((1, 1, 2), "fmg_rbgs_fused"), ((1, 2, 2), "rbgs_l4_fused_rnorm_nodup_cg"),
                                         ((2, 2, 1), "jacobi_l3"), ((2, 1, 1), "jacobi_l3_tb"), ((2, 2, 1), "jacobi_l3_tb"),
                                         ((2, 2, 1), "jacobi_l3_tb_cg"), ((2, 2, 1), "rbgs_l4_cg"), ((2, 2, 1), "rbgs_l4_nodup_cg"), ((2, 1, 1), "rbgs_l4_fused"),
                                         ((2, 2, 1), "rbgs_l4_fused_nodup_cg"), ((2, 1, 1), "rbgs_l4_agg"), ((1, 1, 2), "rbgs_l4_fused_aggx_nodup_cg"), ((1, 1, 2), "jacobi_l3_tb_cg"),
                                         ((1, 2, 2), "rbgs_l4_fused_nodup_cg"),
                                         ((2, 2, 1), "rbgs_l4_fused_agg_nodup_cg"), ((2, 1, 1), "fmg_varcoeff"), ((2, 2, 1), "helmholtz27")])
def test_decomposed_solve_matches_single_block(tmp_path, blocks, case):
    world = blocks[0] * blocks[1] * blocks[2]
    port = _free_port()
    mp.spawn(_worker_solver, args=(world, port, blocks, case, str(tmp_path)), nprocs=world, join=True)
    P, own = _reference_single(case)
    for r in range(world):
        meta = json.load(open(tmp_path / ("res_%d.json" % r)))
        assert meta["it"] == P.iterations
        assert meta["messages"] > 0
        for x, y in zip(meta["res"], P.res_history):
            assert abs(x - y) <= 1e-10 * abs(y) + 1e-13 * P.res_history[0], (meta["res"], P.res_history)
        sol = np.load(tmp_path / ("sol_%d.npy" % r))
        (z0, z1), (y0, y1), (x0, x1) = meta["slices"]
        ref = own[z0:z1, y0:y1, x0:x1]
        assert sol.shape == ref.shape
        assert np.allclose(sol, ref, rtol=1e-9, atol=1e-12), np.abs(sol - ref).max()


# -------------------------------------------------------------------------------------------------
def _worker_27pt(rank, world, port, out_dir):
    """One 27-point stencil application after `communicate`: needs valid edge ghosts."""
    _init(rank, world, port)
    from oracle_ops import OracleOps

    from exastencils_amd.comm import Communicator
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.field import Field, Stencil
    from exastencils_amd.layout import FieldLayout

    ops = OracleOps()
    blocks = (2, 2, 1)
    dom = RectDomain(3, blocks, rank, (1, 1, 2))
    comm = Communicator(dom, ops)
    L = 3
    lay = FieldLayout.node(3, dom.ncells(L), 1)
    U, V = Field("U", L, lay, ops), Field("V", L, lay, ops)
    # global field g(x,y,z) = random per global node; owned nodes filled from it, ghosts left zero
    n = 16
    rng = np.random.RandomState(7)
    G = rng.rand(n + 1, n + 1, n + 1)
    sg, sl = _global_index(dom, lay, L)
    u = U.data().numpy().reshape(lay.shape_zyx)
    u[sl] = G[sg]
    comm.exchange(U)
    offs = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]
    co = [1.0 + 0.1 * i for i in range(27)]
    st = Stencil(offs, co)
    b, e = dom.loop_bounds(lay)
    ops.stencil_op(0, U.lc, U.data(), None, None, V.lc, V.data(), st, 0.0, -1, b, e)
    v = V.data().numpy().reshape(lay.shape_zyx)
    np.save(os.path.join(out_dir, "v_%d.npy" % rank), v[sl])
    json.dump({"slices": [[s.start, s.stop] for s in sg], "b": b, "e": e}, open(os.path.join(out_dir, "m_%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_corner_ghosts_with_six_neighbours(tmp_path):
    port = _free_port()
    mp.spawn(_worker_27pt, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    n = 16
    rng = np.random.RandomState(7)
    G = rng.rand(n + 1, n + 1, n + 1)
    offs = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]
    co = [1.0 + 0.1 * i for i in range(27)]
    ref = np.zeros_like(G)
    acc = None
    for (a, b, c), w in zip(offs, co):
        t = w * G[1 + c:n + c, 1 + b:n + b, 1 + a:n + a]
        acc = t if acc is None else acc + t
    ref[1:n, 1:n, 1:n] = acc
    for r in range(4):
        m = json.load(open(tmp_path / ("m_%d.json" % r)))
        v = np.load(tmp_path / ("v_%d.npy" % r))
        (z0, z1), (y0, y1), (x0, x1) = m["slices"]
        want = ref[z0:z1, y0:y1, x0:x1]
        # compare on the points this block's loop updated (global interior)
        mask = np.zeros_like(want, dtype=bool)
        mask[max(1 - z0, 0):min(n, z1) - z0, max(1 - y0, 0):min(n, y1) - y0, max(1 - x0, 0):min(n, x1) - x0] = True
        assert np.array_equal(v[mask], want[mask])


# -------------------------------------------------------------------------------------------------
def _worker_bench_vcycle(rank, world, port, out_dir):
    """bench.py's V-cycle leg exactly as it runs at N > 1 (fused sweeps with shells, agglomerated coarse levels, Solve
    from the zero state), with the oracle's loops standing in for the kernels and gloo for RCCL."""
    _init(rank, world, port)
    import bench
    from oracle_ops import OracleOps

    from exastencils_amd.comm import Communicator
    from exastencils_amd.domain import RectDomain

    torch.cuda.synchronize = lambda *a, **k: None      # CPU stand-in: nothing is asynchronous here
    ops = OracleOps()
    dom = RectDomain(3, RectDomain.blocks_for(world, 3), rank)
    comm = Communicator(dom, ops, concurrent_ghost_axes=True, consistent_duplicates=True)
    out = bench.vcycle(ops, dom, comm, 6, world)
    json.dump(out, open(os.path.join(out_dir, "v_%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_vcycle_leg_on_two_blocks(tmp_path):
    from oracle import mg

    port = _free_port()
    mp.spawn(_worker_bench_vcycle, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # the same global problem on one block: 1 x 1 x 2 blocks of 64^3 cells = one fragment of 64 x 64 x 128 cells
    O = mg.ProgramA(mg.ConfigA(nd=3, min_level=1, max_level=6, tol=1e-6, nfrag=(1, 1, 1), frag_len=(1, 1, 2)))
    O.setup()
    O.Solve()
    for r in range(2):
        v = json.load(open(tmp_path / ("v_%d.json" % r)))
        assert v["vcycle_agglomerate_level"] == 3 and v["vcycle_fused_rbgs"]
        assert v["solve_iterations"] == O.iterations
        want = O.res_history[-1] / O.res_history[0]
        assert abs(v["solve_residual_reduction"] - want) <= 1e-8 * want


# -------------------------------------------------------------------------------------------------
def _worker_eight(rank, world, port, out_dir, scheme="zy", dist_module=None):
    """The decompositions of a full node -- bench.py's default 1 x 2 x 4 blocks and 2 x 2 x 2 (--blocks cube) --: fused
    red-black sweeps with shells, single-batch ghosts, left-out duplicate exchange, agglomerated coarse levels.  dist_module: the
    bootstrap of a rank that is a thread (tests/filedist.py) instead of a process."""
    if dist_module is None:
        _init(rank, world, port)
    from oracle import mg
    from oracle_ops import OracleOps

    from exastencils_amd.comm import Communicator
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.solver import ConfigL4, SolverFromL4

    mg.lib().orc_set_num_threads(1)
    ops = OracleOps()
    blocks = RectDomain.blocks_for(world, 3, scheme)
    flen = tuple(4 // blocks[d] for d in range(3))
    dom = RectDomain(3, blocks, rank, flen)
    comm = Communicator(dom, ops, concurrent_ghost_axes=True, consistent_duplicates=True, dist_module=dist_module)
    P = SolverFromL4(ConfigL4(nd=3, min_level=0, max_level=3, frag_len=flen, tol=1e-6, fused_coarse=False, fused_rbgs=True,
                              agglomerate_level=1), ops, dom, comm)
    P.setup()
    P.Solve()
    from exastencils_amd.solver import ConfigL3, SolverFromL3

    # the smoother path of bench.py's headline: slotted Jacobi in two-step passes (interior + shell), single-batch ghosts
    Q = SolverFromL3(ConfigL3(nd=3, min_level=0, max_level=3, frag_len=flen, temporal_blocking=True), ops, dom, comm)
    Q.setup()
    Q.Solve()
    json.dump({"res": P.res_history, "it": P.iterations, "blocks": list(blocks), "res_jac": Q.res_history, "it_jac": Q.iterations,
               "dups": bool(comm.check_duplicates(P.Solution[3]))},
              open(os.path.join(out_dir, "e_%d.json" % rank), "w"))
    if dist_module is None:
        dist.barrier()
        dist.destroy_process_group()
    else:
        dist_module.barrier()


@pytest.mark.parametrize("scheme,hosted", [("zy", False), ("cube", False), ("cube", True)], ids=["1x2x4", "2x2x2", "2x2x2_threads"])
def test_eight_blocks_as_on_a_full_node(tmp_path, scheme, hosted):
    """hosted: the 8 ranks as 8 THREADS of this process with the file bootstrap of tests/filedist.py -- the harness with which
    tests/test_gpu_peer.py / test_gpu_multi.py run 8 ranks on a one-GPU box (there over the peer-write transport)."""
    from oracle import mg

    if hosted:
        import threading

        from filedist import FileDist

        errs = []

        def body(r):
            try:
                _worker_eight(r, 8, 0, str(tmp_path), scheme, FileDist(str(tmp_path / "boot"), r, 8, timeout=120.0))
            except BaseException as ex:      # noqa: BLE001
                errs.append((r, repr(ex)))

        ts = [threading.Thread(target=body, args=(r,)) for r in range(8)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
    else:
        port = _free_port()
        mp.spawn(_worker_eight, args=(8, port, str(tmp_path), scheme), nprocs=8, join=True)
    O = mg.ProgramA(mg.ConfigA(nd=3, min_level=0, max_level=3, tol=1e-6, nfrag=(1, 1, 1), frag_len=(4, 4, 4)))
    O.setup()
    O.Solve()
    J = mg.ProgramB(mg.ConfigB(nd=3, min_level=0, max_level=3, nfrag=(1, 1, 1), frag_len=(4, 4, 4)))
    J.setup()
    J.Solve()
    for r in range(8):
        v = json.load(open(tmp_path / ("e_%d.json" % r)))
        assert v["blocks"] == ([1, 2, 4] if scheme == "zy" else [2, 2, 2]) and v["dups"] is True
        assert v["it"] == O.iterations and v["it_jac"] == J.iterations
        for x, y in zip(v["res"], O.res_history):
            assert abs(x - y) <= 1e-10 * abs(y) + 1e-13 * O.res_history[0], (v["res"], O.res_history)
        for x, y in zip(v["res_jac"], J.res_history):
            assert abs(x - y) <= 1e-10 * abs(y) + 1e-13 * J.res_history[0], (v["res_jac"], J.res_history)


# -------------------------------------------------------------------------------------------------
def _worker_dupcheck(rank, world, port, out_dir):
    """Communicator.check_duplicates: True while both owners of every shared plane hold the same bits, False on every rank
    as soon as one value of one shared plane differs."""
    _init(rank, world, port)
    from oracle_ops import OracleOps

    from exastencils_amd.comm import Communicator
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.field import Field
    from exastencils_amd.layout import FieldLayout

    ops = OracleOps()
    dom = RectDomain(3, (1, 2, 2), rank)
    comm = Communicator(dom, ops, consistent_duplicates=True)
    full = Communicator(dom, ops)
    lay = FieldLayout.node(3, dom.ncells(3), 1)
    U = Field("U", 3, lay, ops)
    U.data().copy_(torch.from_numpy(np.random.RandomState(10 + rank).rand(lay.size)))
    before = comm.check_duplicates(U)            # random per-rank data: shared planes differ
    full.exchange(U, None, "dup")                # upstream duplicate exchange makes them agree
    after = comm.check_duplicates(U)
    if rank == world - 1:                        # one value of this block's lower z duplicate plane
        u = U.data().numpy().reshape(lay.shape_zyx)
        u[lay.ref(2), lay.ref(1) + 3, lay.ref(0) + 2] += 1e-9
    broken = comm.check_duplicates(U)
    json.dump({"before": before, "after": after, "broken": broken}, open(os.path.join(out_dir, "d_%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_check_duplicates_assertion_mode(tmp_path):
    port = _free_port()
    mp.spawn(_worker_dupcheck, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    for r in range(4):
        d = json.load(open(tmp_path / ("d_%d.json" % r)))
        assert d == {"before": False, "after": True, "broken": False}, (r, d)
