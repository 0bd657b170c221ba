"""Multi-process, device-resident halo exchange on ONE GPU: the peer-write transport (csrc/examg_peer.hip, HIP IPC) with 2 and 4
ranks started as fresh child processes (tests/peer_worker.py).  RCCL refuses several ranks on one device; this transport does
not, and it is the one the N > 1 bench path uses.

Checked per decomposition (1 x 1 x 2, 1 x 2 x 2 and 1 x 1 x 4 blocks of a 128^3-cell grid; in the last the middle blocks have a
neighbour on both sides of an axis):
  * `communicate` (duplicate + ghost layers, axis by axis and as one batch) restores a scrambled halo of a consistent global field;
  * k overlapped Jacobi pairs (examg_jacobi2_blocks), k overlapped red-black sweeps (examg_rbgs_sweep_blocks), residual +
    restriction (examg_residual_restrict_blocks) and prolongation + correction (examg_prolong_add_blocks): BIT-IDENTICAL to the
    single block running the same statements on the whole grid;
  * all-reduce / all-gather;
  * the V-cycle leg of bench.py (fused sweeps, fused residual + restriction, agglomerated coarse levels, duplicate exchange left
    out): eager == replay from a hipGraph (bitwise), iteration counts and residual histories == the single block's and the
    oracle's within 1e-10 (north_star tolerance), duplicate planes bit-identical on both owners afterwards.
Further down: BASELINE configs[3] (27-entry Helmholtz field, Jacobi pairs, transformed coefficient layout) and configs[4] (FMG start +
fused red-black cycles) as decomposed programs of 2 and 4 processes (1 x 1 x 2, 1 x 2 x 2, 2 x 2 x 1) against the single block and the
oracle program, and the lost-neighbour timeout."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu

PAIRS, SWEEPS = 3, 3
# (total fragment length per dimension, finest level, cycle): 128 cells per dimension either way -- 2 * 2^6 for the decompositions
# with at most two blocks per dimension, 4 * 2^5 for 1 x 1 x 4, whose middle blocks have a neighbour on BOTH sides of z
GRIDS = {2: (6, {"min_level": 2, "tol": 1e-8, "agglomerate_level": 3, "extra": 0}),
         4: (5, {"min_level": 1, "tol": 1e-8, "agglomerate_level": 2, "extra": 0})}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_SINGLE = {}


@pytest.fixture()
def single_block(blocks):
    total = max(blocks)
    if total not in _SINGLE:
        _SINGLE[total] = _single_block(total)
    return _SINGLE[total]


def _single_block(total):
    """The whole grid as one block on the GPU: inputs (saved for the workers) and expected outputs."""
    LEVEL, CYCLE = GRIDS[total]
    import torch

    from exastencils_amd.comm import Communicator
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.field import Field, laplace_fd
    from exastencils_amd.layout import FieldLayout
    from exastencils_amd.ops import HipOps
    from exastencils_amd.solver import ConfigL4, SolverFromL4

    ops = HipOps(0)
    dom = RectDomain(3, (1, 1, 1), 0, (total, total, total))
    L = LEVEL
    nc, ncc = dom.ncells(L), dom.ncells(L - 1)
    lay_u = FieldLayout.node(3, nc, 1, True, True, 0)
    lay_f = FieldLayout.node(3, nc, 0, True, False, 0)
    lay_c = FieldLayout.node(3, ncc, 1, True, True, 0)
    lay_cf = FieldLayout.node(3, ncc, 0, True, False, 0)
    rng = np.random.default_rng(20261004)
    u = rng.uniform(-1.0, 1.0, lay_u.size)
    f = rng.uniform(-1.0, 1.0, lay_f.size) * 1000.0
    uc = rng.uniform(-1.0, 1.0, lay_c.size)
    d = tempfile.mkdtemp(prefix="examg_peer_")
    np.save(os.path.join(d, "u.npy"), u.reshape(lay_u.shape_zyx))
    np.save(os.path.join(d, "f.npy"), f.reshape(lay_f.shape_zyx))
    np.save(os.path.join(d, "uc.npy"), uc.reshape(lay_c.shape_zyx))
    json.dump({"level": L, "pairs": PAIRS, "sweeps": SWEEPS, "cycle": CYCLE, "total_frag": total}, open(os.path.join(d, "spec.json"), "w"))

    A = laplace_fd(3, dom.h(L), "mp")
    w = 0.8 / A.diag
    b, e = dom.loop_bounds(lay_u)
    exp = {}

    def owned(arr, lay, n):
        a = ops.to_host(arr).reshape(lay.shape_zyx)
        sl = tuple(slice(lay.ref(k), lay.ref(k) + n[k] + 1) for k in (2, 1, 0))
        return a[sl].copy()

    x, y, t = ops.from_host(u), ops.from_host(u), ops.from_host(u)
    F = ops.from_host(f)
    for _ in range(PAIRS):
        ops.jacobi2_boxes(lay_u.c_struct(), x, y, t, lay_f.c_struct(), F, A, w, b, e, b, e)
        x, y = y, x
    exp["jacobi"] = owned(x, lay_u, nc)
    x, y = ops.from_host(u), ops.from_host(u)
    for _ in range(SWEEPS):
        ops.rbgs_sweep_fused(lay_u.c_struct(), x, y, lay_f.c_struct(), F, A, w, 0, b, e)
        x, y = y, x
    exp["rbgs"] = owned(x, lay_u, nc)
    r = ops.new_array(lay_u.size)
    fc = ops.new_array(lay_cf.size)
    cb, ce = dom.loop_bounds(lay_cf)
    ops.residual_restrict(lay_u.c_struct(), x, lay_f.c_struct(), F, lay_u.c_struct(), r, A, lay_cf.c_struct(), fc, 1.0, b, e, cb, ce)
    exp["coarse_rhs"] = owned(fc, lay_cf, ncc)
    ops.prolong_add(lay_c.c_struct(), ops.from_host(uc), lay_u.c_struct(), x, b, e)
    exp["prolong"] = owned(x, lay_u, nc)

    cfg = ConfigL4(nd=3, min_level=CYCLE["min_level"], max_level=L, frag_len=(total, total, total), tol=CYCLE["tol"], fused_rbgs=True,
                   fused_residual_restrict=True, fused_residual_norm=True)
    P = SolverFromL4(cfg, ops, dom, Communicator(dom, ops))
    P.setup()
    its = P.Solve()
    exp["cycle"] = owned(P.Solution[L].data(), P.Solution[L].layout, nc)
    torch.cuda.synchronize()

    from oracle import mg

    O = mg.ProgramA(mg.ConfigA(nd=3, min_level=CYCLE["min_level"], max_level=L, frag_len=(total, total, total), tol=CYCLE["tol"]))
    O.setup()
    O.Solve()
    return {"dir": d, "exp": exp, "it": its, "res": list(P.res_history), "oracle_it": O.iterations, "oracle_res": list(O.res_history), "nc": nc, "ncc": ncc}


def _run_ranks(blocks, d, per=1):
    """Start the ranks of a decomposition as fresh child processes (never a re-exec of a process that has touched the GPU).  per = 1: one
    process per rank, torch.distributed / gloo carries the bootstrap (tests/peer_worker.py).  per > 1: `per` ranks per process, one
    thread and stream each, bootstrap through files (tests/ranks_host.py, tests/filedist.py) -- the box allows fewer than 8 processes
    on its card, and 2 x 2 x 2 / 1 x 2 x 4 are what an 8-GPU node runs."""
    world = blocks[0] * blocks[1] * blocks[2]
    port = _free_port()
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.pop("EXAMG_TRANSPORT", None)
    env["EXAMG_PEER_TIMEOUT_MS"] = "30000" if per == 1 else "60000"
    if per > 1:
        # several ranks per process: every rank's streams need hardware queues of their own (HIP shares GPU_MAX_HW_QUEUES = 4 queues
        # among the streams of a process by default; a waiting receive kernel must never sit in front of the send it waits for)
        env["GPU_MAX_HW_QUEUES"] = "8"
    for r in range(world):
        for ext in ("npz", "json"):
            try:
                os.remove(os.path.join(d, "out_%d.%s" % (r, ext)))
            except OSError:
                pass
    procs = []
    if per == 1:
        for r in range(world):
            log = open(os.path.join(d, "log_%d_%d.txt" % (world, r)), "w")
            procs.append((subprocess.Popen([sys.executable, os.path.join(HERE, "peer_worker.py"), str(r), str(world), ",".join(map(str, blocks)), str(port), d],
                                           stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT), log))
    else:
        assert world % per == 0
        boot = tempfile.mkdtemp(prefix="examg_boot_")
        for q in range(world // per):
            log = open(os.path.join(d, "log_%d_%d.txt" % (world, q)), "w")
            procs.append((subprocess.Popen([sys.executable, os.path.join(HERE, "ranks_host.py"), "worker", str(q), str(world // per), str(per),
                                            ",".join(map(str, blocks)), boot, d], stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT), log))
    rcs = []
    for p, log in procs:
        try:
            rcs.append(p.wait(timeout=420))
        except subprocess.TimeoutExpired:
            for q, _ in procs:
                q.kill()
            rcs.append(-9)
        log.close()
    if any(rcs):
        tails = []
        for r in range(len(procs)):
            with open(os.path.join(d, "log_%d_%d.txt" % (world, r))) as fh:
                tails.append("---- process %d (rc %s)\n%s" % (r, rcs[r], fh.read()[-3000:]))
        pytest.fail("peer workers failed:\n" + "\n".join(tails))
    return world


@pytest.mark.parametrize("blocks", [(1, 1, 2), (1, 2, 2), (1, 1, 4)])
def test_peer_transport_multi_process_on_one_gpu(single_block, blocks):
    _check_decomposition(single_block, blocks, 1)


@pytest.mark.skipif(os.environ.get("EXAMG_HOSTED_RANKS") != "1",
                    reason="several ranks per process (threads): opt-in, EXAMG_HOSTED_RANKS=1 (tools/gpu_rehearse_multi.sh)")
@pytest.mark.parametrize("blocks", [(2, 2, 2), (1, 2, 4)])
def test_eight_ranks_as_on_a_full_node(single_block, blocks):
    """The decompositions of an 8-GPU node -- 2 x 2 x 2 (domain/ir/IR_ConnectFragments.scala:46-52: every block has a
    neighbour across an x, a y and a z face; corner ghosts travel through all three axes) and bench.py's default 1 x 2 x 4 (middle
    blocks with neighbours on both sides of z) -- as 8 ranks on the one GPU: 4 processes of 2 ranks each (the box's process guard
    allows no 8 processes on its card), same checks as the 2- and 4-rank decompositions."""
    _check_decomposition(single_block, blocks, 2)


def _check_decomposition(sb, blocks, per):
    d = sb["dir"]
    world = _run_ranks(blocks, d, per)
    nc = tuple(sb["nc"][k] // blocks[k] for k in range(3))
    ncc = tuple(sb["ncc"][k] // blocks[k] for k in range(3))
    outs = [json.load(open(os.path.join(d, "out_%d.json" % r))) for r in range(world)]
    arrs = [np.load(os.path.join(d, "out_%d.npz" % r)) for r in range(world)]
    for r, o in enumerate(outs):
        assert o["transport"] == "peer"
        for k, v in o["checks"].items():
            assert v, "rank %d: check %s failed" % (r, k)
        assert o["dup_consistent"], "rank %d: duplicate planes differ between their two owners after the cycle" % r

    def piece(glob, r, n):
        pos = (r % blocks[0], (r // blocks[0]) % blocks[1], r // (blocks[0] * blocks[1]))
        sl = tuple(slice(pos[k] * n[k], pos[k] * n[k] + n[k] + 1) for k in (2, 1, 0))
        return glob[sl]

    for name, n in (("jacobi", nc), ("rbgs", nc), ("coarse_rhs", ncc), ("prolong", nc)):
        for r in range(world):
            got, want = arrs[r][name], piece(sb["exp"][name], r, n)
            assert got.shape == want.shape
            assert np.array_equal(got, want), "%s: rank %d differs from the single block (max abs %.3e)" % (name, r, np.abs(got - want).max())

    # V-cycle leg: every rank reports the same history; eager == graph replay bit for bit; history == single block == oracle (1e-10)
    for r, o in enumerate(outs):
        assert o["eager"]["res"] == outs[0]["eager"]["res"]
        assert o["graph"]["res"] == o["eager"]["res"], "rank %d: graph replay differs from the eager cycle" % r
        assert o["graph"]["it"] == o["eager"]["it"]
        assert np.array_equal(arrs[r]["cycle_eager"], arrs[r]["cycle_graph"])
    h = outs[0]["eager"]
    assert h["it"] == sb["it"] == sb["oracle_it"]
    for x, y, z in zip(h["res"], sb["res"], sb["oracle_res"]):
        assert abs(x - y) <= 1e-10 * abs(y) + 64 * 2.2e-16 * sb["res"][0]
        assert abs(x - z) <= 1e-10 * abs(z) + 64 * 2.2e-16 * sb["res"][0]
    for r in range(world):
        got, want = arrs[r]["cycle_eager"], piece(sb["exp"]["cycle"], r, nc)
        assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()


# BASELINE configs[3] / configs[4] in their decomposed form (the 8-GPU node is not ours to use: 2 and 4 processes on the one device).
# configs[3]: 27-entry variable-coefficient Helmholtz field under the applied layout transformation, Jacobi pairs (edge / corner
# ghosts of the 27-point stencil arrive through the axis-by-axis exchange); configs[4]: FMG start + fused red-black cycles.
SOLVER_CASES = {
    "helmholtz27": dict(nd=3, min_level=1, max_level=5, smoother="jacobi", omega=0.8, stencil="helmholtz27", restrict_scale=1.0, tol=1e-8,
                        cg_max=512, bc_fn=0, sol_fn=9, coef_fn=7, kappa=10.0, ksq=2.0, rhs_from_solution=True, temporal_blocking=True,
                        coef_entry_fastest=True),
    "fmg_rbgs": dict(nd=3, min_level=1, max_level=5, smoother="rbgs", omega=1.0, stencil="scaled", restrict_scale=1.0, tol=1e-8, cg_max=512,
                     bc_fn=1, fmg=True, fused_rbgs=True),
}
_SOLVER_SINGLE = {}


def _solver_single(case):
    """The same program as ONE block: on the GPU (expected solution) and through the oracle's program (expected history)."""
    if case in _SOLVER_SINGLE:
        return _SOLVER_SINGLE[case]
    from oracle import mg

    from exastencils_amd.ops import HipOps
    from exastencils_amd.solver import ConfigL3, SolverFromL3

    kw = SOLVER_CASES[case]
    ops = HipOps(0)
    P = SolverFromL3(ConfigL3(frag_len=(2, 2, 2), **kw), ops)
    P.setup()
    its = P.Solve()
    S = P.Solution[kw["max_level"]]
    lay, n = S.layout, 2 << kw["max_level"]
    a = ops.to_host(S.data()).reshape(lay.shape_zyx)
    own = a[tuple(slice(lay.ref(d), lay.ref(d) + n + 1) for d in (2, 1, 0))].copy()
    okw = {k: v for k, v in kw.items() if k not in ("temporal_blocking", "coef_entry_fastest", "fused_rbgs")}
    O = mg.ProgramB(mg.ConfigB(nfrag=(1, 1, 1), frag_len=(2, 2, 2), **okw))
    O.setup()
    O.Solve()
    _SOLVER_SINGLE[case] = {"it": its, "res": list(P.res_history), "sol": own, "oracle_it": O.iterations, "oracle_res": list(O.res_history)}
    return _SOLVER_SINGLE[case]


@pytest.mark.parametrize("blocks,case", [((1, 1, 2), "helmholtz27"), ((1, 2, 2), "helmholtz27"), ((1, 1, 2), "fmg_rbgs"), ((2, 2, 1), "fmg_rbgs")])
def test_config3_and_config4_programs_decomposed_over_the_peer_transport(tmp_path, blocks, case):
    sb = _solver_single(case)
    d = str(tmp_path)
    json.dump(SOLVER_CASES, open(os.path.join(d, "cases.json"), "w"))
    world = blocks[0] * blocks[1] * blocks[2]
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EXAMG_PEER_TIMEOUT_MS="30000")
    env.pop("EXAMG_TRANSPORT", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "peer_solver_worker.py"), str(r), str(world), ",".join(map(str, blocks)), str(port), d, case],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=ROOT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=420)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("peer solver workers timed out")
    for r, (p, log) in enumerate(zip(procs, logs)):
        assert p.returncode == 0, "rank %d:\n%s" % (r, log[-3000:])
    n = 2 << SOLVER_CASES[case]["max_level"]
    nc = tuple(n // blocks[k] for k in range(3))
    outs = [json.load(open(os.path.join(d, "%s_%d.json" % (case, r)))) for r in range(world)]
    for r, o in enumerate(outs):
        assert o["transport"] == "peer" and o["exchanges"] > 0
        assert o["dup_consistent"], "rank %d: duplicate planes differ between their owners" % r
        assert o["res"] == outs[0]["res"] and o["it"] == outs[0]["it"]
    h = outs[0]
    assert h["it"] == sb["it"] == sb["oracle_it"]
    r0 = sb["res"][0]
    for x, y, z in zip(h["res"], sb["res"], sb["oracle_res"]):
        assert abs(x - y) <= 1e-10 * abs(y) + 64 * 2.2e-16 * r0, (h["res"], sb["res"])
        assert abs(x - z) <= 1e-10 * abs(z) + 64 * 2.2e-16 * r0, (h["res"], sb["oracle_res"])
    for r in range(world):
        pos = (r % blocks[0], (r // blocks[0]) % blocks[1], r // (blocks[0] * blocks[1]))
        want = sb["sol"][tuple(slice(pos[k] * nc[k], pos[k] * nc[k] + nc[k] + 1) for k in (2, 1, 0))]
        got = np.load(os.path.join(d, "%s_%d.npy" % (case, r)))
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-9 * max(np.abs(want).max(), 1e-300), "rank %d: max abs %.3e" % (r, np.abs(got - want).max())


def test_a_lost_neighbour_ends_in_an_error_not_in_a_spinning_kernel(tmp_path):
    """Rank 0 exchanges once more than rank 1: its receive kernel gives up after EXAMG_PEER_TIMEOUT_MS (0.5 s here), examg_comm_status
    names the wait, and later calls on that communicator return at once instead of waiting again."""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EXAMG_PEER_TIMEOUT_MS="500")
    env.pop("EXAMG_TRANSPORT", None)
    outs = [str(tmp_path / ("r%d.json" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "peer_timeout_worker.py"), str(r), str(port), outs[r]],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=ROOT) for r in range(2)]
    logs = [p.communicate(timeout=300)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    r0 = json.load(open(outs[0]))
    assert r0["transport"] == "peer"
    assert r0["error"] and "receive waited for a message" in r0["error"] and "EXAMG_PEER_TIMEOUT_MS" in r0["error"]
    assert 0.4 < r0["seconds"] < 20.0
    assert r0["second_attempt_seconds"] < 0.4
