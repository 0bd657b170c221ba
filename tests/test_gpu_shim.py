"""The reference-named boundary (shim/): a host program in the shape of generated code that calls ONLY
<fn>_<L>_k<NNN>_wrapper(), exch<Field>_<L>(slot), applyBCs<Field>_<L>(slot) over process-global fieldDeviceData_* arrays
(Compiler/src/exastencils/parallelization/api/cuda/CUDA_Kernel.scala:546-632; communication/ir/IR_CommunicateFunction.scala:
473-480), with libexamg behind those names -- HIP runtime + libexamg only, no Python in that process.  Its printed residual
history must be the oracle's (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4 restated in oracle/mg.py)."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

from oracle import mg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _history(stdout):
    return [float(l[2:]) for l in stdout.splitlines() if l.startswith("# ")]


def _close(a, b, rtol=1e-10):
    assert len(a) == len(b), (a, b)
    floor = 64 * 2.220446049250313e-16 * abs(b[0])
    for x, y in zip(a, b):
        assert abs(x - y) <= rtol * abs(y) + floor, (a, b)


def test_shim_host_matches_oracle():
    import __graft_entry__ as ge

    exe = ge.build_shim(2, 6)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    O = mg.ProgramA(mg.ConfigA(nd=3, min_level=2, max_level=6, tol=1e-6))
    O.setup()
    O.Solve()
    _close(_history(out.stdout), O.res_history)
    printed = [l for l in out.stdout.splitlines() if l and not l.startswith("#") and not l.startswith("iterations")]
    assert mg.compare_with_golden(printed, "\n".join(O.log)) == []


def test_shim_host_512_matches_full_size_fixture():
    """The benchmark's own knowledge (levels 4..9, 512^3) through the reference names, against the own-oracle fixture."""
    import __graft_entry__ as ge

    exe = ge.build_shim(4, 9)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "own", "config3_512.json")))
    _close(_history(out.stdout), [float.fromhex(h) for h in rec["res_history"]])
    assert "iterations %d" % rec["iterations"] in out.stdout


@pytest.mark.parametrize("levels", [(2, 6), (4, 9)])
def test_shim_deferred_launch_mode_prints_the_same_history(levels):
    """EXA_DEFERRED_LAUNCH=1: the same host binary, the same reference-named wrappers -- k000 records, k001 launches ONE out-of-place
    red-black sweep; k002 + k003 one residual + restriction pass; k004 / k005 ride on the first sweep after them
    (shim/exa_poisson3d_kernels.cpp).  Every printed value, in full precision, equals the plain wrappers' bit for bit; fewer launches."""
    import __graft_entry__ as ge

    exe = ge.build_shim(*levels)
    runs = {}
    for mode in ("0", "1"):
        env = dict(os.environ, EXA_DEFERRED_LAUNCH=mode, EXA_TIME_CYCLES="3")
        out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stderr
        timing = [l for l in out.stdout.splitlines() if l.startswith("vcycle_ms")][0].split()
        runs[mode] = (_history(out.stdout), [l for l in out.stdout.splitlines() if not l.startswith("vcycle_ms")], float(timing[1]), int(timing[3]))
    assert runs["1"][0] == runs["0"][0], "full-precision histories differ between the deferred and the plain wrappers"
    assert runs["1"][1] == runs["0"][1]
    assert runs["1"][3] < runs["0"][3], "deferred mode should need fewer launches per cycle: %r" % ({k: v[2:] for k, v in runs.items()},)
    if levels == (4, 9):
        assert runs["1"][2] < runs["0"][2]


def test_shim_two_blocks_over_rccl(tmp_path):
    """Two processes, one per GPU, blocks 1 x 1 x 2: exch<Field>_<L> = examg_exchange over RCCL, reductions through
    examg_allreduce; the root's history must be the oracle's for the same decomposition.  Needs two GPUs."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (one process per GPU; RCCL refuses two ranks on one device)")
    import __graft_entry__ as ge

    exe = ge.build_shim(2, 6)
    idfile = str(tmp_path / "rccl.id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, "1", "1", "2", str(r), idfile], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se
    O = mg.ProgramA(mg.ConfigA(nd=3, min_level=2, max_level=6, tol=1e-6, nfrag=(1, 1, 2)))
    O.setup()
    O.Solve()
    _close(_history(outs[0][0]), O.res_history)


@pytest.mark.parametrize("blocks", [(1, 1, 2), (1, 2, 2)])
def test_shim_blocks_over_the_peer_write_transport_on_one_gpu(tmp_path, blocks):
    """The generated-style C++ host as 2 / 4 processes that SHARE the one GPU: exch<Field>_<L> = examg_exchange on a communicator made
    by examg_comm_create_peer (HIP IPC; handles through files where a generated program would call MPI_Allgather), reductions through
    examg_allreduce on the same regions.  The root's history must be the oracle's for the same decomposition (1e-10)."""
    import __graft_entry__ as ge

    exe = ge.build_shim(2, 6)
    n = blocks[0] * blocks[1] * blocks[2]
    base = str(tmp_path / "peer.handle")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe] + [str(b) for b in blocks] + [str(r), base, "peer"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(n)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se
    O = mg.ProgramA(mg.ConfigA(nd=3, min_level=2, max_level=6, tol=1e-6, nfrag=blocks))
    O.setup()
    O.Solve()
    _close(_history(outs[0][0]), O.res_history)
