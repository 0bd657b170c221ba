"""More than one GPU: bench.py as the driver launches it (torch.distributed.run, one rank per GPU, RCCL) on a small block, and
what must hold there -- the libexamg transport carries the halo exchange, duplicate planes of neighbouring blocks are bit
identical (--check-duplicates), the decomposed V-cycle needs the iterations of the single block.  Skipped on a one-GPU box
(RCCL refuses two ranks on one device); the same code paths run on the CPU with `gloo` in tests/test_distributed_gloo.py and
on one GPU through RCCL-to-self in tests/test_gpu_transport.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(n, extra):
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "10", "--warmup", "2", "--level", "7",
           "--no-cpu-baseline", "--sustained-seconds", "0.5"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("extra", [[], ["--scaling", "strong"]], ids=["weak", "strong"])
def test_bench_two_gpus_over_rccl(extra):
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    r = _bench(2, extra)
    assert r["n_gpus"] == 2 and r["value"] > 0
    assert r["duplicate_planes_bit_identical"] is True
    assert r.get("solve_iterations") in (5, 6, 7), r
    assert r["solve_residual_reduction"] < 1e-6
    assert r["scaling"] == ("strong" if "strong" in extra else "weak")


@pytest.mark.parametrize("n,extra", [(2, []), (4, ["--scaling", "strong"])], ids=["n2_weak", "n4_strong"])
def test_bench_ranks_sharing_one_gpu_over_the_peer_write_transport(n, extra):
    """bench.py exactly as the driver launches it at N > 1, but with the ranks sharing ONE device (`--backend gloo` carries the
    bootstrap and the timing collectives; every halo message moves device to device through HIP IPC): overlapped Jacobi pairs,
    the V-cycle with neighbours replayed from a hipGraph, duplicate planes bit-identical on both owners afterwards."""
    r = _bench(n, ["--backend", "gloo"] + extra)
    assert r["n_gpus"] == n and r["value"] > 0
    assert r["transport"] == "peer"
    assert r["duplicate_planes_bit_identical"] is True
    assert r["vcycle_graph"] is True and "vcycle_error" not in r, r.get("vcycle_error")
    assert r["vcycle_duplicate_planes_bit_identical"] is True
    assert r.get("solve_iterations") in (5, 6, 7), r
    assert r["solve_residual_reduction"] < 1e-6


def test_bench_reports_a_dropped_transport_instead_of_falling_back_silently(monkeypatch):
    """EXAMG_PEER_FORCE_FAIL=1 makes examg_comm_create_peer refuse on every rank: bench.py's probe drops the peer-write transport on
    all ranks together, says so in the line (`transport_notes`) and runs on torch.distributed point-to-point (the only other one that
    takes two ranks on one GPU); the V-cycle with neighbours is then issued eagerly."""
    monkeypatch.setenv("EXAMG_PEER_FORCE_FAIL", "1")
    monkeypatch.delenv("EXAMG_TRANSPORT", raising=False)
    r = _bench(2, ["--backend", "gloo"])
    assert r["transport"] == "torch"
    assert r["transport_notes"] and r["transport_notes"][0]["transport"] == "peer" and "FORCE_FAIL" in r["transport_notes"][0]["reason"]
    assert r["vcycle_graph"] is False and r["duplicate_planes_bit_identical"] is True
    assert r.get("solve_iterations") in (5, 6, 7), r


def _bench_hosted(n, per, extra, level=6):
    """bench.py's `run` as n ranks in n / per processes (tests/ranks_host.py: one thread and stream per rank, bootstrap through files):
    the 8-rank job within the box's limit of processes on the card."""
    import tempfile

    boot = tempfile.mkdtemp(prefix="examg_boot_")
    outp = os.path.join(boot, "bench.json")
    # Two ranks per process = twice the streams (launch + side stream per rank, capture streams) of a product process.  HIP maps the
    # streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4); two streams on one queue run in order, and a receive
    # kernel that waits for the neighbour in the same process would block the very send it waits for.  One rank per process (the
    # product's launch) stays within the default.
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EXAMG_PEER_TIMEOUT_MS="60000", GPU_MAX_HW_QUEUES="8")
    env.pop("EXAMG_TRANSPORT", None)
    args = ["--gpus", str(n), "--steps", "10", "--warmup", "2", "--level", str(level), "--no-cpu-baseline", "--sustained-seconds", "0.5",
            "--settle-steps", "10", "--backend", "file"] + extra
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ranks_host.py"), "bench", str(q), str(n // per), str(per), boot, outp, "--"] + args,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=ROOT) for q in range(n // per)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("hosted bench ranks timed out")
    for q, (p, log) in enumerate(zip(procs, logs)):
        assert p.returncode == 0, "process %d:\n%s" % (q, log[-4000:])
    return json.load(open(outp))


@pytest.mark.skipif(os.environ.get("EXAMG_HOSTED_RANKS") != "1",
                    reason="several ranks per process (threads): opt-in, EXAMG_HOSTED_RANKS=1 (tools/gpu_rehearse_multi.sh)")
@pytest.mark.parametrize("blocks", ["2,2,2", "1,2,4"])
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_eight_ranks_on_one_gpu(blocks, scaling):
    """The exact 8-rank job of the driver's 8-GPU run -- bench.py at N = 8 in both decompositions (--blocks cube = 2 x 2 x 2 of
    SURVEY.md 8e; default 1 x 2 x 4) and both scaling modes -- rehearsed on ONE GPU over the peer-write transport: 4 processes of 2
    ranks each.  No hang, overlapped Jacobi pairs and the V-cycle with neighbours replayed from a hipGraph, duplicate planes
    bit-identical on both owners, the Solve converges in the single block's number of cycles."""
    # weak: a 64^3-cell block per rank; strong: 128^3 cells in total (1 x 2 x 4: blocks of 128 x 64 x 32 cells)
    r = _bench_hosted(8, 2, ["--blocks", blocks, "--scaling", scaling], level=6 if scaling == "weak" else 7)
    assert r["n_gpus"] == 8 and r["value"] > 0
    assert r["config"]["blocks"] == [int(b) for b in blocks.split(",")]
    assert r["transport"] == "peer" and "transport_notes" not in r
    assert r["duplicate_planes_bit_identical"] is True
    assert r["vcycle_graph"] is True and "vcycle_error" not in r, r.get("vcycle_error")
    assert r["vcycle_duplicate_planes_bit_identical"] is True
    assert r.get("solve_iterations") in (5, 6, 7, 8), r
    assert r["solve_residual_reduction"] < 1e-6
    assert r["scaling"] == scaling
