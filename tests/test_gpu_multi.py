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
