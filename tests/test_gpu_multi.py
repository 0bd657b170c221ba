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
           "--no-cpu-baseline", "--check-duplicates"] + extra
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
