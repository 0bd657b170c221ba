#!/usr/bin/env python
"""Peer-write transport, two processes on one GPU: exchanges and gathers before and after the regions are re-allocated for a larger field.
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/lab/regrow_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("EXAMG_PEER_TIMEOUT_MS", "15000")
import numpy as np
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from exastencils_amd.comm import Communicator
from exastencils_amd.domain import RectDomain
from exastencils_amd.field import Field
from exastencils_amd.layout import FieldLayout
from exastencils_amd.ops import HipOps

ops = HipOps(0)
dom = RectDomain(3, (1, 1, world), rank, (2, 2, 2 // world if world <= 2 else 1))
L = 5
nc = dom.ncells(L)
comm = Communicator(dom, ops, concurrent_ghost_axes=True, consistent_duplicates=True, transport="peer")
full = Communicator(dom, ops, transport="peer")
S1 = Field("S1", L, FieldLayout.node(3, nc, 1, True, True, 0), ops, 1, None)
S2 = Field("S2", L, FieldLayout.node(3, nc, 2, True, True, 0), ops, 1, None)
ops.fill_random(S1.data(), 3 + rank); ops.fill_random(S2.data(), 5 + rank)


def stage(name, fn):
    fn()
    ops.synchronize()
    comm.check()
    dist.barrier()
    if rank == 0:
        print("ok:", name, "generation", comm.generation, flush=True)


def gather(n):
    piece = ops.from_host(np.arange(n, dtype=np.float64) + 1000.0 * rank)
    allp = ops.new_array(n * world)
    comm.all_gather([allp[r * n:(r + 1) * n] for r in range(world)], piece)
    ops.synchronize()
    want = np.concatenate([np.arange(n, dtype=np.float64) + 1000.0 * r for r in range(world)])
    assert np.array_equal(ops.to_host(allp), want)


def reduce():
    t = ops.from_host(np.array([1.0 + rank]))
    comm.allreduce(t, "sum")
    assert float(ops.to_host(t)[0]) == world * (world + 1) / 2


stage("small exchanges", lambda: [comm.exchange(S1, None, "ghost", True) for _ in range(3)])
stage("gather 1000", lambda: gather(1000))
stage("reduce", reduce)
stage("larger field, other communicator object (regions grow)", lambda: [full.exchange(S2, None, "all") for _ in range(2)])
stage("larger field", lambda: [comm.exchange(S2, None, "ghost", True) for _ in range(3)])
stage("small exchanges again", lambda: [comm.exchange(S1, None, "ghost", True) for _ in range(3)])
stage("reduce", reduce)
stage("gather 1000 again", lambda: gather(1000))
stage("gather 50000 (regions grow)", lambda: gather(50000))
stage("exchange, reduce, gather", lambda: (comm.exchange(S1, None, "ghost", True), reduce(), gather(50000), comm.exchange(S2, None, "ghost", True)))
dist.barrier()
dist.destroy_process_group()
