#!/usr/bin/env python
"""bench.py -- the reference's headline metric on MI355X.

step      = one application of the smoother function of the generated program
            (Testing/Smoothers/Jac.exa4:125-131: `communicate ghost of Solution<active>`; the 3-D 7-point
            Jacobi loop; `advance`) on one 512^3-cell block per GPU (515^3-double fields in the
            reference layout, 511^3 updated points).
metric    = LU/s (lattice updates per second, the authors' formula Testing/PolyExpl/Jac3Dcc.exa4:58),
            whole job: ranks x points x steps / max-over-ranks time.
            K steps run as passes of three steps on a block without neighbours (examg_jacobi3: 20 steps = 6 x 3 + 2) and of two
            steps across block neighbours (exastencils_amd/smoothers.py: temporal blocking, the reference's contracting-loop
            idea, baseExt/ir/IR_ContractingLoop.scala; bit-identical to K single steps) unless --no-temporal-blocking;
            --temporal-depth 2 keeps the single block on two-step passes.
roofline  = HBM.  `achieved` = compulsory bytes of ONE launch of the dominant kernel / its average duration (HIP events on
            the launch stream).  Compulsory bytes follow the reference's own rule
            (Compiler/src/exastencils/performance/ir/IR_EvaluatePerformanceEstimates.scala:206-215): 8 B x points x distinct
            (field, slot, read/write) streams of the launch = 24 B per point for a Jacobi pass -- also for the two- and three-step
            kernels, which read u and rhs once and write once for TWO / THREE updates per point.  `frac` = achieved / 8 TB/s is
            therefore a fraction of the roofline for every kernel; the per-update figure (24 B per lattice update, which
            exceeds the peak when two updates share a pass) is reported separately as `lu_equivalent_*`.
            `traffic` = fabric-side bytes per launch from rocprofv3 PMC passes (profiles/r02_pmc_kernels.json), given only
            when that profile was taken for this kernel at this block size and layout.
            `roofline_kernels` = the same accounting, timed live, for every hot kernel of the V-cycle and the V-cycle itself.
cpu_baseline = the restated reference CPU path (oracle/examg_oracle.c, generator-shaped OpenMP loops) timed on this host's
            cores on a bounded sample of the same 512^3 workload: Jacobi sweeps (the headline `value`), and per kernel
            (red-black sweep, residual, restriction, prolongation, one V-cycle) as BASELINE.md section 3 lists.

Launch: `python bench.py` (1 GPU) or
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W [--scaling strong]`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_LU = 24.0
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# counter profiles, newest first (tools/gpu_pmc.sh; the commit and, per kernel, the digest of its sources are inside); level 8 / padded
# layouts: <name>_L8.json / <name>_align16.json
PMC_PROFILES = [os.path.join(ROOT, "profiles", n) for n in ("r04_pmc_kernels.json", "r03_pmc_kernels.json")]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--level", type=int, default=9, help="finest level: 2^level cells per dim per GPU (9 => 512^3)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: 2^level cells per dim PER GPU (default); strong: 2^level cells per dim in TOTAL, divided among the blocks")
    ap.add_argument("--align", type=int, default=0,
                    help="row padding of the field layouts in doubles (IR_AddPaddingToFieldLayouts); 0 = the verbatim reference layout")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vcycle", action="store_true")
    ap.add_argument("--settle-steps", type=int, default=150, help="untimed smoother steps before the warm-up (device clocks reach their steady state)")
    ap.add_argument("--halo", choices=("deep", "shell"), default="deep",
                    help="N > 1: two ghost layers and one kernel per pass (deep), or one ghost layer and interior + shell launches (shell)")
    ap.add_argument("--preflight-timeout", type=float, default=180.0, help="N > 1: seconds the first exchange + pass may take")
    ap.add_argument("--no-kernel-table", action="store_true")
    ap.add_argument("--no-temporal-blocking", action="store_true", help="one kernel launch per smoother step")
    ap.add_argument("--temporal-depth", type=int, default=3, choices=(1, 2, 3),
                    help="smoother steps per pass over HBM (3: single block only -- blocks with neighbours have two ghost layers and run pairs)")
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    ap.add_argument("--extras-timeout", type=float, default=240.0, help="seconds the extra measurements may take")
    ap.add_argument("--no-check-duplicates", action="store_true",
                    help="N > 1: skip the bit-for-bit comparison of the shared duplicate planes after the timed steps and after the Solve "
                         "(what leaving their exchange out relies on; on by default)")
    ap.add_argument("--sustained-seconds", type=float, default=5.0,
                    help="after the K timed steps: the same step for at least this long, reported as sustained_* beside the headline (0: skip)")
    ap.add_argument("--blocks", default="",
                    help="N > 1: block decomposition -- 'zy' (default: 8 -> 1x2x4, whole rows / planes in every halo), 'cube' (8 -> 2x2x2, SURVEY.md 8e: "
                         "the reference's domain_rect_numBlocks of its benchmark configurations) or explicit 'bx,by,bz'")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend that carries the bootstrap / timing collectives; 'gloo': several ranks "
                                                      "on ONE GPU (the halo traffic still moves device to device through the peer-write transport)")
    return ap.parse_args(argv)


def host_info():
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count(), "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS")}


def cpu_baseline(seconds: float):
    """Restated reference CPU path on a bounded sample of the same workload (512^3 cells, reference layout) on the hardware
    threads this process is allowed to use: Jacobi sweeps for ~`seconds` s (`value`), then ~2 s per other kernel and one
    V(3,3) cycle of the Benchmark/Poisson3D program."""
    import ctypes as C

    from oracle import mg

    # every loop nest in the generator's default shape: `#pragma omp parallel for schedule(static)` on the OUTER loop only
    # (omp_useCollapse = false, parallelization/api/omp/OMP_Loop.scala:47,103-110) -- oracle/libexamg_oracle_gen.so
    mg.generator_shape(True)
    L = mg.lib()
    L.orc_set_num_threads(min(int(L.orc_num_threads()), mg.cpu_budget()))   # affinity mask capped by the cgroup CPU quota
    n = 512
    lu, lf = mg.Layout.node(3, (n, n, n), 1), mg.Layout.node(3, (n, n, n), 0)
    lcu, lcf = mg.Layout.node(3, (n // 2,) * 3, 1), mg.Layout.node(3, (n // 2,) * 3, 0)
    u, un, f = lu.alloc(), lu.alloc(), lf.alloc()
    uc, fc = lcu.alloc(), lcf.alloc()
    L.orc_fill_random(u.ctypes.data, u.size, 12345)
    L.orc_fill_random(f.ctypes.data, f.size, 777)
    L.orc_fill_random(uc.ctypes.data, uc.size, 5)
    st = mg.laplace_examples(3, (1.0 / n,) * 3)
    sc, luc, lfc, lcuc, lcfc = st.c(), lu.c(), lf.c(), lcu.c(), lcf.c()
    w = 0.8 / st.coefs[0]
    b, e = (C.c_int * 3)(1, 1, 1), (C.c_int * 3)(n, n, n)
    bc, ec = (C.c_int * 3)(1, 1, 1), (C.c_int * 3)(n // 2, n // 2, n // 2)
    ptr = [u.ctypes.data, un.ctypes.data]
    pts, cpts = (n - 1) ** 3, (n // 2 - 1) ** 3

    def timed(fn, budget, min_reps=2):
        fn(0)
        t0 = time.perf_counter()
        k = 0
        while True:
            fn(k)
            k += 1
            dt = time.perf_counter() - t0
            if (dt >= budget and k >= min_reps) or k >= 100000:
                return k, dt

    def jac(i):
        L.orc_jacobi7_const(C.byref(luc), ptr[i % 2], C.byref(lfc), f.ctypes.data, ptr[(i + 1) % 2], C.byref(sc), w, b, e)

    def rbgs(i):
        for colour in (0, 1):
            L.orc_stencil_op(2, C.byref(luc), u.ctypes.data, C.byref(lfc), f.ctypes.data, C.byref(luc), u.ctypes.data, C.byref(sc), w, colour, b, e)

    def residual(i):
        L.orc_stencil_op(1, C.byref(luc), u.ctypes.data, C.byref(lfc), f.ctypes.data, C.byref(luc), un.ctypes.data, C.byref(sc), 0.0, -1, b, e)

    def restrict(i):
        L.orc_restrict(C.byref(luc), un.ctypes.data, C.byref(lcfc), fc.ctypes.data, 1.0, bc, ec)

    def prolong(i):
        L.orc_prolong_add(C.byref(lcuc), uc.ctypes.data, C.byref(luc), u.ctypes.data, b, e)

    k, dt = timed(jac, seconds, 4)
    out = {
        "value": pts * k / dt,
        "unit": "LU/s",
        "cores": int(L.orc_num_threads()),
        "kind": "port",
        "loop_shape": "z-y-x nest, OpenMP pragma on the outer loop only (the generator's default, omp_useCollapse = false), -O3 -fopenmp -ffp-contract=off",
        "sample": "%d Jacobi 7-pt sweeps of 512^3 cells (511^3 updates each), restated generator-shaped OpenMP loop, %.1f s; "
                  "per kernel ~2 s each; one V(3,3) cycle (levels 4..9) of the Benchmark/Poisson3D program" % (k, dt),
    }
    out.update(host_info())
    per = {}
    for name, fn, units, bpu in (("rbgs_sweep", rbgs, pts, 48.0), ("residual", residual, pts, 24.0), ("restrict", restrict, cpts, 72.0),
                                 ("prolong_add", prolong, pts, 17.0)):
        k, dt = timed(fn, 2.0)
        per[name] = {"ms": dt / k * 1e3, "units_per_s": units * k / dt, "algorithmic_gbs": units * bpu * k / dt / 1e9}
    per["jacobi"] = {"ms": 1e3 * pts / out["value"], "units_per_s": out["value"], "algorithmic_gbs": out["value"] * 24.0 / 1e9}
    del u, un, f, uc, fc
    P = mg.ProgramA(mg.ConfigA(nd=3, min_level=4, max_level=9, tol=1e-6))
    P.setup()
    P.mgCycle(9)
    t0 = time.perf_counter()
    P.mgCycle(9)
    per["vcycle"] = {"ms": (time.perf_counter() - t0) * 1e3, "levels": 6}
    out["kernels"] = per
    return out


def pmc_traffic(case, level, align):
    """Fabric-side bytes per launch from the committed counter profile -- only for this kernel case, block size and layout, and only
    while the kernel's sources are the ones that were profiled (tools/pmc_reduce.py: kernel_source_digest): counters taken on another
    build of the kernel describe another kernel, and are reported as stale instead of as `traffic`."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_reduce

    for path in PMC_PROFILES:
        if level != 9:
            path = path.replace(".json", "_L%d.json" % level)
        if align:
            path = path.replace(".json", "_align%d.json" % align)
        try:
            prof = json.load(open(path))
        except (OSError, ValueError):
            continue
        if prof.get("level") != level or prof.get("align", 0) != align:
            continue
        for k in prof.get("kernels", []):
            if k.get("case") == case and "traffic" in k:
                rec = {"traffic": k["traffic"], "fetch_bytes": k["fetch_bytes"], "write_bytes": k["write_bytes"],
                       "source": "profiles/" + os.path.basename(path), "source_commit": prof.get("commit"), "source_kernel": k.get("kernel_name")}
                digest = k.get("source_digest")
                if digest is None or digest != pmc_reduce.kernel_source_digest(case):
                    rec["stale"] = True      # the kernel changed since (or the profile predates the digests): not this build's traffic
                return rec
    return None


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs between the ranks of this host
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with %d ranks" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    local_rank %= max(1, torch.cuda.device_count())      # rehearsal: more ranks than GPUs share the cards
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    run(args, world, rank, local_rank, dist)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run(args, world, rank, local_rank, dist, injected=False):
    """One rank of the benchmark.  `dist`: torch.distributed (main) or an object with the same calls -- tests/filedist.py, with which
    tests/test_gpu_ranks8.py hosts several ranks per process (one thread and stream each) to rehearse the 8-rank job on a one-GPU box
    (`injected=True`: every Communicator of the run bootstraps through it).  Rank 0 prints the JSON line and returns it."""
    import torch

    dist_module = dist if injected else None
    from exastencils_amd.comm import Communicator
    from exastencils_amd.domain import RectDomain
    from exastencils_amd.field import Field, laplace_fd
    from exastencils_amd.layout import FieldLayout
    from exastencils_amd.ops import HipOps

    ops = HipOps(local_rank)
    nd, L = 3, args.level
    blocks = RectDomain.parse_blocks(args.blocks, world, nd)
    if args.scaling == "weak":
        # weak scaling keeps the mesh width: the physical domain grows with the blocks, [0,1]^3 per block (a unit cube cut into
        # 1 x 2 x 4 blocks of 512^3 cells would be an anisotropic mesh, on which point smoothers with full coarsening degrade)
        dom = RectDomain(nd, blocks, rank, hi=tuple(float(b) for b in blocks))
    else:
        # strong scaling: 2^level cells per dimension in total, the unit cube; a block keeps
        # 2^level / blocks[d] cells in dimension d = frag_len[d] * 2^(level - shift) with shift = log2(max blocks)
        shift = max(blocks).bit_length() - 1
        L = args.level - shift
        dom = RectDomain(nd, blocks, rank, tuple(max(blocks) // b for b in blocks))
    # 7-point loops read face ghosts only (one batch per exchange); duplicate planes are computed to the same bits on
    # both sides by every loop of these programs, so their upstream exchange is left out (exastencils_amd/comm.py)
    transport_notes = []
    # At N > 1 nothing may look like a slow benchmark that is in fact a transport which never completes: from here until the first
    # overlapped pass has finished a watchdog ends the process with a diagnosis (exit code 4) after --preflight-timeout seconds.
    preflight_done, comm = None, None
    if world > 1:
        import threading

        preflight_done = threading.Event()

        def preflight_watchdog():
            if not preflight_done.wait(args.preflight_timeout):
                sys.stderr.write("bench.py rank %d: transport probe / first halo exchange did not complete within %g s (transport %r, tried before: %r); "
                                 "EXAMG_TRANSPORT=torch pins the torch.distributed point-to-point path\n"
                                 % (rank, args.preflight_timeout, getattr(comm, "transport", os.environ.get("EXAMG_TRANSPORT")), transport_notes))
                sys.stderr.flush()
                os._exit(4)

        threading.Thread(target=preflight_watchdog, daemon=True).start()
        comm = open_transport(dom, ops, dist, args, transport_notes, dist_module)
    else:
        comm = Communicator(dom, ops, concurrent_ghost_axes=True, consistent_duplicates=True)
    nc = dom.ncells(L)
    if world > 1 and getattr(comm, "transport", None) == "peer":
        # size the peer-write regions ONCE for everything this run exchanges -- the widest face of the finest level (two ghost layers) and
        # the gathered coarse level of the V-cycle leg -- so that no region is re-allocated (a collective: synchronise, barrier, unmap,
        # barrier, map) in the middle of the measurements
        from exastencils_amd.layout import FieldLayout as _FL

        gather = 8
        for d in range(nd):
            gather *= dom.ncells(max(L - 3, 0))[d] + 3
        comm._peer_ensure(comm._max_face_bytes(_FL.node(nd, nc, 2, True, True, args.align), nd), gather)
    # N > 1: two ghost layers on Solution and one on RHS (the layouts' `ghostLayers`; the reference needs as many for its contracting
    # loops, baseExt/ir/IR_ContractingLoop.scala:45-196): a two-step pass is then ONE exchange and ONE kernel per block -- the first stage
    # also covers the neighbour's first plane -- instead of an interior kernel beside a shell of thin launches and two exchanges
    # (--halo shell: that scheme, one ghost layer; exastencils_amd/smoothers.py)
    deep = world > 1 and args.halo == "deep"
    Solution = Field("Solution", L, FieldLayout.node(nd, nc, 2 if deep else 1, True, True, args.align), ops, 2, None)
    RHS = Field("RHS", L, FieldLayout.node(nd, nc, 1, True, True, args.align) if deep else FieldLayout.node(nd, nc, 0, False, False, args.align), ops, 1, None)
    # synthetic data: one global random field, so that the duplicate planes two blocks share hold the same values on both
    for t, seed in ((Solution.data(0), 12345), (Solution.data(1), 12345), (RHS.data(), 777)):
        ops.fill_random(t, seed + (0 if world == 1 else 1000 * rank))
    A = laplace_fd(nd, dom.h(L), "mp")
    w = 0.8 / A.diag
    b, e = dom.loop_bounds(Solution.layout)
    updates = (e[0] - b[0]) * (e[1] - b[1]) * (e[2] - b[2])

    from exastencils_amd.smoothers import jacobi_pair, jacobi_triple

    # steps per pass over HBM: 3 on a block without neighbours (k_three_stage7_lds), 2 with neighbours (two ghost layers; three steps
    # without an exchange would need three), 1 with --no-temporal-blocking
    depth = 1 if args.no_temporal_blocking else min(args.temporal_depth, 3 if world == 1 else 2)

    Tmp = Field("SolutionTmp", L, Solution.layout, ops, 1, None)

    dup_check = None
    if world > 1:
        # make the synthetic fields consistent across blocks with one full exchange: Solution's duplicate planes from the upstream
        # block and its ghosts from the neighbours; RHS (a layout without communication) through a communicating alias of the same
        # array -- both owners of a shared plane then compute the same bits in every step, which `consistent_duplicates` relies on
        # and which is VERIFIED after the timed steps (duplicate_planes_bit_identical)
        full = Communicator(dom, ops, dist_module=dist_module)
        for s_ in (0, 1):
            full.exchange(Solution, s_, "all")
        if deep:
            full.exchange(RHS, None, "all")
        else:
            rhs_alias = Field("RHS", L, FieldLayout.node(nd, nc, 0, True, False, args.align), ops, 1, None)
            rhs_alias.slots[0] = RHS.data()
            full.exchange(rhs_alias, None, "dup")

    def step():
        # Function Smoother@finest: communicate ghost of Solution<active>; Jacobi loop; advance
        comm.exchange(Solution, Solution.active, "ghost", axis_only=True)
        ops.stencil_op(2, Solution.lc, Solution.data(Solution.active), RHS.lc, RHS.data(), Solution.lc,
                       Solution.data(Solution.next), A, w, -1, b, e)
        Solution.advance()

    def steps(k):
        """k smoother applications; consecutive pairs run as one pass over HBM (temporal blocking,
        exastencils_amd/smoothers.py) unless --no-temporal-blocking: same results bit for bit."""
        if depth == 1:
            for _ in range(k):
                step()
            return
        if depth == 3:      # a block without neighbours: three steps per pass (examg_jacobi3), the rest as a pair or a step
            while k >= 3 and (k - 3) != 1:
                jacobi_triple(ops, comm, dom, Solution, RHS, A, w, Tmp)
                k -= 3
        for _ in range(k // 2):
            jacobi_pair(ops, comm, dom, Solution, RHS, A, w, Tmp)
        if k % 2:
            step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    comm.exchange(Solution, Solution.active, "ghost", axis_only=True)
    barrier()
    if world > 1:
        steps(2)               # one overlapped pass with its exchanges on the side stream
        barrier()
        preflight_done.set()
    # Clock settle (untimed, before the W warm-up steps): when load arrives on an idle device the power management overshoots
    # for the first ~30 ms -- passes of 0.67 ms rise to 0.90 ms and come back to 0.66-0.69 ms (tools/pass_times.py) -- so the
    # first ten passes after an idle period run ~20 % below the steady state every longer run sees (1000 steps: 0.329 ms per step,
    # 20 steps from idle: 0.395).  The same smoother steps are run for --settle-steps steps first; the timed region is unchanged.
    steps(2 * (args.settle_steps // 2))
    barrier()
    steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=ops.device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the job's lattice updates per step: every grid point once (duplicate planes at interior faces are computed by both
        # neighbours; the reduction box of a block leaves its lower duplicate planes to the neighbour)
        ub, ue = dom.loop_bounds(Solution.layout, reduction=True)
        unique = (ue[0] - ub[0]) * (ue[1] - ub[1]) * (ue[2] - ub[2])
        t = torch.tensor([float(unique)], dtype=torch.float64, device=t.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_updates = int(t.item())
    else:
        total_updates = updates

    if hasattr(comm, "check"):
        comm.check()           # a wait of the peer-write transport that gave up is an error, not a number
    if world > 1 and not args.no_check_duplicates:
        # after settle + warm-up + timed steps, all without the duplicate exchange: do both owners of every shared plane hold the same bits?
        dup_check = bool(comm.check_duplicates(Solution, Solution.active))
    # the two smoother kernels alone, events on the launch stream -- in the clock state of the timed steps (the sustained leg, which
    # holds the device at full power for seconds, comes afterwards; at N > 1 these launches run without exchanges: timing only)
    stream = torch.cuda.current_stream()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nk = max(10, min(args.steps, 100))
    torch.cuda.synchronize()
    ev0.record(stream)
    for _ in range(nk):
        ops.stencil_op(2, Solution.lc, Solution.data(Solution.active), RHS.lc, RHS.data(), Solution.lc,
                       Solution.data(Solution.next), A, w, -1, b, e)
        Solution.advance()
    ev1.record(stream)
    torch.cuda.synchronize()
    single_ms = ev0.elapsed_time(ev1) / nk
    ev0.record(stream)
    for _ in range(nk):
        ops.jacobi2(Solution.lc, Solution.data(Solution.active), Solution.data(Solution.next), Tmp.data(), RHS.lc, RHS.data(),
                    A, w, b, e)
        Solution.advance()
    ev1.record(stream)
    torch.cuda.synchronize()
    pair_ms = ev0.elapsed_time(ev1) / nk
    triple_ms = None
    if world == 1:
        ev0.record(stream)
        for _ in range(nk):
            ops.jacobi3(Solution.lc, Solution.data(Solution.active), Solution.data(Solution.next), Tmp.data(), RHS.lc, RHS.data(),
                        A, w, b, e)
            Solution.advance()
        ev1.record(stream)
        torch.cuda.synchronize()
        triple_ms = ev0.elapsed_time(ev1) / nk
    # sustained leg: the same step for >= --sustained-seconds, reported beside the K-step value (the driver's SMI samples then
    # see the device under load, and the settle-phase argument is a measurement: a K-step value near this one was taken at the
    # steady-state clocks)
    sustained = None
    if args.sustained_seconds > 0:
        # ~0.25 s of steps between looks at the clock; dt is the all-reduced maximum here, so every rank forms the same chunks
        chunk = max(2, 2 * (int(0.25 / max(dt / args.steps, 1e-6)) // 2))
        if injected:
            # several ranks hosted by ONE process (tests/ranks_host.py): a host thread that runs hundreds of launches ahead of the device
            # can block inside the HIP runtime (full queue) while the launch its kernels wait for belongs to the other thread of the
            # same process.  One rank per process -- the product's launch -- has no such coupling.
            chunk = min(chunk, 16)
        n_sus, t_sus = 0, 0.0
        barrier()
        t1 = time.perf_counter()
        while True:
            steps(chunk)
            n_sus += chunk
            torch.cuda.synchronize()
            go = 1.0 if time.perf_counter() - t1 < args.sustained_seconds else 0.0
            if world > 1:         # every rank runs the same number of chunks
                tg = torch.tensor([go], dtype=torch.float64, device=ops.device if args.backend == "nccl" else "cpu")
                dist.all_reduce(tg, op=dist.ReduceOp.MAX)
                go = float(tg.item())
            if go == 0.0:
                break
        barrier()
        t_sus = time.perf_counter() - t1
        sustained = (n_sus, t_sus)
    compulsory = BYTES_PER_LU * updates          # one pass: read u, read rhs, write u' -- for either kernel
    single_gbs = compulsory / (single_ms * 1e-3) / 1e9
    pair_gbs = compulsory / (pair_ms * 1e-3) / 1e9
    if depth == 1:
        case, kernel_name, kernel_ms, achieved, lus = "jacobi_1step", "k_stencil7_zmarch (one Jacobi step per launch)", single_ms, single_gbs, updates
    elif depth == 3 and triple_ms is not None:
        case, kernel_name, kernel_ms, achieved, lus = ("jacobi_3step", "k_three_stage7_lds (three Jacobi steps per launch)", triple_ms,
                                                       compulsory / (triple_ms * 1e-3) / 1e9, 3 * updates)
    else:
        case, kernel_name, kernel_ms, achieved, lus = "jacobi_2step", "k_two_stage7_lds (two Jacobi steps per launch)", pair_ms, pair_gbs, 2 * updates

    extra = {
        "jacobi_single_step_kernel_ms": single_ms,
        "jacobi_single_step_frac": single_gbs / HBM_PEAK_GBS,
        "jacobi_two_step_kernel_ms": pair_ms,
        "jacobi_two_step_frac": pair_gbs / HBM_PEAK_GBS,
        "temporal_blocking": depth > 1,
        "temporal_depth": depth,
    }
    if triple_ms is not None:
        extra["jacobi_three_step_kernel_ms"] = triple_ms
        extra["jacobi_three_step_frac"] = compulsory / (triple_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    out = None
    if rank == 0:
        pmc = pmc_traffic(case, L, args.align) if nc[0] == nc[1] == nc[2] == (1 << L) else None
        out = {
            "metric": "LU/s",
            "value": total_updates * args.steps / dt,
            "unit": "LU/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": 2 * (args.settle_steps // 2),
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "3D Poisson 7-point Jacobi smoother steps (ghost exchange + sweep + advance), %d x %d x %d cells per GPU, "
                            "%s (%d-double rows)%s"
                            % (nc[0], nc[1], nc[2],
                               "reference field layout" if not args.align else "reference layout model with rows padded to multiples of %d doubles" % args.align,
                               Solution.layout.tot(0),
                               "" if depth == 1 else "; %s consecutive steps per pass over HBM (temporal blocking, bit-identical)" % ("three" if depth == 3 else "two")),
                "blocks": list(dom.num_blocks),
                "updates_per_step_per_gpu": updates,
                "levels": L,
                "align": args.align,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc["traffic"] if (pmc and not pmc.get("stale")) else None,
                "kernel": kernel_name,
                "kernel_ms": kernel_ms,
                "compulsory_bytes_per_launch": compulsory,
                "compulsory_bytes_per_point": BYTES_PER_LU,
                "lu_per_launch": lus,
                # 24 B per lattice UPDATE: with two updates per pass this exceeds the peak -- a throughput
                # statement, not a roofline fraction
                "lu_equivalent_gbs": BYTES_PER_LU * lus / (kernel_ms * 1e-3) / 1e9,
                "lu_equivalent_frac": BYTES_PER_LU * lus / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic_over_compulsory": (pmc["traffic"] / compulsory) if (pmc and not pmc.get("stale")) else None,
                "traffic_source": pmc["source"] if pmc else None,
                "traffic_source_commit": pmc.get("source_commit") if pmc else None,     # the tree the counter profile was taken at
                # True: the kernel's sources changed since that profile -- its counters are not reported as this build's traffic
                "traffic_source_stale": bool(pmc.get("stale")) if pmc else None,
            },
        }
        if dup_check is not None:
            out["duplicate_planes_bit_identical"] = dup_check       # compared after the settle, warm-up and timed steps
        if world > 1:
            out["transport"] = getattr(comm, "transport", None)
            out["halo"] = "deep (two ghost layers: one exchange + one kernel per pass)" if deep else "shell (one ghost layer: interior kernel beside shell launches, two exchanges per pass)"
            if transport_notes:
                out["transport_notes"] = transport_notes      # transports that were tried first and did not pass the probe, and why
        if sustained is not None:
            n_sus, t_sus = sustained
            out["sustained_steps"] = n_sus
            out["sustained_seconds"] = t_sus
            out["sustained_ms_per_step"] = t_sus / n_sus * 1e3
            out["sustained_value"] = total_updates * n_sus / t_sus
        out.update(extra)

    # Extra measurements (kernel table, 256^3 block of configs[1], V-cycle / Solve of config 3).  They run collectives at
    # N > 1; the headline above must survive whatever happens here: a watchdog thread prints it without the extras and ends
    # the process with a non-zero code if they do not finish in time (a blocked collective keeps the main thread in a C call).
    if not args.no_vcycle:
        import threading

        done = threading.Event()

        def watchdog():
            if not done.wait(args.extras_timeout):
                if rank == 0:
                    out["vcycle_error"] = "extras did not finish within %g s" % args.extras_timeout
                    print(json.dumps(out), flush=True)
                os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()
        more = {}
        if world == 1 and not args.no_kernel_table:
            try:
                more["roofline_kernels"] = kernel_table(ops, L, args.align)
            except Exception as ex:
                more["roofline_kernels_error"] = repr(ex)[:300]
        try:
            more.update(config1(ops, world))
        except Exception as ex:
            more["config1_error"] = repr(ex)[:300]
        try:
            more.update(vcycle(ops, dom, comm, L, world, args.align, check_dups=not args.no_check_duplicates, deep_halo=args.halo == "deep"))
        except Exception as ex:  # the headline number must not depend on the extra measurement
            more["vcycle_error"] = repr(ex)[:300]
        if world == 1:
            try:
                torch.cuda.empty_cache()
                more.update(fmg_solve(ops, L, args.align))
            except Exception as ex:
                more["fmg_error"] = repr(ex)[:300]
            try:
                torch.cuda.empty_cache()
                more.update(helmholtz27_cycle(ops, L))
            except Exception as ex:
                more["helmholtz27_error"] = repr(ex)[:300]
            try:
                torch.cuda.empty_cache()
                more.update(shim_cycle(L))
            except Exception as ex:
                more["shim_error"] = repr(ex)[:300]
        done.set()
        if rank == 0:
            if "roofline_kernels" in more and "vcycle_ms" in more:
                cb = more.pop("vcycle_compulsory_bytes")
                more["roofline_kernels"].append({"case": "vcycle_v33_6levels", "kernel": "hipGraph of one mgCycle@finest", "ms": more["vcycle_ms"],
                                                 "compulsory_bytes": cb, "frac": cb / (more["vcycle_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS})
            more.pop("vcycle_compulsory_bytes", None)
            out.update(more)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(out), flush=True)
    return out


def open_transport(dom, ops, dist, args, notes, dist_module=None):
    """The communicator of an N > 1 run: the transport is PROVEN by a real exchange + all-reduce on a small field before anything is
    timed, and the verdict is shared by all ranks (MIN over ranks).  Default order: peer writes through HIP IPC, then RCCL
    send / recv groups, then torch.distributed point-to-point; EXAMG_TRANSPORT pins one.  Nothing is silent: the transport that
    runs is in the JSON line (`transport`), and so is every transport that was tried before it with the reason it was dropped
    (`transport_notes`).  Every Communicator the run creates afterwards takes the same transport (EXAMG_TRANSPORT is set)."""
    import torch

    from exastencils_amd import comm as comm_mod
    from exastencils_amd.comm import Communicator
    from exastencils_amd.field import Field
    from exastencils_amd.layout import FieldLayout

    pinned = os.environ.get("EXAMG_TRANSPORT", "")
    order = [pinned] if pinned in ("peer", "c", "torch") else (["peer", "c", "torch"] if args.backend == "nccl" else ["peer", "torch"])
    world = dom.world_size
    dev = ops.device if args.backend == "nccl" else "cpu"
    # a peer that never answers the probe must leave time for the next transport inside --preflight-timeout (the library's default
    # patience of 120 s is meant for neighbours that write files); the communicators of the run inherit it: ranks of a bench run in step
    os.environ.setdefault("EXAMG_PEER_TIMEOUT_MS", "40000")
    for tr in order:
        os.environ["EXAMG_TRANSPORT"] = tr
        ok, why = 1.0, ""
        try:
            c = Communicator(dom, ops, concurrent_ghost_axes=True, consistent_duplicates=True, transport=tr, dist_module=dist_module)
            probe = Field("probe", 4, FieldLayout.node(dom.nd, dom.ncells(4), 1, True, True, 0), ops, 1, None)
            probe.data().fill_(1.0 + dom.rank)
            full = Communicator(dom, ops, transport=tr, dist_module=dist_module)
            full.exchange(probe, None, "all")
            ops.synchronize()
            # the VALUES that arrived: every ghost face towards a neighbour holds that neighbour's number (a transport whose
            # messages complete but whose bytes are stale -- a cached mapping across GPUs -- fails here, not in a wrong residual later)
            a, lay = probe.host_array(ops), probe.layout
            for d in range(dom.nd):
                for side in (-1, 1):
                    nb = dom.neighbor(d, side)
                    if nb is None:
                        continue
                    g = lay.ref(d) - 1 if side < 0 else lay.ref(d) + dom.ncells(4)[d] + 1
                    sl = [slice(lay.ref(e) + 1, lay.ref(e) + dom.ncells(4)[e]) if e < dom.nd else slice(None) for e in range(3)]
                    sl[d] = g
                    face = a[tuple(reversed(sl))]
                    if not (face == 1.0 + nb).all():
                        raise RuntimeError("ghost face %+d of axis %d holds %r .. %r, not rank %d's value" % (side, d, float(face.min()), float(face.max()), nb))
            t = ops.from_host(__import__("numpy").array([1.0 + dom.rank]))
            c.allreduce(t, "sum")
            ops.synchronize()
            if hasattr(c, "check"):
                c.check()
            if float(ops.to_host(t)[0]) != world * (world + 1) / 2.0:
                raise RuntimeError("all-reduce of 1 + rank over %d ranks gave %r" % (world, float(ops.to_host(t)[0])))
        except Exception as ex:     # noqa: BLE001 -- whatever it was, all ranks move on to the next transport together
            ok, why = 0.0, "%s: %s" % (type(ex).__name__, str(ex)[:300])
        v = torch.tensor([ok], dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        if float(v.item()) == 1.0:
            return c
        notes.append({"transport": tr, "rank": dom.rank, "reason": why or "failed on another rank"})
        comm_mod._SHARED_C_COMMS.clear()       # the next transport starts from a clean slate (the dropped handles are not reused)
    raise SystemExit("bench.py: no block-to-block transport passed its probe: %r" % (notes,))


def kernel_table(ops, level, align):
    """Every hot kernel of the V-cycle at the finest level, one launch each between HIP events: compulsory bytes (reference
    rule), ms, frac of the 8 TB/s roofline, and the PMC traffic of profiles/ when it was taken for this size."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_kernels

    cs, keep = pmc_kernels.cases(ops, level, True, align)
    rows = []
    stream = torch.cuda.current_stream()
    for name, fn, pattern, comp, lus in cs:
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        row = {"case": name, "kernel": pattern, "ms": ms, "compulsory_bytes": comp, "lattice_updates": lus,
               "frac": comp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        pmc = pmc_traffic(name, level, align)
        if pmc and not pmc.get("stale"):
            row["traffic"] = pmc["traffic"]
            row["traffic_over_compulsory"] = pmc["traffic"] / comp
        rows.append(row)
        ops.fill_random(keep["u"], 100)
    del cs, keep
    torch.cuda.empty_cache()
    return rows


def config1(ops, world):
    """BASELINE.json configs[1]: the same smoother step on a 256^3 block (single GPU), events on the launch stream."""
    import torch

    from exastencils_amd.field import laplace_fd
    from exastencils_amd.layout import FieldLayout

    if world != 1:
        return {}
    n = 256
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0, False, False)
    u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 31)
    ops.fill_random(f, 32)
    A = laplace_fd(3, (1.0 / n,) * 3, "mp")
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    Lu, Lf = lu.c_struct(), lf.c_struct()
    stream = torch.cuda.current_stream()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = {}
    for name, fn, steps in (("single_step", lambda x, y: ops.stencil_op(2, Lu, x, Lf, f, Lu, y, A, w, -1, b, e), 1),
                            ("two_step", lambda x, y: ops.jacobi2(Lu, x, y, None, Lf, f, A, w, b, e), 2),
                            ("three_step", lambda x, y: ops.jacobi3(Lu, x, y, None, Lf, f, A, w, b, e), 3)):
        # ~60 ms of the same launches first: a 256^3 launch takes ~75 us, and 50 of them from an idle device would sit entirely inside the
        # power-management transient the headline's settle phase exists for (see main: the first ~30 ms after idle run ~20 % slow)
        for _ in range(800):
            fn(u, un)
            u, un = un, u
        torch.cuda.synchronize()
        reps = 200
        ev0.record(stream)
        for _ in range(reps):
            fn(u, un)
            u, un = un, u
        ev1.record(stream)
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        out["jacobi_256cube_%s_kernel_ms" % name] = ms
        out["jacobi_256cube_%s_lups" % name] = steps * (n - 1) ** 3 / (ms * 1e-3)
        out["jacobi_256cube_%s_frac" % name] = 24.0 * (n - 1) ** 3 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        pmc = pmc_traffic("jacobi_%dstep" % steps, 8, 0)      # profiles/<round>_pmc_kernels_L8.json
        if pmc and not pmc.get("stale"):
            out["jacobi_256cube_%s_traffic" % name] = pmc["traffic"]
            out["jacobi_256cube_%s_traffic_over_compulsory" % name] = pmc["traffic"] / (24.0 * (n - 1) ** 3)
    return out


def vcycle(ops, dom, comm, L, world, align=0, check_dups=True, deep_halo=True):
    """Config 3: one V(3,3) red-black cycle of Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4, 6 levels."""
    import torch

    from exastencils_amd.solver import ConfigL4, SolverFromL4

    # more than one block: the three coarsest levels are gathered and solved redundantly on every rank (solver.py: _agg_cycle)
    agg = L - 3 if world > 1 else None
    # the gathered coarsest level grows with the blocks (16 x 32 x 64 points on 8 GPUs, 5 ms for a single-workgroup CG): it
    # coarsens log2(max blocks per dimension) levels further, back to a few hundred points
    extra = max(dom.num_blocks).bit_length() - 1 if world > 1 else 0
    cfg = ConfigL4(nd=3, min_level=L - 5, max_level=L, tol=1e-6, fused_rbgs=True, fused_residual_restrict=True, agglomerate_level=agg, fused_prolong_min_points=10_000_000, fused_zero_start=True, fused_residual_norm=True,
                   agglomerate_extra_levels=extra, align=align, deep_halo=deep_halo)
    P = SolverFromL4(cfg, ops, dom, comm)

    def rendezvous():
        # every rank has finished recording its graphs before any rank goes on (ranks hosted by one process must not meet another
        # thread's stream capture with a synchronous call; one rank per process: two cheap barriers)
        d = getattr(comm, "dist", None)
        if d is not None:
            d.barrier()

    rendezvous()
    P.setup()
    P._update_residual(L)
    r0 = P.ResNorm(L)
    # N > 1: the peer-write transport's exchanges are ordinary kernels ordered by device-side flags -- the cycle with neighbours
    # replays from a hipGraph on every rank alike; on RCCL / torch.distributed it is issued eagerly (RCCL groups inside a stream
    # capture hang on this stack) and only the agglomerated levels replay from a graph
    use_graph = world == 1 or getattr(comm, "transport", None) == "peer"
    if use_graph:
        P.capture_cycle()
        rendezvous()
        run = P.replay_cycle
    else:
        P.mgCycle(L)
        run = lambda: P.mgCycle(L)
    for _ in range(8):     # ~50 ms of cycles first: the clock transient of a device that was idle (see main) is over
        run()
    n = 10
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    P._update_residual(L)
    r1 = P.ResNorm(L)
    # the reference benchmark's own reported quantity: `totalTimeSolve` (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:263-276):
    # Solve@finest from the zero state to 1e-6 reduction, residual norm on the host after every cycle
    Q = P
    Q.reset()          # zero fields, boundary values: the benchmark's initial state (the captured graph stays valid)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    its = Q.Solve(use_graph=use_graph)
    torch.cuda.synchronize()
    solve_s = time.perf_counter() - t0
    if hasattr(comm, "check"):
        comm.check()
    # compulsory bytes of one cycle as the driver runs it, per level above the coarsest: 24 B per point for every pass of the smoother
    # (6 sweeps = 6 passes, or 4 where three sweeps run as two passes of three colour loops; 16 B for a first sweep that takes the zero
    # field as a constant), residual + restriction (16 B per point + 8 B per coarse point), zeroing the
    # coarse solution (8 B per coarse point, unless left to that sweep), prolongation + correction (16 B per point + 8 B per coarse
    # point; folded into the first post-smoothing sweep: the 8 B per coarse point only)
    comp = 0.0
    for l in range(L - 4, L + 1):
        lb, le = dom.loop_bounds(P.Solution[l].layout)
        lcb, lce = dom.loop_bounds(P.Solution[l - 1].layout)
        p = float((le[0] - lb[0]) * (le[1] - lb[1]) * (le[2] - lb[2]))
        c = float((lce[0] - lcb[0]) * (lce[1] - lcb[1]) * (lce[2] - lcb[2]))
        # passes of the smoother: a sweep (two colour loops) per pass, or -- where the kernel layer takes three colour loops per pass
        # (solver.py: _three_colour_passes) -- three plain sweeps as two passes
        three = bool(getattr(P, "_three_colour_passes", lambda _l: False)(l))
        zero = P._starts_from_zero(l)
        pre = (16.0 + 2 * 24.0) if zero else (2 * 24.0 if three else 3 * 24.0)
        post = 2 * 24.0 if (three and not P._folds_prolongation(l)) else 3 * 24.0
        comp += (pre + post) * p
        comp += 16.0 * p + 8.0 * c
        comp += 0.0 if P._starts_from_zero(l - 1) else 8.0 * c
        comp += 8.0 * c if P._folds_prolongation(l) else 16.0 * p + 8.0 * c
    npts = 1
    b, e = dom.loop_bounds(P.Solution[L].layout)
    for d in range(3):
        npts *= e[d] - b[d]
    return {
        "vcycle_ms": ms,
        "vcycle_levels": 6 + extra,
        "vcycle_residual_reduction": r1 / r0 if r0 else None,
        "vcycle_compulsory_bytes": comp,
        "vcycle_gbs_algorithmic_223B_per_point": 223.0 * npts / (ms * 1e-3) / 1e9,   # SURVEY.md 8d's per-point figure
        "totalTimeSolve": solve_s,                 # seconds
        "totalTimeSolve_ms": solve_s * 1e3,        # the unit the reference's getTotalTime / printJSON reports (timing/ir/IR_GetTime.scala:33-45)
        "solve_iterations": its,
        "solve_residual_reduction": (Q.res_history[-1] / Q.res_history[0]) if Q.res_history and Q.res_history[0] else None,
        "vcycle_graph": use_graph,
        "vcycle_duplicate_planes_bit_identical": (bool(comm.check_duplicates(Q.Solution[L])) if (world > 1 and check_dups) else None),
        "vcycle_fused_rbgs": True,
        "vcycle_agglomerate_level": agg,
    }


def shim_cycle(L):
    """The V-cycle through the reference-named seam: shim/exa_poisson3d_<min>_<max> (generated-style C++ host that calls only
    mgCycle_<L>_k<NNN>_wrapper / exch<F>_<L> / applyBCs<F>_<L>; cuda/CUDA_Kernel.scala:546-632) as a child process, once with the plain
    wrappers (one libexamg call per wrapper) and once with EXA_DEFERRED_LAUNCH=1 (wrappers record, the completing wrapper launches the
    one-pass kernel).  ms per cycle by the host's own clock around device-synchronised cycles, after its Solve."""
    import subprocess

    exe = os.path.join(ROOT, "shim", "exa_poisson3d_%d_%d" % (L - 5, L))
    if L - 5 < 1 or not os.path.exists(exe):
        return {"shim_skipped": "no shim binary for levels %d..%d (build(): 2..6 and 4..9)" % (L - 5, L)}
    out = {}
    for mode, key in (("0", "plain"), ("1", "deferred")):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, EXA_DEFERRED_LAUNCH=mode, EXA_TIME_CYCLES="10"))
        if r.returncode != 0:
            return {"shim_error": ("%s wrappers: rc %d: %s" % (key, r.returncode, r.stderr[-200:]))}
        t = [l for l in r.stdout.splitlines() if l.startswith("vcycle_ms")][0].split()
        out["shim_vcycle_ms_" + key] = float(t[1])
        out["shim_launches_per_cycle_" + key] = int(t[3])
        out["shim_solve_iterations_" + key] = int([l for l in r.stdout.splitlines() if l.startswith("iterations")][0].split()[1])
    out["shim_levels"] = [L - 5, L]
    return out


def fmg_solve(ops, L, align=0):
    """BASELINE configs[4]'s algorithm on one block: full-multigrid start (Testing/FMG/3D_Trigonometric.exa4:189-242) + red-black
    V(3,3) cycles (Testing/Smoothers/RBGS.exa4:125-133) to 1e-6, levels 2..L, from the zero state; FMG start and cycle replayed from
    hipGraphs, residual norm on the host after every cycle."""
    import torch

    from exastencils_amd.solver import ConfigL3, SolverFromL3

    cfg = ConfigL3(nd=3, min_level=2, max_level=L, smoother="rbgs", omega=1.0, stencil="scaled", restrict_scale=1.0, tol=1e-6,
                   cg_max=512, bc_fn=1, fmg=True, fused_rbgs=True, fused_residual_restrict=True, fused_prolong_min_points=10_000_000,
                   fused_zero_start=True, fused_residual_norm=True, fused_coarse=True, align=align)
    P = SolverFromL3(cfg, ops)
    P.setup()
    P.capture()
    best = None
    for _ in range(3):
        P.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        P.Solve(use_graph=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        best = ms if best is None else min(best, ms)
    return {"fmg_solve_ms": best, "fmg_v_cycles": P.iterations, "fmg_levels": L - 1,
            "fmg_residual_reduction": (P.res_history[-1] / P.res_history[0]) if P.res_history and P.res_history[0] else None}


def helmholtz27_cycle(ops, L):
    """BASELINE configs[3]'s operator on one block: -div(a grad u) - k^2 u as a 27-entry stencil field on every level, Jacobi V(3,3)
    cycle of the slotted program (Testing/Smoothers/Jac.exa4 with `Laplace` a stencil field, as Testing/SISC/3D_VarCoeff.exa4 has
    it), 2^L cells per dimension, coarsest level 4^3 cells; the cycle replayed from a hipGraph."""
    import torch

    from exastencils_amd.solver import ConfigL3, SolverFromL3

    need = 60 * 8.0 * float((1 << L) + 3) ** 3 * 8.0 / 7.0      # 27 coefficient planes (twice while they are re-laid out), 2 + 1 + 1 + 2 field arrays, all levels
    free = torch.cuda.mem_get_info(ops.device)[0]
    if need > 0.9 * free:
        return {"helmholtz27_skipped": "needs %.0f GB, %.0f GB free" % (need / 1e9, free / 1e9)}
    cfg = ConfigL3(nd=3, min_level=1, max_level=L - 1, frag_len=(2, 2, 2), smoother="jacobi", omega=0.8, stencil="helmholtz27",
                   restrict_scale=1.0, tol=1e-8, cg_max=512, bc_fn=0, sol_fn=9, coef_fn=7, kappa=10.0, ksq=2.0, rhs_from_solution=True,
                   fused_coarse=True,
                   # LayoutTransformations { transform LaplaceCoeff with [x, y, z, i] => [i, x, y, z] }: the 27 entries of a point contiguous
                   coef_entry_fastest=True,
                   # temporal blocking on the coefficient stream (csrc/kernels_sf27pair.hip; same bits as the separate loops): per level 2 + (1 +
                   # residual) before and 2 + 1 after the coarse-grid correction -- four passes over the coefficients instead of seven
                   temporal_blocking=True, fused_smooth_residual=True)
    P = SolverFromL3(cfg, ops)
    P.setup()
    r0 = P._residual_and_norm(cfg.max_level)
    P.capture()
    run = P._graphs["cycle"].replay
    for _ in range(2):
        run()
    n = 4
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    r1 = P._residual_and_norm(cfg.max_level)
    # compulsory bytes per cycle and level above the coarsest.  PER PASS (what the kernels of this cycle can at best move): the four
    # passes over the coefficients -- pair of steps 240 B per point (216 coefficients + u + rhs + result), step + residual 248, pair 240,
    # single step 240 -- plus restriction (8 B per point + 8 B per coarse point), zeroing the coarse solution, prolongation + correction
    # (16 B per point + 8 B per coarse point).  PER UPDATE (SURVEY.md 8d's accounting: 24 + 8 * 27 B for each of the 7 loops): a throughput
    # statement, above the per-pass figure because two loops share a pass.
    comp_pass = comp_update = 0.0
    for l in range(cfg.min_level + 1, cfg.max_level + 1):
        lb, le = P.domain.loop_bounds(P.Solution[l].layout)
        cb, ce = P.domain.loop_bounds(P.Solution[l - 1].layout)
        p = float((le[0] - lb[0]) * (le[1] - lb[1]) * (le[2] - lb[2]))
        c = float((ce[0] - cb[0]) * (ce[1] - cb[1]) * (ce[2] - cb[2]))
        transfers = (8.0 * p + 8.0 * c) + 8.0 * c + (16.0 * p + 8.0 * c)
        comp_pass += (240.0 + 248.0 + 240.0 + 240.0) * p + transfers
        comp_update += 7 * 240.0 * p + transfers
    return {"helmholtz27_vcycle_ms": ms, "helmholtz27_levels": cfg.max_level - cfg.min_level + 1,
            "helmholtz27_coefficient_layout": "entry-fastest (LayoutTransformations [x,y,z,i] => [i,x,y,z])",
            "helmholtz27_passes_per_level": "pair, step + residual | pair, step (7 loops of 240 B per point in 4 passes over the coefficients)",
            "helmholtz27_residual_reduction_6_cycles": r1 / r0 if r0 else None,
            "helmholtz27_vcycle_frac": comp_pass / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,                  # per-pass bytes: a roofline fraction
            "helmholtz27_vcycle_lu_equivalent_frac": comp_update / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}  # per-update bytes: throughput, not a fraction


if __name__ == "__main__":
    main()
