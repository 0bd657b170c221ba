"""Host-side mirror of the generated multigrid programs: the functions the reference generator emits
into User/User_<fn>.cpp (mgCycle_<lvl>, Solve_<lvl>, ResNorm_<lvl>, ...), written against the kernel
layer (`ops`) and `communicate` / `apply bc` exactly where the ExaSlang-4 programs have them.

  SolverFromL4  Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4 (= Examples/Poisson/3D_FD_Poisson_fromL4.exa4)
                and Examples/Poisson/2D_FD_Poisson_fromL4.exa4: RBGS V(3,3), CG coarse solve.
  SolverFromL3  Testing/Smoothers/{Jac,RBGS}.exa4, Testing/CommBasic/PureMPI.exa4,
                Testing/SISC/3D_{Const,Var}Coeff.exa4, Testing/FMG/3D_*.exa4: slotted Jacobi (or RBGS),
                UpResidual / Restriction / Correction / VCycle / FMG functions.

Every loop is a libexamg kernel launch on the current HIP stream; reductions stay on the device until
the program needs the number on the host (`repeat until` conditions, prints).  On a single block the
coarse-grid CG runs as one persistent kernel (examg_cg_coarse), which makes a whole V-cycle free of
host round trips, so it can be captured into a hipGraph (`capture_cycle`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import threading as _threading

from .comm import Communicator
from .domain import RectDomain
from .field import (FN_POLY3D, FN_ZERO, Field, Stencil, laplace_fd, laplace_unit, stencil_field_offsets)
from .layout import FieldLayout

# stream captures of one process run one at a time (a host with several blocks per process -- one thread each -- captures per block;
# nothing executes during a capture, so no block waits for another inside it)
_CAPTURE_LOCK = _threading.Lock()

APPLY, RESIDUAL, SMOOTH = 0, 1, 2


def reduced_prec(x: float) -> str:
    """printWithReducedPrec (Compiler/src/exastencils/util/ir/IR_ResolvePrintWithReducedPrec.scala:50-71):
    std::cout at precision 4, fewer digits towards the 1e-12 zero threshold (config/Knowledge.scala:293-305)."""
    if x <= 1.0e-12:
        return "EFFECTIVELY ZERO"
    if x <= 1.0e-11:
        prec = 1
    elif x <= 9.999999999999999e-11:
        prec = 2
    elif x <= 9.999999999999999e-10:
        prec = 3
    else:
        prec = 4
    return "%.*g" % (prec, x)


class _Program:
    """Shared plumbing: domain, kernel layer, communicator, the three statement kinds."""

    def __init__(self, nd: int, min_level: int, max_level: int, frag_len, ops, domain: Optional[RectDomain], comm):
        if ops is None:
            from .ops import HipOps

            ops = HipOps()          # raises without libexamg.so / GPU: no fallback
        self.ops = ops
        self.domain = domain or RectDomain(nd, (1, 1, 1), 0, frag_len)
        self.comm = comm or Communicator(self.domain, ops)
        self.nd, self.min_level, self.max_level = nd, min_level, max_level
        self.levels = list(range(min_level, max_level + 1))
        self.log: List[str] = []
        self.res_history: List[float] = []
        self.err_history: List[float] = []
        self.cg_iters: List[int] = []
        self._graphs: Dict = {}

    # `loop over <field>` bounds
    def bounds(self, f: Field, reduction: bool = False):
        return self.domain.loop_bounds(f.layout, reduction)

    # `communicate [dup|ghost of] <field>`  ->  exch<Field>_<level>(slot)
    def communicate(self, f: Field, slot: Optional[int] = None, what: str = "all", axis_only: bool = False):
        """axis_only: the loop that follows reads face ghosts only (5/7-point stencil) -- a communicator created with
        concurrent_ghost_axes then sends all axes in one batch (exastencils_amd/comm.py)."""
        self.comm.exchange(f, slot, what, axis_only)

    @staticmethod
    def _faces_only(A: Stencil) -> bool:
        return all(sum(1 for c in o if c != 0) <= 1 for o in A.offsets)

    # `apply bc to <field>`  ->  applyBCs<Field>_<level>(slot)
    def apply_bc(self, f: Field, slot: Optional[int] = None):
        if f.bc_fn is None:
            return
        mask = self.domain.face_mask()
        if mask:
            self.ops.apply_dirichlet(f.lc, f.data(slot), self.domain.geom(f.level), f.bc_fn, f.bc_params, mask)

    def _dot_host(self, x: Field, y: Field, over: Field, xslot=None, yslot=None) -> float:
        b, e = self.bounds(over, reduction=True)
        t = self.ops.dot(x.lc, x.data(xslot), y.lc, y.data(yslot), b, e)
        return self.comm.reduce_value(t, "sum")

    def _single_block(self) -> bool:
        """No neighbour across any face (a periodic dimension makes a lone block its own neighbour: the paths with exchanges)."""
        return self.domain.world_size == 1 and not any(self.domain.periodic)

    def _report_cg_limit(self):
        """One-call coarse solves count on the device how often the CG loop ran out of iterations (info[3]); the message the generated
        function prints at that point is appended to the log when the host next looks (end of Solve)."""
        info = getattr(self, "_cg_info", None)
        if info is None or not getattr(self.cfg, "fused_coarse", False):
            return
        n = int(self.ops.to_host(info)[3])
        for _ in range(n - getattr(self, "_cg_limit_seen", 0)):
            self.log.append("Maximum number of cgs iterations (%d) was exceeded" % self.cfg.cg_max)
        self._cg_limit_seen = n


# =================================================================================================
# Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4
# =================================================================================================


@dataclass
class ConfigL4:
    nd: int = 3
    min_level: int = 2
    max_level: int = 9
    frag_len: Tuple[int, int, int] = (1, 1, 1)
    omega: float = 0.8
    n_smooth: int = 3
    tol: float = 1.0e-6
    max_it: int = 100
    cg_max: int = 128
    cg_tol: float = 0.001
    bc_fn: int = FN_POLY3D
    rhs_fn: Optional[int] = None
    sol_fn: Optional[int] = None
    align: int = 0
    fused_coarse: bool = True     # single block: mgCycle@coarsest as one persistent kernel
    fused_rbgs: bool = False      # one-pass red-black sweep (out of place, pointer swap)
    # single block + fused_rbgs: three plain sweeps in a row -- six colour loops -- run as TWO passes of three colour loops each
    # (examg_rbgs_colours3) where the kernel layer has the one-pass form for the level (examg_three_stage_eligible)
    fused_rbgs3: bool = True
    fused_residual_restrict: bool = False   # single block: `Residual = ...` + restriction as one pass, fine residual not stored
    # single block + fused_rbgs: the correction loop is folded into the first post-smoothing sweep on levels with at least this
    # many points (0 = never).  MI355X: 512^3 1.16 -> 0.87 ms; below ~5*10^7 points the fields sit in the Infinity Cache, the
    # separate correction loop is cheap and the fold does not pay (256^3: 0.141 -> 0.150 ms)
    fused_prolong_min_points: int = 0
    # single block + fused_rbgs: `Solution@coarser = 0` is not executed where the first pre-smoothing sweep of that level is a
    # one-pass sweep: it takes the zero field as a constant (examg_rbgs_sweep_fused_zero: no zeroing loop, 16 instead of 24 B per point)
    fused_zero_start: bool = False
    # Solve@finest: `Residual = RHS - A * Solution` + ResNorm as one pass that does not store the residual (nothing reads
    # Residual@finest between two of the loop's updates: every path of mgCycle starts with its own residual pass); also on blocks
    # with neighbours (the exchange first, then one pass over the reduction's box, then the all-reduce)
    fused_residual_norm: bool = False
    overlap_transfers: bool = True            # blocks > 1: residual / restriction as interior + shell around their halo exchange
    agglomerate_level: Optional[int] = None   # blocks > 1: levels <= this are solved redundantly on every rank (see _agg_cycle)
    agglomerate_extra_levels: int = 0         # the gathered hierarchy coarsens this many levels below min_level
    graph_agglomerated: bool = True           # GPU: the cycle on the gathered levels (no communication) replays from a hipGraph
    deep_halo: bool = False                   # blocks > 1: Solution with TWO ghost layers and RHS with one (`ghostLayers` of the layouts): the one-pass sweeps then need one exchange and no shell (smoothers.deep_halo_boxes)


class SolverFromL4(_Program):
    def __init__(self, cfg: ConfigL4, ops=None, domain: Optional[RectDomain] = None, comm=None):
        super().__init__(cfg.nd, cfg.min_level, cfg.max_level, cfg.frag_len, ops, domain, comm)
        self.cfg = cfg
        nd, dom, ops = cfg.nd, self.domain, self.ops
        lo, hi = cfg.min_level, cfg.max_level
        self.Solution: Dict[int, Field] = {}
        self.RHS: Dict[int, Field] = {}
        self.Residual: Dict[int, Field] = {}
        self.Laplace: Dict[int, Stencil] = {}
        self._sol_alt: Dict[int, object] = {}
        self._three_ok: Dict[int, bool] = {}
        self._sweep_tmp: Dict[int, Field] = {}
        for l in self.levels:
            nc = dom.ncells(l)
            with_comm = FieldLayout.node(nd, nc, 1, True, True, cfg.align)     # Layout NodeWithComm (...exa4:13-16)
            no_ghost = FieldLayout.node(nd, nc, 0, True, False, cfg.align)     # Layout NodeNoGhost  (...exa4:18-21)
            # (the levels that run distributed sweeps: above the coarsest and above the agglomerated ones, whose gather buffers assume one ghost layer)
            deep = cfg.deep_halo and not self._single_block() and l != lo and (cfg.agglomerate_level is None or l > cfg.agglomerate_level)
            sol_lay = FieldLayout.node(nd, nc, 2, True, True, cfg.align) if deep else with_comm
            rhs_lay = with_comm if deep else no_ghost
            self.Solution[l] = Field("Solution", l, sol_lay, ops, 1, cfg.bc_fn if l == hi else FN_ZERO)   # :24-25
            self.RHS[l] = Field("RHS", l, rhs_lay, ops, 1, None)                                           # :27
            self.Residual[l] = Field("Residual", l, no_ghost if l == lo else with_comm, ops, 1, FN_ZERO)    # :29-30
            self.Laplace[l] = laplace_fd(nd, dom.h(l), "mp", "pow")                                         # :39-47
            if cfg.fused_rbgs and l != lo:
                self._sol_alt[l] = ops.new_array(sol_lay.size)
        nc = dom.ncells(lo)
        self.cgTmp0 = Field("cgTmp0", lo, FieldLayout.node(nd, nc, 1, True, True, cfg.align), ops, 1, FN_ZERO)   # :32
        self.cgTmp1 = Field("cgTmp1", lo, FieldLayout.node(nd, nc, 0, True, False, cfg.align), ops, 1, None)     # :33
        self._cg_info = ops.new_array(4)
        self._agg = None
        k = cfg.agglomerate_level
        if k is not None and (dom.world_size > 1 or any(dom.periodic)) and lo <= k < hi:
            # The coarse levels of a decomposed hierarchy are latency-bound (a 64^3 block per GPU and six exchanges per
            # sweep): from level k down every rank gathers the whole level (one all-gather of the restricted right-hand
            # side, a few MB over xGMI) and runs the remaining cycle on it as ONE block -- fused sweeps, persistent CG
            # kernel, no further communication -- then keeps its own part of the correction.  Same statements as the
            # distributed cycle; only the order in which the coarse-grid CG's reductions add differs.
            whole = RectDomain(nd, (1, 1, 1), 0, tuple(dom.num_blocks[d] * dom.frag_len[d] for d in range(3)), dom.lo, dom.hi)
            # The whole coarsest level is `blocks` times larger than one block's (16 x 32 x 64 points on 8 GPUs): the gathered
            # hierarchy may go on coarsening (agglomerate_extra_levels) so that the CG again sees a grid of a few hundred
            # points -- the cycle then has more levels than on one GPU, as a larger problem should.
            acfg = ConfigL4(nd=nd, min_level=max(0, lo - cfg.agglomerate_extra_levels), max_level=k, frag_len=whole.frag_len, omega=cfg.omega, n_smooth=cfg.n_smooth,
                            cg_max=cfg.cg_max, cg_tol=cfg.cg_tol, bc_fn=FN_ZERO, align=cfg.align, fused_coarse=True,
                            fused_rbgs=cfg.fused_rbgs)
            self._agg = SolverFromL4(acfg, ops, whole, Communicator(whole, ops))
            self._agg.setup()
            if cfg.graph_agglomerated and hasattr(ops, "torch") and getattr(getattr(ops, "device", None), "type", "cpu") != "cpu":
                # warm-up and capture run the cycle once for real: with every field zero the coarse CG would start from a zero
                # residual (alpha = 0 / 0) and leave NaNs and a spurious "iteration limit" count behind -- give it a right-hand
                # side, then put the state back (arrays are zeroed in place: the captured graph stays valid)
                AF = self._agg.RHS[k]
                ab, ae = self._agg.bounds(AF)
                ops.set(AF.lc, AF.data(), 1.0, ab, ae)
                self._agg.capture_cycle()
                self._agg.reset()
                self._agg._cg_info.zero_()
                self._agg._cg_limit_seen = 0
            nc = dom.ncells(k)
            n_own = 1
            n_halo = 1
            for d in range(nd):
                n_own *= nc[d] + 1
                n_halo *= nc[d] + 3
            self._agg_send = ops.new_array(n_own)
            self._agg_recv_all = ops.new_array(n_own * dom.world_size)     # one array: consecutive pieces, one per rank
            self._agg_recv = [self._agg_recv_all[r * n_own:(r + 1) * n_own] for r in range(dom.world_size)]
            self._agg_back = ops.new_array(n_halo)

    def _agg_cycle(self, k: int):
        dom, ops, A = self.domain, self.ops, self._agg
        nd, nc = dom.nd, dom.ncells(k)
        F, S = self.RHS[k], self.Solution[k]
        own_b = [0, 0, 0]
        own_e = [nc[d] + 1 if d < nd else 1 for d in range(3)]
        ops.pack(F.lc, F.data(), self._agg_send, own_b, own_e)
        self.comm.all_gather(self._agg_recv, self._agg_send)
        AF, AS = A.RHS[k], A.Solution[k]
        for r in range(dom.world_size):
            pos = RectDomain(nd, dom.num_blocks, r, dom.frag_len).pos
            gb = [pos[d] * nc[d] if d < nd else 0 for d in range(3)]
            ge = [gb[d] + nc[d] + 1 if d < nd else 1 for d in range(3)]
            ops.unpack(AF.lc, AF.data(), self._agg_recv[r], gb, ge)
        b, e = A.bounds(AS)
        ops.set(AS.lc, AS.data(), 0.0, b, e)
        A.apply_bc(AS)
        capturing = hasattr(ops, "torch") and ops.torch.cuda.is_available() and ops.torch.cuda.is_current_stream_capturing()
        if "cycle" in A._graphs and not capturing:
            A.replay_cycle()      # the gathered levels are one block without communication: ~60 launches as one graph replay
        else:
            A.mgCycle(k)
        # own part of the correction, ghost layers included (the neighbours' inner planes come straight from the whole level)
        lb = [-S.layout.ghost[d] if d < nd else 0 for d in range(3)]
        le = [nc[d] + 1 + S.layout.ghost[d] if d < nd else 1 for d in range(3)]
        gb = [dom.pos[d] * nc[d] + lb[d] if d < nd else 0 for d in range(3)]
        ge = [dom.pos[d] * nc[d] + le[d] if d < nd else 1 for d in range(3)]
        ops.pack(AS.lc, AS.data(), self._agg_back, gb, ge)
        ops.unpack(S.lc, S.data(), self._agg_back, lb, le)

    # Function ResNorm@(coarsest and finest) : Real  (...exa4:113-119)
    def ResNorm(self, l: int) -> float:
        return math.sqrt(self._dot_host(self.Residual[l], self.Residual[l], self.Residual[l]))

    def _residual_and_norm(self, l: int) -> float:
        """`Residual = RHS - A * Solution` (statement of Solve@finest) followed by ResNorm()."""
        cfg = self.cfg
        if not (cfg.fused_residual_norm and hasattr(self.ops, "residual_norm2")):
            self._update_residual(l)
            return self.ResNorm(l)
        # nothing reads Residual@finest before the cycle writes it again (its own residual pass comes first on every path of
        # mgCycle, with or without neighbours): the squares are summed where the residual would be stored, over the reduction's
        # box (duplicate planes at interior faces count once), then all-reduced
        S, R, A = self.Solution[l], self.Residual[l], self.Laplace[l]
        self.communicate(S, axis_only=self._faces_only(A))
        b, e = self.bounds(R, reduction=True)
        t = self.ops.residual_norm2(S.lc, S.data(), self.RHS[l].lc, self.RHS[l].data(), A, b, e, R.lc, R.data())
        return math.sqrt(self.comm.reduce_value(t, "sum"))

    def _deep_residual_restrict(self, l: int) -> bool:
        """`communicate Solution; Residual = RHS - A * Solution; communicate Residual; RHS@coarser = Restriction * Residual` on a block with
        neighbours whose layouts carry two ghost layers of Solution and one of RHS: the residual on the neighbour's first plane -- what
        `communicate Residual` would bring -- is evaluated here from the same bits, so the one-pass kernel runs on the WHOLE coarse box
        (its fine footprint reaches one point across every interior face) after one exchange; no shell, no residual array."""
        S, F, R, Fc, A = self.Solution[l], self.RHS[l], self.Residual[l], self.RHS[l - 1], self.Laplace[l]
        dom = self.domain
        faces = [(d, side) for d in range(dom.nd) for side in (-1, 1) if dom.neighbor(d, side) is not None]
        if (not self.cfg.deep_halo or not faces or any(S.layout.ghost[d] < 2 or F.layout.ghost[d] < 1 for d, _ in faces) or
                not hasattr(self.ops, "residual_restrict_one_pass")):
            return False
        fb, fe = [list(x) for x in self.bounds(R)]
        for d, side in faces:
            if side < 0:
                fb[d] -= 1
            else:
                fe[d] += 1
        b, e = self.bounds(Fc)
        if not self.ops.residual_restrict_one_pass(S.lc, F.lc, A, Fc.lc, fb, fe, b, e):
            return False
        self.communicate(S)          # axis by axis: the restriction's footprint reads edge and corner ghosts
        self.ops.residual_restrict(S.lc, S.data(), F.lc, F.data(), R.lc, R.data(), A, Fc.lc, Fc.data(), 1.0, fb, fe, b, e)
        return True

    def _rhs_ghosts(self, l: int):
        """deep_halo: the right-hand side of level l changed -- its ghost layer, which the first stage of the one-pass sweeps reads on the
        neighbour's first plane, follows (one exchange per level and cycle)."""
        F = self.RHS[l]
        if self.cfg.deep_halo and not self._single_block() and F.layout.communicates_ghost and max(F.layout.ghost) > 0:
            self.communicate(F, None, "ghost")

    def _update_residual(self, l: int):
        S, R = self.Solution[l], self.Residual[l]
        b, e = self.bounds(R)
        A = self.Laplace[l]

        def loop(bb, ee):
            self.ops.stencil_op(RESIDUAL, S.lc, S.data(), self.RHS[l].lc, self.RHS[l].data(), R.lc, R.data(), A, 0.0, -1, bb, ee)

        if self.cfg.overlap_transfers and not self._single_block() and self._faces_only(A):
            # `communicate Solution` = duplicate layers (in sequence: the loop reads them) + ghost layers (overlapped)
            from .smoothers import overlapped_loop

            self.communicate(S, None, "dup")
            overlapped_loop(self.ops, self.domain, b, e, lambda: self.communicate(S, None, "ghost", axis_only=True), loop)
        else:
            self.communicate(S, axis_only=self._faces_only(A))
            loop(b, e)
        self.apply_bc(R)

    # Function Application (...exa4:251-277): init part
    def setup(self):
        cfg, hi = self.cfg, self.cfg.max_level
        if cfg.rhs_fn is not None:      # InitRHS@finest (Examples/Poisson/2D_FD_Poisson_fromL4.exa4:231-235)
            f = self.RHS[hi]
            b, e = self.bounds(f)
            self.ops.fill_fn(f.lc, f.data(), self.domain.geom(hi), cfg.rhs_fn, (), b, e)
        self._rhs_ghosts(hi)
        for l in self.levels:           # finest: the program's `apply bc`; coarser: homogeneous values the cycle relies on (mgCycle: static_bc)
            self.apply_bc(self.Solution[l])
        self._init_alt_shells()

    def _init_alt_shells(self):
        """The second Solution array of the fused red-black sweep carries the same Dirichlet shell as the field."""
        mask = self.domain.face_mask()
        for l, alt in self._sol_alt.items():
            S = self.Solution[l]
            if S.bc_fn is not None and mask:
                self.ops.apply_dirichlet(S.lc, alt, self.domain.geom(l), S.bc_fn, S.bc_params, mask)

    def reset(self):
        """Back to the state after initFieldsWithZero + setup(): arrays are zeroed in place (device pointers, and with
        them a captured graph, stay valid)."""
        for l in self.levels:
            for t in self.Solution[l].slots + self.RHS[l].slots + self.Residual[l].slots:
                t.zero_()
            if l in self._sol_alt:
                self._sol_alt[l].zero_()
        for t in self.cgTmp0.slots + self.cgTmp1.slots:
            t.zero_()
        self.log, self.res_history, self.err_history, self.cg_iters = [], [], [], []
        self.setup()

    # Function Solve@finest (...exa4:121-150)
    def Solve(self, use_graph: bool = False) -> int:
        cfg, hi = self.cfg, self.cfg.max_level
        initRes = self._residual_and_norm(hi)
        curRes = initRes
        self.res_history.append(initRes)
        self.log.append(reduced_prec(initRes))
        curIt = 0
        while not (curIt >= cfg.max_it or curRes <= cfg.tol * initRes):
            curIt += 1
            if use_graph:
                self.replay_cycle()
            else:
                self.mgCycle(hi)
            if cfg.sol_fn is not None:  # PrintError@finest (2-D example :83-95)
                S = self.Solution[hi]
                b, e = self.bounds(S)
                t = self.ops.max_err_fn(S.lc, S.data(), self.domain.geom(hi), cfg.sol_fn, (), b, e)
                err = self.comm.reduce_value(t, "max")
                self.err_history.append(err)
                self.log.append(reduced_prec(err))
            curRes = self._residual_and_norm(hi)
            self.res_history.append(curRes)
            self.log.append(reduced_prec(curRes))
        self.iterations = curIt
        self._report_cg_limit()
        return curIt

    # repeat 3 times { color with { (i0+i1+i2) % 2, communicate; loop over Solution {...}; apply bc } }  (:204-213)
    def _one_pass_sweep(self, l: int) -> bool:
        """Does the kernel layer run the red-black sweep of level l as one pass (examg_two_stage_eligible)?"""
        if not hasattr(self, "_one_pass"):
            self._one_pass = {}
        if l not in self._one_pass:
            S = self.Solution[l]
            b, e = self.bounds(S)
            self._one_pass[l] = not hasattr(self.ops, "two_stage_eligible") or \
                self.ops.two_stage_eligible(S.lc, self.RHS[l].lc, self.Laplace[l], b, e, b, e)
        return self._one_pass[l]

    def _starts_from_zero(self, l: int) -> bool:
        """Is `Solution@l = 0` (in mgCycle@(l+1)) left to the first pre-smoothing sweep of level l?"""
        cfg = self.cfg
        return bool(cfg.fused_zero_start and cfg.fused_rbgs and self._single_block() and l != cfg.min_level and l < cfg.max_level and
                    cfg.n_smooth >= 1 and self.Solution[l].bc_fn == FN_ZERO and
                    not (self._agg is not None and l == cfg.agglomerate_level) and self._one_pass_sweep(l))

    def _folds_prolongation(self, l: int) -> bool:
        """Is `Solution@l += Prolongation * Solution@(l-1)` folded into the first post-smoothing sweep of level l?"""
        cfg = self.cfg
        if not (cfg.fused_prolong_min_points > 0 and cfg.fused_rbgs and self._single_block() and cfg.n_smooth >= 1):
            return False
        if cfg.n_smooth % 3 == 0 and self._three_colour_passes(l):
            # where three sweeps run as two passes of three colour loops, the separate correction loop + those passes beat the folded
            # first sweep + a pass per remaining sweep (tools/ab_post.py, 512^3, one process: cycle 4.875 -> 4.784 ms; same bits)
            return False
        S = self.Solution[l]
        b, e = self.bounds(S)
        # the fold pays where the pass is bandwidth-bound and large (a read-modify-write loop less) and where the level is launch-bound
        # (rows shorter than 64 points: one kernel less, csrc/kernels_small.hip); in between the separate correction is faster
        if (e[0] - b[0]) * (e[1] - b[1]) * (e[2] - b[2]) < cfg.fused_prolong_min_points and (e[0] - b[0]) >= 64:
            return False
        # the kernel layer decides whether its one-pass kernel takes these arguments; without it the entry point runs the plain
        # loops on a copy, which costs more than the separate calls
        return self._one_pass_sweep(l)

    def _three_colour_passes(self, l: int) -> bool:
        """Does the kernel layer run three colour loops of level l in one pass (examg_rbgs_colours3 with its one-pass kernel)?"""
        if not (self.cfg.fused_rbgs3 and hasattr(self.ops, "three_stage_eligible")):
            return False
        hit = self._three_ok.get(l)
        if hit is None:
            S, F = self.Solution[l], self.RHS[l]
            b, e = self.bounds(S)
            hit = self._three_ok[l] = bool(self.ops.three_stage_eligible(S.lc, F.lc, self.Laplace[l], list(b), list(e)))
        return hit

    def _smooth(self, l: int, correction_from: Optional[Field] = None, zero_input: bool = False):
        S, F, A = self.Solution[l], self.RHS[l], self.Laplace[l]
        w = self.cfg.omega / A.diag           # `0.8 / diag(Laplace)`, folded to a literal by the generator
        b, e = self.bounds(S)
        if self.cfg.fused_rbgs and self._single_block():
            # one pass per sweep; the exchange is empty on one block.  The sweep writes the loop's box only and
            # `apply bc` would re-write the same position-only Dirichlet values each time: both arrays get their
            # shell once, the per-colour `apply bc` calls become no-ops and are dropped -- no bit changes
            # (_init_alt_shells, called from setup())
            if correction_from is None and not self._one_pass_sweep(l):
                # short rows (coarse levels): the entry point would copy the field and run the two colour loops on the copy --
                # the loops in place are the same statements with one launch less per sweep (launch-bound levels)
                for _ in range(self.cfg.n_smooth):
                    for colour in (0, 1):
                        self.ops.stencil_op(SMOOTH, S.lc, S.data(), F.lc, F.data(), S.lc, S.data(), A, w, colour, b, e)
                return
            it = 0
            while it < self.cfg.n_smooth:
                alt = self._sol_alt[l]
                plain = not (it == 0 and (correction_from is not None or zero_input))
                if plain and self.cfg.n_smooth - it >= 3 and self._three_colour_passes(l):
                    # three sweeps = six colour loops 0 1 0 1 0 1 = two passes of three (the second starts with colour 1): 2 x 24 B per point
                    # instead of 3 x 24, the same arithmetic in the same order
                    for first in (0, 1):
                        alt = self._sol_alt[l]
                        self.ops.rbgs_colours3(S.lc, S.data(), alt, F.lc, F.data(), A, w, first, b, e)
                        self._sol_alt[l], S.slots[0] = S.slots[0], alt
                    it += 3
                    continue
                if it == 0 and correction_from is not None:
                    # the correction loop, whose box is the sweep's box, rides along (examg_rbgs_sweep_fused_prolong)
                    Sc = correction_from
                    self.ops.rbgs_sweep_fused_prolong(S.lc, S.data(), alt, F.lc, F.data(), A, w, 0, b, e, Sc.lc, Sc.data())
                elif it == 0 and zero_input:
                    self.ops.rbgs_sweep_fused_zero(S.lc, alt, F.lc, F.data(), A, w, 0, b, e)
                else:
                    self.ops.rbgs_sweep_fused(S.lc, S.data(), alt, F.lc, F.data(), A, w, 0, b, e)
                self._sol_alt[l], S.slots[0] = S.slots[0], alt
                it += 1
            return
        assert correction_from is None and not zero_input
        if self.cfg.fused_rbgs:
            # blocks with neighbours: fused deep interior + two-point shell with its two exchanges on a side stream
            # (exastencils_amd/smoothers.py: rbgs_sweep); the three arrays carry the Dirichlet planes of the physical faces
            from .smoothers import rbgs_sweep

            tmp = self._sweep_tmp.get(l)
            if tmp is None:
                tmp = self._sweep_tmp[l] = Field("SolutionSweepTmp", l, S.layout, self.ops, 1, S.bc_fn, S.bc_params)
                self.apply_bc(tmp)      # once: the values are functions of the position, and no loop writes those planes
            for _ in range(self.cfg.n_smooth):
                self._sol_alt[l] = rbgs_sweep(self.ops, self.comm, self.domain, S, F, A, w, self._sol_alt[l], tmp, 0, tmp_planes_valid=True)
            return
        for _ in range(self.cfg.n_smooth):
            for colour in (0, 1):
                self.communicate(S, axis_only=self._faces_only(A))
                self.ops.stencil_op(SMOOTH, S.lc, S.data(), F.lc, F.data(), S.lc, S.data(), A, w, colour, b, e)
                self.apply_bc(S)

    # Function mgCycle@(all but coarsest) (...exa4:203-249)
    def mgCycle(self, l: int, solution_is_zero: bool = False):
        if self._agg is not None and l == self.cfg.agglomerate_level:
            return self._agg_cycle(l)
        if l == self.cfg.min_level:
            return self.mgCycle_coarsest(l, solution_is_zero)
        ops = self.ops
        self._smooth(l, zero_input=solution_is_zero)
        R, Fc = self.Residual[l], self.RHS[l - 1]
        if self.cfg.fused_residual_restrict and self._single_block():
            # nothing reads Residual@l between here and its next update: residual and restriction in one pass, bit-identical
            S = self.Solution[l]
            self.communicate(S)
            fb, fe = self.bounds(R)
            b, e = self.bounds(Fc)
            ops.residual_restrict(S.lc, S.data(), self.RHS[l].lc, self.RHS[l].data(), R.lc, R.data(), self.Laplace[l], Fc.lc,
                                  Fc.data(), 1.0, fb, fe, b, e)
        elif self.cfg.fused_residual_restrict and self._deep_residual_restrict(l):
            pass      # deep halos: one exchange of Solution (two ghost layers) and the one-pass kernel on the whole coarse box
        elif (self.cfg.fused_residual_restrict and self.cfg.overlap_transfers and not self._single_block() and
              self._faces_only(self.Laplace[l]) and hasattr(self.comm, "c_residual_restrict") and
              self.comm.c_residual_restrict(self.Solution[l], self.RHS[l], R, self.Laplace[l], Fc, 1.0, *self.bounds(R), *self.bounds(Fc),
                                            axis_only=True, overlap=True)):
            # blocks with neighbours, library transport: the four statements as ONE call (examg_residual_restrict_blocks) -- one-pass
            # kernel on the coarse box shrunk by one point at interior faces, shell + both exchanges on the communicator's side stream
            self.apply_bc(R)
        else:
            self._update_residual(l)
            b, e = self.bounds(Fc)
            if self.cfg.overlap_transfers and not self._single_block():
                from .smoothers import overlapped_loop

                self.communicate(R, None, "dup")
                overlapped_loop(ops, self.domain, b, e, lambda: self.communicate(R, None, "ghost"),
                                lambda bb, ee: ops.restrict(R.lc, R.data(), Fc.lc, Fc.data(), 1.0, bb, ee))
            else:
                self.communicate(R)
                ops.restrict(R.lc, R.data(), Fc.lc, Fc.data(), 1.0, b, e)
        self._rhs_ghosts(l - 1)
        Sc, S = self.Solution[l - 1], self.Solution[l]
        b, e = self.bounds(Sc)
        # single block with the one-pass sweeps: every loop of the cycle writes inner points only and the Dirichlet values are
        # functions of the position, written once by setup() -- `apply bc` would re-write the same bits (as in _smooth)
        # (blocks with neighbours alike: exchanges bring the neighbour's copies of the same position-only values)
        static_bc = self.cfg.fused_rbgs
        # ... and a coarser level that starts with a one-pass sweep reads its zero Solution as a constant (its boundary values are 0)
        zero_start = self._starts_from_zero(l - 1) or (l - 1 == self.cfg.min_level and self._coarsest_starts_from_zero())
        if not zero_start:
            ops.set(Sc.lc, Sc.data(), 0.0, b, e)
        if not static_bc:
            self.apply_bc(Sc)
        self.mgCycle(l - 1, solution_is_zero=zero_start)
        b, e = self.bounds(S)
        if self._folds_prolongation(l):
            # `communicate Solution@coarser` and `apply bc Solution` are empty / re-write the same values on a single block
            self.communicate(Sc)
            self._smooth(l, correction_from=Sc)
            return
        side = ops.side_stream() if (self.cfg.overlap_transfers and not self._single_block() and hasattr(ops, "side_stream")) else None
        if side is not None and hasattr(self.comm, "c_prolong_add") and self.comm.c_prolong_add(Sc, S, b, e, overlap=True):
            pass      # library transport: exchange + kernel as ONE call (examg_prolong_add_blocks), same split as below
        elif side is not None:
            # `communicate Solution@coarser`: the interpolation of node values reads duplicate and inner points of the coarse
            # block only (fine node i lies between coarse nodes i/2 and (i+1)/2, both inside [0, n]) -- the ghost part of the
            # exchange runs beside the kernel, the duplicate part (which the kernel reads) before it
            torch = ops.torch
            self.communicate(Sc, None, "dup")
            main = torch.cuda.current_stream(ops.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self.communicate(Sc, None, "ghost")
            ops.prolong_add(Sc.lc, Sc.data(), S.lc, S.data(), b, e)
            main.wait_stream(side)
        else:
            self.communicate(Sc)
            ops.prolong_add(Sc.lc, Sc.data(), S.lc, S.data(), b, e)
        if not static_bc:
            self.apply_bc(S)
        self._smooth(l)

    # Function mgCycle@coarsest (...exa4:152-201)
    def _coarsest_starts_from_zero(self) -> bool:
        """Is `Solution@coarsest = 0` (in mgCycle of the level above) left to the one-kernel coarse solve (EXAMG_CG_ZERO_START)?"""
        cfg = self.cfg
        lo = cfg.min_level
        return bool(cfg.fused_zero_start and cfg.fused_coarse and cfg.fused_rbgs and self._single_block() and self.Solution[lo].bc_fn == FN_ZERO and
                    self.Laplace[lo].cfield is None and not (self._agg is not None and lo == cfg.agglomerate_level))

    def mgCycle_coarsest(self, l: int, solution_is_zero: bool = False):
        ops, A = self.ops, self.Laplace[l]
        Sol, Res, F, p_, Ap = self.Solution[l], self.Residual[l], self.RHS[l], self.cgTmp0, self.cgTmp1
        if self.cfg.fused_coarse and self._single_block():
            from .lib import CG_ZERO_START

            b, e = self.bounds(Sol)
            ops.cg_coarse(Sol.lc, Sol.data(), F.lc, F.data(), Res.lc, Res.data(), p_.lc, p_.data(), Ap.lc, Ap.data(), A,
                          self.domain.geom(l), self.domain.face_mask(), self.cfg.cg_max, self.cfg.cg_tol, b, e, self._cg_info,
                          flags=CG_ZERO_START if solution_is_zero else 0)
            return
        assert not solution_is_zero
        self._update_residual(l)
        # `alphaNom = sum(Residual^2)` of an iteration is the sum under the square root of the norm taken just before it
        # (same kernel, same data, same all-reduce): it is carried over instead of being reduced a second time -- one
        # host-visible reduction less per iteration, same bits
        rr = self._dot_host(Res, Res, Res)
        curRes = math.sqrt(rr)
        initRes = curRes
        b, e = self.bounds(p_)
        ops.axpby(Res.lc, Res.data(), p_.lc, p_.data(), 1.0, 0.0, b, e)        # cgTmp0 = Residual
        self.apply_bc(p_)
        for step in range(self.cfg.cg_max):
            self.communicate(p_, axis_only=self._faces_only(A))
            b, e = self.bounds(Ap)
            ops.stencil_op(APPLY, p_.lc, p_.data(), None, None, Ap.lc, Ap.data(), A, 0.0, -1, b, e)
            alphaNom = rr
            alphaDenom = self._dot_host(p_, Ap, p_)
            alpha = alphaNom / alphaDenom if alphaDenom != 0.0 else float("nan")
            b, e = self.bounds(Sol)
            ops.axpby(p_.lc, p_.data(), Sol.lc, Sol.data(), alpha, 1.0, b, e)  # Solution += alpha * cgTmp0
            self.apply_bc(Sol)
            b, e = self.bounds(Res)
            ops.axpby(Ap.lc, Ap.data(), Res.lc, Res.data(), -alpha, 1.0, b, e)  # Residual -= alpha * cgTmp1
            self.apply_bc(Res)
            rr = self._dot_host(Res, Res, Res)
            nextRes = math.sqrt(rr)
            if nextRes <= self.cfg.cg_tol * initRes:
                self.cg_iters.append(step + 1)
                return
            beta = (nextRes * nextRes) / (curRes * curRes)
            b, e = self.bounds(p_)
            ops.axpby(Res.lc, Res.data(), p_.lc, p_.data(), 1.0, beta, b, e)   # cgTmp0 = Residual + beta * cgTmp0
            self.apply_bc(p_)
            curRes = nextRes
        self.cg_iters.append(self.cfg.cg_max)
        self.log.append("Maximum number of cgs iterations (%d) was exceeded" % self.cfg.cg_max)

    # -- hipGraph capture of one V-cycle (single block, fused coarse solve: no host round trips) ----
    def capture_cycle(self):
        """One mgCycle@finest of a single block as a hipGraph (fused coarse solve: no host round trip).  Cycles with block
        neighbours are not captured: RCCL point-to-point groups inside a stream capture hung on this stack (ROCm 7.2, one-GPU
        self-exchange test, round 2) -- there the comm-free part, the agglomerated coarse levels, is what gets captured
        (_agg_cycle)."""
        if not self.cfg.fused_coarse:
            raise RuntimeError("graph capture needs the fused coarse solve")
        if not self._single_block():
            # blocks with neighbours: the peer-write transport orders its messages with device-side flags and counters, so the
            # exchanges are ordinary kernels that replay; every rank captures and replays the same cycle.  The coarsest levels must
            # be the agglomerated ones (their CG is the persistent kernel; a distributed CG returns to the host twice per iteration)
            if getattr(self.comm, "transport", None) != "peer":
                raise RuntimeError("graph capture of a cycle with block neighbours needs the peer-write transport")
            if self._agg is None:
                raise RuntimeError("graph capture of a cycle with block neighbours needs agglomerated coarse levels (agglomerate_level)")
        torch = self.ops.torch
        hi = self.cfg.max_level
        s = torch.cuda.Stream(self.ops.device)
        s.wait_stream(torch.cuda.current_stream(self.ops.device))
        with torch.cuda.stream(s):
            self.mgCycle(hi)      # warm-up outside capture (lazy allocations)
        torch.cuda.current_stream(self.ops.device).wait_stream(s)
        # A cycle with an ODD number of out-of-place passes on a level (three sweeps as two passes of three colour loops + the three
        # post-smoothing sweeps: five) leaves that level's solution in the other array: the cycle is then recorded twice -- from either
        # assignment of the arrays -- and the replays alternate, moving the host's pointers as the recorded cycle would have
        graphs, roles = [], [self._array_roles()]
        for _ in range(2):
            g = torch.cuda.CUDAGraph()
            # thread_local: other threads of the process (RCCL's proxy threads at N > 1) may issue HIP calls during the capture
            with _CAPTURE_LOCK, torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.mgCycle(hi)
            graphs.append(g)
            roles.append(self._array_roles())
            if roles[-1] == roles[0]:
                break
        if roles[-1] != roles[0]:
            raise RuntimeError("graph capture: the arrays of a level do not return to their roles after two cycles")
        self._graphs["cycle"] = graphs
        self._cycle_roles = roles
        self._graph_generation = getattr(self.comm, "generation", 0)
        return graphs[0]

    def _array_roles(self):
        """Which array is the solution and which the spare one, per level (the fused sweeps swap them)."""
        return [(l, self.Solution[l].slots[0].data_ptr(), self._sol_alt[l].data_ptr() if self._sol_alt.get(l) is not None else 0)
                for l in sorted(self.Solution)]

    def replay_cycle(self):
        if getattr(self.comm, "generation", 0) != getattr(self, "_graph_generation", 0):
            raise RuntimeError("the peer-write regions were re-allocated after this cycle was captured (a larger field was exchanged since): capture again")
        graphs = self._graphs["cycle"]
        if len(graphs) == 1:
            graphs[0].replay()
            return
        # two recordings, one per assignment of the arrays (capture_cycle): take the one that starts from the present assignment (cycles
        # run outside the graph in between move it too), then move the host's pointers as the recorded cycle would have
        now = self._array_roles()
        phase = 0 if now == self._cycle_roles[0] else 1
        if now != self._cycle_roles[phase]:
            raise RuntimeError("replay_cycle: the arrays are in neither of the two recorded assignments: capture again")
        graphs[phase].replay()
        for (l, s0, _a0), (_l, s1, _a1) in zip(self._cycle_roles[phase], self._cycle_roles[phase + 1]):
            if s0 != s1:
                S = self.Solution[l]
                self._sol_alt[l], S.slots[0] = S.slots[0], self._sol_alt[l]


# =================================================================================================
# Testing/Smoothers/Jac.exa4 and relatives
# =================================================================================================


@dataclass
class ConfigL3:
    nd: int = 3
    min_level: int = 0
    max_level: int = 4
    frag_len: Tuple[int, int, int] = (1, 1, 1)
    smoother: str = "jacobi"            # 'jacobi' (2 slots) | 'rbgs'
    omega: float = 0.8
    n_smooth: int = 3
    stencil: str = "unit"               # 'unit' | 'scaled' | 'varcoeff'
    restrict_scale: float = 4.0
    tol: float = 1.0e-5
    max_it: int = 100
    cg_max: int = 512
    cg_tol: float = 0.001
    bc_fn: int = FN_POLY3D
    rhs_fn: Optional[int] = None
    sol_fn: Optional[int] = None
    coef_fn: Optional[int] = None
    kappa: float = 10.0
    fmg: bool = False
    align: int = 0
    temporal_blocking: bool = False   # pairs of Jacobi steps in one pass (exastencils_amd/smoothers.py)
    fused_residual_restrict: bool = False   # single block: UpResidual + Restriction as one pass (fine residual not stored)
    fused_rbgs: bool = False          # red-black sweeps as one out-of-place pass (with neighbours: fused interior + shell)
    fused_prolong_min_points: int = 0 # single block: Correction folded into the first post-smoothing pass (pair of Jacobi steps with temporal_blocking, red-black sweep with fused_rbgs; ConfigL4)
    fused_zero_start: bool = False    # single block + fused_rbgs: SetSolution@coarser(0) left to the first pre-smoothing sweep of the coarser level (ConfigL4)
    fused_residual_norm: bool = False # single block + fused_residual_restrict: UpResidual@finest + NormResidual of the Solve loop in one pass (ConfigL4)
    fused_coarse: bool = False        # single block: VCycle_0@coarsest as one persistent kernel (examg_cg_coarse_variant)
    ksq: float = 0.0                  # stencil 'helmholtz27': shift k^2 of  -div(a grad u) - k^2 u  (config 4)
    # stencil fields: the coefficient field under `LayoutTransformations { transform LaplaceCoeff with [x, y, z, i] => [i, x, y, z] }`
    # (layoutTransformation/, Testing/LayoutTrafo/*.exa4): entries of a point contiguous -- ONE coefficient stream per sweep
    coef_entry_fastest: bool = False
    # single block + Jacobi: the last pre-smoothing step and UpResidual@current as one pass (examg_jacobi_residual: 27-entry record fields
    # share the coefficient stream between the two loops; same bits as the two statements)
    fused_smooth_residual: bool = False
    rhs_from_solution: bool = False   # RHS = A * sol_fn (discrete manufactured solution)
    # InitSolution of Testing/Opts/base.exa4:166-170: Solution@finest = (double)std::rand()/RAND_MAX, drawn by every process of the
    # reference's process grid after std::srand(mpiRank) (exastencils_amd/crand.py); None: Solution starts at zero
    init_rand_procs: Optional[Tuple[int, int, int]] = None


class SolverFromL3(_Program):
    def __init__(self, cfg: ConfigL3, ops=None, domain: Optional[RectDomain] = None, comm=None):
        super().__init__(cfg.nd, cfg.min_level, cfg.max_level, cfg.frag_len, ops, domain, comm)
        self.cfg = cfg
        nd, dom, ops = cfg.nd, self.domain, self.ops
        lo, hi = cfg.min_level, cfg.max_level
        nslots = 2 if cfg.smoother == "jacobi" else 1
        prm = (cfg.kappa,)
        self.Solution: Dict[int, Field] = {}
        self.RHS: Dict[int, Field] = {}
        self.Residual: Dict[int, Field] = {}
        self.Laplace: Dict[int, Stencil] = {}
        for l in self.levels:
            nc = dom.ncells(l)
            basic = FieldLayout.node(nd, nc, 1, True, True, cfg.align)       # BasicComm / CommFullTempBlockable
            nocomm = FieldLayout.node(nd, nc, 0, False, False, cfg.align)    # NoComm / CommPartTempBlockable / NoCommSF
            self.Solution[l] = Field("Solution", l, basic, ops, nslots, cfg.bc_fn if l == hi else FN_ZERO, prm)
            self.RHS[l] = Field("RHS", l, nocomm, ops, 1, None)
            self.Residual[l] = Field("Residual", l, basic, ops, 1, FN_ZERO)
            if cfg.stencil == "unit":
                self.Laplace[l] = laplace_unit(nd)
            elif cfg.stencil == "scaled":
                self.Laplace[l] = laplace_fd(nd, dom.h(l), "pm", "mul")
            elif cfg.stencil == "helmholtz27":   # config 4: 27-entry stencil field (examg_init_helmholtz27)
                from .field import helmholtz27_offsets

                cf = ops.new_array(27 * nocomm.size)
                b, e = dom.loop_bounds(nocomm)
                ops.init_helmholtz27(nocomm.c_struct(), cf, dom.geom(l), cfg.coef_fn, (cfg.kappa, cfg.ksq), b, e)
                self.Laplace[l] = Stencil(helmholtz27_offsets(), [], cf, nocomm)
                if cfg.coef_entry_fastest:
                    self.Laplace[l] = self.Laplace[l].entry_fastest(ops)
                    del cf
            else:   # InitLaplace@l (Testing/SISC/3D_VarCoeff.exa4:206-217)
                cf = ops.new_array((2 * nd + 1) * nocomm.size)
                b, e = dom.loop_bounds(nocomm)
                ops.init_varcoeff7(nocomm.c_struct(), cf, dom.geom(l), cfg.coef_fn, prm, b, e)
                self.Laplace[l] = Stencil(stencil_field_offsets(nd), [], cf, nocomm)
                if cfg.coef_entry_fastest:
                    self.Laplace[l] = self.Laplace[l].entry_fastest(ops)
                    del cf
        nc = dom.ncells(lo)
        self._func_dir: Dict[int, bool] = {}       # level -> its boundary planes hold SetFuncDir's values (FMG start), not the field's bc
        self._rb_alt, self._rb_tmp, self._one_pass = {}, {}, {}
        self._cg_info = ops.new_array(4)
        self.VecP = Field("VecP", lo, FieldLayout.node(nd, nc, 1, True, True, cfg.align), ops, 1, FN_ZERO)
        self.VecGradP = Field("VecGradP", lo, FieldLayout.node(nd, nc, 0, False, False, cfg.align), ops, 1, None)

    def _w(self, l: int) -> float:
        if self.cfg.stencil in ("varcoeff", "helmholtz27"):
            return self.cfg.omega                       # kernel forms (1.0 / diag) * omega per point
        return (1.0 / self.Laplace[l].diag) * self.cfg.omega

    # Function UpResidual@all
    def UpResidual(self, l: int):
        S, R, F = self.Solution[l], self.Residual[l], self.RHS[l]
        self.communicate(S, S.active)
        b, e = self.bounds(R)
        self.ops.stencil_op(RESIDUAL, S.lc, S.data(), F.lc, F.data(), R.lc, R.data(), self.Laplace[l], 0.0, -1, b, e)

    # Function NormResidual_0@(finest, coarsest) : Real
    def NormResidual(self, l: int) -> float:
        R = self.Residual[l]
        return math.sqrt(self._dot_host(R, R, R))

    def _residual_and_norm(self, l: int) -> float:
        """`UpResidual@finest ( )` followed by `NormResidual_0@finest ( )` (Function Solve)."""
        cfg, A = self.cfg, self.Laplace[l]
        if not (cfg.fused_residual_norm and cfg.fused_residual_restrict and self._single_block() and A.cfield is None and
                l != cfg.min_level and hasattr(self.ops, "residual_norm2")):
            self.UpResidual(l)
            return self.NormResidual(l)
        # nothing reads Residual@finest before the cycle's own residual pass writes it again: the squares are summed where the
        # residual would be stored (SolverFromL4._residual_and_norm)
        S, R, F = self.Solution[l], self.Residual[l], self.RHS[l]
        self.communicate(S, S.active)
        b, e = self.bounds(R, reduction=True)
        t = self.ops.residual_norm2(S.lc, S.data(), F.lc, F.data(), A, b, e, R.lc, R.data())
        return math.sqrt(self.comm.reduce_value(t, "sum"))

    # Function NormError_0@finest : Real
    def NormError(self, l: int) -> float:
        S = self.Solution[l]
        b, e = self.bounds(S, reduction=True)
        t = self.ops.max_err_fn(S.lc, S.data(), self.domain.geom(l), self.cfg.sol_fn, (self.cfg.kappa,), b, e)
        return self.comm.reduce_value(t, "max")

    # Function Smoother@((coarsest + 1) to finest)
    def _one_pass_sweep(self, l: int) -> bool:
        """Does the kernel layer run the red-black sweep of level l as one pass (examg_two_stage_eligible)?"""
        if l not in self._one_pass:
            S = self.Solution[l]
            b, e = self.bounds(S)
            self._one_pass[l] = self.nd == 3 and (not hasattr(self.ops, "two_stage_eligible") or
                                                  self.ops.two_stage_eligible(S.lc, self.RHS[l].lc, self.Laplace[l], b, e, b, e))
        return self._one_pass[l]

    def _sweep_arrays(self, l: int):
        """Second Solution array of the out-of-place red-black sweeps (+ the scratch field of the shell on blocks with neighbours)."""
        S = self.Solution[l]
        if l not in self._rb_alt:
            self._rb_alt[l] = self.ops.new_array(S.layout.size)
            if not self._single_block():
                self._rb_tmp[l] = Field("SolutionSweepTmp", l, S.layout, self.ops, 1, None)
            self._boundary_planes(l, self._rb_alt[l])
        return self._rb_alt[l]

    def _boundary_planes(self, l: int, array):
        """Write the values Solution@l currently carries on the physical faces -- SetFuncDir's while the FMG start works on the
        level, the field's bc otherwise; both are functions of the position -- into `array` (second array of the sweeps)."""
        S = self.Solution[l]
        if self._func_dir.get(l):
            self._set_func_dir(l, array)
        elif S.bc_fn is not None and self.domain.face_mask():
            self.ops.apply_dirichlet(S.lc, array, self.domain.geom(l), S.bc_fn, S.bc_params, self.domain.face_mask())

    def Smoother(self, l: int, correction_from: Optional[Field] = None, zero_input: bool = False):
        S, F, A = self.Solution[l], self.RHS[l], self.Laplace[l]
        b, e = self.bounds(S)
        if self.cfg.smoother == "jacobi":       # Testing/Smoothers/Jac.exa4:125-131
            assert correction_from is None and not zero_input
            self.communicate(S, S.active, "ghost")
            self.ops.stencil_op(SMOOTH, S.lc, S.data(S.active), F.lc, F.data(), S.lc, S.data(S.next), A, self._w(l), -1, b, e)
            S.advance()
        elif self.cfg.fused_rbgs and A.cfield is None and not (self._single_block() and not self._one_pass_sweep(l)):
            # both colour loops in one out-of-place pass; the second array carries the field's boundary planes, rewritten
            # together with them (setup, SetFuncDir / ResetBC of the FMG start).  (Single block, rows too short for the
            # one-pass kernel: the entry point would copy the field and run the colour loops on the copy -- the loops in place,
            # below, are the same statements with one launch less.)
            alt = self._sweep_arrays(l)
            if self._single_block():
                # the exchange is empty; the correction loop before the sweep / the zero field the sweep starts from ride along
                w = self._w(l)
                if correction_from is not None:
                    Sc = correction_from
                    self.ops.rbgs_sweep_fused_prolong(S.lc, S.data(), alt, F.lc, F.data(), A, w, 0, b, e, Sc.lc, Sc.data())
                elif zero_input:
                    self.ops.rbgs_sweep_fused_zero(S.lc, alt, F.lc, F.data(), A, w, 0, b, e)
                else:
                    self.ops.rbgs_sweep_fused(S.lc, S.data(), alt, F.lc, F.data(), A, w, 0, b, e)
                self._rb_alt[l], S.slots[S.active] = S.slots[S.active], alt
                return
            assert correction_from is None and not zero_input
            from .smoothers import rbgs_sweep

            self.communicate(S, S.active, "dup")       # the ghost part of `communicate Solution` is inside rbgs_sweep
            self._rb_alt[l] = rbgs_sweep(self.ops, self.comm, self.domain, S, F, A, self._w(l), alt, self._rb_tmp[l], 0)
        else:                                   # Testing/Smoothers/RBGS.exa4:125-133
            assert correction_from is None and not zero_input
            for colour in (0, 1):
                self.communicate(S, S.active)
                self.ops.stencil_op(SMOOTH, S.lc, S.data(), F.lc, F.data(), S.lc, S.data(), A, self._w(l), colour, b, e)

    # `repeat n times { Smoother@current ( ) }`
    def _folds_prolongation(self, l: int) -> bool:
        """Is Correction@l folded into the first post-smoothing pass (pair of Jacobi steps / red-black sweep)?"""
        cfg = self.cfg
        jac = cfg.temporal_blocking and cfg.smoother == "jacobi" and cfg.n_smooth >= 2
        rb = cfg.fused_rbgs and cfg.smoother == "rbgs" and cfg.n_smooth >= 1
        if not (cfg.fused_prolong_min_points > 0 and (jac or rb) and self._single_block() and self.nd == 3 and
                self.Laplace[l].cfield is None):
            return False
        S = self.Solution[l]
        b, e = self.bounds(S)
        # the fold pays where the pass is bandwidth-bound and large (a read-modify-write loop less) and where the level is launch-bound
        # (rows shorter than 64 points: one kernel less, csrc/kernels_small.hip); in between the separate correction is faster
        if (e[0] - b[0]) * (e[1] - b[1]) * (e[2] - b[2]) < cfg.fused_prolong_min_points and (e[0] - b[0]) >= 64:
            return False
        return self._one_pass_sweep(l)

    def _starts_from_zero(self, l: int) -> bool:
        """Is `SetSolution@l ( 0 )` (in VCycle@(l+1)) left to the first pre-smoothing sweep of level l?  Its boundary planes must be
        the zeros of the field's bc (not the values SetFuncDir puts there while the FMG start works ON that level)."""
        cfg = self.cfg
        return bool(cfg.fused_zero_start and cfg.fused_rbgs and cfg.smoother == "rbgs" and self._single_block() and
                    l != cfg.min_level and l < cfg.max_level and cfg.n_smooth >= 1 and self.Solution[l].bc_fn == FN_ZERO and
                    not self._func_dir.get(l) and self.Laplace[l].cfield is None and self._one_pass_sweep(l))

    def Smoothers(self, l: int, n: int, correction_from: Optional[Field] = None, zero_input: bool = False):
        cfg = self.cfg
        if not (cfg.temporal_blocking and cfg.smoother == "jacobi"):
            for i in range(n):
                self.Smoother(l, correction_from if i == 0 else None, zero_input and i == 0)
            return
        assert not zero_input
        # pairs of Smoother calls as one pass over HBM (exastencils_amd/smoothers.py), bit-identical
        from .smoothers import jacobi_pair

        S = self.Solution[l]
        if not hasattr(self, "_pair_tmp"):
            self._pair_tmp = {}
        tmp = self._pair_tmp.get(l)
        if tmp is None:
            tmp = self._pair_tmp[l] = Field("SolutionTmp", l, S.layout, self.ops, 1, S.bc_fn, S.bc_params)
            self.apply_bc(tmp)
        k = n
        while k >= 2:
            jacobi_pair(self.ops, self.comm, self.domain, S, self.RHS[l], self.Laplace[l], self._w(l), tmp,
                        correction_from=correction_from)
            correction_from = None
            k -= 2
        if k:
            self.Smoother(l)

    # Function Restriction / Correction / SetSolution
    def Restriction(self, l: int):
        R, Fc = self.Residual[l], self.RHS[l - 1]
        self.communicate(R, None, "ghost")
        b, e = self.bounds(Fc)
        self.ops.restrict(R.lc, R.data(), Fc.lc, Fc.data(), self.cfg.restrict_scale, b, e)

    def Correction(self, l: int):
        Sc, S = self.Solution[l - 1], self.Solution[l]
        self.communicate(Sc, Sc.active, "ghost")
        b, e = self.bounds(S)
        self.ops.prolong_add(Sc.lc, Sc.data(), S.lc, S.data(), b, e)

    def SetSolution(self, l: int, v: float):
        S = self.Solution[l]
        b, e = self.bounds(S)
        self.ops.set(S.lc, S.data(), v, b, e)

    # Function VCycle@((coarsest + 1) to finest)
    def VCycle(self, l: int, solution_is_zero: bool = False):
        if l == self.cfg.min_level:
            self._flush_deferred_correction()
            return self.VCycle_0(l)
        if (self.cfg.fused_smooth_residual and self.cfg.smoother == "jacobi" and self.cfg.n_smooth >= 1 and self._single_block() and
                not solution_is_zero and hasattr(self.ops, "jacobi_residual")):
            # `repeat n times { Smoother@current }` + `UpResidual@current`: the last Smoother call and the residual loop in one pass.
            # Both slots of Solution hold the same boundary values (apply_bc writes every slot), which is what the residual reads
            # around the box.
            self.Smoothers(l, self.cfg.n_smooth - 1)
            S, R, F = self.Solution[l], self.Residual[l], self.RHS[l]
            b, e = self.bounds(S)
            assert (b, e) == self.bounds(R)
            self.ops.jacobi_residual(S.lc, S.data(S.active), S.data(S.next), F.lc, F.data(), R.lc, R.data(), self.Laplace[l], self._w(l), b, e)
            S.advance()
            self.Restriction(l)
            zero_start = self._starts_from_zero(l - 1)
            if not zero_start:
                self.SetSolution(l - 1, 0.0)
            self.VCycle(l - 1, solution_is_zero=zero_start)
            self.Correction(l)
            self.Smoothers(l, self.cfg.n_smooth)
            return
        if getattr(self, "_deferred_correction", None) == l and not solution_is_zero:
            # first cycle on this level after the FMG start came up from below: its Correction in the first sweep, then ResetBC@coarser
            self._deferred_correction = None
            Sc = self.Solution[l - 1]
            self.communicate(Sc, Sc.active, "ghost")
            self.Smoother(l, correction_from=Sc)
            self.ResetBC(l - 1)
            self.Smoothers(l, self.cfg.n_smooth - 1)
        else:
            self._flush_deferred_correction()
            self.Smoothers(l, self.cfg.n_smooth, zero_input=solution_is_zero)
        if self.cfg.fused_residual_restrict and self._single_block():
            # UpResidual@current + Restriction@current: nothing reads Residual@current before UpResidual writes it again
            S, R, F, Fc = self.Solution[l], self.Residual[l], self.RHS[l], self.RHS[l - 1]
            self.communicate(S, S.active)
            fb, fe = self.bounds(R)
            b, e = self.bounds(Fc)
            self.ops.residual_restrict(S.lc, S.data(), F.lc, F.data(), R.lc, R.data(), self.Laplace[l], Fc.lc, Fc.data(),
                                       self.cfg.restrict_scale, fb, fe, b, e)
        else:
            self.UpResidual(l)
            self.Restriction(l)
        zero_start = self._starts_from_zero(l - 1)
        if not zero_start:
            self.SetSolution(l - 1, 0.0)
        self.VCycle(l - 1, solution_is_zero=zero_start)
        if self._folds_prolongation(l):
            Sc = self.Solution[l - 1]
            self.communicate(Sc, Sc.active, "ghost")      # Correction@current's exchange (empty on a single block)
            self.Smoothers(l, self.cfg.n_smooth, correction_from=Sc)
        else:
            self.Correction(l)
            self.Smoothers(l, self.cfg.n_smooth)

    # Function VCycle_0@coarsest (Testing/Smoothers/Jac.exa4:75-109)
    def VCycle_0(self, l: int):
        ops, A = self.ops, self.Laplace[l]
        S, R, P, GP = self.Solution[l], self.Residual[l], self.VecP, self.VecGradP
        if self.cfg.fused_coarse and self._single_block() and hasattr(ops, "cg_coarse"):
            # the whole function as one persistent workgroup: alpha from the squared norm, no `apply bc` in this solver
            # (include/examg.h: examg_cg_coarse_variant); reductions in the kernel's fixed order, iteration count stays on the device
            from .lib import CG_ALPHA_FROM_NORM, CG_NO_BC

            b, e = self.bounds(S)
            ops.cg_coarse(S.lc, S.data(), self.RHS[l].lc, self.RHS[l].data(), R.lc, R.data(), P.lc, P.data(), GP.lc, GP.data(), A,
                          self.domain.geom(l), self.domain.face_mask(), self.cfg.cg_max, self.cfg.cg_tol, b, e, self._cg_info,
                          flags=CG_ALPHA_FROM_NORM | CG_NO_BC)
            return
        self.UpResidual(l)
        self.communicate(R)
        res = self.NormResidual(l)
        initialRes = res
        b, e = self.bounds(P)
        ops.axpby(R.lc, R.data(), P.lc, P.data(), 1.0, 0.0, b, e)
        for step in range(self.cfg.cg_max):
            self.communicate(P)
            b, e = self.bounds(P)
            ops.stencil_op(APPLY, P.lc, P.data(), None, None, GP.lc, GP.data(), A, 0.0, -1, b, e)
            alphaDenom = self._dot_host(P, GP, P)
            alpha = (res * res) / alphaDenom if alphaDenom != 0.0 else float("nan")
            b, e = self.bounds(S)
            ops.axpby(P.lc, P.data(), S.lc, S.data(), alpha, 1.0, b, e)
            ops.axpby(GP.lc, GP.data(), R.lc, R.data(), -alpha, 1.0, b, e)
            nextRes = self.NormResidual(l)
            if nextRes <= self.cfg.cg_tol * initialRes:
                self.cg_iters.append(step + 1)
                return
            beta = (nextRes * nextRes) / (res * res)
            b, e = self.bounds(P)
            ops.axpby(R.lc, R.data(), P.lc, P.data(), 1.0, beta, b, e)
            res = nextRes
        self.cg_iters.append(self.cfg.cg_max)
        self.log.append("Maximum number of cgs iterations (%d) was exceeded" % self.cfg.cg_max)

    # -- FMG (Testing/FMG/3D_Trigonometric.exa4:189-242) -------------------------------------------
    def SetFuncDir(self, l: int):
        """`loop over Solution<s> only dup [dir] on boundary`: duplicate plane of each physical face, DLB..DRE
        tangentially (baseExt/ir/IR_LoopOverPointsInOneFragment.scala:57-70)."""
        S = self.Solution[l]
        for s_ in range(S.num_slots):
            self._set_func_dir(l, S.data(s_))
        self._func_dir[l] = True
        if l in self._rb_alt:
            self._boundary_planes(l, self._rb_alt[l])

    def _set_func_dir(self, l: int, array):
        S, dom = self.Solution[l], self.domain
        mask = dom.face_mask()
        if mask:      # every physical face in one launch (six loops in the program text)
            self.ops.fill_dup_faces(S.lc, array, dom.geom(l), self.cfg.bc_fn, (self.cfg.kappa,), mask)

    def InitRHS(self, l: int):
        F = self.RHS[l]
        b, e = self.bounds(F)
        if self.cfg.rhs_fn is not None:
            self.ops.fill_fn(F.lc, F.data(), self.domain.geom(l), self.cfg.rhs_fn, (self.cfg.kappa,), b, e)
        else:
            self.ops.set(F.lc, F.data(), 0.0, b, e)

    def ResetBC(self, l: int):
        for s_ in range(self.Solution[l].num_slots):
            self.apply_bc(self.Solution[l], s_)
        self._func_dir[l] = False
        if l in self._rb_alt:
            self._boundary_planes(l, self._rb_alt[l])

    def FMG(self, l: int):
        self.SetFuncDir(l)
        self.InitRHS(l)
        self.VCycle(l)
        if self._folds_prolongation(l + 1) and self.cfg.smoother == "rbgs":
            # Correction@(l+1) rides along with the first pre-smoothing sweep of VCycle@(l+1) (the statements in between -- ResetBC@l,
            # SetFuncDir@(l+1), InitRHS@(l+1) -- neither read nor write what the correction loop writes); ResetBC@l, which changes the
            # boundary values the correction reads on level l, waits for it (VCycle, _deferred_correction)
            self._deferred_correction = l + 1
        else:
            self.Correction(l + 1)
            self.ResetBC(l)
        if l != self.cfg.max_level - 1:
            self.FMG(l + 1)

    def _flush_deferred_correction(self):
        l = getattr(self, "_deferred_correction", None)
        if l is not None:
            self._deferred_correction = None
            self.Correction(l)
            self.ResetBC(l - 1)

    # Function Application: init part
    def setup(self):
        cfg, hi = self.cfg, self.cfg.max_level
        if cfg.rhs_from_solution:
            # RHS = A * u_exact on the finest level: the discrete solution is then sol_fn itself
            S, F = self.Solution[hi], self.RHS[hi]
            lay = S.layout
            tmp = self.ops.new_array(lay.size)
            gb = [lay.idx("GLB", d) if d < self.nd else 0 for d in range(3)]
            ge = [lay.idx("GRE", d) if d < self.nd else 1 for d in range(3)]
            self.ops.fill_fn(S.lc, tmp, self.domain.geom(hi), cfg.sol_fn, (cfg.kappa,), gb, ge)
            b, e = self.bounds(F)
            self.ops.stencil_op(APPLY, S.lc, tmp, None, None, F.lc, F.data(), self.Laplace[hi], 0.0, -1, b, e)
        elif cfg.rhs_fn is not None:
            self.InitRHS(hi)
        if cfg.init_rand_procs is not None:       # Function InitSolution (Testing/Opts/base.exa4:166-170)
            from .crand import random_start

            S = self.Solution[hi]
            random_start(self.ops, S, S.active, self.domain, cfg.init_rand_procs)
        for l in self.levels:
            for s_ in range(self.Solution[l].num_slots):
                self.apply_bc(self.Solution[l], s_)
        self.apply_bc(self.VecP)
        if cfg.fused_rbgs and cfg.smoother == "rbgs":
            # the second array of the out-of-place sweeps belongs to the set-up, not to the first cycle
            for l in self.levels[1:]:
                if self.Laplace[l].cfield is None and not (self._single_block() and not self._one_pass_sweep(l)):
                    if l in self._rb_alt:
                        self._boundary_planes(l, self._rb_alt[l])      # after reset()
                    else:
                        self._sweep_arrays(l)
        if cfg.temporal_blocking and cfg.smoother == "jacobi":
            # the scratch fields of the two-step passes are part of the set-up (a 43 GB allocation inside Solve shows up there)
            if not hasattr(self, "_pair_tmp"):
                self._pair_tmp = {}
            for l in self.levels[1:]:
                if l not in self._pair_tmp:
                    S = self.Solution[l]
                    self._pair_tmp[l] = Field("SolutionTmp", l, S.layout, self.ops, 1, S.bc_fn, S.bc_params)
                    self.apply_bc(self._pair_tmp[l])

    # -- hipGraph capture (single block, one-call coarse solve: nothing in FMG / VCycle returns to the host) --------------------
    def _pointer_state(self):
        st = []
        for l in self.levels:
            S = self.Solution[l]
            st.append((S.active, tuple(t.data_ptr() for t in S.slots), self._rb_alt[l].data_ptr() if l in self._rb_alt else 0))
        return st

    def reset(self):
        """Back to the state after initFieldsWithZero + setup(): arrays are zeroed in place (device pointers, and with them the
        captured graphs, stay valid)."""
        arrays = list(self.VecP.slots) + list(self.VecGradP.slots) + list(self._rb_alt.values())
        for l in self.levels:
            arrays += self.Solution[l].slots + self.RHS[l].slots + self.Residual[l].slots
        for f in list(self._rb_tmp.values()) + list(getattr(self, "_pair_tmp", {}).values()):
            arrays += f.slots
        for t in arrays:
            t.zero_()
        for f in getattr(self, "_pair_tmp", {}).values():
            self.apply_bc(f)
        self._func_dir = {}
        self._deferred_correction = None
        self.log, self.res_history, self.err_history, self.cg_iters = [], [], [], []
        self.setup()

    def capture(self):
        """Record the FMG start (if the program has one) and one VCycle@finest as hipGraphs -- what a compiled host issues in a few
        microseconds per launch -- and go back to the initial state; Solve(use_graph=True) replays them."""
        cfg = self.cfg
        if not (self._single_block() and cfg.fused_coarse and hasattr(self.ops, "torch")):
            raise RuntimeError("graph capture needs a single block, the one-call coarse solve and the HIP kernel layer")
        torch, dev = self.ops.torch, self.ops.device
        hi, lo = cfg.max_level, cfg.min_level

        def phases():
            out = []
            if cfg.fmg and hi > lo:
                out.append(("fmg", lambda: self.FMG(lo)))
                if self._folds_prolongation(hi) and cfg.smoother == "rbgs":
                    out.append(("cycle_first", lambda: self.VCycle(hi)))      # carries the FMG start's last Correction and ResetBC
            out.append(("cycle", lambda: self.VCycle(hi)))
            return out

        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _, fn in phases():        # warm-up outside capture (lazy allocations)
                before = self._pointer_state()
                fn()
                if self._pointer_state() != before:
                    raise RuntimeError("graph capture needs an even number of array swaps per level and phase")
        torch.cuda.current_stream(dev).wait_stream(side)
        for name, fn in phases():
            g = torch.cuda.CUDAGraph()
            with _CAPTURE_LOCK, torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
            self._graphs[name] = g
        self.reset()

    # Function Solve
    def Solve(self, use_graph: bool = False) -> int:
        cfg, hi = self.cfg, self.cfg.max_level
        resStart = self._residual_and_norm(hi)
        res = resStart
        self.res_history.append(res)
        self.log.append(reduced_prec(res))
        if cfg.fmg:
            if use_graph:
                self._graphs["fmg"].replay()
            else:
                self.FMG(cfg.min_level)
        numIt = 0
        first = cfg.fmg and use_graph and "cycle_first" in self._graphs
        if first and (res < cfg.tol * resStart or cfg.max_it <= 0):
            raise RuntimeError("Solve(use_graph=True): no cycle follows the FMG start, whose last correction the first cycle's graph carries")
        while not (res < cfg.tol * resStart or numIt >= cfg.max_it):
            numIt += 1
            if use_graph:
                self._graphs["cycle_first" if first else "cycle"].replay()
                first = False
            else:
                self.VCycle(hi)
            res = self._residual_and_norm(hi)
            self.res_history.append(res)
            if cfg.sol_fn is not None:
                err = self.NormError(hi)
                self.err_history.append(err)
                self.log.append(reduced_prec(err))
            else:
                self.log.append(reduced_prec(res))
        self._flush_deferred_correction()      # no cycle ran after the FMG start
        self._report_cg_limit()
        self.log.append(str(numIt))
        self.iterations = numIt
        return numIt
