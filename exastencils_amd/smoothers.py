"""Temporally blocked smoother steps (two Jacobi steps per pass over HBM), also across block neighbours.

Reference idea: `repeat n times with contraction` / IR_ContractingLoop (Compiler/src/exastencils/baseExt/ir/
IR_ContractingLoop.scala; Testing/PolyExpl/Jac3Dcc.exa4:27) and the `Comm*TempBlockable` layouts of the
generated-from-L3 programs.  The reference widens the ghost layers to block in time; here the ghost layers stay
one deep (the drop-in layout) and the block is split into a deep interior, which needs no halo at all for two steps,
and a two-point shell along the interior faces:

  main stream   u_out = J(J(u)) on the loop's box shrunk by 2 at interior faces; the first step is evaluated on the box
                shrunk by 1 (examg_jacobi2_boxes) -- reads no ghost value, starts immediately
  side stream   1. communicate ghost of u                      (as every Smoother call does)
                2. tmp = J(u) on the three planes next to every interior face       (thin launches)
                3. communicate ghost of tmp                    (the neighbours' first-step values)
                4. u_out = J(tmp) on the two planes next to every interior face     (thin launches)
  join          (events)

Two exchanges per two steps -- as many as two plain Smoother calls -- all of it (pack, RCCL send/recv over xGMI, unpack,
thin kernels) overlapped with the one large kernel, and bit-identical results.  This is the reference's core/boundary
split (Compiler/src/exastencils/baseExt/ir/IR_LoopOverPointsInOneFragment.scala:143-222,
experimental_splitLoopsForAsyncComm) applied to a pair of steps.  On a single block only the main-stream launch remains.
"""
from __future__ import annotations

SMOOTH = 2


def deep_halo_boxes(ops, domain, S, F, A, b, e, faces):
    """Temporal blocking across block neighbours by DEEPER HALOS instead of a recomputed shell -- the reference's own way
    (baseExt/ir/IR_ContractingLoop.scala:45-196 on a layout with `ghostLayers` >= 2): with two exchanged ghost layers of the input (and
    one of the right-hand side) the first stage runs on the loop's box grown by one point across every interior face -- the neighbour's
    first plane is computed here too, from the same bits with the same kernel -- and the second stage on the loop's own box: ONE exchange
    and ONE kernel per pass, no scratch field, nothing beside the kernel that competes with it.  Returns the first-stage box, or None
    when the layouts have no room for it (then the shell scheme below runs)."""
    lay, flay = S.layout, F.layout
    if not faces or any(lay.ghost[d] < 2 or flay.ghost[d] < 1 for d, _ in faces) or not (lay.communicates_ghost and flay.communicates_ghost):
        return None
    b1, e1 = list(b), list(e)
    for d, side in faces:
        if side < 0:
            b1[d] = b[d] - 1
        else:
            e1[d] = e[d] + 1
    if hasattr(ops, "two_stage_eligible") and not ops.two_stage_eligible(S.lc, F.lc, A, b1, e1, list(b), list(e)):
        return None
    return b1, e1


def _edges_matter(faces) -> bool:
    """More than one axis with neighbours: the first stage on a ghost plane reads edge ghosts, which only the axis-by-axis exchange fills."""
    return len({d for d, _ in faces}) > 1



def jacobi_triple(ops, comm, domain, S, F, A, w: float, tmp_field) -> bool:
    """Three applications of `Smoother@current` on field S (2 slots) in ONE pass (examg_jacobi3: temporal blocking of depth 3), on a block
    without neighbours: reads slot <active>, writes the other slot, one advance -- the slot three advances make active.  Returns False
    (nothing done) on a block with neighbours: three steps without an exchange would need three ghost layers; the caller runs a pair and
    a step there."""
    nd = domain.nd
    if any(domain.neighbor(d, side) is not None for d in range(nd) for side in (-1, 1)) or not hasattr(ops, "jacobi3"):
        return False
    b, e = domain.loop_bounds(S.layout)
    if hasattr(ops, "three_stage_eligible") and not ops.three_stage_eligible(S.lc, F.lc, A, list(b), list(e)):
        return False        # the entry point would run a step through `tmp` (with a copy of the box) and a pair: the caller's pair + step is cheaper
    axis_only = all(sum(1 for c in o if c != 0) <= 1 for o in A.offsets)
    comm.exchange(S, S.active, "ghost", axis_only)      # empty on a single block
    ops.jacobi3(S.lc, S.data(S.active), S.data(S.next), tmp_field.data(), F.lc, F.data(), A, w, b, e)
    S.advance()
    return True


def jacobi_pair(ops, comm, domain, S, F, A, w: float, tmp_field, overlap: bool = True, correction_from=None):
    """Two applications of `Smoother@current` (Testing/Smoothers/Jac.exa4:125-131) on field S (2 slots):
    reads slot <active>, leaves the result in the slot two `advance`s would make active (the same one),
    reached through one advance of the out-of-place pass.  tmp_field: scratch field of S's layout whose
    Dirichlet shell holds S's boundary values.  correction_from (single block only): the coarser Solution field whose
    prolongation `Correction@current` adds to S just before the pair -- folded into the pass (examg_jacobi2_prolong)."""
    nd = domain.nd
    b, e = domain.loop_bounds(S.layout)
    src, dst = S.active, S.next
    faces = [(d, side) for d in range(nd) for side in (-1, 1) if domain.neighbor(d, side) is not None]
    axis_only = all(sum(1 for c in o if c != 0) <= 1 for o in A.offsets)   # 5/7-point: face ghosts suffice

    def shrunk(k):
        bb, ee = list(b), list(e)
        for d, side in faces:
            if side < 0:
                bb[d] = b[d] + k
            else:
                ee[d] = e[d] - k
        return bb, ee

    def slab(d, side, k):
        """The k planes of the loop's box next to face (d, side), tangentially the whole box."""
        sb, se = list(b), list(e)
        if side < 0:
            se[d] = min(b[d] + k, e[d])
        else:
            sb[d] = max(e[d] - k, b[d])
        return sb, se

    def interior():
        b1, e1 = shrunk(1)
        b2, e2 = shrunk(2)
        if all(e2[d] > b2[d] for d in range(nd)):
            ops.jacobi2_boxes(S.lc, S.data(src), S.data(dst), tmp_field.data(), F.lc, F.data(), A, w, b1, e1, b2, e2)

    def shell():
        comm.exchange(S, src, "ghost", axis_only)
        # tmp's duplicate planes on PHYSICAL faces are read by the second step of the slabs (tangential neighbours) and
        # written by nobody: bring them over from the source slot, whatever boundary values the program put there
        # (Dirichlet function, or SetFuncDir values during an FMG start)
        lay = S.layout
        for d in range(nd):
            for side in (-1, 1):
                if domain.neighbor(d, side) is not None:
                    continue
                pb = [lay.idx("DLB", t) if t < nd else 0 for t in range(3)]
                pe = [lay.idx("DRE", t) if t < nd else 1 for t in range(3)]
                pb[d], pe[d] = (lay.idx("DLB", d), lay.idx("DLE", d)) if side < 0 else (lay.idx("DRB", d), lay.idx("DRE", d))
                ops.axpby(S.lc, S.data(src), tmp_field.lc, tmp_field.data(), 1.0, 0.0, pb, pe)
        for d, side in faces:
            sb, se = slab(d, side, 3)
            ops.stencil_op(SMOOTH, S.lc, S.data(src), F.lc, F.data(), tmp_field.lc, tmp_field.data(), A, w, -1, sb, se)
        comm.exchange(tmp_field, None, "ghost", axis_only)
        for d, side in faces:
            sb, se = slab(d, side, 2)
            ops.stencil_op(SMOOTH, tmp_field.lc, tmp_field.data(), F.lc, F.data(), S.lc, S.data(dst), A, w, -1, sb, se)

    if not faces:
        comm.exchange(S, src, "ghost", axis_only)      # empty on a single block
        if correction_from is not None:
            Sc = correction_from
            ops.jacobi2_prolong(S.lc, S.data(src), S.data(dst), tmp_field.data(), F.lc, F.data(), A, w, b, e, Sc.lc, Sc.data())
        else:
            ops.jacobi2_boxes(S.lc, S.data(src), S.data(dst), tmp_field.data(), F.lc, F.data(), A, w, b, e, b, e)
        S.advance()
        return
    assert correction_from is None, "the folded correction needs a block without neighbours"
    deep = deep_halo_boxes(ops, domain, S, F, A, b, e, faces)
    if deep is not None:
        comm.exchange(S, src, "ghost", axis_only and not _edges_matter(faces))
        ops.jacobi2_boxes(S.lc, S.data(src), S.data(dst), tmp_field.data(), F.lc, F.data(), A, w, deep[0], deep[1], list(b), list(e))
        S.advance()
        return
    # product path on GPUs: the whole choreography below as ONE library call (csrc/examg_comm.hip: pass_blocks) -- the Python
    # form that follows is the same sequence statement by statement; it serves the CPU kernel layer (gloo tests) and is what
    # the library call is tested against (tests/test_gpu_transport.py)
    if hasattr(comm, "c_pass") and comm.c_pass("jacobi2", S, S.data(src), S.data(dst), tmp_field.data(), F, A, w, 0, b, e, axis_only, overlap):
        S.advance()
        return

    # The fallback of examg_jacobi2_boxes (short rows on coarse levels, other stencils) uses tmp as scratch for its whole
    # first step: it must then finish before the shell work writes tmp -- sequential order, no overlap.
    # The kernel layer decides (ONE place: examg_two_stage_eligible -- stencil kind AND entry order, row length, box inside the
    # allocation): only then may the shell work on tmp run concurrently on the side stream.
    b1, e1 = shrunk(1)
    b2, e2 = shrunk(2)
    fused = hasattr(ops, "two_stage_eligible") and all(e2[d] > b2[d] for d in range(nd)) and \
        ops.two_stage_eligible(S.lc, F.lc, A, b1, e1, b2, e2)
    side_stream = ops.side_stream() if (overlap and fused and hasattr(ops, "side_stream")) else None
    if side_stream is not None:
        torch = ops.torch
        main = torch.cuda.current_stream(ops.device)
        side_stream.wait_stream(main)                # everything issued so far (u, rhs) is visible to the side stream
        with torch.cuda.stream(side_stream):
            shell()
        interior()
        main.wait_stream(side_stream)
    else:
        interior()
        shell()
    S.advance()


def rbgs_sweep(ops, comm, domain, S, F, A, w: float, alt, tmp_field, first: int = 0, overlap: bool = True, tmp_planes_valid: bool = False):
    """One red-black sweep of `repeat { color with { (i0+i1+i2) % 2, communicate S; loop over S { S += w (F - A S) };
    apply bc to S } }` (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:204-213) on a block WITH neighbours, out of place
    from S's array into `alt`; returns the array that is free afterwards (S's former one) -- the two change roles.

    Deep interior (loop box shrunk by one point for the first colour, two for the second): one fused pass that needs no
    ghost value.  Two-point shell along the interior faces: exchange S, first colour on three planes into tmp, exchange
    tmp, second colour on two planes into alt -- thin launches on a side stream, concurrent with the interior pass.
    Both arrays and tmp carry S's Dirichlet values on the physical faces (caller: once, `apply bc` values are
    position-only); results are bit-identical to the two in-place half sweeps."""
    nd = domain.nd
    b, e = domain.loop_bounds(S.layout)
    src = S.data()
    faces = [(d, side) for d in range(nd) for side in (-1, 1) if domain.neighbor(d, side) is not None]
    axis_only = all(sum(1 for c in o if c != 0) <= 1 for o in A.offsets)
    if not faces:
        comm.exchange(S, None, "ghost", axis_only)
        ops.rbgs_sweep_fused(S.lc, src, alt, F.lc, F.data(), A, w, first, b, e)
        S.slots[S.active] = alt
        return src

    def shrunk(k):
        bb, ee = list(b), list(e)
        for d, side in faces:
            if side < 0:
                bb[d] = b[d] + k
            else:
                ee[d] = e[d] - k
        return bb, ee

    def slab(d, side, k):
        sb, se = list(b), list(e)
        if side < 0:
            se[d] = min(b[d] + k, e[d])
        else:
            sb[d] = max(e[d] - k, b[d])
        return sb, se

    deep = deep_halo_boxes(ops, domain, S, F, A, b, e, faces)
    if deep is not None:
        comm.exchange(S, None, "ghost", axis_only and not _edges_matter(faces))
        scratch = None if hasattr(ops, "two_stage_eligible") else tmp_field.data()      # the CPU kernel layer runs the two loops through a copy
        ops.rbgs_sweep_fused_boxes(S.lc, src, alt, scratch, F.lc, F.data(), A, w, first, deep[0], deep[1], list(b), list(e))
        S.slots[S.active] = alt
        return src
    if hasattr(comm, "c_pass") and comm.c_pass("rbgs", S, src, alt, tmp_field.data(), F, A, w, first, b, e, axis_only, overlap, tmp_planes_valid):
        S.slots[S.active] = alt        # one library call did interior + shell (see jacobi_pair)
        return src
    b1, e1 = shrunk(1)
    b2, e2 = shrunk(2)
    # eligibility of the one-pass kernel is the kernel layer's decision (examg_two_stage_eligible); without it the fallback
    # needs tmp as scratch on the main stream and everything runs in sequence
    fused = hasattr(ops, "two_stage_eligible") and all(e2[d] > b2[d] for d in range(nd)) and \
        ops.two_stage_eligible(S.lc, F.lc, A, b1, e1, b2, e2)

    def interior(scratch):
        if all(e2[d] > b2[d] for d in range(nd)):
            ops.rbgs_sweep_fused_boxes(S.lc, src, alt, scratch, F.lc, F.data(), A, w, first, b1, e1, b2, e2)

    def shell():
        lay, tmp = S.layout, tmp_field.data()
        comm.exchange(S, None, "ghost", axis_only)
        for d in range(nd if not tmp_planes_valid else 0):     # tmp's physical-face planes: S's (read tangentially by the second colour)
            for side in (-1, 1):
                if domain.neighbor(d, side) is not None:
                    continue
                pb = [lay.idx("DLB", t) if t < nd else 0 for t in range(3)]
                pe = [lay.idx("DRE", t) if t < nd else 1 for t in range(3)]
                pb[d], pe[d] = (lay.idx("DLB", d), lay.idx("DLE", d)) if side < 0 else (lay.idx("DRB", d), lay.idx("DRE", d))
                ops.axpby(S.lc, src, S.lc, tmp, 1.0, 0.0, pb, pe)
        for d, side in faces:                    # first colour on three planes: tmp = S, then the colour's points (reads S only)
            sb, se = slab(d, side, 3)
            ops.axpby(S.lc, src, S.lc, tmp, 1.0, 0.0, sb, se)
            ops.stencil_op(SMOOTH, S.lc, src, F.lc, F.data(), S.lc, tmp, A, w, first, sb, se)
        comm.exchange(tmp_field, None, "ghost", axis_only)
        for d, side in faces:                    # second colour on two planes: alt = tmp, then the colour's points (reads tmp only)
            sb, se = slab(d, side, 2)
            ops.axpby(S.lc, tmp, S.lc, alt, 1.0, 0.0, sb, se)
            ops.stencil_op(SMOOTH, S.lc, tmp, F.lc, F.data(), S.lc, alt, A, w, 1 - first, sb, se)

    side_stream = ops.side_stream() if (overlap and fused and hasattr(ops, "side_stream")) else None
    if side_stream is not None:
        torch = ops.torch
        main = torch.cuda.current_stream(ops.device)
        side_stream.wait_stream(main)
        with torch.cuda.stream(side_stream):
            shell()
        interior(None)
        main.wait_stream(side_stream)
    else:
        # the fallback of the fused pass uses tmp as scratch for its first half sweep: it finishes before the shell writes tmp
        interior(tmp_field.data())
        shell()
    S.slots[S.active] = alt
    return src


def overlapped_loop(ops, domain, b, e, exchange_ghost, kernel, overlap: bool = True):
    """`communicate ghost of f; loop over g { ... reads f within one point ... }` on a block with neighbours, as the
    reference's core/boundary split does it (Compiler/src/exastencils/baseExt/ir/IR_LoopOverPointsInOneFragment.scala:143-222,
    experimental_splitLoopsForAsyncComm): the loop's box shrunk by one point at every interior face reads no ghost value
    and starts at once; the exchange (pack, RCCL send / recv, unpack) runs on the side stream meanwhile; the one-point
    shell -- disjoint slabs, so that accumulating loops (`+=`) stay correct -- follows when the halo is in.  Used for
    the residual (reads Solution's ghosts) and for the restriction (its coarse box shrunk by one point reads no fine ghost).
    `kernel(begin, end)` launches the loop on a sub-box; same bits as exchange + one launch over [b, e)."""
    nd = domain.nd
    faces = [(d, side) for d in range(nd) for side in (-1, 1) if domain.neighbor(d, side) is not None]
    if not faces:
        exchange_ghost()
        kernel(b, e)
        return
    ib, ie = list(b), list(e)
    for d, side in faces:
        if side < 0:
            ib[d] = b[d] + 1
        else:
            ie[d] = e[d] - 1
    if any(ie[d] - ib[d] < 1 for d in range(nd)):      # a block this thin has no interior: plain order
        exchange_ghost()
        kernel(b, e)
        return
    side_stream = ops.side_stream() if (overlap and hasattr(ops, "side_stream")) else None
    if side_stream is not None:
        torch = ops.torch
        main = torch.cuda.current_stream(ops.device)
        side_stream.wait_stream(main)
        with torch.cuda.stream(side_stream):
            exchange_ghost()
        kernel(ib, ie)
        main.wait_stream(side_stream)
    else:
        exchange_ghost()
        kernel(ib, ie)
    lo, hi = list(b), list(e)
    for d in range(nd):
        for side in (-1, 1):
            if (d, side) not in faces:
                continue
            sb, se = list(lo), list(hi)
            if side < 0:
                se[d] = b[d] + 1
            else:
                sb[d] = e[d] - 1
            kernel(sb, se)
        lo[d], hi[d] = ib[d], ie[d]
