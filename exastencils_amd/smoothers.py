"""Temporally blocked smoother steps (two Jacobi steps per pass over HBM), also across block neighbours.

Reference idea: `repeat n times with contraction` / IR_ContractingLoop (Compiler/src/exastencils/baseExt/ir/
IR_ContractingLoop.scala; Testing/PolyExpl/Jac3Dcc.exa4:27) and the `Comm*TempBlockable` layouts of the
generated-from-L3 programs.  The reference widens the ghost layers to block in time; here the ghost layers stay
one deep (the drop-in layout) and the second step of the duplicate planes at interior faces is finished separately:

  1. communicate ghost of u                                   (as every Smoother call does)
  2. u_out = J(J(u)) on the loop's box minus the duplicate planes at interior faces   (examg_jacobi2_boxes;
     the first step is evaluated on the whole box, so everything the second step needs there is local)
  3. tmp = J(u) on the two planes next to every interior face   (thin launches)
  4. communicate ghost of tmp                                  (the neighbours' first-step values)
  5. u_out = J(tmp) on the duplicate planes at interior faces   (thin launches)

Two exchanges per two steps -- as many as two plain Smoother calls -- and bit-identical results.  On a single block
steps 3-5 vanish and 1 is empty.

Overlap: steps 3 and 4 (thin launches, pack, RCCL send/recv over xGMI, unpack) run on a side stream while the main
stream executes step 2, the only large kernel; step 5 waits for both (events).  This is the reference's core/boundary
split (Compiler/src/exastencils/baseExt/ir/IR_LoopOverPointsInOneFragment.scala:143-222,
experimental_splitLoopsForAsyncComm) with the roles swapped: the interior needs no halo at all here.
"""
from __future__ import annotations

SMOOTH = 2


def jacobi_pair(ops, comm, domain, S, F, A, w: float, tmp_field, overlap: bool = True):
    """Two applications of `Smoother@current` (Testing/Smoothers/Jac.exa4:125-131) on field S (2 slots):
    reads slot <active>, leaves the result in the slot two `advance`s would make active (the same one),
    reached through one advance of the out-of-place pass.  tmp_field: scratch field of S's layout whose
    Dirichlet shell holds S's boundary values."""
    nd = domain.nd
    b, e = domain.loop_bounds(S.layout)
    src, dst = S.active, S.next
    faces = [(d, side) for d in range(nd) for side in (-1, 1) if domain.neighbor(d, side) is not None]
    axis_only = all(sum(1 for c in o if c != 0) <= 1 for o in A.offsets)   # 5/7-point: face ghosts suffice
    comm.exchange(S, src, "ghost", axis_only)
    b2, e2 = list(b), list(e)
    for d, side in faces:
        if side < 0:
            b2[d] = b[d] + 1
        else:
            e2[d] = e[d] - 1
    def first_step_on_face_slabs_and_exchange():
        for d, side in faces:
            sb, se = list(b), list(e)
            if side < 0:
                se[d] = b[d] + 2
            else:
                sb[d] = e[d] - 2
            ops.stencil_op(SMOOTH, S.lc, S.data(src), F.lc, F.data(), tmp_field.lc, tmp_field.data(), A, w, -1, sb, se)
        comm.exchange(tmp_field, None, "ghost", axis_only)

    side_stream = ops.side_stream() if (faces and overlap and hasattr(ops, "side_stream")) else None
    # The fallback of examg_jacobi2_boxes (short rows on coarse levels) uses tmp as scratch for the whole first step:
    # it must then run BEFORE the face slabs and the exchange write tmp, hence no overlap for those boxes.
    canonical7 = nd == 3 and A.cfield is None and len(A.offsets) == 7 and axis_only
    if side_stream is not None and ((e2[0] - b2[0]) < 64 or not canonical7):
        side_stream = None
    if side_stream is not None:
        torch = ops.torch
        main = torch.cuda.current_stream(ops.device)
        side_stream.wait_stream(main)                # ghosts of u are in place
        with torch.cuda.stream(side_stream):
            first_step_on_face_slabs_and_exchange()
        ops.jacobi2_boxes(S.lc, S.data(src), S.data(dst), tmp_field.data(), F.lc, F.data(), A, w, b, e, b2, e2)
        main.wait_stream(side_stream)
    else:
        ops.jacobi2_boxes(S.lc, S.data(src), S.data(dst), tmp_field.data(), F.lc, F.data(), A, w, b, e, b2, e2)
        if faces:
            first_step_on_face_slabs_and_exchange()
    if faces:
        for d, side in faces:
            sb, se = list(b), list(e)
            if side < 0:
                se[d] = b[d] + 1
            else:
                sb[d] = e[d] - 1
            ops.stencil_op(SMOOTH, tmp_field.lc, tmp_field.data(), F.lc, F.data(), S.lc, S.data(dst), A, w, -1, sb, se)
    S.advance()
