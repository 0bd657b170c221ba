#!/usr/bin/env python
"""Launch every hot kernel of the V-cycle a few times at one level, in a fixed order, for rocprofv3 counter passes:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcK_FETCH -- python3 $R/tools/pmc_kernels.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcK_WRITE -- python3 $R/tools/pmc_kernels.py
    python3 $R/tools/pmc_kernels.py --time          # un-profiled timing pass (HIP events), gpurun_out/pmcK_times.json
    python3 $R/tools/pmc_reduce.py                  # -> profiles/r02_pmc_kernels.json

Each case is one kernel launch; `compulsory` = distinct field.slot reads/writes x 8 B x points of the launch (the reference's
own rule, Compiler/src/exastencils/performance/ir/IR_EvaluatePerformanceEstimates.scala:206-215)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cases(ops, level, with27=True, align=0):
    from exastencils_amd.field import Stencil, helmholtz27_offsets, laplace_fd
    from exastencils_amd.layout import FieldLayout
    from exastencils_amd.lib import GeomC

    n = 1 << level
    nc, ncc = (n, n, n), (n // 2,) * 3
    lu, lf = FieldLayout.node(3, nc, 1, True, True, align), FieldLayout.node(3, nc, 0, True, False, align)
    luc, lfc = FieldLayout.node(3, ncc, 1, True, True, align), FieldLayout.node(3, ncc, 0, True, False, align)
    u, un, r = (ops.new_array(lu.size) for _ in range(3))
    f = ops.new_array(lf.size)
    uc, fc = ops.new_array(luc.size), ops.new_array(lfc.size)
    for i, t in enumerate((u, un, f, r, uc, fc)):
        ops.fill_random(t, 100 + i)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    bc, ec = [1, 1, 1], [n // 2] * 3
    L, F, Lc, Fc = lu.c_struct(), lf.c_struct(), luc.c_struct(), lfc.c_struct()
    pts, cpts = (n - 1) ** 3, (n // 2 - 1) ** 3
    out = ops.new_scalar()
    # name, launch, kernel-name pattern, compulsory bytes per launch, lattice updates per launch
    cs = [
        # rows of 400 .. 512 points (level 9) run the row-marching kernel, others the 128-point-window kernel
        ("jacobi_1step", lambda: ops.stencil_op(2, L, u, F, f, L, un, A, w, -1, b, e), "k_stencil7_rowmarch<2" if 400 <= n - 1 <= 512 else "k_stencil7_zmarch<2", 24 * pts, pts),
        ("residual", lambda: ops.stencil_op(1, L, u, F, f, L, r, A, 0.0, -1, b, e), "k_stencil7_rowmarch<1" if 400 <= n - 1 <= 512 else "k_stencil7_zmarch<1", 24 * pts, pts),
        ("rbgs_half_sweep", lambda: ops.stencil_op(2, L, u, F, f, L, u, A, w, 0, b, e), "k_stencil7_zmarch<2", 24 * pts, pts // 2),
        ("jacobi_2step", lambda: ops.jacobi2(L, u, un, None, F, f, A, w, b, e), "k_two_stage7_lds<0, false, 8, true, 1, 0", 24 * pts, 2 * pts),
        ("jacobi_3step", lambda: ops.jacobi3(L, u, un, None, F, f, A, w, b, e), "k_three_stage7_lds<0, 8, true, 3, false", 24 * pts, 3 * pts),
        ("rbgs_fused_sweep", lambda: ops.rbgs_sweep_fused(L, u, un, F, f, A, w, 0, b, e), "k_two_stage7_lds<0, true, 8, true, 1, 0", 24 * pts, pts),
        ("rbgs_3colours", lambda: ops.rbgs_colours3(L, u, un, F, f, A, w, 0, b, e), "k_three_stage7_lds<0, 8, true, 3, true", 24 * pts, 3 * (pts // 2)),
        ("rbgs_fused_sweep_prolong", lambda: ops.rbgs_sweep_fused_prolong(L, u, un, F, f, A, w, 0, b, e, Lc, uc), "k_two_stage7_lds<0, true, 8, true, 1, 1",
         24 * pts + 8 * cpts, pts),
        ("rbgs_fused_sweep_zero", lambda: ops.rbgs_sweep_fused_zero(L, un, F, f, A, w, 0, b, e), "k_two_stage7_lds<0, true, 8, true, 1, 2", 16 * pts, pts),
        ("residual_restrict", lambda: ops.residual_restrict(L, u, F, f, L, r, A, Fc, fc, 1.0, b, e, bc, ec), "k_residual_restrict3",
         16 * pts + 8 * cpts, pts),
        ("restrict", lambda: ops.restrict(L, r, Fc, fc, 1.0, bc, ec), "k_restrict3_wide", 8 * pts + 8 * cpts, cpts),
        ("prolong_add", lambda: ops.prolong_add(Lc, uc, L, u, b, e), "k_prolong_add3_pairs", 16 * pts + 8 * cpts, pts),
        ("residual_norm", lambda: ops.residual_norm2(L, u, F, f, A, b, e, out=out), "k_stencil7_zmarch<3", 16 * pts, pts),
        ("dot_norm", lambda: ops.dot(L, r, L, r, b, e, out), "k_dot_rows", 8 * pts, pts),
    ]
    if align == 0:
        # the red-black half sweep with Solution and RHS under the colour split `[x, y, z] => [x / 2, y, z, x % 2]` (LayoutTransformations,
        # Testing/LayoutTrafo/rbgs.exa4:2): the points of a colour and their x neighbours are contiguous -- 32 B per update instead of 48
        lus, lfs = lu.split_x(), lf.split_x()
        us, fs = ops.new_array(lus.size), ops.new_array(lfs.size)
        ops.transform_field(L, u, lus.c_struct(), us)
        ops.transform_field(F, f, lfs.c_struct(), fs)
        Ls, Fs = lus.c_struct(), lfs.c_struct()
        cs.insert(3, ("rbgs_half_sweep_colour_split", lambda: ops.stencil_op(2, Ls, us, Fs, fs, Ls, us, A, w, 0, b, e), "k_rbgs_half_split7",
                      16 * pts, pts // 2))
    if align == 0 and level >= 8:
        # the one-step kernel on the padded layout the reference produces with data_alignFieldPointers (rows of 544 doubles at
        # 512^3, field/ir/IR_AddPaddingToFieldLayouts.scala:36-41): every 16-byte access aligned, no partial lines at window edges
        lup, lfp = FieldLayout.node(3, nc, 1, True, True, 16), FieldLayout.node(3, nc, 0, True, False, 16)
        up, unp, fp = ops.new_array(lup.size), ops.new_array(lup.size), ops.new_array(lfp.size)
        ops.fill_random(up, 201)
        ops.fill_random(fp, 202)
        Lp, Fp = lup.c_struct(), lfp.c_struct()
        cs.insert(1, ("jacobi_1step_padded_rows", lambda: ops.stencil_op(2, Lp, up, Fp, fp, Lp, unp, A, w, -1, b, e), "k_stencil7_zmarch<2",
                      24 * pts, pts))
    if with27:
        nocomm = FieldLayout.node(3, nc, 0, False, False, align)
        cf = ops.new_array(27 * nocomm.size)
        g = GeomC()
        for d in range(3):
            g.h[d] = 1.0 / n
        ops.init_helmholtz27(nocomm.c_struct(), cf, g, 7, (10.0, 2.0), [0, 0, 0], [n + 1] * 3)
        A27 = Stencil(helmholtz27_offsets(), [], cf, nocomm)
        Fn = nocomm.c_struct()
        f27 = ops.new_array(nocomm.size)
        ops.fill_random(f27, 7)
        cs.append(("jacobi_27entry_field", lambda: ops.stencil_op(2, L, u, Fn, f27, L, un, A27, 0.8, -1, b, e),
                   "k_stencilfield_unrolled<2, 27>", (24 + 8 * 27) * pts, pts))
        # the same field under `LayoutTransformations { transform <coefficients> with [x, y, z, i] => [i, x, y, z] }`: one coefficient stream
        A27t = A27.entry_fastest(ops)
        cs.append(("jacobi_27entry_field_entry_fastest", lambda: ops.stencil_op(2, L, u, Fn, f27, L, un, A27t, 0.8, -1, b, e),
                   "k_stencilfield27_rec<2>", (24 + 8 * 27) * pts, pts))
        # temporal blocking on the records (csrc/kernels_sf27pair.hip): both loops of a pair share the coefficients of a point
        r27 = ops.new_array(lu.size)
        cs.append(("jacobi_27entry_two_steps", lambda: ops.jacobi2(L, u, un, None, Fn, f27, A27t, 0.8, b, e),
                   "k_sf27_two_stage_r2<2>", (24 + 8 * 27) * pts, 2 * pts))
        cs.append(("jacobi_27entry_step_residual", lambda: ops.jacobi_residual(L, u, un, Fn, f27, L, r27, A27t, 0.8, b, e),
                   "k_sf27_two_stage_r2<1>", (32 + 8 * 27) * pts, 2 * pts))
    return cs, dict(u=u)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--time", action="store_true", help="timing pass: 20 launches per case between HIP events")
    ap.add_argument("--no27", action="store_true")
    ap.add_argument("--align", type=int, default=0)
    args = ap.parse_args()
    import torch

    from exastencils_amd.ops import HipOps

    ops = HipOps(0)
    cs, keep = cases(ops, args.level, not args.no27, args.align)
    res = []
    for name, fn, pattern, comp, lu in cs:
        fn()
        torch.cuda.synchronize()
        reps = 20 if args.time else args.reps
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res.append(dict(case=name, kernel=pattern, compulsory_bytes=comp, lattice_updates=lu, ms=ms, level=args.level, align=args.align))
        if args.time:
            print("%-22s %8.4f ms  %6.0f GB/s compulsory  frac %.3f" % (name, ms, comp / ms / 1e6, comp / ms / 1e6 / 8000.0), flush=True)
        ops.fill_random(keep["u"], 100)
    if args.time:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(res, open(os.path.join(ROOT, "gpurun_out", "pmcK_times_L%d%s.json" % (args.level, "_a%d" % args.align if args.align else "")), "w"), indent=1)


if __name__ == "__main__":
    main()
