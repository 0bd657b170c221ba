#!/usr/bin/env python
"""Time every hot kernel of the V-cycle at one level with events on the launch stream (GPU box).
    python tools/kernel_times.py [--level 9]
Prints ms, LU/s and algorithmic GB/s (SURVEY.md 8d byte counts) per kernel; writes gpurun_out/kernel_times.json."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    from exastencils_amd.field import laplace_fd
    from exastencils_amd.layout import FieldLayout
    from exastencils_amd.ops import HipOps
    from exastencils_amd.lib import GeomC

    ops = HipOps(0)
    n = 1 << args.level
    nc, ncc = (n, n, n), (n // 2,) * 3
    lu, lf = FieldLayout.node(3, nc, 1), FieldLayout.node(3, nc, 0)
    luc, lfc = FieldLayout.node(3, ncc, 1), FieldLayout.node(3, ncc, 0)
    u, un, f, r = (ops.new_array(lu.size) for _ in range(2)), None, None, None
    u, un = list(u)
    f, r = ops.new_array(lf.size), ops.new_array(lu.size)
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
    g = GeomC()
    for d in range(3):
        g.h[d] = 1.0 / n
    cases = [
        ("jacobi", lambda: ops.stencil_op(2, L, u, F, f, L, un, A, w, -1, b, e), pts, 24),
        ("residual", lambda: ops.stencil_op(1, L, u, F, f, L, r, A, 0.0, -1, b, e), pts, 24),
        ("rbgs colour 0 (in place)", lambda: ops.stencil_op(2, L, u, F, f, L, u, A, w, 0, b, e), pts, 24),
        ("rbgs colour 1 (in place)", lambda: ops.stencil_op(2, L, u, F, f, L, u, A, w, 1, b, e), pts, 24),
        ("rbgs fused sweep", lambda: ops.rbgs_sweep_fused(L, u, un, F, f, A, w, 0, b, e), pts, 24),
        ("jacobi2 (two steps)", lambda: ops.jacobi2(L, u, un, None, F, f, A, w, b, e), 2 * pts, 24),
        ("residual_restrict", lambda: ops.residual_restrict(L, u, F, f, L, r, A, Fc, fc, 1.0, b, e, bc, ec), pts, 16 + 1),
        ("restrict", lambda: ops.restrict(L, r, Fc, fc, 1.0, bc, ec), cpts, 72),
        ("prolong_add", lambda: ops.prolong_add(Lc, uc, L, u, b, e), pts, 17),
        ("dot", lambda: ops.dot(L, r, L, r, b, e, out), pts, 8),
        ("axpby y+=a x", lambda: ops.axpby(L, r, L, un, 0.5, 1.0, b, e), pts, 24),
        ("set", lambda: ops.set(L, un, 0.0, b, e), pts, 8),
        ("apply_dirichlet", lambda: ops.apply_dirichlet(L, u, g, 1, (), 63), 6 * (n + 3) ** 2, 8),
    ]
    res = {}
    for name, fn, units, bpu in cases:
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        res[name] = dict(ms=ms, units_per_s=units / ms * 1e3, gbs=units * bpu / ms / 1e6)
        print("%-28s %9.4f ms  %10.3e units/s  %8.0f GB/s algorithmic" % (name, ms, units / ms * 1e3, units * bpu / ms / 1e6), flush=True)
        # keep values bounded
        ops.fill_random(u, 100)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "kernel_times_L%d.json" % args.level), "w"), indent=1)


if __name__ == "__main__":
    main()
