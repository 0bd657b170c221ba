#!/usr/bin/env python
"""A 1024^3 block on one GPU (8.7 GB per array): the fused kernels against their unfused forms, bit for bit, and their run
times -- index arithmetic beyond 2^31 bytes per array, 32-bit relative offsets of the two-step kernel, 288 GB HBM."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exastencils_amd.field import laplace_fd
from exastencils_amd.layout import FieldLayout
from exastencils_amd import lib
from exastencils_amd.ops import HipOps

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ops = HipOps(0, lib.DBG_LIB_PATH)      # debug build: examg_debug_restrict selects the restriction kernel
lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
lc = FieldLayout.node(3, (n // 2,) * 3, 0)
u, a, b_, t, f = (ops.new_array(lu.size) for _ in range(4)), None, None, None, ops.new_array(lf.size)
u, a, b_, t = list(u)
fc1, fc2 = ops.new_array(lc.size), ops.new_array(lc.size)
ops.fill_random(u, 1); ops.fill_random(f, 2)
A = laplace_fd(3, (1.0 / n,) * 3)
w = 0.8 / A.diag
b, e = [1, 1, 1], [n, n, n]
L, F, Lc = lu.c_struct(), lf.c_struct(), lc.c_struct()


def timed(fn, k=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


out = {"n": n, "array_GB": lu.size * 8 / 1e9}
# two Jacobi steps: fused vs two launches (the intermediate array carries the same boundary planes, as `apply bc` leaves them)
t.copy_(u)
ops.stencil_op(2, L, u, F, f, L, t, A, w, -1, b, e)
ops.stencil_op(2, L, t, F, f, L, a, A, w, -1, b, e)
b_.copy_(u)       # shell of the output array as the program leaves it
a_shell = a.clone(); a_shell.copy_(u); ops.stencil_op(2, L, t, F, f, L, a_shell, A, w, -1, b, e)
ops.jacobi2(L, u, b_, None, F, f, A, w, b, e)
out["jacobi2_equals_two_steps"] = bool(torch.equal(a_shell, b_))
out["jacobi_step_ms"] = timed(lambda: ops.stencil_op(2, L, u, F, f, L, t, A, w, -1, b, e))
out["jacobi2_ms"] = timed(lambda: ops.jacobi2(L, u, b_, None, F, f, A, w, b, e))
# red-black sweep: fused vs two in-place half sweeps
a.copy_(u)
for c in (0, 1):
    ops.stencil_op(2, L, a, F, f, L, a, A, w, c, b, e)
b_.copy_(u)
ops.rbgs_sweep_fused(L, u, b_, F, f, A, w, 0, b, e)
out["rbgs_fused_equals_half_sweeps"] = bool(torch.equal(a, b_))
out["rbgs_fused_ms"] = timed(lambda: ops.rbgs_sweep_fused(L, u, b_, F, f, A, w, 0, b, e))
# restriction: wide kernel vs one thread per point
bc, ec = [1, 1, 1], [n // 2] * 3
ops.restrict(L, u, Lc, fc1, 1.0, bc, ec)
ops.L.examg_debug_restrict(0)
ops.restrict(L, u, Lc, fc2, 1.0, bc, ec)
ops.L.examg_debug_restrict(1)
out["restrict_wide_equals_plain"] = bool(torch.equal(fc1, fc2))
out["restrict_ms"] = timed(lambda: ops.restrict(L, u, Lc, fc1, 1.0, bc, ec))
# correction + sweep in one pass vs correction loop, then the fused sweep; sweep of the zero field vs zeroing + fused sweep
lcu = FieldLayout.node(3, (n // 2,) * 3, 1)
uc = ops.new_array(lcu.size)
ops.fill_random(uc, 3)
Lcu = lcu.c_struct()
a.copy_(u)
ops.prolong_add(Lcu, uc, L, a, b, e)
t.copy_(u)
ops.rbgs_sweep_fused(L, a, t, F, f, A, w, 0, b, e)
b_.copy_(u)
ops.rbgs_sweep_fused_prolong(L, u, b_, F, f, A, w, 0, b, e, Lcu, uc)
out["sweep_with_folded_correction_equals_two_launches"] = bool(torch.equal(t, b_))
out["sweep_with_folded_correction_ms"] = timed(lambda: ops.rbgs_sweep_fused_prolong(L, u, b_, F, f, A, w, 0, b, e, Lcu, uc))
a.zero_(); t.zero_(); b_.zero_()
ops.rbgs_sweep_fused(L, a, t, F, f, A, w, 0, b, e)
ops.rbgs_sweep_fused_zero(L, b_, F, f, A, w, 0, b, e)
out["zero_field_sweep_equals_sweep_of_zeros"] = bool(torch.equal(t, b_))
# residual + norm in one pass vs residual loop + reduction
ops.stencil_op(1, L, u, F, f, L, a, A, 0.0, -1, b, e)
want = ops.scalar_value(ops.dot(L, a, L, a, b, e))
got = ops.scalar_value(ops.residual_norm2(L, u, F, f, A, b, e))
out["residual_norm2_rel_diff"] = abs(got - want) / want
# one Jacobi step and the residual with the one-step kernel against the generic kernel, residual + restriction against its two loops
import ctypes as C
ops.stencil_op(2, L, u, F, f, L, a, A, w, -1, b, e)
ops.L.examg_debug_force_generic(1)
ops.stencil_op(2, L, u, F, f, L, b_, A, w, -1, b, e)
ops.L.examg_debug_force_generic(0)
out["one_step_kernel_equals_generic"] = bool(torch.equal(a, b_))
bc_, ec_ = [1, 1, 1], [n // 2] * 3
ops.residual_restrict(L, u, F, f, L, t, A, Lc, fc1, 1.0, b, e, bc_, ec_)
ops.stencil_op(1, L, u, F, f, L, t, A, 0.0, -1, b, e)
ops.restrict(L, t, Lc, fc2, 1.0, bc_, ec_)
out["residual_restrict_equals_two_loops"] = bool(torch.equal(fc1, fc2))
out["residual_restrict_ms"] = timed(lambda: ops.residual_restrict(L, u, F, f, L, t, A, Lc, fc1, 1.0, b, e, bc_, ec_))
# prolongation + correction: pair kernel against the one-thread-per-point kernel
lcu = FieldLayout.node(3, (n // 2,) * 3, 1)
ucs = ops.new_array(lcu.size)
ops.fill_random(ucs, 5)
a.copy_(u); b_.copy_(u)
ops.prolong_add(lcu.c_struct(), ucs, L, a, b, e)
out["prolong_ms"] = timed(lambda: ops.prolong_add(lcu.c_struct(), ucs, L, b_, b, e))
# 7-entry stencil field: z-march kernel against the generic kernel (skipped above 800^3: 7 coefficient planes of a 1024^3 block are 60 GB)
if n <= 800:
    from exastencils_amd.field import Stencil, stencil_field_offsets
    ops.L.examg_debug_stencilfield.argtypes = [C.c_int, C.c_int]
    cf = ops.new_array(7 * lf.size)
    ops.fill_random(cf, 3)
    cf += 3.0
    sf = Stencil(stencil_field_offsets(3), [], cf, lf)
    ops.stencil_op(2, L, u, F, f, L, a, sf, 0.8, -1, b, e)
    ops.L.examg_debug_stencilfield(-1, 0)
    ops.stencil_op(2, L, u, F, f, L, b_, sf, 0.8, -1, b, e)
    ops.L.examg_debug_stencilfield(1, 0)
    out["stencilfield7_kernel_equals_generic"] = bool(torch.equal(a, b_))
    out["stencilfield7_ms"] = timed(lambda: ops.stencil_op(2, L, u, F, f, L, a, sf, 0.8, -1, b, e))
pts = (n - 1) ** 3
out["jacobi2_lups"] = 2 * pts / (out["jacobi2_ms"] * 1e-3)
out["jacobi2_algorithmic_gbs"] = 48.0 * pts / (out["jacobi2_ms"] * 1e-3) / 1e9
print(json.dumps(out))
