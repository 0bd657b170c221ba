"""Kernel-call layer: one method per emitted-loop kind, forwarding to the C ABI of libexamg.so.

PyTorch is plumbing here: device memory (float64 tensors), the current HIP stream (so launches can be
captured by torch.cuda.graph and timed by events on that stream) and torch.distributed.  All
arithmetic happens in the HIP kernels; there is no CPU fallback -- constructing HipOps without the
built library or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

from . import lib as _lib
from .field import Stencil, fn_expr
from .lib import ExamgError, check, ivec


class HipOps:
    name = "hip"

    def __init__(self, device: Optional[int] = None, lib_path: Optional[str] = None):
        import torch

        self.torch = torch
        self.L = _lib.load(lib_path)   # raises if libexamg.so is missing
        if not torch.cuda.is_available():
            raise ExamgError("HipOps needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        if self.L.examg_device_count() < 1:
            raise ExamgError("libexamg sees no HIP device")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._work = torch.empty(int(self.L.examg_reduce_work_bytes()) // 8, dtype=torch.float64, device=self.device)

    # -- plumbing ------------------------------------------------------------------------------
    def new_array(self, n: int):
        return self.torch.zeros(int(n), dtype=self.torch.float64, device=self.device)

    def new_scalar(self):
        return self.torch.zeros(1, dtype=self.torch.float64, device=self.device)

    @staticmethod
    def ptr(t) -> int:
        return t.data_ptr()

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)

    def side_stream(self):
        """A second HIP stream for halo traffic that overlaps the interior kernels."""
        if getattr(self, "_side", None) is None:
            # high priority: short shell kernels beside a pass that fills the chip (csrc/examg_comm.hip: overlap_of)
            import os

            self._side = self.torch.cuda.Stream(self.device, priority=0 if os.environ.get("EXAMG_SIDE_PRIORITY") == "0" else -1)
        return self._side

    def to_host(self, t):
        return t.detach().cpu().numpy()

    def from_host(self, a):
        return self.torch.from_numpy(a).to(self.device)

    # -- stencil loops ----------------------------------------------------------------------------
    def stencil_op(self, mode: int, lu, u, lf, rhs, ld, dst, st: Stencil, w: float, colour: int, begin, end):
        sc = st.c_struct(self.ptr)
        check(self.L.examg_stencil_op(mode, C.byref(lu), self.ptr(u), C.byref(lf) if lf is not None else None,
                                      self.ptr(rhs) if rhs is not None else None, C.byref(ld), self.ptr(dst),
                                      C.byref(sc), float(w), int(colour), ivec(begin), ivec(end), self._stream()),
              "examg_stencil_op")

    def rbgs_sweep_fused(self, lu, u_in, u_out, lf, rhs, st: Stencil, w: float, first: int, begin, end):
        sc = st.c_struct(self.ptr)
        check(self.L.examg_rbgs_sweep_fused(C.byref(lu), self.ptr(u_in), self.ptr(u_out), C.byref(lf), self.ptr(rhs),
                                            C.byref(sc), float(w), int(first), ivec(begin), ivec(end), self._stream()),
              "examg_rbgs_sweep_fused")

    def rbgs_sweep_fused_boxes(self, lu, u_in, u_out, tmp, lf, rhs, st: Stencil, w: float, first: int, begin1, end1, begin2, end2):
        sc = st.c_struct(self.ptr)
        check(self.L.examg_rbgs_sweep_fused_boxes(C.byref(lu), self.ptr(u_in), self.ptr(u_out), self.ptr(tmp) if tmp is not None else None,
                                                  C.byref(lf), self.ptr(rhs), C.byref(sc), float(w), int(first), ivec(begin1), ivec(end1),
                                                  ivec(begin2), ivec(end2), self._stream()), "examg_rbgs_sweep_fused_boxes")

    def jacobi2(self, lu, u_in, u_out, tmp, lf, rhs, st: Stencil, w: float, begin, end):
        sc = st.c_struct(self.ptr)
        check(self.L.examg_jacobi2(C.byref(lu), self.ptr(u_in), self.ptr(u_out), self.ptr(tmp) if tmp is not None else None,
                                   C.byref(lf), self.ptr(rhs), C.byref(sc), float(w), ivec(begin), ivec(end), self._stream()),
              "examg_jacobi2")

    def jacobi3(self, lu, u_in, u_out, tmp, lf, rhs, st: Stencil, w: float, begin, end):
        """Three Jacobi steps in one pass where the kernel applies (examg_jacobi3)."""
        sc = st.c_struct(self.ptr)
        check(self.L.examg_jacobi3(C.byref(lu), self.ptr(u_in), self.ptr(u_out), self.ptr(tmp) if tmp is not None else None,
                                   C.byref(lf), self.ptr(rhs), C.byref(sc), float(w), ivec(begin), ivec(end), self._stream()),
              "examg_jacobi3")

    def rbgs_colours3(self, lu, u_in, u_out, lf, rhs, st: Stencil, w: float, first: int, begin, end):
        """Three colour loops (first, other, first) in one pass where the kernel applies (examg_rbgs_colours3)."""
        sc = st.c_struct(self.ptr)
        check(self.L.examg_rbgs_colours3(C.byref(lu), self.ptr(u_in), self.ptr(u_out), C.byref(lf), self.ptr(rhs), C.byref(sc), float(w), int(first),
                                         ivec(begin), ivec(end), self._stream()), "examg_rbgs_colours3")

    def three_stage_eligible(self, lu, lf, st: Stencil, begin, end) -> bool:
        sc = st.c_struct(self.ptr)
        return bool(self.L.examg_three_stage_eligible(C.byref(lu), C.byref(lf), C.byref(sc), ivec(begin), ivec(end)))

    def jacobi_residual(self, lu, u_in, u_out, lf, rhs, lr, res, st: Stencil, w: float, begin, end):
        """One Jacobi step and the residual of its result in one pass (examg_jacobi_residual)."""
        sc = st.c_struct(self.ptr)
        check(self.L.examg_jacobi_residual(C.byref(lu), self.ptr(u_in), self.ptr(u_out), C.byref(lf), self.ptr(rhs), C.byref(lr), self.ptr(res),
                                           C.byref(sc), float(w), ivec(begin), ivec(end), self._stream()), "examg_jacobi_residual")

    def jacobi2_boxes(self, lu, u_in, u_out, tmp, lf, rhs, st: Stencil, w: float, begin1, end1, begin2, end2):
        sc = st.c_struct(self.ptr)
        check(self.L.examg_jacobi2_boxes(C.byref(lu), self.ptr(u_in), self.ptr(u_out), self.ptr(tmp) if tmp is not None else None,
                                         C.byref(lf), self.ptr(rhs), C.byref(sc), float(w), ivec(begin1), ivec(end1),
                                         ivec(begin2), ivec(end2), self._stream()), "examg_jacobi2_boxes")

    def rbgs_sweep_fused_prolong(self, lu, u_in, u_out, lf, rhs, st: Stencil, w: float, first: int, begin, end, lc, uc):
        sc = st.c_struct(self.ptr)
        check(self.L.examg_rbgs_sweep_fused_prolong(C.byref(lu), self.ptr(u_in), self.ptr(u_out), C.byref(lf), self.ptr(rhs), C.byref(sc),
                                                    float(w), int(first), ivec(begin), ivec(end), C.byref(lc), self.ptr(uc), self._stream()),
              "examg_rbgs_sweep_fused_prolong")

    def rbgs_sweep_fused_zero(self, lu, u_out, lf, rhs, st: Stencil, w: float, first: int, begin, end):
        sc = st.c_struct(self.ptr)
        check(self.L.examg_rbgs_sweep_fused_zero(C.byref(lu), self.ptr(u_out), C.byref(lf), self.ptr(rhs), C.byref(sc), float(w),
                                                 int(first), ivec(begin), ivec(end), self._stream()), "examg_rbgs_sweep_fused_zero")

    def jacobi2_prolong(self, lu, u_in, u_out, tmp, lf, rhs, st: Stencil, w: float, begin, end, lc, uc):
        sc = st.c_struct(self.ptr)
        check(self.L.examg_jacobi2_prolong(C.byref(lu), self.ptr(u_in), self.ptr(u_out), self.ptr(tmp) if tmp is not None else None,
                                           C.byref(lf), self.ptr(rhs), C.byref(sc), float(w), ivec(begin), ivec(end), C.byref(lc),
                                           self.ptr(uc), self._stream()), "examg_jacobi2_prolong")

    def two_stage_eligible(self, lu, lf, st: Stencil, begin1, end1, begin2, end2) -> bool:
        """Will jacobi2_boxes / rbgs_sweep_fused_boxes run their one-pass kernel (True) or the fallback that writes `tmp`?"""
        sc = st.c_struct(self.ptr)
        return bool(self.L.examg_two_stage_eligible(C.byref(lu), C.byref(lf), C.byref(sc), ivec(begin1), ivec(end1), ivec(begin2), ivec(end2)))

    def residual_restrict_one_pass(self, lu, lf, st: Stencil, lc, fbegin, fend, cbegin, cend) -> bool:
        """Will residual_restrict run its one-pass kernel for these boxes (True) or residual + restriction through the residual array?"""
        sc = st.c_struct(self.ptr)
        return bool(self.L.examg_residual_restrict_one_pass(C.byref(lu), C.byref(lf), C.byref(sc), C.byref(lc), ivec(fbegin), ivec(fend), ivec(cbegin),
                                                            ivec(cend)))

    # -- inter-grid -------------------------------------------------------------------------------
    def restrict(self, lfine, rf, lc, fc, scale: float, begin, end):
        check(self.L.examg_restrict(C.byref(lfine), self.ptr(rf), C.byref(lc), self.ptr(fc), float(scale), ivec(begin),
                                    ivec(end), self._stream()), "examg_restrict")

    def residual_restrict(self, lu, u, lf, rhs, lr, res, st: Stencil, lc, fc, scale: float, fbegin, fend, cbegin, cend):
        sc = st.c_struct(self.ptr)
        check(self.L.examg_residual_restrict(C.byref(lu), self.ptr(u), C.byref(lf), self.ptr(rhs), C.byref(lr) if lr is not None else None,
                                             self.ptr(res) if res is not None else None, C.byref(sc), C.byref(lc), self.ptr(fc), float(scale),
                                             ivec(fbegin), ivec(fend), ivec(cbegin), ivec(cend), self._stream()), "examg_residual_restrict")

    def prolong_add(self, lc, uc, lfine, uf, begin, end):
        check(self.L.examg_prolong_add(C.byref(lc), self.ptr(uc), C.byref(lfine), self.ptr(uf), ivec(begin), ivec(end),
                                       self._stream()), "examg_prolong_add")

    # -- BLAS-1 -----------------------------------------------------------------------------------
    def set(self, l, x, v: float, begin, end):
        check(self.L.examg_set(C.byref(l), self.ptr(x), float(v), ivec(begin), ivec(end), self._stream()), "examg_set")

    def axpby(self, lx, x, ly, y, a: float, b: float, begin, end):
        check(self.L.examg_axpby(C.byref(lx), self.ptr(x), C.byref(ly), self.ptr(y), float(a), float(b), ivec(begin),
                                 ivec(end), self._stream()), "examg_axpby")

    def axpby_dev(self, lx, x, ly, y, a: float, b: float, which: int, sign: float, num, den, begin, end):
        check(self.L.examg_axpby_dev(C.byref(lx), self.ptr(x), C.byref(ly), self.ptr(y), float(a), float(b), int(which),
                                     float(sign), self.ptr(num), self.ptr(den), ivec(begin), ivec(end), self._stream()),
              "examg_axpby_dev")

    # -- reductions (result stays on the device) ---------------------------------------------------
    def dot(self, lx, x, ly, y, begin, end, out=None):
        out = self.new_scalar() if out is None else out
        check(self.L.examg_dot(C.byref(lx), self.ptr(x), C.byref(ly), self.ptr(y), ivec(begin), ivec(end), self.ptr(out),
                               self.ptr(self._work), self._stream()), "examg_dot")
        return out

    def residual_norm2(self, lu, u, lf, rhs, st: Stencil, begin, end, lr=None, res=None, out=None):
        """sum over the box of (rhs - A u)^2 on the device, the residual not stored (examg_residual_norm2)"""
        out = self.new_scalar() if out is None else out
        sc = st.c_struct(self.ptr)
        check(self.L.examg_residual_norm2(C.byref(lu), self.ptr(u), C.byref(lf), self.ptr(rhs), C.byref(sc), ivec(begin), ivec(end),
                                          C.byref(lr) if lr is not None else None, self.ptr(res) if res is not None else None,
                                          self.ptr(out), self.ptr(self._work), self._stream()), "examg_residual_norm2")
        return out

    def max_err_fn(self, l, x, geom, fn: int, params: Sequence[float], begin, end, out=None):
        out = self.new_scalar() if out is None else out
        return self.max_err_expr(l, x, geom, fn_expr(fn, params), begin, end, out)

    def max_err_expr(self, l, x, geom, expr, begin, end, out=None):
        out = self.new_scalar() if out is None else out
        check(self.L.examg_max_err_expr(C.byref(l), self.ptr(x), C.byref(geom), C.byref(expr), ivec(begin), ivec(end), self.ptr(out),
                                        self.ptr(self._work), self._stream()), "examg_max_err_expr")
        return out

    def fill_expr(self, l, x, geom, expr, begin, end):
        check(self.L.examg_fill_expr(C.byref(l), self.ptr(x), C.byref(geom), C.byref(expr), ivec(begin), ivec(end), self._stream()),
              "examg_fill_expr")

    def apply_dirichlet_expr(self, l, x, geom, expr, face_mask: int):
        check(self.L.examg_apply_dirichlet_expr(C.byref(l), self.ptr(x), C.byref(geom), C.byref(expr), int(face_mask), self._stream()),
              "examg_apply_dirichlet_expr")

    def scalar_value(self, t) -> float:
        """Host value of a device scalar (the reference's 8-byte D2H copy after a reduction)."""
        return float(t.item())

    # -- boundary / init ----------------------------------------------------------------------------
    def fill_fn(self, l, x, geom, fn: int, params: Sequence[float], begin, end):
        self.fill_expr(l, x, geom, fn_expr(fn, params), begin, end)

    def fill_dup_faces(self, l, x, geom, fn: int, params: Sequence[float], face_mask: int):
        """`loop over F only dup [dir] on boundary { F = fn }` for every physical face of the mask, one launch."""
        check(self.L.examg_fill_dup_faces_expr(C.byref(l), self.ptr(x), C.byref(geom), C.byref(fn_expr(fn, params)), int(face_mask), self._stream()),
              "examg_fill_dup_faces_expr")

    def apply_dirichlet(self, l, x, geom, fn: int, params: Sequence[float], face_mask: int):
        self.apply_dirichlet_expr(l, x, geom, fn_expr(fn, params), face_mask)

    def init_varcoeff7(self, lc, cf, geom, coef_fn: int, params: Sequence[float], begin, end):
        check(self.L.examg_init_varcoeff7(C.byref(lc), self.ptr(cf), C.byref(geom), C.byref(fn_expr(coef_fn, params)),
                                          ivec(begin), ivec(end), self._stream()), "examg_init_varcoeff7")

    def init_helmholtz27(self, lc, cf, geom, coef_fn: int, params: Sequence[float], begin, end):
        prm = list(params) + [0.0, 0.0]
        check(self.L.examg_init_helmholtz27(C.byref(lc), self.ptr(cf), C.byref(geom), C.byref(fn_expr(coef_fn, params)), float(prm[1]),
                                            ivec(begin), ivec(end), self._stream()), "examg_init_helmholtz27")

    def transform_stencilfield(self, lc, nent: int, src, dst, to_entry_fastest: bool):
        """`transform <coefficient field> with [x, y, z, i] => [i, x, y, z]` (or its inverse) applied to the data."""
        check(self.L.examg_transform_stencilfield(C.byref(lc), int(nent), self.ptr(src), self.ptr(dst), 1 if to_entry_fastest else 0,
                                                  self._stream()), "examg_transform_stencilfield")

    def transform_field(self, lsrc, src, ldst, dst):
        """The same field under another layout transformation (`transform <field> with [x, y, z] => [x / 2, y, z, x % 2]` or back)."""
        check(self.L.examg_transform_field(C.byref(lsrc), self.ptr(src), C.byref(ldst), self.ptr(dst), self._stream()), "examg_transform_field")

    # -- halo ------------------------------------------------------------------------------------------
    def pack(self, l, x, buf, begin, end):
        check(self.L.examg_pack(C.byref(l), self.ptr(x), self.ptr(buf), ivec(begin), ivec(end), self._stream()), "examg_pack")

    def unpack(self, l, x, buf, begin, end):
        check(self.L.examg_unpack(C.byref(l), self.ptr(x), self.ptr(buf), ivec(begin), ivec(end), self._stream()),
              "examg_unpack")

    # -- coarse solve ---------------------------------------------------------------------------------
    def cg_coarse(self, lu, sol, lf, rhs, lr, res, lp, p, lq, ap, st: Stencil, geom, face_mask: int, max_it: int,
                  rel_tol: float, begin, end, info, flags: int = 0):
        """flags: CG_ALPHA_FROM_NORM | CG_NO_BC (include/examg.h: the layer-3 generator's form of the solver)."""
        sc = st.c_struct(self.ptr)
        check(self.L.examg_cg_coarse_variant(C.byref(lu), self.ptr(sol), C.byref(lf), self.ptr(rhs), C.byref(lr), self.ptr(res),
                                             C.byref(lp), self.ptr(p), C.byref(lq), self.ptr(ap), C.byref(sc), C.byref(geom),
                                             int(face_mask), int(max_it), float(rel_tol), ivec(begin), ivec(end), int(flags),
                                             self.ptr(info), self._stream()), "examg_cg_coarse_variant")

    def fill_random(self, x, seed: int):
        check(self.L.examg_fill_random(self.ptr(x), int(x.numel()), int(seed), self._stream()), "examg_fill_random")

    # -- external fields -------------------------------------------------------------------------------
    def copy_to_external(self, l_int, x_int, l_ext, dest):
        check(self.L.examg_copy_to_external(C.byref(l_int), self.ptr(x_int), C.byref(l_ext), self.ptr(dest), self._stream()),
              "examg_copy_to_external")

    def copy_from_external(self, l_ext, src, l_int, x_int):
        check(self.L.examg_copy_from_external(C.byref(l_ext), self.ptr(src), C.byref(l_int), self.ptr(x_int), self._stream()),
              "examg_copy_from_external")
