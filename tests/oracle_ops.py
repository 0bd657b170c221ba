"""TEST-ONLY kernel layer backed by the CPU oracle (oracle/examg_oracle.c) on CPU torch tensors.

It lets the product's host logic (exastencils_amd.solver / comm / domain) run without a GPU -- in the
`-m "not gpu"` suite and in the world_size-2 gloo tests -- with the oracle standing in for the HIP
kernels.  It is never imported by the package itself.
"""
import ctypes as C

import numpy as np
import torch

from oracle import mg


def _lp(l):
    return C.cast(C.pointer(l), C.POINTER(mg.LayoutC))


def _gp(g):
    return C.cast(C.pointer(g), C.POINTER(mg.GeomC))


def _iv(v):
    return (C.c_int * 3)(*[int(x) for x in v])


def _p4(params):
    v = list(params) + [0.0] * (4 - len(params))
    return (C.c_double * 4)(*v[:4])


class OracleOps:
    name = "oracle"

    def __init__(self):
        self.L = mg.lib()
        self.torch = torch
        self.device = torch.device("cpu")

    def new_array(self, n):
        return torch.zeros(int(n), dtype=torch.float64)

    def new_scalar(self):
        return torch.zeros(1, dtype=torch.float64)

    @staticmethod
    def ptr(t):
        return t.data_ptr()

    def synchronize(self):
        pass

    def to_host(self, t):
        return t.numpy()

    def from_host(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def _st(self, st):
        s = mg.StencilC()
        s.nent = len(st.offsets)
        s.diag = st.diag_index
        for k, o in enumerate(st.offsets):
            for d in range(3):
                s.off[k][d] = o[d]
            s.coef[k] = st.coefs[k] if st.coefs else 0.0
        if st.cfield is not None:
            s.cfield = st.cfield.data_ptr()
            lc = st.clayout.c_struct()
            C.memmove(C.byref(s.clayout), C.byref(lc), C.sizeof(mg.LayoutC))
        else:
            s.cfield = None
        s.wform = int(getattr(st, "wform", 0))
        return s

    def stencil_op(self, mode, lu, u, lf, rhs, ld, dst, st, w, colour, begin, end):
        sc = self._st(st)
        self.L.orc_stencil_op(mode, _lp(lu), self.ptr(u), _lp(lf) if lf is not None else None,
                              self.ptr(rhs) if rhs is not None else None, _lp(ld), self.ptr(dst), C.byref(sc), float(w),
                              int(colour), _iv(begin), _iv(end))

    def rbgs_sweep_fused(self, lu, u_in, u_out, lf, rhs, st, w, first, begin, end):
        u_out.copy_(u_in)
        for c in (first, 1 - first):
            self.stencil_op(2, lu, u_out, lf, rhs, lu, u_out, st, w, c, begin, end)

    def rbgs_sweep_fused_boxes(self, lu, u_in, u_out, tmp, lf, rhs, st, w, first, begin1, end1, begin2, end2):
        tmp.copy_(u_in)
        self.stencil_op(2, lu, tmp, lf, rhs, lu, tmp, st, w, first, begin1, end1)
        self.axpby(lu, tmp, lu, u_out, 1.0, 0.0, begin2, end2)
        self.stencil_op(2, lu, tmp, lf, rhs, lu, u_out, st, w, 1 - first, begin2, end2)

    def jacobi2(self, lu, u_in, u_out, tmp, lf, rhs, st, w, begin, end):
        self.stencil_op(2, lu, u_in, lf, rhs, lu, tmp, st, w, -1, begin, end)
        self.stencil_op(2, lu, tmp, lf, rhs, lu, u_out, st, w, -1, begin, end)

    def jacobi3(self, lu, u_in, u_out, tmp, lf, rhs, st, w, begin, end):
        """Three loops one after the other; outside the box u_out keeps what it held (the shells the steps read are u_in's)."""
        a, b = u_in.clone(), u_in.clone()
        self.stencil_op(2, lu, u_in, lf, rhs, lu, a, st, w, -1, begin, end)
        self.stencil_op(2, lu, a, lf, rhs, lu, b, st, w, -1, begin, end)
        self.stencil_op(2, lu, b, lf, rhs, lu, a, st, w, -1, begin, end)
        self.axpby(lu, a, lu, u_out, 1.0, 0.0, begin, end)

    def rbgs_colours3(self, lu, u_in, u_out, lf, rhs, st, w, first, begin, end):
        work = u_in.clone()
        for c in (first, 1 - first, first):
            self.stencil_op(2, lu, work, lf, rhs, lu, work, st, w, c, begin, end)
        self.axpby(lu, work, lu, u_out, 1.0, 0.0, begin, end)

    def jacobi_residual(self, lu, u_in, u_out, lf, rhs, lr, res, st, w, begin, end):
        self.stencil_op(2, lu, u_in, lf, rhs, lu, u_out, st, w, -1, begin, end)
        self.stencil_op(1, lu, u_out, lf, rhs, lr, res, st, 0.0, -1, begin, end)

    def jacobi2_boxes(self, lu, u_in, u_out, tmp, lf, rhs, st, w, begin1, end1, begin2, end2):
        tmp.copy_(u_in)
        self.stencil_op(2, lu, u_in, lf, rhs, lu, tmp, st, w, -1, begin1, end1)
        self.stencil_op(2, lu, tmp, lf, rhs, lu, u_out, st, w, -1, begin2, end2)

    def rbgs_sweep_fused_prolong(self, lu, u_in, u_out, lf, rhs, st, w, first, begin, end, lc, uc):
        u_out.copy_(u_in)
        self.prolong_add(lc, uc, lu, u_out, begin, end)
        for c in (first, 1 - first):
            self.stencil_op(2, lu, u_out, lf, rhs, lu, u_out, st, w, c, begin, end)

    def rbgs_sweep_fused_zero(self, lu, u_out, lf, rhs, st, w, first, begin, end):
        u_out.zero_()
        for c in (first, 1 - first):
            self.stencil_op(2, lu, u_out, lf, rhs, lu, u_out, st, w, c, begin, end)

    def jacobi2_prolong(self, lu, u_in, u_out, tmp, lf, rhs, st, w, begin, end, lc, uc):
        u_out.copy_(u_in)
        tmp.copy_(u_in)
        self.prolong_add(lc, uc, lu, u_out, begin, end)
        self.stencil_op(2, lu, u_out, lf, rhs, lu, tmp, st, w, -1, begin, end)
        self.stencil_op(2, lu, tmp, lf, rhs, lu, u_out, st, w, -1, begin, end)

    def restrict(self, lfine, rf, lc, fc, scale, begin, end):
        self.L.orc_restrict(_lp(lfine), self.ptr(rf), _lp(lc), self.ptr(fc), float(scale), _iv(begin), _iv(end))

    def residual_restrict(self, lu, u, lf, rhs, lr, res, st, lc, fc, scale, fbegin, fend, cbegin, cend):
        self.stencil_op(1, lu, u, lf, rhs, lr, res, st, 0.0, -1, fbegin, fend)
        self.restrict(lr, res, lc, fc, scale, cbegin, cend)

    def prolong_add(self, lc, uc, lfine, uf, begin, end):
        self.L.orc_prolong_add(_lp(lc), self.ptr(uc), _lp(lfine), self.ptr(uf), _iv(begin), _iv(end))

    def set(self, l, x, v, begin, end):
        self.L.orc_set(_lp(l), self.ptr(x), float(v), _iv(begin), _iv(end))

    def axpby(self, lx, x, ly, y, a, b, begin, end):
        self.L.orc_axpby(_lp(lx), self.ptr(x), _lp(ly), self.ptr(y), float(a), float(b), _iv(begin), _iv(end))

    def dot(self, lx, x, ly, y, begin, end, out=None):
        out = self.new_scalar() if out is None else out
        out[0] = self.L.orc_dot(_lp(lx), self.ptr(x), _lp(ly), self.ptr(y), _iv(begin), _iv(end))
        return out

    def residual_norm2(self, lu, u, lf, rhs, st, begin, end, lr=None, res=None, out=None):
        import torch

        n = 1
        for d in range(3):
            n *= sum(getattr(lu, k)[d] for k in ("pad_l", "ghost_l", "dup_l", "inner", "dup_r", "ghost_r", "pad_r"))
        tmp = torch.zeros(n, dtype=torch.float64)
        self.stencil_op(1, lu, u, lf, rhs, lu, tmp, st, 0.0, -1, begin, end)
        return self.dot(lu, tmp, lu, tmp, begin, end, out)

    def cg_coarse(self, lu, sol, lf, rhs, lr, res, lp, p, lq, ap, st, geom, face_mask, max_it, rel_tol, begin, end, info, flags=0):
        """The statements examg_cg_coarse fuses (Benchmark/Poisson3D/3D_FD_Poisson_fromL4.exa4:152-201), one oracle loop each;
        flags 1: alpha = res * res / alphaDenom, 2: no `apply bc` (the layer-3 generator's form, Testing/Smoothers/Jac.exa4:75-109), 4: the zero field
        as the start."""
        import math

        def bc(l, x):
            if face_mask and not (flags & 2):
                self.apply_dirichlet(l, x, geom, 0, (), face_mask)

        def norm():
            return math.sqrt(self.scalar_value(self.dot(lr, res, lr, res, begin, end)))

        if flags & 4:        # EXAMG_CG_ZERO_START: `Solution = 0` rides along
            self.set(lu, sol, 0.0, begin, end)
        self.stencil_op(1, lu, sol, lf, rhs, lr, res, st, 0.0, -1, begin, end)
        bc(lr, res)
        cur = init = norm()
        self.axpby(lr, res, lp, p, 1.0, 0.0, begin, end)
        bc(lp, p)
        steps = max_it
        for it in range(max_it):
            self.stencil_op(0, lp, p, None, None, lq, ap, st, 0.0, -1, begin, end)
            num = cur * cur if flags & 1 else self.scalar_value(self.dot(lr, res, lr, res, begin, end))
            den = self.scalar_value(self.dot(lp, p, lq, ap, begin, end))
            alpha = num / den if den != 0.0 else float("nan")
            self.axpby(lp, p, lu, sol, alpha, 1.0, begin, end)
            bc(lu, sol)
            self.axpby(lq, ap, lr, res, -alpha, 1.0, begin, end)
            bc(lr, res)
            nxt = norm()
            if nxt <= rel_tol * init:
                steps = it + 1
                break
            beta = (nxt * nxt) / (cur * cur)
            self.axpby(lr, res, lp, p, 1.0, beta, begin, end)
            bc(lp, p)
            cur = nxt
        else:
            info[3] += 1           # the loop ran out: the statement after it prints "Maximum number of cgs iterations ... exceeded"
        info[0] = steps

    def max_err_fn(self, l, x, geom, fn, params, begin, end, out=None):
        out = self.new_scalar() if out is None else out
        out[0] = self.L.orc_max_err_fn(_lp(l), self.ptr(x), _gp(geom), int(fn), _p4(params), _iv(begin), _iv(end))
        return out

    # -- expression programs (include/examg.h EXAMG_OP_*): evaluated with numpy, test-only -----------------------------
    @staticmethod
    def _eval_program(prog, x, y, z):
        import numpy as np

        un = {"neg": np.negative, "sin": np.sin, "cos": np.cos, "exp": np.exp, "sinh": np.sinh, "cosh": np.cosh, "sqrt": np.sqrt,
              "tan": np.tan, "log": np.log, "fabs": np.abs, "tanh": np.tanh}
        bi = {"+": np.add, "-": np.subtract, "*": np.multiply, "/": np.divide, "pow": np.power, "max": np.maximum, "min": np.minimum}
        st = []
        for name, c in prog:
            if name == "const":
                st.append(np.full_like(x, c, dtype=np.float64) + 0.0 * x)
            elif name in ("x", "y", "z"):
                st.append({"x": x, "y": y, "z": z}[name] + 0.0 * x)
            elif name in un:
                st[-1] = un[name](st[-1])
            else:
                b = st.pop()
                st[-1] = bi[name](st[-1], b)
        return st[0]

    def _box_points(self, l, geom, begin, end):
        import numpy as np

        i2, i1, i0 = np.meshgrid(np.arange(begin[2], end[2]), np.arange(begin[1], end[1]), np.arange(begin[0], end[0]), indexing="ij")
        x = i0 * geom.h[0] + geom.pos_begin[0]
        y = i1 * geom.h[1] + geom.pos_begin[1]
        z = i2 * geom.h[2] + geom.pos_begin[2]
        ref = [l.pad_l[d] + l.ghost_l[d] for d in range(3)]
        tot = [l.pad_l[d] + l.ghost_l[d] + l.dup_l[d] + l.inner[d] + l.dup_r[d] + l.ghost_r[d] + l.pad_r[d] for d in range(3)]
        sl = tuple(slice(begin[d] + ref[d], end[d] + ref[d]) for d in (2, 1, 0))
        return x, y, z, sl, (tot[2], tot[1], tot[0])

    def fill_expr(self, l, x, geom, expr, begin, end):
        if any(end[d] <= begin[d] for d in range(3)):
            return
        px, py, pz, sl, shape = self._box_points(l, geom, begin, end)
        x.numpy().reshape(shape)[sl] = self._eval_program(expr.program, px, py, pz)

    def max_err_expr(self, l, x, geom, expr, begin, end, out=None):
        import numpy as np

        out = self.new_scalar() if out is None else out
        out[0] = 0.0
        if all(end[d] > begin[d] for d in range(3)):
            px, py, pz, sl, shape = self._box_points(l, geom, begin, end)
            out[0] = float(np.max(np.abs(x.numpy().reshape(shape)[sl] - self._eval_program(expr.program, px, py, pz))))
        return out

    def apply_dirichlet_expr(self, l, x, geom, expr, face_mask):
        nd = l.nd
        for d in range(nd):
            for side in (0, 1):
                if not face_mask & (1 << (2 * d + side)):
                    continue
                b, e = [0, 0, 0], [1, 1, 1]
                for t in range(nd):
                    if t == d:
                        b[t], e[t] = (0, l.dup_l[t]) if side == 0 else (l.dup_l[t] + l.inner[t], l.dup_l[t] + l.inner[t] + l.dup_r[t])
                    else:
                        b[t], e[t] = -l.ghost_l[t], l.dup_l[t] + l.inner[t] + l.dup_r[t] + l.ghost_r[t]
                self.fill_expr(l, x, geom, expr, b, e)

    def scalar_value(self, t):
        return float(t.item())

    def fill_fn(self, l, x, geom, fn, params, begin, end):
        self.L.orc_fill_fn(_lp(l), self.ptr(x), _gp(geom), int(fn), _p4(params), _iv(begin), _iv(end))

    def fill_dup_faces(self, l, x, geom, fn, params, face_mask):
        """One fill per face: duplicate plane, tangentially DLB..DRE (`only dup [dir] on boundary`)."""
        nd = l.nd
        for d in range(nd):
            for side in (0, 1):
                if not (face_mask >> (2 * d + side)) & 1:
                    continue
                b, e = [0, 0, 0], [1, 1, 1]
                for t in range(nd):
                    if t == d:
                        b[t] = 0 if side == 0 else l.dup_l[t] + l.inner[t]
                        e[t] = b[t] + (l.dup_l[t] if side == 0 else l.dup_r[t])
                    else:
                        b[t], e[t] = 0, l.dup_l[t] + l.inner[t] + l.dup_r[t]
                self.fill_fn(l, x, geom, fn, params, b, e)

    def apply_dirichlet(self, l, x, geom, fn, params, face_mask):
        nd = l.nd
        for d in range(nd):
            for side in (0, 1):
                if not (face_mask >> (2 * d + side)) & 1:
                    continue
                b, e = [0, 0, 0], [1, 1, 1]
                for t in range(nd):
                    if t == d:
                        if side == 0:
                            b[t], e[t] = 0, l.dup_l[t]
                        else:
                            b[t] = l.dup_l[t] + l.inner[t]
                            e[t] = b[t] + l.dup_r[t]
                    else:
                        b[t] = -l.ghost_l[t]
                        e[t] = l.dup_l[t] + l.inner[t] + l.dup_r[t] + l.ghost_r[t]
                self.fill_fn(l, x, geom, fn, params, b, e)

    def init_varcoeff7(self, lc, cf, geom, coef_fn, params, begin, end):
        self.L.orc_init_varcoeff7(_lp(lc), self.ptr(cf), _gp(geom), int(coef_fn), _p4(params), _iv(begin), _iv(end))

    def init_helmholtz27(self, lc, cf, geom, coef_fn, params, begin, end):
        self.L.orc_init_helmholtz27(_lp(lc), self.ptr(cf), _gp(geom), int(coef_fn), _p4(params), _iv(begin), _iv(end))

    def pack(self, l, x, buf, begin, end):
        self.L.orc_pack(_lp(l), self.ptr(x), self.ptr(buf), _iv(begin), _iv(end))

    def unpack(self, l, x, buf, begin, end):
        self.L.orc_unpack(_lp(l), self.ptr(x), self.ptr(buf), _iv(begin), _iv(end))

    def fill_random(self, x, seed):
        self.L.orc_fill_random(self.ptr(x), int(x.numel()), int(seed))
