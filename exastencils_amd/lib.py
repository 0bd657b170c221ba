"""ctypes binding of libexamg.so (the C ABI declared in include/examg.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950).
No fallback: a missing library raises ImportError-like RuntimeError at load().
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EXAMG_LIB", os.path.join(_HERE, "libexamg.so"))   # EXAMG_LIB: experimental builds (tools/)
MAXE = 27


class LayoutC(C.Structure):
    _fields_ = [("nd", C.c_int32)] + [
        (n, C.c_int32 * 3) for n in ("pad_l", "ghost_l", "dup_l", "inner", "dup_r", "ghost_r", "pad_r")
    ] + [("transform", C.c_int32)]


class StencilC(C.Structure):
    _fields_ = [
        ("nent", C.c_int32),
        ("diag", C.c_int32),
        ("off", (C.c_int32 * 3) * MAXE),
        ("coef", C.c_double * MAXE),
        ("cfield", C.c_void_p),
        ("clayout", LayoutC),
        ("ctransform", C.c_int32),
        ("wform", C.c_int32),
    ]


class GeomC(C.Structure):
    _fields_ = [("pos_begin", C.c_double * 3), ("h", C.c_double * 3)]


MAX_EXPR = 256
OPS = {"const": 0, "x": 1, "y": 2, "z": 3, "+": 4, "-": 5, "*": 6, "/": 7, "neg": 8, "sin": 9, "cos": 10, "exp": 11, "sinh": 12,
       "cosh": 13, "sqrt": 14, "pow": 15, "tan": 16, "log": 17, "fabs": 18, "max": 19, "min": 20, "tanh": 21}


class ExprC(C.Structure):
    """examg_expr_t: a postfix program over the node position (include/examg.h)."""
    _fields_ = [("n", C.c_int32), ("op", C.c_int32 * MAX_EXPR), ("c", C.c_double * MAX_EXPR)]

    @staticmethod
    def from_program(prog):
        """prog: list of (opcode name, constant or None)."""
        if not 1 <= len(prog) <= MAX_EXPR:
            raise ValueError("expression program with %d instructions (limit %d)" % (len(prog), MAX_EXPR))
        e = ExprC()
        e.n = len(prog)
        for i, (name, c) in enumerate(prog):
            e.op[i] = OPS[name]
            e.c[i] = float(c) if c is not None else 0.0
        e.program = list(prog)
        return e


class ExamgError(RuntimeError):
    pass


_lib = None
_libs = {}
DBG_LIB_PATH = os.path.join(_HERE, "libexamg_dbg.so")   # -DEXAMG_DEBUG_HOOKS build: variant selection for the parity tests

# every symbol include/examg.h declares
SYMBOLS = [
    "examg_version", "examg_last_error", "examg_device_count", "examg_stencil_op", "examg_jacobi",
    "examg_rbgs_colour", "examg_residual", "examg_rbgs_sweep_fused", "examg_rbgs_sweep_fused_boxes", "examg_jacobi2", "examg_jacobi3", "examg_rbgs_colours3", "examg_three_stage_eligible", "examg_jacobi2_boxes", "examg_jacobi_residual", "examg_rbgs_sweep_fused_prolong", "examg_jacobi2_prolong", "examg_rbgs_sweep_fused_zero", "examg_two_stage_eligible", "examg_restrict", "examg_residual_restrict", "examg_prolong_add",
    "examg_set", "examg_axpby", "examg_axpby_dev", "examg_reduce_work_bytes", "examg_dot", "examg_residual_norm2",
    "examg_fill_expr", "examg_apply_dirichlet_expr", "examg_fill_dup_faces_expr", "examg_max_err_expr", "examg_init_varcoeff7", "examg_init_helmholtz27", "examg_pack", "examg_unpack",
    "examg_cg_coarse", "examg_cg_coarse_variant", "examg_fill_random", "examg_copy_to_external", "examg_copy_from_external",
    "examg_comm_unique_id", "examg_comm_create", "examg_comm_destroy", "examg_comm_rank", "examg_comm_size",
    "examg_exchange_workspace_bytes", "examg_exchange", "examg_allreduce", "examg_allgather",
    "examg_jacobi2_blocks", "examg_rbgs_sweep_blocks", "examg_crand_seed", "examg_crand_draw_host",
    "examg_transform_stencilfield", "examg_transform_field", "examg_layout_size", "examg_comm_peer_release", "examg_residual_restrict_one_pass", "examg_residual_restrict_blocks", "examg_prolong_add_blocks",
    "examg_comm_create_peer", "examg_comm_peer_alloc", "examg_comm_peer_connect", "examg_comm_peer_slab_bytes",
    "examg_comm_peer_gather_bytes", "examg_comm_status",
]

COMM_ID_BYTES = 128
PEER_HANDLE_BYTES = 128
EXCH_DUP, EXCH_GHOST, EXCH_ALL, EXCH_CONCURRENT_AXES = 1, 2, 3, 4
PASS_TMP_PLANES_VALID = 8
CG_ALPHA_FROM_NORM, CG_NO_BC, CG_ZERO_START = 1, 2, 4


class CrandStateC(C.Structure):
    """examg_crand_state_t"""
    _fields_ = [("r", C.c_uint32 * 31), ("k", C.c_int32)]


class NeighborsC(C.Structure):
    """examg_neighbors_t"""
    _fields_ = [("rank", (C.c_int32 * 2) * 3)]


def load(path=None):
    """Load libexamg.so (or the library at `path`) and declare the prototypes.  Raises if the HIP library is not built."""
    global _lib
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ExamgError(
            "%s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback" % path)
    # torch first: libexamg.so needs libamdhip64.so.7 and must bind to the one HIP runtime of the process,
    # the copy PyTorch ships and loads (two runtimes in one process do not both see the device)
    import torch  # noqa: F401

    L = C.CDLL(path)
    vp, ip = C.c_void_p, C.POINTER(C.c_int32)
    lp, sp, gp, dp = C.POINTER(LayoutC), C.POINTER(StencilC), C.POINTER(GeomC), C.POINTER(C.c_double)
    L.examg_version.restype = C.c_int
    L.examg_last_error.restype = C.c_char_p
    L.examg_device_count.restype = C.c_int
    L.examg_stencil_op.argtypes = [C.c_int, lp, vp, lp, vp, lp, vp, sp, C.c_double, C.c_int, ip, ip, vp]
    L.examg_jacobi.argtypes = [lp, vp, vp, lp, vp, sp, C.c_double, ip, ip, vp]
    L.examg_rbgs_colour.argtypes = [lp, vp, lp, vp, sp, C.c_double, C.c_int, ip, ip, vp]
    L.examg_residual.argtypes = [lp, vp, lp, vp, lp, vp, sp, ip, ip, vp]
    L.examg_rbgs_sweep_fused.argtypes = [lp, vp, vp, lp, vp, sp, C.c_double, C.c_int, ip, ip, vp]
    L.examg_jacobi2.argtypes = [lp, vp, vp, vp, lp, vp, sp, C.c_double, ip, ip, vp]
    L.examg_jacobi3.argtypes = [lp, vp, vp, vp, lp, vp, sp, C.c_double, ip, ip, vp]
    L.examg_rbgs_colours3.argtypes = [lp, vp, vp, lp, vp, sp, C.c_double, C.c_int, ip, ip, vp]
    L.examg_three_stage_eligible.argtypes = [lp, lp, sp, ip, ip]
    L.examg_jacobi_residual.argtypes = [lp, vp, vp, lp, vp, lp, vp, sp, C.c_double, ip, ip, vp]
    L.examg_jacobi2_boxes.argtypes = [lp, vp, vp, vp, lp, vp, sp, C.c_double, ip, ip, ip, ip, vp]
    L.examg_rbgs_sweep_fused_boxes.argtypes = [lp, vp, vp, vp, lp, vp, sp, C.c_double, C.c_int, ip, ip, ip, ip, vp]
    L.examg_rbgs_sweep_fused_prolong.argtypes = [lp, vp, vp, lp, vp, sp, C.c_double, C.c_int, ip, ip, lp, vp, vp]
    L.examg_rbgs_sweep_fused_zero.argtypes = [lp, vp, lp, vp, sp, C.c_double, C.c_int, ip, ip, vp]
    L.examg_jacobi2_prolong.argtypes = [lp, vp, vp, vp, lp, vp, sp, C.c_double, ip, ip, lp, vp, vp]
    L.examg_two_stage_eligible.argtypes = [lp, lp, sp, ip, ip, ip, ip]
    L.examg_restrict.argtypes = [lp, vp, lp, vp, C.c_double, ip, ip, vp]
    L.examg_residual_restrict.argtypes = [lp, vp, lp, vp, lp, vp, sp, lp, vp, C.c_double, ip, ip, ip, ip, vp]
    L.examg_prolong_add.argtypes = [lp, vp, lp, vp, ip, ip, vp]
    L.examg_set.argtypes = [lp, vp, C.c_double, ip, ip, vp]
    L.examg_axpby.argtypes = [lp, vp, lp, vp, C.c_double, C.c_double, ip, ip, vp]
    L.examg_axpby_dev.argtypes = [lp, vp, lp, vp, C.c_double, C.c_double, C.c_int, C.c_double, vp, vp, ip, ip, vp]
    L.examg_reduce_work_bytes.restype = C.c_size_t
    L.examg_dot.argtypes = [lp, vp, lp, vp, ip, ip, vp, vp, vp]
    L.examg_residual_norm2.argtypes = [lp, vp, lp, vp, sp, ip, ip, lp, vp, vp, vp, vp]
    ep = C.POINTER(ExprC)
    L.examg_fill_expr.argtypes = [lp, vp, gp, ep, ip, ip, vp]
    L.examg_apply_dirichlet_expr.argtypes = [lp, vp, gp, ep, C.c_uint32, vp]
    L.examg_fill_dup_faces_expr.argtypes = [lp, vp, gp, ep, C.c_uint32, vp]
    L.examg_max_err_expr.argtypes = [lp, vp, gp, ep, ip, ip, vp, vp, vp]
    L.examg_init_varcoeff7.argtypes = [lp, vp, gp, ep, ip, ip, vp]
    L.examg_init_helmholtz27.argtypes = [lp, vp, gp, ep, C.c_double, ip, ip, vp]
    L.examg_transform_stencilfield.argtypes = [lp, C.c_int, vp, vp, C.c_int, vp]
    L.examg_transform_field.argtypes = [lp, vp, lp, vp, vp]
    L.examg_layout_size.argtypes = [lp]
    L.examg_layout_size.restype = C.c_int64
    L.examg_pack.argtypes = [lp, vp, vp, ip, ip, vp]
    L.examg_unpack.argtypes = [lp, vp, vp, ip, ip, vp]
    L.examg_cg_coarse.argtypes = [lp, vp, lp, vp, lp, vp, lp, vp, lp, vp, sp, gp, C.c_uint32, C.c_int, C.c_double, ip, ip, vp, vp]
    L.examg_cg_coarse_variant.argtypes = [lp, vp, lp, vp, lp, vp, lp, vp, lp, vp, sp, gp, C.c_uint32, C.c_int, C.c_double, ip, ip,
                                          C.c_uint32, vp, vp]
    L.examg_copy_to_external.argtypes = [lp, vp, lp, vp, vp]
    L.examg_copy_from_external.argtypes = [lp, vp, lp, vp, vp]
    L.examg_fill_random.argtypes = [vp, C.c_int64, C.c_uint64, vp]
    L.examg_comm_unique_id.argtypes = [vp]
    L.examg_comm_create.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int]
    L.examg_comm_destroy.argtypes = [vp]
    L.examg_comm_rank.argtypes = [vp]
    L.examg_comm_size.argtypes = [vp]
    L.examg_exchange_workspace_bytes.argtypes = [lp]
    L.examg_exchange_workspace_bytes.restype = C.c_size_t
    L.examg_exchange.argtypes = [vp, lp, vp, C.POINTER(NeighborsC), C.c_int, vp, C.c_size_t, vp]
    L.examg_allreduce.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    nbp = C.POINTER(NeighborsC)
    L.examg_crand_seed.argtypes = [C.POINTER(CrandStateC), C.c_uint32]
    L.examg_crand_draw_host.argtypes = [C.POINTER(CrandStateC), vp, C.c_int64]
    L.examg_jacobi2_blocks.argtypes = [vp, nbp, lp, vp, vp, vp, lp, vp, sp, C.c_double, ip, ip, C.c_int, vp, C.c_size_t, C.c_int, vp]
    L.examg_rbgs_sweep_blocks.argtypes = [vp, nbp, lp, vp, vp, vp, lp, vp, sp, C.c_double, C.c_int, ip, ip, C.c_int, vp, C.c_size_t, C.c_int, vp]
    L.examg_allgather.argtypes = [vp, vp, vp, C.c_int64, vp]
    L.examg_residual_restrict_one_pass.argtypes = [lp, lp, sp, lp, ip, ip, ip, ip]
    L.examg_residual_restrict_blocks.argtypes = [vp, nbp, lp, vp, lp, vp, lp, vp, sp, lp, vp, C.c_double, ip, ip, ip, ip, C.c_int, vp, C.c_size_t,
                                                 C.c_int, vp]
    L.examg_prolong_add_blocks.argtypes = [vp, nbp, lp, vp, lp, vp, ip, ip, C.c_int, vp, C.c_size_t, C.c_int, vp]
    L.examg_comm_create_peer.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    L.examg_comm_peer_alloc.argtypes = [vp, C.c_size_t, C.c_size_t, vp]
    L.examg_comm_peer_connect.argtypes = [vp, vp]
    L.examg_comm_peer_slab_bytes.argtypes = [vp]
    L.examg_comm_peer_slab_bytes.restype = C.c_size_t
    L.examg_comm_peer_gather_bytes.argtypes = [vp]
    L.examg_comm_peer_gather_bytes.restype = C.c_size_t
    L.examg_comm_status.argtypes = [vp, vp]
    for name in SYMBOLS:
        fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
        if name not in ("examg_version", "examg_last_error", "examg_device_count", "examg_reduce_work_bytes", "examg_exchange_workspace_bytes",
                        "examg_comm_peer_slab_bytes", "examg_comm_peer_gather_bytes"):
            fn.restype = C.c_int
    if hasattr(L, "examg_debug_force_generic"):      # debug build only
        L.examg_debug_force_generic.argtypes = [C.c_int]
        L.examg_debug_force_generic.restype = C.c_int
    _libs[path] = L
    if path == LIB_PATH:
        _lib = L
    return L


def check(rc: int, what: str = ""):
    if rc != 0:
        raise ExamgError("%s failed (%d): %s" % (what or "libexamg call", rc, load().examg_last_error().decode()))


def ivec(v):
    return (C.c_int32 * 3)(*[int(x) for x in v])


def dvec4(v):
    v = list(v) + [0.0] * (4 - len(v))
    return (C.c_double * 4)(*v[:4])
