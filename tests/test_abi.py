"""CPU-side checks of the drop-in boundary: libexamg.so loads and exports every symbol include/examg.h
declares (no compute calls here -- there is no GPU in the CPU suite), and the header/struct mirror agree."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "examg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(examg_[a-z0-9_]+)\s*\(", src)))


def _exported(path):
    """Dynamic symbols a shared library defines whose names start with examg_ (C linkage)."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({line.split()[-1] for line in out.splitlines() if line.split() and line.split()[-1].startswith("examg_")})


def test_library_exports_exactly_the_declared_symbols():
    """Both directions: every function include/examg.h declares is exported, and the product library exports no examg_*
    symbol the header does not declare (variant-selection hooks live in the debug build only)."""
    import __graft_entry__ as ge

    ge.build_examg()
    from exastencils_amd import lib

    L = C.CDLL(lib.LIB_PATH)
    declared = _header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), "libexamg.so does not export %s" % name
    assert sorted(lib.SYMBOLS) == declared, "exastencils_amd.lib.SYMBOLS is out of sync with include/examg.h"
    assert _exported(lib.LIB_PATH) == declared, "libexamg.so exports symbols include/examg.h does not declare"


def test_debug_build_adds_only_debug_hooks():
    import __graft_entry__ as ge

    ge.build_examg_dbg()
    from exastencils_amd import lib

    extra = sorted(set(_exported(lib.DBG_LIB_PATH)) - set(_header_functions()))
    assert extra and all(name.startswith("examg_debug_") for name in extra), extra


def test_struct_mirrors_match_header_sizes():
    from exastencils_amd import lib

    assert C.sizeof(lib.LayoutC) == 4 * 23          # nd, 7 x [3], transform
    assert C.sizeof(lib.GeomC) == 8 * 6
    # nent, diag, off[27][3] (int32) | coef[27] (double, 8-aligned) | pointer | layout (+ tail padding)
    assert lib.StencilC.coef.offset == 4 * (2 + 81) + 4
    assert lib.StencilC.cfield.offset == lib.StencilC.coef.offset + 8 * 27


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from exastencils_amd import lib

    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        lib.load()
    except lib.ExamgError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when libexamg.so is absent")


def test_hipops_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        return
    from exastencils_amd.lib import ExamgError
    from exastencils_amd.ops import HipOps

    try:
        HipOps()
    except ExamgError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("HipOps() must raise without a GPU")


def test_shim_host_references_only_reference_names():
    """shim/exa_poisson3d_host.cpp -- the generated-style host -- must bind nothing of libexamg but the communicator bootstrap
    of main(): every loop goes through <fn>_<L>_k<NNN>_wrapper / exch<Field>_<L> / applyBCs<Field>_<L> names."""
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        obj = os.path.join(td, "host.o")
        subprocess.run(["g++", "-c", "-x", "c++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "shim"), "-DEXA_MIN_LEVEL=2", "-DEXA_MAX_LEVEL=6",
                        os.path.join(ROOT, "shim", "exa_poisson3d_host.cpp"), "-o", obj], check=True, capture_output=True)
        und = subprocess.run(["nm", "-u", obj], capture_output=True, text=True, check=True).stdout.split()
    lib_syms = sorted(s for s in und if s.startswith("examg_"))
    assert set(lib_syms) <= {"examg_comm_unique_id", "examg_last_error", "examg_device_count"}, lib_syms
    wrappers = [s for s in und if s.endswith("_wrapper")]
    assert any(s.startswith("mgCycle_6_k00") for s in wrappers) and any(s.startswith("ResNorm_") for s in wrappers)
    assert any(s.startswith("exchSolution_") for s in und) and any(s.startswith("applyBCsSolution_") for s in und)
