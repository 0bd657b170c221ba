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


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge

    ge.build_examg()
    from exastencils_amd import lib

    L = C.CDLL(lib.LIB_PATH)
    declared = _header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), "libexamg.so does not export %s" % name
    assert sorted(lib.SYMBOLS) == declared, "exastencils_amd.lib.SYMBOLS is out of sync with include/examg.h"


def test_struct_mirrors_match_header_sizes():
    from exastencils_amd import lib

    assert C.sizeof(lib.LayoutC) == 4 * 22
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
