"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/mfvi_hip.h declares, and the ctypes table binds exactly those (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mfvi_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mfvi_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "mfvi_forward" in syms and "mfvi_backward" in syms and "mfvi_kl" in syms and len(syms) >= 20


def test_library_exports_every_declared_symbol():
    import mfvi_dip_mia_amd as M
    M.build()
    lib = ctypes.CDLL(M._lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), "libmfvi_hip.so does not export " + s


def test_ctypes_table_matches_header():
    import mfvi_dip_mia_amd as M
    assert sorted(M._lib.SIGNATURES) == declared_symbols()
    L = M._lib.lib()
    assert L.mfvi_abi_version() == 6
    assert isinstance(L.mfvi_last_error(), bytes)


def test_missing_library_fails_loudly(monkeypatch):
    import mfvi_dip_mia_amd as M
    monkeypatch.setattr(M._lib, "_lib", None)
    monkeypatch.setattr(M._lib, "LIB_PATH", "/nonexistent/libmfvi_hip.so")
    with pytest.raises(M._lib.MfviError, match="no CPU fallback"):
        M._lib.lib()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(ROOT, "mfvi-dip-mia_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.lower() or f == "common.h" and "CPU oracle" in src, os.path.join(dp, f)
