"""The C-ABI library loads without a GPU and exports every symbol include/sdvar_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "sdvar_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sdvar_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    from sdvar_amd import engine as E
    lib = E.load_library()
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/sdvar_hip.h but not exported by libsdvar_hip.so"
        assert s in E._SIGNATURES, f"{s} has no ctypes prototype in sdvar_amd/engine.py"
    assert set(E._SIGNATURES) == set(syms)
    assert lib.sdvar_abi_version() == 1


def test_argument_errors_are_reported_without_gpu():
    from sdvar_amd import engine as E
    lib = E.load_library()
    h = ctypes.c_void_p()
    d = E._ModelDesc()
    d.depth, d.n_stages, d.vocab, d.cvae, d.num_classes, d.max_batch, d.max_chunk_stages = 0, 10, 4096, 32, 1000, 1, 1
    rc = lib.sdvar_model_create(ctypes.byref(d), ctypes.byref(h))
    assert rc == 1 and b"depth" in lib.sdvar_last_error()
    assert lib.sdvar_kv_len(None) == -1


def test_missing_library_fails_loudly(tmp_path):
    from sdvar_amd import engine as E
    import pytest
    saved = E._lib
    E._lib = None
    try:
        with pytest.raises(E.SdvarError):
            E.load_library(str(tmp_path / "nope.so"))
    finally:
        E._lib = saved


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sdvar_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports the oracle"
