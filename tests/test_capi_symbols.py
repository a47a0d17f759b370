"""CPU checks of the boundary: the gfx950 library builds, loads, and exports every symbol include/dicey_gpu.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dicey_amd", "csrc"), "-s", "-j4"])
    from dicey_amd import _capi
    return _capi.load()


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "dicey_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dg_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported(lib):
    from dicey_amd import _capi
    names = declared_symbols()
    assert names, "header parse failed"
    assert sorted(_capi.SYMBOLS) == names
    for n in names:
        assert hasattr(lib, n), n


def test_abi_version_and_error_path(lib):
    assert lib.dg_abi_version() == 7
    h = ctypes.c_void_p()
    rc = lib.dg_index_open(b"/nonexistent/x.fm9", 0, 0, ctypes.byref(h))
    assert rc != 0 and not h.value  # DG_ENODEV here (no GPU) or DG_EIO on a GPU box; never a silent success
    assert lib.dg_last_error()


def test_package_has_no_fallback_when_library_is_missing(tmp_path, monkeypatch):
    from dicey_amd import _capi
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_capi, "_lib", None)
    with pytest.raises(ImportError):
        _capi.load()


def test_product_sources_never_touch_the_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "dicey_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(d, f), errors="replace").read()
                if "oracle" in txt.lower():
                    bad.append(os.path.join(d, f))
    assert not bad, bad
