"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol that
include/gigl_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

from gigl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gigl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gigl_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.gigl_version() >= 100


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    ctx = ctypes.c_void_p()
    rc = lib.gigl_ctx_create(0, ctypes.byref(ctx))
    assert rc == -5 and not ctx.value  # GIGL_E_NO_DEVICE, never a CPU fallback
    from gigl_amd.engine import HipEngine
    with pytest.raises(RuntimeError):
        HipEngine(0)


def test_product_package_never_imports_oracle():
    """the oracle is test infrastructure: nothing under gigl_amd/ may import, link or exec it"""
    for dp, _, files in os.walk(os.path.join(ROOT, "gigl_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "gigl_oracle" not in src and "libgigl_oracle" not in src, f
