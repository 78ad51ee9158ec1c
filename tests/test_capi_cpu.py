"""CPU-side checks of the drop-in boundary: the HIP library loads without a GPU and exports
every symbol include/warpx_amd.h declares; argument validation happens before any launch."""
import ctypes as C
import os

import pytest

from warpx_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_capi.PRODUCT_LIB):
        import __graft_entry__ as g
        g.build()
    return _capi.load_product()


def test_every_declared_symbol_is_exported(lib):
    syms = _capi.declared_symbols(os.path.join(ROOT, "include", "warpx_amd.h"))
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib._dll, s)]
    assert not missing, missing


def test_binder_covers_the_header(lib):
    syms = set(_capi.declared_symbols(os.path.join(ROOT, "include", "warpx_amd.h")))
    bound = set(lib.names)
    assert syms <= bound, sorted(syms - bound)


def test_version_and_error_string(lib):
    assert b"gfx950" in lib.version()
    assert isinstance(lib.last_error(), bytes)


def test_invalid_arguments_are_rejected_before_launch(lib):
    from warpx_amd.containers import FieldArray, field_triplet
    # null pointers inside the views -> WXA_ERR_INVALID_ARG, never a launch
    f = (_capi.FieldView * 3)()
    rc = lib._evolve_b(f, f, 1e-16, (C.c_double * 3)(1, 1, 1), None)
    assert rc == -1
    assert b"bad field view" in lib.last_error()
    p = _capi.ParticleView()
    p.np = 10  # np > 0 with null arrays
    g = _capi.GridGeom()
    rc = lib._gather_push(C.byref(p), f, f, C.byref(g), 1.0, 1.0, 1.0, 3, 1, 0, None)
    assert rc == -1
    with pytest.raises(_capi.WxaError):
        lib.gather_push(C.byref(p), f, f, C.byref(g), 1.0, 1.0, 1.0, 7, 1, 0, None)


def test_product_does_not_reference_the_oracle():
    """The product path may not import, link or execute anything under oracle/."""
    import subprocess
    out = subprocess.run(["nm", "-D", _capi.PRODUCT_LIB], capture_output=True, text=True).stdout
    assert "orc_" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "warpx_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(root, fn), errors="ignore").read()
                assert "liboracle" not in text and "oracle_lib" not in text, fn
