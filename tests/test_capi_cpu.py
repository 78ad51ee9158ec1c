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


def test_ctypes_structs_match_the_header(tmp_path):
    """The ctypes mirrors in warpx_amd/_capi.py have the size and field offsets the C compiler gives the structs
    of include/warpx_amd.h (a field added on one side only would shift everything behind it)."""
    import ctypes as C
    import subprocess
    pairs = [("wxa_sim_config", _capi.SimConfig), ("wxa_field_view", _capi.FieldView),
             ("wxa_particle_view", _capi.ParticleView), ("wxa_grid_geom", _capi.GridGeom),
             ("wxa_moving_window", _capi.MovingWindow), ("wxa_plasma_injector", _capi.PlasmaInjector),
             ("wxa_laser_antenna", _capi.LaserAntenna), ("wxa_injected_momentum", _capi.InjectedMomentum), ("wxa_laser_push_params", _capi.LaserPushParams),
             ("wxa_comm", _capi.Comm)]
    lines = ['#include "warpx_amd.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(void) {"]
    for cname, cls in pairs:
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for fname, *_ in cls._fields_:
            lines.append(f'  printf(" %zu", offsetof({cname}, {fname}));')
        lines.append('  printf("\\n");')
    lines.append("  return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().splitlines()
    for (cname, cls), line in zip(pairs, out):
        parts = line.split()
        assert parts[0] == cname
        want = [int(v) for v in parts[1:]]
        got = [C.sizeof(cls)] + [getattr(cls, f[0]).offset for f in cls._fields_]
        assert got == want, (cname, got, want)
