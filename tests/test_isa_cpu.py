"""Checks on the gfx950 ISA of the tile kernels (hipcc cross-compiles here; no GPU needed).

Round 4 found the fp32 order-2 deposition build wrong on the MI355X while the same source passed on the CPU execution
model: phase D chose between three component bodies with a wave-uniform scalar, the compiler merged the last `ds_add` of
the three into one shared block, and the register holding that block's LDS address was an implicit-def on the edge coming
from the third body (the last deposit of every jz stencil went to a stale address).  The kernels no longer contain such a
choice (one pass per component, compile-time constants); this test keeps it that way: no basic block of the tile kernels
may consist of a lone LDS / global atomic whose address is a bare register -- the shape of that merged tail."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ATOMIC = re.compile(r"^\s*(ds_add_(rtn_)?f(32|64)|global_atomic_add_f(32|64))\s+v\d+(, v\[\d+:\d+\]|, v\d+)(, off)?\s*$")


def _merged_tails(asm):
    found, kernel, label_at = [], "", -10
    for n, line in enumerate(asm.splitlines()):
        if line.startswith("_ZN3wxa") and line.rstrip().endswith(":") or line.startswith("_ZN3wxa") and "; @" in line:
            kernel = line.split(":")[0][:90]
        if line.startswith(".LBB"):
            label_at = n
        elif ATOMIC.match(line) and n - label_at <= 3:
            found.append((kernel, n + 1, line.strip()))
    return found


def test_the_detector_sees_the_shape_it_looks_for():
    asm = "_ZN3wxa1kEv: ; @_ZN3wxa1kEv\n.LBB2_330:\n\tv_cvt_f32_f64_e32 v2, v[28:29]\n\tds_add_f32 v68, v2\n.LBB2_331:\n\tds_add_f32 v64, v72 offset:14880\n"
    assert [f[2] for f in _merged_tails(asm)] == ["ds_add_f32 v68, v2"]


@pytest.mark.skipif(not shutil.which(HIPCC) and not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src", ["deposit_tile.hip", "gather_tile.hip"])
def test_no_atomic_in_a_shared_tail_block(tmp_path, src):
    out = tmp_path / (src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "--cuda-device-only", "-S",
           os.path.join(ROOT, "warpx_amd", "csrc", src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    found = _merged_tails(out.read_text())
    assert not found, found[:5]
