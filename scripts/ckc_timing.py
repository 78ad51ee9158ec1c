"""EvolveB with the CKC stencil at 256^3: the plain kernel (WXA_CKC_PLAIN=1) against the LDS-tiled one, back-to-back
launches between one pair of events; ms per launch and the fraction of the 8 TB/s HBM peak that the algorithmic
72 B per cell reach. 
    python scripts/ckc_timing.py [ncell] [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from warpx_amd import _capi, load_product
from warpx_amd.sim import WarpXSim

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lib = load_product()
L = 40e-6
sim = WarpXSim(lib, (n, n, n), (-L / 2,) * 3, (L / 2,) * 3, nox=3, galerkin=1, use_filter=1, sort_interval=3)
E = (_capi.FieldView * 3)(*[sim.field_view(c) for c in ("Ex", "Ey", "Ez")])
B = (_capi.FieldView * 3)(*[sim.field_view(c) for c in ("Bx", "By", "Bz")])
dinv = (C.c_double * 3)(*[1.0 / d for d in sim.dx])
dx3 = (C.c_double * 3)(*sim.dx)
co = [(C.c_double * 5)() for _ in range(3)]
lib.ckc_stencil_coefficients(dx3, *co)
for rnd in range(2):
    for plain, var in [(1, -1)] + [(0, v) for v in (-1, 0, 2, 3, 4, 5, 6, 7, 8, 9)]:
        os.environ["WXA_CKC_PLAIN"] = str(plain)
        os.environ["WXA_CKC_VARIANT"] = str(var)
        call = lambda: lib.evolve_b_ckc(E, B, 1e-17, *co, None)
        call(); call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"round {rnd} {'plain' if plain else 'tiled variant %d' % var}: EvolveB(CKC) {ms:.4f} ms ({100 * 72.0 * n ** 3 / 1e9 / (ms * 1e-3) / 8000.0:.1f} % of 8 TB/s)", flush=True)
os.environ["WXA_STENCIL_VARIANT"] = "-1"
call = lambda: lib.evolve_b(E, B, 0.0, dinv, None)
call(); call()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    call()
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"for comparison, Yee EvolveB {ms:.4f} ms ({100 * 72.0 * n ** 3 / 1e9 / (ms * 1e-3) / 8000.0:.1f} %)")
