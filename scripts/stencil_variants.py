"""Sweep of the EvolveB / EvolveE kernel configurations (WXA_STENCIL_VARIANT) on 256^3 fields with the host layer's
padded layout: back-to-back launches between one pair of events per configuration; prints ms per launch and the
fraction of the 8 TB/s HBM peak the algorithmic bytes (72 / 96 B per cell) reach.
    python scripts/stencil_variants.py [ncell] [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from warpx_amd import _capi, load_product, plasma
from warpx_amd.sim import WarpXSim

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(14))
lib = load_product()
L = 40e-6
sim = WarpXSim(lib, (n, n, n), (-L / 2,) * 3, (L / 2,) * 3, nox=3, galerkin=1, use_filter=1, sort_interval=3)
E = (_capi.FieldView * 3)(*[sim.field_view(c) for c in ("Ex", "Ey", "Ez")])
B = (_capi.FieldView * 3)(*[sim.field_view(c) for c in ("Bx", "By", "Bz")])
J = (_capi.FieldView * 3)(*[sim.field_view(c) for c in ("jx", "jy", "jz")])
dinv = (C.c_double * 3)(*[1.0 / d for d in sim.dx])
best = {}
for rnd in range(2):
    for v in variants:
        os.environ["WXA_STENCIL_VARIANT"] = str(v)
        res = []
        for name, call, bpc in (("B", lambda: lib.evolve_b(E, B, 0.0, dinv, None), 72.0),
                                ("E", lambda: lib.evolve_e(E, B, J, 0.0, dinv, None), 96.0)):
            call(); call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                call()
            e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1) / reps
            res.append((ms, bpc * n ** 3 / 1e9 / (ms * 1e-3) / 8000.0))
        print(f"round {rnd} variant {v:2d}: EvolveB {res[0][0]:.4f} ms ({100 * res[0][1]:.1f} %)   EvolveE {res[1][0]:.4f} ms ({100 * res[1][1]:.1f} %)", flush=True)
