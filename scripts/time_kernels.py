"""Times single C-ABI calls (deposit, gather, sort) on a config-2-sized synthetic plasma with
torch.cuda events on the null stream.  python scripts/time_kernels.py [ncell] [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import device_uniform_plasma
from warpx_amd import _capi, load_product, plasma
from warpx_amd.containers import STAG, FieldArray, ParticleArrays, field_triplet, grid_geom

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = load_product()
dev = "cuda"
L = 40e-6
ncell = (n, n, n)
dx = [L / n] * 3
parts = device_uniform_plasma(ncell, (-L / 2,) * 3, (L / 2,) * 3, (2, 2, 2), 1e25, 0.01, 1, (0, 0, 0), ncell, dev)
pa = ParticleArrays(parts.shape[1], dev); pa.data = parts
srt = ParticleArrays(pa.np, dev)
ws = C.c_void_p(); lib.workspace_create(C.byref(ws))
d3 = lambda v: (C.c_double * 3)(*v)
i3 = lambda v: (C.c_int32 * 3)(*v)


def timed(fn, reps=reps):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts) / len(ts)


sort = lambda: lib.sort_particles_by_cell(C.byref(pa.view), C.byref(srt.view), d3((-L / 2,) * 3), d3([1 / d for d in dx]),
                                          i3((0, 0, 0)), i3(ncell), ws, None)
print("sort            min/avg ms", timed(sort))
J = [FieldArray(ncell, STAG[c], (5, 5, 5), dev, pad=True) for c in ("jx", "jy", "jz")]
E = [FieldArray(ncell, STAG[c], (4, 4, 4), dev, pad=True) for c in ("Ex", "Ey", "Ez")]
B = [FieldArray(ncell, STAG[c], (4, 4, 4), dev, pad=True) for c in ("Bx", "By", "Bz")]
gJ = grid_geom((-L / 2,) * 3, dx, (0, 0, 0), (4, 4, 4))
gE = grid_geom((-L / 2,) * 3, dx, (0, 0, 0), (4, 4, 4))
dt = 1.0 / (np.sqrt(3.0) / dx[0] * plasma.C_LIGHT)
q, m = -plasma.Q_E, plasma.M_E
dep = lambda: lib.deposit_current(C.byref(srt.view), field_triplet(J), C.byref(gJ), q, dt, -0.5 * dt, 3, 0, ws, None)
print("deposit (tiles) min/avg ms", timed(dep))
depg = lambda: lib.deposit_current(C.byref(srt.view), field_triplet(J), C.byref(gJ), q, dt, -0.5 * dt, 3, 0, None, None)
if n <= 128:
    print("deposit (global atomics) min/avg ms", timed(depg, 2))
gat = lambda: lib.gather_push_ws(C.byref(srt.view), field_triplet(E), field_triplet(B), C.byref(gE), q, m, 0.0, 3, 1, 0, 0, ws, None)
print("gather+push (tiles, dt=0) min/avg ms", timed(gat))
gat0 = lambda: lib.gather_push_ws(C.byref(srt.view), field_triplet(E), field_triplet(B), C.byref(gE), q, m, 0.0, 3, 1, 0, 0, None, None)
print("gather+push (global loads) min/avg ms", timed(gat0))
