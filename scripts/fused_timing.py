"""A/B timing on the MI355X (dev build: WXA_PRODUCT_LIB=warpx_amd/libwarpx_amd_dev.so): PushPX + DepositCurrent as two
kernels (wxa_gather_push_ws, wxa_deposit_current) against wxa_debug_push_and_deposit (one kernel over the LDS tiles) on the
same particle state -- the bench regime rebuilt standalone:
n^3 cells, 8 particles per cell at random positions (Poisson occupancy), u_th = 0.01 c, order 3, energy-conserving gather,
Esirkepov, Boris; the tile is sorted and then drifted by `stale` steps, as the steps between two sorts see it.
    python scripts/fused_timing.py [ncell=256] [stale=1] [repeats=3]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from warpx_amd import _capi, load_product, plasma
from warpx_amd.containers import FieldArray, ParticleArrays, STAG, field_triplet, grid_geom

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
stale = int(sys.argv[2]) if len(sys.argv) > 2 else 1
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lib = load_product()
fused = lib._dll.wxa_debug_push_and_deposit   # -DWXA_DEV_VARIANTS builds only
fused.restype = C.c_int
fused.argtypes = [_capi._PPV, _capi._FV3, _capi._FV3, _capi._FV3, _capi._PGG, _capi._PGG, C.c_double, C.c_double, C.c_double,
                  C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = "cuda"
L = 40e-6
ncell = (n, n, n)
dx = np.array([L / n] * 3)
dt = 1.0 / (np.sqrt(np.sum(1.0 / dx ** 2)) * plasma.C_LIGHT)
npart = 8 * n ** 3
g = torch.Generator(device=dev)
g.manual_seed(7)
src = ParticleArrays(npart, dev)
for d in range(3):
    src.data[d] = -L / 2 + L * torch.rand(npart, dtype=torch.float64, device=dev, generator=g)
src.data[3].fill_(1e25 * (L / n) ** 3 / 8.0)
for d in range(3):
    src.data[4 + d] = 0.01 * plasma.C_LIGHT * torch.randn(npart, dtype=torch.float64, device=dev, generator=g)
srt = ParticleArrays(npart, dev)
ws = C.c_void_p()
lib.workspace_create(C.byref(ws))
d3 = lambda v: (C.c_double * 3)(*[float(x) for x in v])
lib.sort_particles_by_cell(C.byref(src.view), C.byref(srt.view), d3((-L / 2,) * 3), d3(1.0 / dx), (C.c_int32 * 3)(0, 0, 0),
                           (C.c_int32 * 3)(*ncell), ws, None)
torch.cuda.synchronize()
del src
# what `stale` steps of free streaming do to the sorted tile (positions kept inside the domain, as the wrap would)
gam = torch.sqrt(1.0 + (srt.data[4] ** 2 + srt.data[5] ** 2 + srt.data[6] ** 2) / plasma.C_LIGHT ** 2)
for d in range(3):
    srt.data[d] += stale * dt * srt.data[4 + d] / gam
    srt.data[d].clamp_(-L / 2, L / 2 - 1e-12)
del gam
saved = srt.data.clone()
order, ng_eb, ng_depos, ng_j = 3, 4, 4, 5
rng = np.random.default_rng(3)
E = [FieldArray(ncell, STAG[nm], (ng_eb,) * 3, dev, pad=True) for nm in ("Ex", "Ey", "Ez")]
B = [FieldArray(ncell, STAG[nm], (ng_eb,) * 3, dev, pad=True) for nm in ("Bx", "By", "Bz")]
J = [FieldArray(ncell, STAG[nm], (ng_j,) * 3, dev, pad=True) for nm in ("jx", "jy", "jz")]
for f, s in list(zip(E, (1e8,) * 3)) + list(zip(B, (1.0,) * 3)):
    f.storage.copy_(s * torch.randn(f.storage.shape, dtype=torch.float64, device=dev, generator=g))
ge = grid_geom((-L / 2,) * 3, dx, (0, 0, 0), (ng_eb,) * 3)
gj = grid_geom((-L / 2,) * 3, dx, (0, 0, 0), (ng_depos,) * 3)
q, m = -plasma.Q_E, plasma.M_E
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]


def two_kernels():
    ev[0].record()
    lib.gather_push_ws(C.byref(srt.view), field_triplet(E), field_triplet(B), C.byref(ge), q, m, dt, order, 1,
                       _capi.PUSHER_BORIS, 1, ws, None)
    ev[1].record()
    lib.deposit_current(C.byref(srt.view), field_triplet(J), C.byref(gj), q, dt, -0.5 * dt, order, _capi.DEPOSIT_ESIRKEPOV,
                        ws, None)
    ev[2].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])


def one_kernel():
    ev[0].record()
    fused(C.byref(srt.view), field_triplet(E), field_triplet(B), field_triplet(J), C.byref(ge), C.byref(gj),
          q, m, dt, -0.5 * dt, order, 1, _capi.PUSHER_BORIS, _capi.DEPOSIT_ESIRKEPOV, ws, None)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1])


def jsum():
    return [float(f.storage.abs().sum()) for f in J]


for r in range(repeats + 1):   # the first pass warms up
    srt.data.copy_(saved)
    for f in J:
        f.storage.zero_()
    torch.cuda.synchronize()
    a, b = two_kernels()
    ja = jsum()
    pa = float(srt.data[:3].abs().sum())
    srt.data.copy_(saved)
    for f in J:
        f.storage.zero_()
    torch.cuda.synchronize()
    c = one_kernel()
    jb = jsum()
    pb = float(srt.data[:3].abs().sum())
    if r:
        print(f"{n}^3, stale {stale}: gather+push {a:.3f} + deposit {b:.3f} = {a + b:.3f} ms | one kernel {c:.3f} ms | "
              f"sum|J| rel diff {max(abs(x - y) / x for x, y in zip(ja, jb)):.1e}, sum|x| rel diff {abs(pa - pb) / pa:.1e}")
lib.workspace_destroy(ws)
