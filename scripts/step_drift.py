"""Per-chunk step time over a long run (clock / workload drift): python scripts/step_drift.py [sort_interval]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import device_uniform_plasma
from warpx_amd import load_product, plasma
from warpx_amd.containers import ParticleArrays
from warpx_amd.sim import WarpXSim
si = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = 256
L = 40e-6
lib = load_product()
sim = WarpXSim(lib, (n,) * 3, (-L / 2,) * 3, (L / 2,) * 3, nox=3, galerkin=1, use_filter=1, sort_interval=si)
parts = device_uniform_plasma((n,) * 3, (-L / 2,) * 3, (L / 2,) * 3, (2, 2, 2), 1e25, 0.01, 1, (0, 0, 0), (n,) * 3, "cuda")
pa = ParticleArrays(parts.shape[1], "cuda"); pa.data = parts
sim.add_species(-plasma.Q_E, plasma.M_E, pa)
del parts, pa
sim.evolve(4)
torch.cuda.synchronize()
for chunk in range(10):
    t0 = time.perf_counter()
    sim.evolve(16)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 16 * 1e3
    print(f"steps {4 + chunk * 16:4d}-{4 + chunk * 16 + 15:4d}: {dt:.3f} ms/step (incl. 2 PushP per 16 steps)")
sim.enable_timers(True)
sim.timers(reset=True)
sim.evolve(8)
torch.cuda.synchronize()
print({k: (round(v[0] / max(v[1], 1), 3), v[1]) for k, v in sim.timers(reset=True).items()})
import ctypes as C
for sid in (0,):
    v = sim.particle_view(sid)
    from warpx_amd.distributed import _as_tensor
    ux = _as_tensor(v.ux, 8 * int(v.np), True).view(torch.float64)
    print("u_rms/c", float(torch.sqrt(torch.mean(ux * ux))) / plasma.C_LIGHT)
