"""Clocks of the gather tile kernel in the thermalised bench regime (256^3, 8 ppc): where a workgroup's time goes.
Needs the profile build:  WXA_DEPOSIT_PROFILE=1 WXA_LIB_OUT=warpx_amd/libwarpx_amd_prof.so python -m warpx_amd.build
    python scripts/gather_profile.py [ncell]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import device_uniform_plasma
os.environ["WXA_PRODUCT_LIB"] = os.path.join(ROOT, "warpx_amd", "libwarpx_amd_prof.so")
from warpx_amd import _capi, load_product, plasma
from warpx_amd.containers import ParticleArrays
from warpx_amd.sim import WarpXSim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = 40e-6
lib = load_product()
raw = C.CDLL(os.environ["WXA_PRODUCT_LIB"])
sim = WarpXSim(lib, (n,) * 3, (-L / 2,) * 3, (L / 2,) * 3, nox=3, galerkin=1, use_filter=1, sort_interval=3)
parts = device_uniform_plasma((n,) * 3, (-L / 2,) * 3, (L / 2,) * 3, (2, 2, 2), 1e25, 0.01, 12345, (0, 0, 0), (n,) * 3, "cuda")
pa = ParticleArrays(parts.shape[1], "cuda"); pa.data = parts
sim.add_species(-plasma.Q_E, plasma.M_E, pa)
del parts, pa
sim.evolve(43)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
raw.wxa_debug_gather_profile(out, 1)
steps = 6
sim.enable_timers(True); sim.timers(reset=True)
sim.evolve(steps)
torch.cuda.synchronize()
ph = sim.timers(reset=True)
raw.wxa_debug_gather_profile(out, 1)
o = [int(v) for v in out]
wg, trips = max(o[2], 1), max(o[6], 1)
print(f"GatherAndPush {ph['GatherAndPush'][0] / ph['GatherAndPush'][1]:.3f} ms per launch (with the profile's waits)")
print(f"per workgroup: staging {o[0] / wg:.0f} cycles, particle loop of wave 0 {o[1] / wg:.0f} cycles, {o[6] / wg:.2f} trips of wave 0")
print(f"per trip of wave 0: particle loads {o[3] / trips:.0f}, shapes + LDS gather {o[4] / trips:.0f}, "
      f"momentum loads + push + stores {o[5] / trips:.0f} cycles")
