"""Phase clocks of the deposition tile kernels (rows / waves variants) in the thermalised bench regime.
Needs a profile build:  WXA_DEPOSIT_PROFILE=1 WXA_LIB_OUT=warpx_amd/libwarpx_amd_prof.so python -m warpx_amd.build --force
(then rebuild the product with --force).   python scripts/deposit_profile2.py <variant> [ncell]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import device_uniform_plasma
os.environ["WXA_PRODUCT_LIB"] = os.path.join(ROOT, "warpx_amd", "libwarpx_amd_prof.so")
from warpx_amd import _capi, load_product, plasma
from warpx_amd.containers import ParticleArrays
from warpx_amd.sim import WarpXSim
variant = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L = 40e-6
lib = load_product()
raw = C.CDLL(os.environ["WXA_PRODUCT_LIB"])
sim = WarpXSim(lib, (n,) * 3, (-L / 2,) * 3, (L / 2,) * 3, nox=3, galerkin=1, use_filter=1, sort_interval=3)
parts = device_uniform_plasma((n,) * 3, (-L / 2,) * 3, (L / 2,) * 3, (2, 2, 2), 1e25, 0.01, 12345, (0, 0, 0), (n,) * 3, "cuda")
pa = ParticleArrays(parts.shape[1], "cuda"); pa.data = parts
sim.add_species(-plasma.Q_E, plasma.M_E, pa)
del parts, pa
sim.evolve(40)
os.environ["WXA_DEPOSIT_VARIANT"] = variant
sim.evolve(3)
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
raw.wxa_debug_deposit_profile(out, 1)
steps = 6
sim.evolve(steps)
torch.cuda.synchronize()
raw.wxa_debug_deposit_profile(out, 1)
tot = sum(out[:6])   # the workgroup clocks of the phases (6, 7, 8: wave 0's chunk clocks)
print(f"variant {variant}: total workgroup-clock {tot:.3e} per {steps} launches")
for i in range(6):
    print(f"   phase {i}: {100.0 * out[i] / tot:5.1f} %")
print("   counters:", [int(out[i]) for i in range(8, 16)])
if out[9]:
    print(f"   wave 0 spends {100.0 * out[9] / out[2]:.1f} % of phase 2 in its chunk loop (the rest: waiting for the slowest wave)")
if out[8]:   # wave 0's chunks: load wait and the rest, cycles per chunk
    print(f"   wave 0, per chunk: loads issued -> arrived {out[6] / out[8]:.0f} cycles, coordinates .. last atomic retired "
          f"{out[7] / out[8]:.0f} cycles, {out[8] / steps:.0f} chunks per launch")
