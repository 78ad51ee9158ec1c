"""Phase clocks of deposit_tile_kernel in the thermalised regime.
Needs a profile build:  WXA_DEPOSIT_PROFILE=1 python -m warpx_amd.build --force"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import device_uniform_plasma
from warpx_amd import load_product, plasma
from warpx_amd.containers import ParticleArrays
from warpx_amd.sim import WarpXSim
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n, L = 256, 40e-6
from warpx_amd import _capi
if os.environ.get("WXA_PRODUCT_LIB"):
    _capi.PRODUCT_LIB = os.path.abspath(os.environ["WXA_PRODUCT_LIB"])
lib = load_product()
raw = C.CDLL(_capi.PRODUCT_LIB)
sim = WarpXSim(lib, (n,) * 3, (-L / 2,) * 3, (L / 2,) * 3, nox=3, galerkin=1, use_filter=1, sort_interval=4)
parts = device_uniform_plasma((n,) * 3, (-L / 2,) * 3, (L / 2,) * 3, (2, 2, 2), 1e25, 0.01, 1, (0, 0, 0), (n,) * 3, "cuda")
pa = ParticleArrays(parts.shape[1], "cuda"); pa.data = parts
sim.add_species(-plasma.Q_E, plasma.M_E, pa)
del parts, pa
names = ["zero+stage+key", "item lists", "fast pass", "deferred flush", "append/direct", "write-back"]
for label, steps in (("cold lattice", 4), ("after %d steps" % pre, 4)):
    if label != "cold lattice":
        sim.evolve(pre)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    raw.wxa_debug_deposit_profile(out, 1)
    sim.evolve(steps)
    torch.cuda.synchronize()
    raw.wxa_debug_deposit_profile(out, 1)
    tot = sum(out[:6])
    print(label, "total block-cycles %.3e  -> %.2f ms/launch at 256 blocks, 2.4 GHz" % (tot, tot / steps / 256 / 2.4e9 * 1e3))
    for i, nm in enumerate(names):
        print("   %-16s %5.1f %%" % (nm, 100.0 * out[i] / tot))
    trips = max(out[10], 1)
    print("   per launch: trips %.0f, fast items/trip %.1f, slow items/trip %.1f, consumed/trip %.1f, staged/trip %.1f, "
          "flushes %.0f with %.1f items each" % (out[10] / steps, out[8] / trips, out[9] / trips, out[11] / trips,
                                                   out[12] / trips, out[13] / steps, out[14] / max(out[13], 1)))
