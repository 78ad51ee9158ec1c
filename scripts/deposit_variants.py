"""HISTORICAL (round 2): the variant numbers it drives no longer exist in deposit_tile.hip; kept as the record of how
profiles/round2/* was produced.  Use scripts/variants.py (any WXA_* switch of a -DWXA_DEV_VARIANTS build).

A/B timing of the LDS-tile deposition configurations (WXA_DEPOSIT_VARIANT, deposit_tile.hip) inside the bench
workload: 256^3, 8 ppc, order 3, Esirkepov, thermalised by the pre-roll.  Prints the CurrentDeposition phase time
per launch (HIP events on the kernels' stream) and the whole step for every variant.

    python scripts/deposit_variants.py [--ncell 256] [--steps 6] [--variants 0,1,2,...]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import device_uniform_plasma
from warpx_amd import _capi, load_product, plasma
from warpx_amd.containers import ParticleArrays
from warpx_amd.sim import WarpXSim, field_energy

ap = argparse.ArgumentParser()
ap.add_argument("--ncell", type=int, default=256)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--preroll", type=int, default=40)
ap.add_argument("--variants", default="0,1,2,3,4,5,6,7")
args = ap.parse_args()
lib = load_product()
n = args.ncell
L = 40e-6
sim = WarpXSim(lib, (n, n, n), (-L / 2,) * 3, (L / 2,) * 3, nox=3, galerkin=None, particle_pusher=_capi.PUSHER_BORIS,
               current_deposition=_capi.DEPOSIT_ESIRKEPOV, use_filter=1, cfl=1.0, sort_interval=3)
parts = device_uniform_plasma((n, n, n), (-L / 2,) * 3, (L / 2,) * 3, (2, 2, 2), 1e25, 0.01, 12345, (0, 0, 0), (n, n, n), "cuda")
pa = ParticleArrays(parts.shape[1], "cuda")
pa.data = parts
sim.add_species(-plasma.Q_E, plasma.M_E, pa)
del parts, pa
sim.evolve(args.preroll)
torch.cuda.synchronize()
out = {}
for v in [int(x) for x in args.variants.split(",")]:
    os.environ["WXA_DEPOSIT_VARIANT"] = str(v)
    sim.evolve(3)                      # one sort cycle untimed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.evolve(args.steps)
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / args.steps * 1e3
    sim.enable_timers(True)
    sim.timers(reset=True)
    sim.evolve(args.steps)
    torch.cuda.synchronize()
    ph = sim.timers(reset=True)
    sim.enable_timers(False)
    ms, cnt = ph["CurrentDeposition"]
    ee, eb = field_energy(sim)
    out[v] = {"deposit_ms": ms / cnt, "step_ms": step_ms, "field_energy": ee + eb}
    print(f"variant {v}: CurrentDeposition {ms / cnt:.3f} ms/launch, step {step_ms:.3f} ms, E+B energy {ee + eb:.6e}", flush=True)
print(json.dumps(out))
