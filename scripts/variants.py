"""A/B timing of environment-selected kernel variants inside the bench workload (256^3, 8 ppc, order 3, Esirkepov, Boris,
filter, sort every 3rd step, thermalised by the pre-roll): ONE simulation, the variants one after the other, every phase's
HIP-event time per launch.  Needs a build with -DWXA_DEV_VARIANTS (the production library ignores the switches):

    WXA_EXTRA_DEFS=-DWXA_DEV_VARIANTS WXA_LIB_OUT=warpx_amd/libwarpx_amd_dev.so python -m warpx_amd.build --force
    WXA_PRODUCT_LIB=warpx_amd/libwarpx_amd_dev.so python scripts/variants.py base WXA_GATHER_RB=0 WXA_GATHER_RB=3 ...

A spec is "base" (no switch) or NAME=value[,NAME2=value2]; --repeat runs the list several times (box drift)."""
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
ap.add_argument("specs", nargs="+")
ap.add_argument("--ncell", type=int, default=256)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--preroll", type=int, default=40)
ap.add_argument("--repeat", type=int, default=1)
ap.add_argument("--deposition", default="esirkepov")
ap.add_argument("--no-step-time", action="store_true")
args = ap.parse_args()
lib = load_product()
n = args.ncell
L = 40e-6
dep = _capi.DEPOSIT_ESIRKEPOV if args.deposition == "esirkepov" else _capi.DEPOSIT_DIRECT
sim = WarpXSim(lib, (n, n, n), (-L / 2,) * 3, (L / 2,) * 3, nox=3, galerkin=None, particle_pusher=_capi.PUSHER_BORIS,
               current_deposition=dep, use_filter=1, cfl=1.0, sort_interval=3)
parts = device_uniform_plasma((n, n, n), (-L / 2,) * 3, (L / 2,) * 3, (2, 2, 2), 1e25, 0.01, 12345, (0, 0, 0), (n, n, n), "cuda")
pa = ParticleArrays(parts.shape[1], "cuda")
pa.data = parts
sim.add_species(-plasma.Q_E, plasma.M_E, pa)
del parts, pa
sim.evolve(args.preroll)
torch.cuda.synchronize()
touched = set()
out = []
for rep in range(args.repeat):
    for spec in args.specs:
        for k in touched:
            os.environ.pop(k, None)
        if spec != "base":
            for kv in spec.split(","):
                k, v = kv.split("=")
                os.environ[k] = v
                touched.add(k)
        sim.evolve(3)                      # one sort cycle untimed
        torch.cuda.synchronize()
        step_ms = float("nan")
        if not args.no_step_time:
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
        ee, eb = field_energy(sim)
        row = {"spec": spec, "step_ms": step_ms, "field_energy": ee + eb}
        row.update({k: (ms / cnt if cnt else 0.0) for k, (ms, cnt) in ph.items()})
        out.append(row)
        print(f"{spec:40s} step {step_ms:7.3f} | gather {row['GatherAndPush']:.3f} deposit {row['CurrentDeposition']:.3f} "
              f"sync {row['SyncCurrent']:.3f} B {row['EvolveB']:.3f} E {row['EvolveE']:.3f} redistribute "
              f"{row['Redistribute']:.3f} | E+B {ee + eb:.6e}", flush=True)
print(json.dumps(out))
