"""Seconds per step of the CPU oracle stepper with its per-phase timers (test infrastructure; how long the parity tests'
reference runs take on a box):  python scripts/oracle_time.py [ncell] [steps] [random]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tests.oracle_lib import load_oracle
from warpx_amd import plasma
from warpx_amd.sim import WarpXSim
orc = load_oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = 40e-6
parts = plasma.uniform_plasma((n,) * 3, (-L / 2,) * 3, (L / 2,) * 3, (2, 2, 2), 1e25, 0.01, seed=1)
if len(sys.argv) > 3:   # random order: what the benchmark-regime test hands over
    perm = np.random.default_rng(0).permutation(len(parts[0]))
    parts = [p[perm] for p in parts]
sim = WarpXSim(orc, (n,) * 3, (-L / 2,) * 3, (L / 2,) * 3, nox=3, galerkin=1, use_filter=1, sort_interval=3)
sim.add_species(-plasma.Q_E, plasma.M_E, parts)
sim.evolve(1)
sim.enable_timers(True); sim.timers(reset=True)
t0 = time.perf_counter(); sim.evolve(steps); t1 = time.perf_counter()
print(f"{n}^3: {(t1 - t0) / steps:.3f} s/step, threads {orc.num_threads()}", {k: round(v[0] / max(v[1], 1), 1) for k, v in sim.timers().items()})
