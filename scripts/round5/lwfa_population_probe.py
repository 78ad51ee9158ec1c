"""Round 5: where the particles of BASELINE config 5 sit once the window is full -- per cell and per tile of 8^3 cells --
and what that means for the deposition's work lists (deposit_tile.hip: four pairs per cell direct, pairs 5 .. 12 through
the tail table of 1024 pairs per tile, the rest as excess chunks).
    python scripts/round5/lwfa_population_probe.py [nx ny nz ppc]"""
import math
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from warpx_amd import load_product
from warpx_amd.sim import WarpXSim
nx, ny, nz, ppc = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (256, 256, 512, 2))]
torch.cuda.set_device(0)
lib = load_product()
deck = os.path.join(ROOT, "tests", "decks", "laser_wakefield_boosted_3d.inputs")
sim = WarpXSim.from_inputs(lib, deck, overrides=[f"amr.n_cell={nx} {ny} {nz}", f"electrons.num_particles_per_cell_each_dim={ppc} {ppc} {ppc}",
                                                 "warpx.sort_intervals=1", "max_step=1000000"])
sim.set_synchronize_at_end(False)
gamma = 5.0
beta = math.sqrt(1.0 - 1.0 / gamma ** 2)
lz = 16e-6 * gamma * (1.0 + beta)
dx = np.array([60e-6 / nx, 60e-6 / ny, lz / nz])
fill = int(1.15 * lz / ((1.0 + beta) * dx.min())) + 1
sim.evolve(fill)
torch.cuda.synchronize()
p = sim.particles(0)
live = p[3] != 0.0
print(f"after {fill} steps: {p.shape[1]} particles, {int(live.sum())} with weight")
pos = p[:3][:, live]
lo = np.array([-30e-6, -30e-6, pos[2].min()])
idx = [np.clip(((pos[d] - lo[d]) / dx[d]).astype(np.int64), 0, (nx, ny, nz)[d] - 1) for d in range(3)]
cell = (idx[2] * ny + idx[1]) * nx + idx[0]
cnt = np.bincount(cell, minlength=nx * ny * nz)
n = cnt.sum()
print("cells: max %d; particles in cells of <= 8: %.3f, 9-24: %.3f, 25-100: %.3f, 101-1000: %.3f, > 1000: %.3f" % (
    cnt.max(), cnt[cnt <= 8].sum() / n, cnt[(cnt > 8) & (cnt <= 24)].sum() / n, cnt[(cnt > 24) & (cnt <= 100)].sum() / n,
    cnt[(cnt > 100) & (cnt <= 1000)].sum() / n, cnt[cnt > 1000].sum() / n))
c3 = cnt.reshape(nz // 8, 8, ny // 8, 8, nx // 8, 8)
tile = c3.sum(axis=(1, 3, 5))
print("tiles: %d, mean %.0f, max %d; particles in tiles of <= 4096: %.3f, 4097-8192: %.3f, 8193-32768: %.3f, > 32768: %.3f (%d tiles)" % (
    tile.size, tile.mean(), tile.max(), tile[tile <= 4096].sum() / n, tile[(tile > 4096) & (tile <= 8192)].sum() / n,
    tile[(tile > 8192) & (tile <= 32768)].sum() / n, tile[tile > 32768].sum() / n, int((tile > 32768).sum())))
pairs = (c3 + 1) // 2
tail = np.clip(pairs - 4, 0, 8).sum(axis=(1, 3, 5))       # pairs 5 .. 12 of a cell: the tail table, 1024 per tile
excess = np.clip(pairs - 12, 0, None).sum(axis=(1, 3, 5))
over = np.clip(tail - 1024, 0, None)
print("tail table: %d tiles beyond 1024 pairs, %.3f of all particles in the pairs that do not fit (deferred list, then global atomics)" % (
    int((over > 0).sum()), 2.0 * over.sum() / n))
print("excess pairs (beyond 24 in a cell): %.3f of all particles; direct part: %.3f; tail: %.3f" % (
    2.0 * excess.sum() / n, 2.0 * np.clip(pairs, 0, 4).sum() / n, 2.0 * tail.sum() / n))
# along z (the window's length): particles per z slab of 8 cells
slab = tile.sum(axis=(1, 2))
print("particles per slab of 8 cells along z (x 1e6):", " ".join("%.1f" % (v / 1e6) for v in slab))
sim.close()
