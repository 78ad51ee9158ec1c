"""Round 5: how the sums of BTD snapshot 3 (the reference golden file test_3d_laser_acceleration_btd.json) move under 1e-5\nperturbations of the deck, next to the residual against the golden file (CPU backend, 13 s per run).\n    python scripts/round5/btd_sensitivity.py > profiles/round5/r5_btd_sensitivity.txt"""
import json, os, sys
sys.path.insert(0, "/root/repo")
from tests.oracle_lib import load_host_cpu
from tests.helpers import btd_snapshot_checksum
from warpx_amd.sim import WarpXSim
M_E, M_P = 9.1093837015e-31, 1.67262192369e-27
lib = load_host_cpu()
gold = json.load(open("/root/repo/tests/golden/laser_acceleration_btd_3d_checksums.json"))["checksums"]
deck = "/root/reference/Examples/Tests/boosted_diags/inputs_test_3d_laser_acceleration_btd"
def run(ov):
    sim = WarpXSim.from_inputs(lib, deck, overrides=ov) if ov else WarpXSim.from_inputs(lib, deck)
    sim.evolve(sim.max_step)
    got = btd_snapshot_checksum(sim, 3, ("electrons", "ions", "beam"), (M_E, M_P, M_E))
    sim.close()
    return got
base = run([])
keys = [("lev=0", k) for k in ("Ex","Ey","Ez","Bx","By","Bz","jx","jy","jz","rho")] + [("electrons", "particle_momentum_x"), ("electrons","particle_position_x")]
print("%-34s" % "residual vs golden", " ".join("%+9.2e" % ((base[g][k]-gold[g][k])/gold[g][k]) for g,k in keys))
for name, ov in [
    ("e_max x (1+1e-5)", ["laser1.e_max=2.00002e12"]),
    ("density x (1+1e-5)", ["electrons.density=3.500035e24", "ions.density=3.500035e24"]),
    ("electron density only x(1+1e-5)", ["electrons.density=3.500035e24"]),
    ("cfl 1-1e-5", ["warpx.cfl=0.99999"]),
    ("gamma_boost 10.0001", ["warpx.gamma_boost=10.0001"]),
    ("laser t_peak +1e-5 rel", ["laser1.profile_t_peak=40.0004e-15"]),
    ("laser position -0.1000001e-6", ["laser1.position=0. 0. -0.100001e-6"]),
    ("plasma zmax .00300003", ["electrons.zmax=.00300003", "ions.zmax=.00300003"]),
    ("wavelength x(1+1e-5)", ["laser1.wavelength=0.8100081e-6"]),
]:
    try:
        g2 = run(ov)
        print("%-34s" % name, " ".join("%+9.2e" % ((g2[g][k]-base[g][k])/base[g][k]) for g,k in keys))
    except Exception as e:
        print(name, "failed", e)
