"""Round 5: the reference's back-transformed golden file against this library on the CPU backend, sum by sum.
    python scripts/round5/btd_golden_residuals.py
    WXA_REFERENCE_CORNERS=0 WXA_PEC_RHO_FOLD_GUARD_COLUMNS=1 python scripts/round5/btd_golden_residuals.py   # round 4's behaviour
profiles/round5/r5_btd_golden_residuals.txt holds the three runs (before, with the rho fold over the valid points only, with the
reference's corners as well)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers import btd_snapshot_checksum  # noqa: E402
from tests.oracle_lib import load_host_cpu  # noqa: E402
from warpx_amd.sim import WarpXSim  # noqa: E402

M_E, M_P = 9.1093837015e-31, 1.67262192369e-27
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "laser_acceleration_btd_3d_checksums.json")))
sim = WarpXSim.from_inputs(load_host_cpu(), os.path.join(ROOT, "tests", "decks", "laser_wakefield_btd_3d.inputs"))
sim.evolve(sim.max_step)
got = btd_snapshot_checksum(sim, 3, ("electrons", "ions", "beam"), (M_E, M_P, M_E))
for grp in ("lev=0", "electrons"):
    for k, v in gold["checksums"][grp].items():
        if k in got[grp]:
            print("%-10s %-24s rel %+ .3e" % (grp, k, (got[grp][k] - v) / v if v else 0.0))
sim.close()
