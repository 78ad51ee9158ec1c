"""Worker for the multi-brick deck test: every rank builds its brick of the simulation an inputs file
describes (wxa_sim_create_from_inputs with nbricks / coord) on the CPU build of the host layer, runs the
deck's max_step over the gloo transport, and rank 0 adds up the per-brick checksums.

    python -m torch.distributed.run --nproc-per-node N tests/deck_worker.py NBX NBY NBZ DECK OUT
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402,F401
import torch.distributed as dist  # noqa: E402

from tests.oracle_lib import load_hip_on_cpu, load_host_cpu  # noqa: E402
from warpx_amd.distributed import TorchBrickTransport, brick_coord  # noqa: E402
from warpx_amd.sim import WarpXSim  # noqa: E402


def main():
    nb = tuple(int(v) for v in sys.argv[1:4])
    deck, out = sys.argv[4], sys.argv[5]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    transport = TorchBrickTransport(on_device=False)
    # WXA_WORKER_LIB=hipcpu: the product's .hip sources on the CPU execution model instead of the CPU kernels
    load = load_hip_on_cpu if os.environ.get("WXA_WORKER_LIB") == "hipcpu" else load_host_cpu
    # WXA_TEST_OVERRIDES: "a=b;c=d" appended to the deck, as on the reference's command line
    over = tuple(v for v in os.environ.get("WXA_TEST_OVERRIDES", "").split(";") if v)
    # WXA_TEST_DIAGNOSTICS=1: the deck's diagnostics are written (the test suite's default is not to)
    diags = os.environ.get("WXA_TEST_DIAGNOSTICS") == "1"
    if nb == (0, 0, 0):     # let the library choose the bricks for comm.nranks
        sim = WarpXSim.from_inputs(load(), deck, overrides=over, comm=transport.comm, diagnostics=diags)
    else:
        assert world == nb[0] * nb[1] * nb[2]
        sim = WarpXSim.from_inputs(load(), deck, overrides=over, nbricks=nb, coord=brick_coord(rank, nb),
                                   comm=transport.comm, diagnostics=diags)
    # WXA_TEST_MAX_STEP: a shorter run for brick-against-one-brick comparisons (both sides stop at the same step)
    sim.evolve(min(sim.max_step, int(os.environ.get("WXA_TEST_MAX_STEP", sim.max_step))))
    # WXA_TEST_PLOTFILE: one plotfile for all bricks (a collective call: every brick writes its grid, brick 0 the headers)
    if os.environ.get("WXA_TEST_PLOTFILE"):
        sim.write_plotfile(os.environ["WXA_TEST_PLOTFILE"])
    gathered = [None] * world
    dist.gather_object(sim.checksum(), gathered if rank == 0 else None, dst=0)
    if rank == 0:
        total = {}
        for part in gathered:
            for group, vals in part.items():
                for key, val in vals.items():
                    total.setdefault(group, {}).setdefault(key, 0.0)
                    total[group][key] += val
        json.dump(total, open(out, "w"))
    # WXA_TEST_BTD_OUT: the lab-frame snapshots of the deck's BackTransformed diagnostic, the bricks' shares put together
    btd_out = os.environ.get("WXA_TEST_BTD_OUT")
    if btd_out:
        import numpy as np
        nsnap = int(os.environ.get("WXA_TEST_BTD_NUM", "1"))
        mine = []
        for i in range(nsnap):
            mine.append({"box": sim.btd_box(i), "info": sim.btd_info(i),
                         "data": {c: sim.btd_snapshot(i, c) for c in WarpXSim.BTD_COMPONENTS},
                         "particles": [sim.btd_particles(i, s) for s in range(len(sim.species_names))]})
        shares = [None] * world
        dist.gather_object(mine, shares if rank == 0 else None, dst=0)
        if rank == 0:
            out = {}
            for i in range(nsnap):
                hi = np.max([sh[i]["box"][1] for sh in shares], axis=0)
                lo = np.min([sh[i]["box"][0] for sh in shares], axis=0)
                for c in WarpXSim.BTD_COMPONENTS:
                    whole = np.zeros(tuple(int(v) for v in hi - lo + 1))
                    for sh in shares:
                        (i0, j0, _), (i1, j1, _) = sh[i]["box"]
                        whole[i0 - lo[0]:i1 - lo[0] + 1, j0 - lo[1]:j1 - lo[1] + 1, :] += sh[i]["data"][c]
                    out[f"s{i}_{c}"] = whole
                for s in range(len(sim.species_names)):
                    out[f"s{i}_particles{s}"] = np.concatenate([sh[i]["particles"][s] for sh in shares], axis=1)
                out[f"s{i}_slices"] = np.array([sh[i]["info"]["slices"] for sh in shares])
                out[f"s{i}_zlab"] = np.array([sh[i]["info"]["z_lab"] for sh in shares])
            np.savez(btd_out, **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
