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
    if nb == (0, 0, 0):     # let the library choose the bricks for comm.nranks
        sim = WarpXSim.from_inputs(load(), deck, comm=transport.comm)
    else:
        assert world == nb[0] * nb[1] * nb[2]
        sim = WarpXSim.from_inputs(load(), deck, nbricks=nb, coord=brick_coord(rank, nb), comm=transport.comm)
    # WXA_TEST_MAX_STEP: a shorter run for brick-against-one-brick comparisons (both sides stop at the same step)
    sim.evolve(min(sim.max_step, int(os.environ.get("WXA_TEST_MAX_STEP", sim.max_step))))
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
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
