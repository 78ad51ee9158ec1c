"""Run a WarpX inputs file on the HIP library, the way the reference's executable is used:

    python -m warpx_amd.run <inputs_file> [name=value ...] [--checksum out.json]

Evolves for the deck's max_step, writes the deck's diagnostics as the reference does (diag_type = Full with
format = plotfile -> <file_prefix><step>/, the reduced diagnostics FieldEnergy / ParticleEnergy / ParticleMomentum /
ParticleNumber -> diags/reducedfiles/<name>.txt) and the reference-format regression checksum of the final
state (Regression/Checksum/checksum.py) as JSON.  The deck reader, the step loop and the checksum are
the library's (include/warpx_amd.h: wxa_sim_create_from_inputs, wxa_sim_evolve,
wxa_sim_checksum_json); this file only parses the command line.
"""
from __future__ import annotations

import argparse
import json
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("inputs")
    ap.add_argument("overrides", nargs="*", help="name=value, applied after the file")
    ap.add_argument("--checksum", default=None, help="write the checksum JSON here (default: stdout)")
    ap.add_argument("--max-step", type=int, default=None, help="instead of the deck's max_step")
    ap.add_argument("--no-diagnostics", action="store_true",
                    help="do not write the deck's diagnostics (plotfiles under <diag>.file_prefix, diags/reducedfiles/*.txt)")
    args = ap.parse_args(argv)

    import os

    import torch

    from . import load_product
    from .sim import WarpXSim
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    lib = load_product()   # raises when the HIP library is missing: no fallback
    comm = None
    if world > 1:          # one process per GPU (python -m torch.distributed.run --nproc-per-node N -m warpx_amd.run ...)
        import torch.distributed as dist

        from .distributed import TorchBrickTransport
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        transport = TorchBrickTransport(on_device=True)
        comm = transport.comm
    # the library chooses the bricks; the deck's Full (plotfile) and reduced diagnostics are written under diags/
    sim = WarpXSim.from_inputs(lib, args.inputs, args.overrides, comm=comm, diagnostics=not args.no_diagnostics)
    steps = args.max_step if args.max_step is not None else sim.max_step
    if steps is None or steps < 0:
        raise SystemExit("max_step is not set in the inputs file: pass --max-step")
    sim.evolve(steps)
    total = sim.checksum()                    # this brick's share of every sum
    if world > 1:
        parts = [None] * world
        dist.gather_object(total, parts if rank == 0 else None, dst=0)
        if rank == 0:
            total = {g: {k: sum(p[g][k] for p in parts) for k in vals} for g, vals in parts[0].items()}
    if rank == 0:
        text = json.dumps(total, indent=2)
        if args.checksum:
            with open(args.checksum, "w") as f:
                f.write(text + "\n")
        else:
            print(text)
    sim.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
