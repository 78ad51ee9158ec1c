"""Run a WarpX inputs file on the HIP library, the way the reference's executable is used:

    python -m warpx_amd.run <inputs_file> [name=value ...] [--checksum out.json]

Evolves for the deck's max_step and writes the reference-format regression checksum of the final
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
    args = ap.parse_args(argv)

    from . import load_product
    from .sim import WarpXSim
    lib = load_product()   # raises when the HIP library is missing: no fallback
    sim = WarpXSim.from_inputs(lib, args.inputs, args.overrides)
    steps = args.max_step if args.max_step is not None else sim.max_step
    if steps is None or steps < 0:
        raise SystemExit("max_step is not set in the inputs file: pass --max-step")
    sim.evolve(steps)
    text = json.dumps(sim.checksum(), indent=2)
    if args.checksum:
        with open(args.checksum, "w") as f:
            f.write(text + "\n")
    else:
        print(text)
    sim.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
