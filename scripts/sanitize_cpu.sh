#!/bin/bash
# Host layer + CPU restatement under AddressSanitizer / UBSan (no GPU needed): builds a sanitized copy of
# tests/host_cpu's library in /tmp and runs a few decks through it.  Last run clean at the end of round 2 (round 1:
# (decks with walls, PEC, laser + moving window, direct deposition, device-side injection mirror; a 2-brick
# gloo run with overlap_halo was checked the same way by swapping the library in); round 2 added the lens, boosted-frame,
# plotfile and evolve-in-pieces paths.
set -eu
cd "$(dirname "$0")/.."
g++ -O1 -g -march=x86-64-v3 -std=c++17 -fPIC -fopenmp -ffp-contract=off -fsanitize=address,undefined \
    -fno-omit-frame-pointer -shared -Wl,-Bsymbolic -o /tmp/libhost_cpu_asan.so \
    tests/host_cpu/host_cpu.cpp oracle/pic_oracle.cpp
cat > /tmp/asan_run.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from warpx_amd import _capi
from warpx_amd.sim import WarpXSim
lib = _capi.CLib('/tmp/libhost_cpu_asan.so', "hst_", _capi._INPUTS_SIGS, kernels=False)
for deck, ov in (("particle_walls_3d.inputs", []), ("laser_injection_3d.inputs", ["max_step=6"]),
                 ("langmuir_beam_direct_3d.inputs", ["max_step=6"]), ("pec_two_particles_3d.inputs", []),
                 ("uniform_plasma_3d.inputs", ["amr.n_cell=16 16 16", "max_step=4"]),
                 # round 2: repeated plasma lens (lab and boosted), boosted injection behind the window, boosted antenna,
                 # the boosted laser-wakefield deck, plotfile writer, evolve in pieces
                 ("plasma_lens_3d.inputs", []), ("plasma_lens_boosted_3d.inputs", []),
                 ("boosted_injection_3d.inputs", ["max_step=12"]), ("boosted_laser_3d.inputs", ["max_step=10"]),
                 ("laser_wakefield_boosted_3d.inputs", ["max_step=12"])):
    sim = WarpXSim.from_inputs(lib, os.path.join("tests/decks", deck), overrides=ov)
    sim.set_synchronize_at_end(False)
    sim.evolve(sim.max_step // 2)
    sim.evolve(sim.max_step - sim.max_step // 2)
    sim.synchronize()
    sim.checksum()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        sim.write_plotfile(os.path.join(d, "plt"))
    sim.close()
    print(deck, "clean")
PY
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) OMP_NUM_THREADS=2 \
    python /tmp/asan_run.py
