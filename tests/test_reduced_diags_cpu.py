"""Reduced diagnostics of the host layer (warpx_amd/csrc/host/ReducedDiags.hpp: FieldEnergy, ParticleEnergy,
ParticleMomentum, ParticleNumber -- the quantities BASELINE.json's parity gate is stated in) on the CPU build of the
host layer: the reference's file format, its intervals syntax and call sites, the values against numpy on the fields
and particles themselves and against the oracle stepper's own rows, decks, and bricks over gloo.

The reference's own test of these (Examples/Tests/reduced_diags/analysis_reduced_diags_impl.py) does the same thing:
it recomputes the energies from the full output and compares with the last row of the text files."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.oracle_lib import load_host_cpu
from warpx_amd import _capi, plasma
from warpx_amd.sim import WarpXSim, field_energy, particle_moments
from tests.ports import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = 40e-6


@pytest.fixture(scope="module")
def host_cpu():
    return load_host_cpu()


def read_table(path):
    """(header columns, rows) of a reduced-diagnostics file: the reference's own analysis reads them with
    np.genfromtxt, which skips the '#' line."""
    lines = open(path).read().splitlines()
    assert lines[0].startswith("#")
    return lines[0][1:].split(" "), np.atleast_2d(np.genfromtxt(path))


def two_species_sim(lib, n_cell=(16, 12, 12), **kw):
    sim = WarpXSim(lib, n_cell, (-L / 2,) * 3, (L / 2,) * 3, nox=3, use_filter=1, sort_interval=2, **kw)
    a = plasma.uniform_plasma(n_cell, (-L / 2,) * 3, (L / 2,) * 3, (1, 1, 2), 1e25, 0.05, seed=3)
    b = plasma.uniform_plasma(n_cell, (-L / 2,) * 3, (L / 2,) * 3, (1, 2, 1), 2e25, 0.002, seed=4)
    b[6] = b[6] + 0.01 * plasma.C_LIGHT   # a drift: a momentum that is not round-off
    ids = [sim.add_species(-plasma.Q_E, plasma.M_E, a), sim.add_species(plasma.Q_E, 1836.0 * plasma.M_E, b)]
    return sim, ids


def test_files_rows_and_values(host_cpu, oracle, tmp_path):
    """Four diagnostics on a two-species run: header rows as the reference's constructors write them, one row per
    step of the intervals (row 0 before the first step), step / time columns, 14 digits; the last row against numpy on
    the arrays, every row against the rows the oracle stepper writes for the same run."""
    rows = {}
    for lib, sub in ((host_cpu, "host/"), (oracle, "oracle/")):
        path = str(tmp_path / sub) + "/"
        os.makedirs(path, exist_ok=True)
        sim, ids = two_species_sim(lib)
        sim.add_reduced_diag("EF", "FieldEnergy", "1", path)
        sim.add_reduced_diag("EP", "ParticleEnergy", "2", path)
        sim.add_reduced_diag("PP", "ParticleMomentum", "1:5:3", path)
        sim.add_reduced_diag("NP", "ParticleNumber", "6", path)
        sim.add_reduced_diag("quiet", "FieldEnergy", "1", None)   # no file
        sim.evolve(4)
        sim.evolve(2)   # a second call: no second "row 0"
        rows[sub] = {n: np.atleast_2d(np.genfromtxt(path + n + ".txt")) for n in ("EF", "EP", "PP", "NP")}
        if lib is host_cpu:
            assert not os.path.exists(path + "quiet.txt")
            ee, eb = field_energy(sim)
            mom = [particle_moments(sim, i) for i in ids]
            n_live = [sim.particles(i).shape[1] for i in ids]
            dt = sim.dt
            assert np.allclose(sim.reduced_diag("quiet"), [ee + eb, ee, eb], rtol=1e-13)
            # compute_now evaluates without writing a row
            assert np.array_equal(sim.reduced_diag("EF", compute_now=True), sim.reduced_diag("quiet"))
            assert len(open(path + "EF.txt").read().splitlines()) == 1 + 7
        sim.close()
    path = str(tmp_path / "host/") + "/"
    head, ef = read_table(path + "EF.txt")
    assert head == ["[0]step()", "[1]time(s)", "[2]total_lev0(J)", "[3]E_lev0(J)", "[4]B_lev0(J)"]
    assert list(ef[:, 0]) == [0, 1, 2, 3, 4, 5, 6]
    assert np.allclose(ef[:, 1], np.arange(7) * dt, rtol=1e-14, atol=0)
    assert np.allclose(ef[-1, 2:], [ee + eb, ee, eb], rtol=1e-13)
    assert ef[0, 2] == 0.0 and np.all(np.diff(ef[:3, 2]) > 0)   # the fields start at zero and grow from the thermal noise
    # 14 digits, scientific (ReducedDiags.cpp:108), the step as an integer
    first = open(path + "EF.txt").read().splitlines()[1].split(" ")
    assert first[0] == "0" and all(len(v.split("e")[0]) == 16 and "e" in v for v in first[1:])

    head, ep = read_table(path + "EP.txt")
    assert head == ["[0]step()", "[1]time(s)", "[2]total(J)", "[3]species0(J)", "[4]species1(J)", "[5]total_mean(J)",
                    "[6]species0_mean(J)", "[7]species1_mean(J)"]
    assert list(ep[:, 0]) == [0, 2, 4, 6]
    ek, w = [m["ekin"] for m in mom], [m["weight"] for m in mom]
    assert np.allclose(ep[-1, 2:], [ek[0] + ek[1], ek[0], ek[1], (ek[0] + ek[1]) / (w[0] + w[1]), ek[0] / w[0], ek[1] / w[1]],
                       rtol=1e-13)

    head, pp = read_table(path + "PP.txt")
    assert head[2:5] == ["[2]total_x(kg*m/s)", "[3]total_y(kg*m/s)", "[4]total_z(kg*m/s)"]
    assert head[5] == "[5]species0_x(kg*m/s)" and head[11] == "[11]total_mean_x(kg*m/s)" and head[-1] == "[19]species1_mean_z(kg*m/s)"
    assert list(pp[:, 0]) == [1, 4]   # "1:5:3"
    # the drifting species: its z momentum is far from round-off, the mean is sum / weight
    sim_last = pp[-1]
    assert abs(sim_last[10]) > 1e2 * abs(sim_last[8])   # (the transverse sums are thermal noise, ~ 1 / sqrt(N))
    assert np.isclose(sim_last[19], sim_last[10] / w[1], rtol=1e-12)
    assert np.isclose(sim_last[4], sim_last[7] + sim_last[10], rtol=1e-14)

    head, npn = read_table(path + "NP.txt")
    assert head == ["[0]step()", "[1]time(s)", "[2]total_macroparticles()", "[3]species0_macroparticles()",
                    "[4]species1_macroparticles()", "[5]total_weight()", "[6]species0_weight()", "[7]species1_weight()"]
    assert list(npn[:, 0]) == [0, 6]
    assert list(npn[-1, 2:5]) == [sum(n_live), n_live[0], n_live[1]]
    assert np.allclose(npn[-1, 5:], [w[0] + w[1], w[0], w[1]], rtol=1e-13)

    # the oracle stepper's rows (its own formulas, its own schedule): same steps, same numbers
    for n in ("EF", "EP", "PP", "NP"):
        a, b = rows["host/"][n], rows["oracle/"][n]
        assert a.shape == b.shape, n
        assert np.array_equal(a[:, 0], b[:, 0])
        scale = np.max(np.abs(b[:, 2:]), axis=0)
        if n == "PP":   # the components that are round-off sums of +- terms: against the size of the terms
            scale = np.maximum(scale, 1e-6 * np.max(np.abs(b[:, 2:])))
        assert np.all(np.abs(a[:, 2:] - b[:, 2:]) <= 1e-10 * scale), (n, np.max(np.abs(a[:, 2:] - b[:, 2:]) / scale))


def test_intervals_syntax(host_cpu, tmp_path):
    """utils::parser::IntervalsParser (IntervalsParser.cpp:17-107): period | start:stop | start:stop:period, comma
    separated; a period of 0 never fires."""
    path = str(tmp_path) + "/"
    sim, _ = two_species_sim(host_cpu, n_cell=(12, 12, 12))
    for name, iv in (("a", "3"), ("b", "2:4"), ("c", "::4"), ("d", "1:2, 5:"), ("e", "0"), ("f", "7:")):
        sim.add_reduced_diag(name, "ParticleNumber", iv, path)
    sim.evolve(8)
    sim.close()
    steps = {}
    for n in "abcdef":
        lines = open(path + n + ".txt").read().splitlines()
        steps[n] = [int(line.split(" ")[0]) for line in lines[1:]]
    assert steps == {"a": [0, 3, 6], "b": [2, 3, 4], "c": [0, 4, 8], "d": [1, 2, 5, 6, 7, 8], "e": [], "f": [7, 8]}


def test_refusals(host_cpu, tmp_path):
    sim, _ = two_species_sim(host_cpu, n_cell=(12, 12, 12))
    with pytest.raises(_capi.WxaError, match="not a valid type"):
        sim.add_reduced_diag("x", "FieldProbe", "1", None)
    with pytest.raises(_capi.WxaError, match="valid syntax"):
        sim.add_reduced_diag("x", "FieldEnergy", "1:2:3:4", None)
    with pytest.raises(_capi.WxaError, match="cannot read"):
        sim.add_reduced_diag("x", "FieldEnergy", "often", None)
    sim.add_reduced_diag("x", "FieldEnergy", "1", None)
    with pytest.raises(_capi.WxaError, match="defined twice"):
        sim.add_reduced_diag("x", "ParticleEnergy", "1", None)
    with pytest.raises(_capi.WxaError, match="no diagnostic named"):
        sim.reduced_diag("y")
    sim.close()


RD_LINES = ["warpx_amd.write_diagnostics=1", "diag1.intervals=100000:", "diag1.dump_last_timestep=0", "warpx.reduced_diags_names=EF EP PP NP FP", "EF.type=FieldEnergy", "EF.intervals=2", "EP.type=ParticleEnergy",
            "PP.type=ParticleMomentum", "PP.intervals=3", "NP.type=ParticleNumber", "FP.type=FieldProbe", "FP.intervals=1"]


def test_deck_with_reduced_diags(host_cpu, tmp_path):
    """warpx.reduced_diags_names in a deck: the four types are produced under <name>.path with the species' names in the
    header; a type outside the path (FieldProbe) is output this library does not write, as before."""
    deck = os.path.join(ROOT, "tests", "decks", "langmuir_multi_3d.inputs")
    path = str(tmp_path) + "/"
    over = ["my_constants.nx=16", "max_step=6"] + RD_LINES + [f"{n}.path={path}" for n in ("EF", "EP", "PP", "NP")]
    sim = WarpXSim.from_inputs(host_cpu, deck, overrides=over)
    sim.evolve(sim.max_step)
    sim.dx = [L / 16] * 3   # what the numpy checkers need to know about a deck-built run: lx / nx, the species' masses
    sim.species = [(-plasma.Q_E, plasma.M_E), (plasma.Q_E, plasma.M_E)]
    ee, eb = field_energy(sim)
    ek = [particle_moments(sim, i)["ekin"] for i in range(2)]
    sim.close()
    head, ef = read_table(path + "EF.txt")
    assert list(ef[:, 0]) == [0, 2, 4, 6]
    assert np.allclose(ef[-1, 2:], [ee + eb, ee, eb], rtol=1e-13)
    head, ep = read_table(path + "EP.txt")
    assert head[3:5] == ["[3]electrons(J)", "[4]positrons(J)"] and head[6] == "[6]electrons_mean(J)"
    assert ep.shape[0] == 7 and np.allclose(ep[-1, 3:5], ek, rtol=1e-13)
    # the Langmuir oscillation: kinetic energy goes into the field; the sum is the same at the two rows where momenta
    # and fields are at the same time (before the first step, after the last; in between the rows hold the leap-frog
    # momenta half a step behind, as the reference's do)
    tot = ep[::2, 2] + ef[:, 2]
    assert abs(tot[-1] - tot[0]) < 1e-2 * tot[0] and ef[-1, 3] > 0.1 * (ep[0, 2] - ep[-1, 2]) > 0
    assert not os.path.exists(path + "FP.txt")
    # <name>.frequency went away in the reference (ReducedDiags::BackwardCompatibility); the default path
    with pytest.raises(_capi.WxaError, match="frequency"):
        WarpXSim.from_inputs(host_cpu, deck, overrides=over + ["EF.frequency=2"])
    with pytest.raises(_capi.WxaError, match="type must be set"):
        WarpXSim.from_inputs(host_cpu, deck, overrides=over + ["warpx.reduced_diags_names=EF ZZ"])


def test_particle_number_counts_live_particles(host_cpu, tmp_path):
    """Absorbing walls retire particles in place until the next sort: ParticleNumber counts the live ones
    (TotalNumberOfParticles, ParticleNumber.cpp:113-117)."""
    deck = os.path.join(ROOT, "tests", "decks", "particle_walls_3d.inputs")
    path = str(tmp_path) + "/"
    sim = WarpXSim.from_inputs(host_cpu, deck, overrides=["warpx_amd.write_diagnostics=1",
                                                          "warpx.reduced_diags_names=NP", "NP.type=ParticleNumber",
                                                          f"NP.path={path}"])
    sim.evolve(sim.max_step)
    sim.close()
    head, t = read_table(path + "NP.txt")
    assert head[3:6] == ["[3]reflecting_particles_macroparticles()", "[4]absorbing_particles_macroparticles()",
                         "[5]periodic_particles_macroparticles()"]
    assert t[0, 2] == 7 and list(t[0, 3:6]) == [2, 3, 2]
    assert list(t[-1, 3:6]) == [2, 1, 2] and t[-1, 2] == 5           # two of the three left through the absorbing walls
    assert list(t[-1, 7:10]) == [2.0, 1.0, 2.0]                      # ... and took their weight with them
    assert np.all(np.diff(t[:, 4]) <= 0)


@pytest.mark.parametrize("nb,port", [((1, 1, 2), 29671), ((2, 2, 1), 29672)])
def test_bricks_write_the_rows_of_one_brick(host_cpu, tmp_path, nb, port):
    """The same deck on 2 and 4 bricks over gloo: brick 0 writes the files, the rows are the sums over the bricks
    (ReduceRealSum: one message each way with every other brick) -- equal to the one-brick rows; a nodal point on a
    face between bricks is counted once (amrex's owner mask in MultiFab::norm2)."""
    deck = os.path.join(ROOT, "tests", "decks", "langmuir_multi_3d.inputs")
    base = ["my_constants.nx=16", "max_step=5", "algo.particle_shape=3"] + RD_LINES
    one, many = str(tmp_path / "one") + "/", str(tmp_path / "many") + "/"
    sim = WarpXSim.from_inputs(host_cpu, deck, overrides=base + [f"{n}.path={one}" for n in ("EF", "EP", "PP", "NP")])
    sim.evolve(sim.max_step)
    sim.close()
    n = nb[0] * nb[1] * nb[2]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port(port)), os.path.join(ROOT, "tests", "deck_worker.py"),
           *[str(v) for v in nb], deck, str(tmp_path / "sum.json")]
    over = ";".join(base + [f"{d}.path={many}" for d in ("EF", "EP", "PP", "NP")])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, OMP_NUM_THREADS="2", WXA_TEST_OVERRIDES=over))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for d in ("EF", "EP", "PP", "NP"):
        ha, a = read_table(one + d + ".txt")
        hb, b = read_table(many + d + ".txt")
        assert ha == hb and a.shape == b.shape, d
        assert np.array_equal(a[:, :2], b[:, :2])
        scale = np.maximum(np.max(np.abs(a[:, 2:]), axis=0), 1e-6 * np.max(np.abs(a[:, 2:])))
        if d == "PP":
            # the deck's momenta add up to zero (+- pairs): the sums are round-off, measured against the size of their
            # terms -- sum w |p| <= sqrt(sum w . sum w p^2) = sqrt(W 2 m Ekin) -- and the means against that over W
            ekin, wtot = read_table(one + "EP.txt")[1][0, 2], read_table(one + "NP.txt")[1][0, 5]
            scale = np.full(a.shape[1] - 2, np.sqrt(wtot * 2.0 * plasma.M_E * ekin))
            scale[9:] /= wtot
        assert np.all(np.abs(a[:, 2:] - b[:, 2:]) <= 1e-11 * scale), (d, np.max(np.abs(a[:, 2:] - b[:, 2:]) / scale))
    assert not os.path.exists(many + "FP.txt")
