"""Pins the CPU oracle's PEC field boundary (first "next" row of SURVEY.md 8(f)) against the reference's
own golden checksums and analysis for Examples/Tests/pec/inputs_test_3d_pec_field.  CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests import pec_case
from warpx_amd.sim import WarpXSim

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def pec_run(oracle):
    sim = pec_case.make_sim(oracle)
    sim.evolve(pec_case.MAX_STEP)
    return sim


def test_golden_checksums(oracle, pec_run):
    gold = json.load(open(os.path.join(HERE, "golden", "pec_field_3d_checksums.json")))
    for name, want in gold["checksums"]["lev=0"].items():
        v = pec_run.field_view(name)
        got = oracle.cell_centered_abs_sum(C.byref(v))
        print(f"{name}: got {got:.16e} want {want:.16e} rel {abs(got - want) / want:.2e}")
        assert abs(got - want) / want < gold["rtol"]


def test_reference_analysis(pec_run):
    """Examples/Tests/pec/analysis_pec.py: standing wave of twice the incident amplitude (1 %), Ey = 0 on the walls."""
    ey = pec_run.field_valid("Ey")
    e_th = 2.0 * pec_case.EY_IN
    assert abs(ey.max() - e_th) / e_th < 0.01
    assert abs(ey.min() + e_th) / e_th < 0.01
    assert np.all(ey[:, :, 0] == 0.0) and np.all(ey[:, :, -1] == 0.0)


def test_host_layer_reproduces_the_oracle_stepper(oracle, pec_run):
    """The product's C++ host layer (WarpX::EvolveB/EvolveE -> ApplyB/EfieldBoundary, BrickComm with a
    non-periodic direction) on the CPU kernels against the independent oracle stepper: bit for bit."""
    from tests.oracle_lib import load_host_cpu
    sim = pec_case.make_sim(load_host_cpu())
    sim.evolve(pec_case.MAX_STEP)
    # valid points bit for bit.  (Guards are not compared: the host layer lets the guards behind a wall travel with the
    # periodic / brick fills of the other directions, so a point behind a wall and beyond a periodic face holds the
    # mirror of the current field; the oracle stepper, like amrex::FillBoundary, leaves it one exchange old.)
    for name in ("Ex", "Ey", "Ez", "Bx", "By", "Bz"):
        assert np.array_equal(sim.field_valid(name), pec_run.field_valid(name)), name


# ---- particles next to a PEC wall: Examples/Tests/pec/inputs_test_3d_pec_particle -----------------
# Quantities of the golden file that are round-off residue in the reference itself and cannot be
# pinned: By, jz and every z component (zero by symmetry: 1e-18 of the field scale and below), and jx --
# the particles move 1e-14 cell per step in x, less than the spacing of doubles at x ~ 133 cells
# (2.8e-14), so the Esirkepov displacement x_new - x_old is quantised to whole ulps and jx comes out as
# exactly 2x the reference's value here (one ulp against half): it measures the last bit of the position
# arithmetic, not the deposition.
PINNED_FIELDS = ("Ex", "Ey", "Ez", "Bx", "Bz", "jy")
PINNED_MOMENTS = ("momentum_x", "momentum_y", "position_x", "position_y")


@pytest.fixture(scope="module")
def pec_particle_run(oracle):
    sim, e, p = pec_case.make_particle_sim(oracle)
    sim.evolve(pec_case.P_MAX_STEP)
    return sim, e, p


def test_particle_golden_checksums(oracle, pec_particle_run):
    from warpx_amd.sim import particle_moments
    sim, e, p = pec_particle_run
    gold = json.load(open(os.path.join(HERE, "golden", "pec_particle_3d_checksums.json")))
    ref = gold["checksums"]
    for name in PINNED_FIELDS:
        got = oracle.cell_centered_abs_sum(C.byref(sim.field_view(name)))
        want = ref["lev=0"][name]
        print(f"{name}: got {got:.16e} want {want:.16e} rel {abs(got - want) / want:.2e}")
        assert abs(got - want) / want < gold["rtol"], name
    for sid, species in ((e, "electron"), (p, "proton")):
        m = particle_moments(sim, sid)
        for key in PINNED_MOMENTS:
            kind, ax = key.split("_")
            got = m["abs_" + kind][ "xyz".index(ax)]
            want = ref[species]["particle_" + key]
            print(f"{species}.{key}: got {got:.16e} want {want:.16e} rel {abs(got - want) / want:.2e}")
            assert abs(got - want) / want < gold["rtol"], (species, key)
        assert m["weight"] == ref[species]["particle_weight"]


def test_particle_case_on_the_host_layer(oracle, pec_particle_run):
    """Host layer (SyncCurrentAndRho -> ApplyJfieldBoundary, PEC gather guards, no wrap along x) on the CPU
    kernels against the independent oracle stepper."""
    from tests.oracle_lib import load_host_cpu
    from warpx_amd.sim import particle_moments
    ref, e, p = pec_particle_run
    sim, e2, p2 = pec_case.make_particle_sim(load_host_cpu())
    sim.evolve(pec_case.P_MAX_STEP)
    for name in PINNED_FIELDS:
        a, b = sim.field_valid(name), ref.field_valid(name)
        assert np.max(np.abs(a - b)) <= 1e-12 * np.max(np.abs(b)), name
    for s1, s2 in ((e2, e), (p2, p)):
        m1, m2 = particle_moments(sim, s1), particle_moments(ref, s2)
        for i in (0, 1):
            assert abs(m1["abs_momentum"][i] - m2["abs_momentum"][i]) <= 1e-10 * m2["abs_momentum"][i]
            assert abs(m1["abs_position"][i] - m2["abs_position"][i]) <= 1e-10 * m2["abs_position"][i]


# ---- particle walls: Examples/Tests/boundaries/inputs_test_3d_particle_boundaries --------------------
def _boundaries_report(sim, ids):
    from warpx_amd.sim import particle_moments
    out = {}
    for sid, species in zip(ids, ("reflecting_particles", "absorbing_particles", "periodic_particles")):
        m = particle_moments(sim, sid)
        out[species] = {"particle_weight": m["weight"]}
        for i, ax in enumerate("xyz"):
            out[species]["particle_momentum_" + ax] = m["abs_momentum"][i]
            out[species]["particle_position_" + ax] = m["abs_position"][i]
    return out


def _check_boundaries(got):
    gold = json.load(open(os.path.join(HERE, "golden", "particle_boundaries_3d_checksums.json")))
    for species, vals in got.items():
        for key, val in vals.items():
            want = gold["checksums"][species][key]
            assert abs(val - want) <= gold["rtol"] * abs(want), (species, key, val, want)   # zeros must be exact


@pytest.mark.parametrize("which", ["oracle", "host_layer"])
def test_particle_boundaries_golden(oracle, which):
    """Reflecting (x), absorbing (y) and periodic (z) walls: the reference's golden particle checksums after
    8 steps, on the oracle stepper and on the product's host layer over the CPU kernels (retire-in-place,
    compaction before the particles are handed out)."""
    if which == "oracle":
        lib = oracle
    else:
        from tests.oracle_lib import load_host_cpu
        lib = load_host_cpu()
    sim, r, a, p = pec_case.make_boundaries_sim(lib)
    sim.evolve(pec_case.B_MAX_STEP)
    _check_boundaries(_boundaries_report(sim, (r, a, p)))
    assert sim.particles(a).shape[1] == 1      # two of the three absorbing_particles were lost


# ---- moving window + continuous injection + laser antenna + PEC walls:
#      Examples/Physics_applications/laser_acceleration/inputs_test_3d_laser_acceleration ---------------
def _lwfa_report(oracle, sim, electrons):
    from warpx_amd.sim import particle_moments
    out = {"lev=0": {}, "electrons": {}}
    sim.compute_rho()
    for name in ("Bx", "By", "Bz", "Ex", "Ey", "Ez", "jx", "jy", "jz", "rho"):
        out["lev=0"][name] = oracle.cell_centered_abs_sum(C.byref(sim.field_view(name)))
    m = particle_moments(sim, electrons)
    for i, ax in enumerate("xyz"):
        out["electrons"]["particle_momentum_" + ax] = m["abs_momentum"][i]
        out["electrons"]["particle_position_" + ax] = m["abs_position"][i]
    out["electrons"]["particle_weight"] = m["weight"]
    return out


def check_lwfa_against_golden(got):
    gold = json.load(open(os.path.join(HERE, "golden", "laser_acceleration_3d_checksums.json")))
    worst = 0.0
    for group in ("lev=0", "electrons"):
        for key, val in got[group].items():
            want = gold["checksums"][group][key]
            rel = abs(val - want) / abs(want)
            worst = max(worst, rel)
            print(f"{group}.{key}: got {val:.16e} want {want:.16e} rel {rel:.2e}")
            assert rel < gold["rtol"], (group, key, val, want)
    print("worst relative deviation", worst)


@pytest.mark.parametrize("which", ["oracle", "host_layer"])
def test_laser_acceleration_golden(oracle, which):
    """The reference's 3-D laser-wakefield regression (BASELINE config 5 in small: 32x32x256, order 3, window
    moving at c, Gaussian antenna, continuous injection, PEC walls, filter): every field, current, rho and
    particle checksum of the golden file at the reference's tolerance, on the oracle stepper and on the
    product's C++ host layer driving the CPU restatement's kernels."""
    if which == "oracle":
        lib = oracle
    else:
        from tests.oracle_lib import load_host_cpu
        lib = load_host_cpu()
    sim, e = pec_case.make_lwfa_sim(lib)
    assert sim.particles(e).shape[1] == 21780          # 22 x 22 columns x 45 planes with 0 <= z < 12 um
    sim.evolve(pec_case.L_MAX_STEP)
    check_lwfa_against_golden(_lwfa_report(oracle, sim, e))
    assert sim.particles(e).shape[1] == 69212           # 98 planes injected while the window advanced


def test_boosted_laser_wakefield_host_layer_against_the_independent_oracle_stepper(oracle):
    """BASELINE config 5 in small (tests/decks/laser_wakefield_boosted_3d.inputs: gamma = 5, window at c, CKC, Vay, order 3,
    filter, NCI corrector, drifting antenna, boosted continuous injection) -- the product's host layer reading the deck
    (on the CPU kernels) against the INDEPENDENT oracle stepper set up call by call (tests/pec_case.make_boosted_lwfa_sim:
    its own moving window, injection front, antenna, PEC walls, NCI filter and schedule): every regression checksum at
    the reference's 1e-9.  No golden file of the reference pins a 3-D boosted run (its boosted decks draw random beams)."""
    from tests.oracle_lib import load_host_cpu
    from tests.test_inputs_cpu import compare_with_golden
    deck = os.path.join(HERE, "decks", "laser_wakefield_boosted_3d.inputs")
    host = WarpXSim.from_inputs(load_host_cpu(), deck)
    assert host.max_step == pec_case.BOOST_MAX_STEP
    host.evolve(host.max_step)
    got = host.checksum()
    host.close()
    sim, e = pec_case.make_boosted_lwfa_sim(oracle)
    sim.evolve(pec_case.BOOST_MAX_STEP)
    want = _lwfa_report(oracle, sim, e)
    npart = sim.particles(e).shape[1]
    sim.close()
    got["lev=0"].pop("part_per_cell", None)
    worst = compare_with_golden(got, want, 1e-9)
    print("boosted wakefield deck, host layer vs oracle stepper: worst relative deviation", worst, "particles", npart)
    assert npart > 10000
