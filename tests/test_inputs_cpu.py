"""The input-deck front end of the host layer (warpx_amd/csrc/host/WarpXInputs.hpp, Parser.hpp, Checksum.hpp)
on the CPU build of the host layer: expression evaluator, deck syntax, refusal of parameters outside the
path, and whole decks -- ours under tests/decks and, when the reference checkout is present, the
reference's own example decks unmodified -- run to the reference's golden checksums with the library's own
checksum writer (no oracle reducer in the loop)."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from tests.oracle_lib import load_host_cpu
from warpx_amd import _capi
from warpx_amd.sim import WarpXSim

HERE = os.path.dirname(os.path.abspath(__file__))
DECKS = os.path.join(HERE, "decks")
REFERENCE = "/root/reference"


@pytest.fixture(scope="module")
def lib():
    return load_host_cpu()


def _eval(lib, expr, **variables):
    names = (C.c_char_p * max(len(variables), 1))(*[k.encode() for k in variables])
    vals = (C.c_double * max(len(variables), 1))(*variables.values())
    out = C.c_double()
    lib.parser_eval(expr.encode(), len(variables), names, vals, C.byref(out))
    return out.value


@pytest.mark.parametrize("expr,want", [
    ("1+2*3", 7.0), ("(1+2)*3", 9.0), ("2**3**2", 512.0), ("2^3", 8.0), ("-2^2", -4.0), ("2^-1", 0.5),
    ("10/4/5", 0.5), ("1.e-6*2", 2e-6), ("3 - -2", 5.0), ("+4", 4.0), ("1 - 2 - 3", -4.0),
    ("(3>2) * (2<1)", 0.0), ("(3>=3) + (2<=1) + (1==1) + (1!=1)", 2.0), ("1 < 2 and 2 < 3", 1.0), ("0 or 0", 0.0),
    ("sqrt(16) + abs(-2) + fabs(-1)", 7.0), ("if(2>1, 10, 20)", 10.0), ("if(0, 10, 20)", 20.0),
    ("min(3, 4) + max(3, 4) + pow(2, 10)", 1031.0), ("heaviside(-1, 0.5) + heaviside(0, 0.5) + heaviside(2, 0.5)", 1.5),
    ("floor(2.7) + ceil(2.2)", 5.0), ("fmod(7, 4)", 3.0),
    ("sin(pi/2) + cos(0) + exp(0) + log(1) + log10(100) + tanh(0) + atan2(0, 1)", 5.0),
    ("2*pi*clight/clight", 2 * math.pi), ("q_e/m_e", 1.602176634e-19 / 9.1093837015e-31),
    ("epsilon0*mu0*clight^2", 8.8541878128e-12 * 1.25663706212e-06 * 299792458.0 ** 2),
])
def test_parser_values(lib, expr, want):
    assert _eval(lib, expr) == pytest.approx(want, rel=1e-15, abs=1e-300)


def test_parser_variables_and_errors(lib):
    k = 2.0 * 2.0 * math.pi / 40e-6
    got = _eval(lib, "0.01 * sin(k*x) * cos(k*y) * cos(k*z)", k=k, x=3e-6, y=-7e-6, z=1.1e-5)
    assert got == pytest.approx(0.01 * math.sin(k * 3e-6) * math.cos(k * -7e-6) * math.cos(k * 1.1e-5), rel=1e-15)
    assert _eval(lib, "((1.e5*sin(2*pi*(z)/wavelength)) * (z<z2) * (z>z1))", z=1e-7, wavelength=1e-6, z1=-2e-6, z2=2e-6) \
        == pytest.approx(1e5 * math.sin(2 * math.pi * 1e-7 / 1e-6), rel=1e-15)
    for bad in ("1 +", "foo(1)", "unknown_name * 2", "(1", "sqrt(1, 2)", "1 2"):
        with pytest.raises(_capi.WxaError):
            _eval(lib, bad)


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


MINIMAL = """
max_step = 2
amr.n_cell = 8 8 8
geometry.dims = 3
geometry.prob_lo = -1. -1. -1.
geometry.prob_hi =  1.  1.  1.
boundary.field_lo = periodic periodic periodic
boundary.field_hi = periodic periodic periodic
"""


def test_deck_syntax_includes_overrides_and_constants(lib, tmp_path):
    _write(tmp_path, "base.inputs", MINIMAL + "my_constants.n = 4*m   # m is defined by the including file\n")
    deck = _write(tmp_path, "top.inputs", "FILE = base.inputs\nmy_constants.m = 3\namr.n_cell = n n 2*n  # overrides the base\n"
                                          "warpx.cfl=0.5\n")
    sim = WarpXSim.from_inputs(lib, deck)
    v = sim.field_view("Ex")
    assert (v.n[0] - 2 * v.ng[0], v.n[1] - 2 * v.ng[1] - 1, v.n[2] - 2 * v.ng[2] - 1) == (12, 12, 24)
    dt_half = sim.dt
    assert sim.max_step == 2 and sim.species_names == []
    sim.close()
    sim = WarpXSim.from_inputs(lib, deck, overrides=["warpx.cfl = 1.0", "max_step=7"])
    assert sim.dt == pytest.approx(2 * dt_half, rel=1e-14) and sim.max_step == 7
    sim.close()


def test_ckc_solver_from_the_inputs_file(lib, tmp_path):
    """algo.maxwell_solver = ckc: dt = cfl min(dx) / c (CartesianCKCAlgorithm::ComputeMaxDt)."""
    deck = _write(tmp_path, "ckc.inputs", MINIMAL + "algo.maxwell_solver = ckc\nwarpx.cfl = 0.9\n")
    sim = WarpXSim.from_inputs(lib, deck)
    assert sim.dt == pytest.approx(0.9 * 0.25 / 299792458.0, rel=1e-15)
    sim_yee = WarpXSim.from_inputs(lib, _write(tmp_path, "yee.inputs", MINIMAL + "warpx.cfl = 0.9\n"))
    assert sim.dt > sim_yee.dt * 1.5          # cubic cells: sqrt(3) apart
    sim.close()
    sim_yee.close()


@pytest.mark.parametrize("extra,needle", [
    ("algo.maxwell_solver = psatd", "maxwell_solver"),
    ("warpx.gamma_boost = 10.", "boost_direction"),                      # the frame is on the path, its direction is mandatory
    ("warpx.gamma_boost = 10.\nwarpx.boost_direction = x", "boost must be in the z direction"),
    ("boundary.field_lo = pml pml pml", "pml"),
    ("warpx.do_pml = 1", "do_pml"),
    ("amr.max_level = 1", "max_level"),
    ("geometry.dims = 2", "dims"),
    ("warpx.grid_type = collocated", "grid_type"),
    ("algo.field_gathering = momentum-conserving", "field_gathering"),
    ("particles.species_names = e\ne.charge = -q_e\ne.mass = m_e\ne.injection_style = NRandomPerCell\nalgo.particle_shape = 1",
     "injection_style"),
    ("particles.species_names = e\ne.charge = -q_e\ne.mass = m_e\ne.injection_style = nuniformpercell\n"
     "e.num_particles_per_cell_each_dim = 1 1 1\ne.profile = constant\ne.density = 1.\n"
     "e.momentum_distribution_type = maxwell_boltzmann\nalgo.particle_shape = 1", "momentum_distribution_type"),
    ("particles.species_names = e\ne.charge = -q_e\ne.mass = m_e\ne.injection_style = singleparticle\n"
     "e.single_particle_pos = 0 0 0\ne.single_particle_u = 0 0 0\ne.single_particle_weight = 1", "particle_shape"),
    ("warpx.some_new_feature = 1", "some_new_feature"),
    ("my_constants.a = b\nmy_constants.b = a", "my_constants"),
])
def test_parameters_outside_the_path_are_refused_by_name(lib, tmp_path, extra, needle):
    deck = _write(tmp_path, "bad.inputs", MINIMAL + extra + "\n")
    with pytest.raises(_capi.WxaError) as e:
        WarpXSim.from_inputs(lib, deck)
    assert needle in str(e.value)


# golden fixture, quantities excluded because they are round-off residue in the reference itself
OUR_DECKS = [
    ("langmuir_multi_3d.inputs", "langmuir_multi_3d_checksums.json", ()),
    ("langmuir_beam_direct_3d.inputs", "langmuir_multi_picmi_3d_checksums.json", ()),
    ("pec_standing_wave_3d.inputs", "pec_field_3d_checksums.json", ()),
    ("pec_two_particles_3d.inputs", "pec_particle_3d_checksums.json",
     ("By", "jx", "jz", "particle_momentum_z", "particle_position_z")),
    ("particle_walls_3d.inputs", "particle_boundaries_3d_checksums.json", ()),
    ("laser_wakefield_3d.inputs", "laser_acceleration_3d_checksums.json", ()),
    ("laser_injection_3d.inputs", "laser_injection_3d_checksums.json", ()),
    # Higuera-Cary pusher, constant external fields on the particle, 10^4 steps: every digit, including the
    # round-off residues in x and px (the CPU kernels keep the reference's operation order)
    ("particle_pusher_3d.inputs", "particle_pusher_3d_checksums.json", ()),
    # <species>.do_classical_radiation_reaction: Boris + radiation reaction in a constant external B
    ("radiation_reaction_3d.inputs", "radiation_reaction_3d_checksums.json", ()),
    # particles.*_ext_particle_init_style = repeated_plasma_lens: lab frame, lenses shorter than a step (residence
    # correction), and a frame boosted by gamma = 2 (ConvertLabParamsToBoost, MapParticletoBoostedFrame, the lens
    # evaluated in the lab frame and its fields transformed back)
    ("plasma_lens_3d.inputs", "plasma_lens_3d_checksums.json", ()),
    ("plasma_lens_short_3d.inputs", "plasma_lens_short_3d_checksums.json", ()),
    ("plasma_lens_boosted_3d.inputs", "plasma_lens_boosted_3d_checksums.json", ()),
]


def compare_with_golden(got, gold_checksums, rtol, skip=()):
    worst = 0.0
    for group, vals in gold_checksums.items():
        for key, want in vals.items():
            if key in skip or key in ("particle_initialenergy", "particle_regionofinterest"):
                continue
            val = got[group][key]
            rel = abs(val - want) / abs(want) if want != 0 else abs(val)
            worst = max(worst, rel)
            assert rel < rtol, (group, key, val, want, rel)
    return worst


@pytest.mark.parametrize("deck,golden,skip", OUR_DECKS)
def test_decks_reach_the_reference_golden_checksums(lib, deck, golden, skip):
    gold = json.load(open(os.path.join(HERE, "golden", golden)))
    sim = WarpXSim.from_inputs(lib, os.path.join(DECKS, deck))
    sim.evolve(sim.max_step)
    worst = compare_with_golden(sim.checksum(), gold["checksums"], gold["rtol"], skip)
    print(deck, "worst relative deviation", worst)
    sim.close()


REFERENCE_DECKS = [
    ("Examples/Tests/langmuir/inputs_test_3d_langmuir_multi", "test_3d_langmuir_multi", ()),
    ("Examples/Tests/pec/inputs_test_3d_pec_field", "test_3d_pec_field", ()),
    ("Examples/Tests/pec/inputs_test_3d_pec_particle", "test_3d_pec_particle",
     ("By", "jx", "jz", "particle_momentum_z", "particle_position_z")),
    ("Examples/Tests/boundaries/inputs_test_3d_particle_boundaries", "test_3d_particle_boundaries", ()),
    ("Examples/Physics_applications/laser_acceleration/inputs_test_3d_laser_acceleration", "test_3d_laser_acceleration", ()),
    ("Examples/Tests/laser_injection/inputs_test_3d_laser_injection", "test_3d_laser_injection", ()),
    ("Examples/Tests/particle_pusher/inputs_test_3d_particle_pusher", "test_3d_particle_pusher", ()),
    ("Examples/Tests/radiation_reaction/inputs_test_3d_radiation_reaction", "test_3d_radiation_reaction", ()),
    ("Examples/Tests/plasma_lens/inputs_test_3d_plasma_lens", "test_3d_plasma_lens", ()),
    ("Examples/Tests/plasma_lens/inputs_test_3d_plasma_lens_short", "test_3d_plasma_lens_short", ()),
    ("Examples/Tests/plasma_lens/inputs_test_3d_plasma_lens_boosted", "test_3d_plasma_lens_boosted", ()),
]


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference checkout is not on this machine")
@pytest.mark.parametrize("deck,name,skip", REFERENCE_DECKS)
def test_the_reference_decks_run_unmodified(lib, deck, name, skip):
    """The reference's own example inputs files, read where they lie, against its own golden JSON files."""
    gold = json.load(open(os.path.join(REFERENCE, "Regression/Checksum/benchmarks_json", name + ".json")))
    sim = WarpXSim.from_inputs(lib, os.path.join(REFERENCE, deck))
    sim.evolve(sim.max_step)
    compare_with_golden(sim.checksum(), gold, 1e-9, skip)
    sim.close()


M_E, M_P = 9.1093837015e-31, 1.67262192369e-27
BTD_DECKS = [os.path.join(DECKS, "laser_wakefield_btd_3d.inputs")] + (
    [os.path.join(REFERENCE, "Examples/Tests/boosted_diags/inputs_test_3d_laser_acceleration_btd")] if os.path.isdir(REFERENCE) else [])


@pytest.mark.parametrize("deck", BTD_DECKS, ids=["our_deck", "reference_deck_unmodified"][:len(BTD_DECKS)])
def test_back_transformed_snapshot_against_the_reference_golden_file(lib, deck):
    """The whole boosted path at once -- gamma = 10, CKC, Vay, order 3, NCI corrector, moving window, PEC along z, Gaussian
    antenna, continuous injection of a moving plasma, a Gaussian beam, max_step from warpx.zmax_plasma_to_compute_max_step,
    back-transformed fields and particles -- against the reference's golden file for lab-frame snapshot 3
    (Regression/Checksum/benchmarks_json/test_3d_laser_acceleration_btd.json).  This test found the BackTransformed
    diagnostics sampling the fields after the window shift instead of before it (WarpXEvolve.cpp:241): 17 ... 97 % on the
    field sums; with the reference's order E and B agreed to 1.5e-5, jz and rho to 1.3e-3 (round 4).  Round 5 found what was
    left: two behaviours of the reference that depend on the box layout and that this library had replaced by
    layout-independent ones -- the guard points behind a wall AND beyond a periodic face, which amrex::FillBoundary leaves to
    the next PEC pass (one step old; host/BrickComm.hpp), and the charge in the guard columns next to a wall, which
    PEC::ApplyReflectiveBoundarytoRhofield does not fold (WarpX::ApplyRhofieldBoundary).  With the reference's behaviour:
    **every field sum to 3e-10, the back-transformed electrons to 6e-10 -- the reference's own 1e-9**, which is what the
    fixture now states."""
    from tests.helpers import btd_snapshot_checksum, compare_btd_with_golden
    gold = json.load(open(os.path.join(HERE, "golden", "laser_acceleration_btd_3d_checksums.json")))
    sim = WarpXSim.from_inputs(lib, deck)
    assert sim.max_step == 84   # computeMaxStepBoostAccelerator (WarpXInitData.cpp:820-855)
    sim.evolve(sim.max_step)
    info = sim.btd_info(3)
    assert info["n"] == (32, 32, 64) and info["slices"] == 50
    got = btd_snapshot_checksum(sim, 3, ("electrons", "ions", "beam"), (M_E, M_P, M_E))
    worst = compare_btd_with_golden(got, gold)
    print(os.path.basename(deck), "worst relative deviation per group", worst)
    assert "ions" not in got   # the at-rest ions of the snapshot's slab have all left through the lower wall: no entry
    sim.close()


def test_what_the_btd_golden_file_weighs(lib):
    """Where the sums of test_3d_laser_acceleration_btd.json come from (round 5, looking for the 1e-5 ... 1e-3 left against
    it).  Every lab-frame snapshot is swept from the top of the boosted domain to its bottom; its lowest slice (k_lab = 13 of
    13 ... 62) is taken in the cell layer next to the lower PEC wall, where the plasma streams out at -beta c: that one
    slice carries a fifth of sum|rho| and sum|jz| and a quarter of sum|Ex| and sum|By| -- and there rho_lab and jz_lab
    are one quantity, gamma jz' (rho_lab c = beta jz_lab to 0.2 %: the boosted-frame charge next to the wall is nothing
    against the unbalanced current), which is why the golden file's rho and jz miss by the same 1.25e-3 / 1.29e-3 while
    the fields miss by 1e-5.  Also excluded in round 5 (scripts of profiles/round5/README.md): the beta of the Lorentz
    transform (0.995 instead of sqrt(1 - 1/gamma^2) would explain rho and Ey at once -- it moves Ex by 6e-5 the wrong
    way), and 1e-5 perturbations of amplitude, density, cfl, gamma_boost, t_peak, wavelength, antenna position, zmax:
    none has the signature (rho and jz a hundred times the fields).  The wall slice was the right place: what was left
    turned out to be the reference's treatment of the guard points and guard columns NEXT TO THAT WALL (see
    test_back_transformed_snapshot_against_the_reference_golden_file); the shares asserted here are unchanged."""
    deck = BTD_DECKS[0]
    sim = WarpXSim.from_inputs(lib, deck)
    sim.evolve(sim.max_step)
    c, beta = 299792458.0, math.sqrt(1.0 - 1.0 / 100.0)
    share = {}
    for name in ("rho", "jz", "Ex", "By", "Ey", "Bx", "Ez", "jx"):
        prof = np.abs(sim.btd_snapshot(3, name)).sum(axis=(0, 1))
        filled = np.nonzero(prof)[0]
        assert filled.min() == 13 and filled.max() == 62
        share[name] = prof[13] / prof.sum()
    print("share of the slice next to the wall:", {k: round(float(v), 3) for k, v in share.items()})
    assert 0.15 < share["rho"] < 0.3 and 0.15 < share["jz"] < 0.3 and abs(share["rho"] - share["jz"]) < 0.02
    assert share["Ex"] > 0.2 and share["By"] > 0.2 and share["Ey"] > 0.12 and share["Bx"] > 0.12
    assert share["Ez"] < 0.06 and share["jx"] < 0.03
    rho, jz = sim.btd_snapshot(3, "rho"), sim.btd_snapshot(3, "jz")
    wall = np.abs(rho[:, :, 13] * c - beta * jz[:, :, 13]).sum() / np.abs(rho[:, :, 13] * c).sum()
    bulk = np.abs(rho[:, :, 30] * c - beta * jz[:, :, 30]).sum() / np.abs(rho[:, :, 30] * c).sum()
    assert wall < 5e-3 and bulk > 1e-2, (wall, bulk)
    sim.close()


def test_the_two_reference_behaviours_behind_the_btd_residual(lib, monkeypatch):
    """What separated this library from the reference's back-transformed golden file until round 5, switched back on: with
    the layout-independent corners (WXA_REFERENCE_CORNERS=0: guard points behind a wall and beyond a periodic face travel
    with the exchange instead of being left one step old to the next PEC pass, as amrex::FillBoundary leaves them) and the
    rho fold over the guard columns as well (WXA_PEC_RHO_FOLD_GUARD_COLUMNS=1: PEC::ApplyReflectiveBoundarytoRhofield folds a
    box's valid points only, WarpX_PEC.cpp:697) the run misses the golden file by exactly the round-4 signature -- Ex and By
    + 1.5e-5, Ey and Bx - 5.5e-6, jx + 4.3e-5, jz and rho + 1.3e-3 -- and with the rho fold alone switched back only jz and rho
    move (to 4e-6).  The defaults meet the file at 1e-9: test_back_transformed_snapshot_against_the_reference_golden_file."""
    from tests.helpers import btd_snapshot_checksum
    gold = json.load(open(os.path.join(HERE, "golden", "laser_acceleration_btd_3d_checksums.json")))["checksums"]["lev=0"]

    def residuals(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sim = WarpXSim.from_inputs(lib, BTD_DECKS[0])
        sim.evolve(sim.max_step)
        got = btd_snapshot_checksum(sim, 3, ("electrons", "ions", "beam"), (M_E, M_P, M_E))["lev=0"]
        sim.close()
        return {k: (got[k] - v) / v for k, v in gold.items()}

    both = residuals({"WXA_REFERENCE_CORNERS": "0", "WXA_PEC_RHO_FOLD_GUARD_COLUMNS": "1"})
    want = {"Ex": 1.464e-5, "By": 1.502e-5, "Ey": -5.33e-6, "Bx": -5.86e-6, "jx": 4.307e-5, "jz": 1.245e-3, "rho": 1.294e-3}
    for k, v in want.items():
        assert abs(both[k] - v) < 0.02 * abs(v), (k, both[k], v)
    corners_only = residuals({"WXA_REFERENCE_CORNERS": "0", "WXA_PEC_RHO_FOLD_GUARD_COLUMNS": "0"})
    for k in ("Ex", "By", "Ey", "Bx", "jx"):
        assert abs(corners_only[k] - want[k]) < 0.02 * abs(want[k]), (k, corners_only[k])
    assert abs(corners_only["rho"]) < 1e-5 and abs(corners_only["jz"]) < 1e-5, corners_only


def test_the_gaussian_beam_is_not_what_limits_the_btd_pin(lib):
    """The deck without its 10^-14 C beam and with another seed: the field and electron sums of snapshot 3 move by less than
    1e-6 (measured: 5e-7 on the electrons' px without the beam, 1e-8 on the fields, 1e-10 between seeds) -- which is why
    the reference's golden file can be met at 1e-9 with another generator than AMReX's (and why the 1e-5 ... 1e-3 that
    were left against it until round 5 were not the beam's random numbers)."""
    from tests.helpers import btd_snapshot_checksum
    deck = BTD_DECKS[0]
    runs = []
    for ov in ([], ["warpx.random_seed=7"], ["beam.npart=0"]):
        sim = WarpXSim.from_inputs(lib, deck, overrides=ov) if ov else WarpXSim.from_inputs(lib, deck)
        sim.evolve(sim.max_step)
        runs.append(btd_snapshot_checksum(sim, 3, ("electrons", "ions", "beam"), (M_E, M_P, M_E)))
        sim.close()
    for other in runs[1:]:
        for group in ("lev=0", "electrons"):
            for key, v in runs[0][group].items():
                assert abs(other[group][key] - v) <= 1e-6 * abs(v), (group, key, v, other[group][key])
    assert "beam" not in runs[2] and runs[1]["beam"]["particle_weight"] == runs[0]["beam"]["particle_weight"]


def test_gaussian_beam_injection(lib, tmp_path):
    """injection_style = gaussian_beam (PhysicalParticleContainer.cpp:503-677): weights q_tot / (npart q), moments of the
    positions and momenta, the cuts, the 4- and 8-fold symmetrisation, and the same beam on any brick layout."""
    base = """max_step = 1
amr.n_cell = 16 16 16
geometry.dims = 3
geometry.prob_lo = -8.e-6 -8.e-6 -8.e-6
geometry.prob_hi =  8.e-6  8.e-6  8.e-6
boundary.field_lo = periodic periodic periodic
boundary.field_hi = periodic periodic periodic
algo.particle_shape = 1
particles.species_names = beam
beam.charge = -q_e
beam.mass = m_e
beam.injection_style = gaussian_beam
beam.x_rms = 1.e-6
beam.y_rms = 0.5e-6
beam.z_rms = 1.e-6
beam.x_m = 1.e-6
beam.y_m = -1.e-6
beam.z_m = 0.5e-6
beam.npart = 20000
beam.q_tot = -1.e-12
beam.momentum_distribution_type = gaussian
beam.ux_m = 0.1
beam.uz_m = 10.
beam.ux_th = 0.01
beam.uy_th = 0.02
beam.uz_th = 0.5
"""
    deck = tmp_path / "beam.inputs"
    deck.write_text(base)
    c, q_e = 299792458.0, 1.602176634e-19
    sim = WarpXSim.from_inputs(lib, str(deck))
    p = sim.particles(0)
    n = p.shape[1]
    assert n == 20000 and np.all(p[3] == p[3][0]) and abs(p[3][0] * n * q_e / 1e-12 - 1.0) < 1e-12
    for row, mean, rms in ((0, 1e-6, 1e-6), (1, -1e-6, 0.5e-6), (2, 0.5e-6, 1e-6), (4, 0.1 * c, 0.01 * c), (5, 0.0, 0.02 * c), (6, 10 * c, 0.5 * c)):
        assert abs(p[row].mean() - mean) < 4 * rms / math.sqrt(n), (row, p[row].mean(), mean)
        assert abs(p[row].std() / rms - 1.0) < 0.03, (row, p[row].std(), rms)
    ref = p[:, np.lexsort(p[:3])]
    sim.close()
    cut = WarpXSim.from_inputs(lib, str(deck), overrides=["beam.x_cut=1.", "beam.z_cut=0.5"])
    pc = cut.particles(0)
    assert 0.2 * n < pc.shape[1] < 0.35 * n   # erf(1/sqrt 2) * erf(0.5/sqrt 2) = 0.683 * 0.383 = 0.261
    assert np.all(np.abs(pc[0] - 1e-6) <= 1e-6) and np.all(np.abs(pc[2] - 0.5e-6) <= 0.5e-6) and pc[3][0] == p[3][0]
    cut.close()
    for order in (4, 8):
        sym = WarpXSim.from_inputs(lib, str(deck), overrides=["beam.do_symmetrize=1", f"beam.symmetrization_order={order}",
                                                               "beam.x_m=0", "beam.y_m=0"])
        ps = sym.particles(0)
        assert ps.shape[1] == n and abs(ps[3].sum() / (p[3][0] * n) - 1.0) < 1e-12
        assert abs(ps[0].sum()) < 1e-9 * np.abs(ps[0]).sum() and abs(ps[4].sum() - 0.1 * c * 0) < 1e-9 * np.abs(ps[4]).sum()
        sym.close()


def lens_orbit_errors(sim, sid=0, gamma_boost=1.0, short=False):
    """The gate of the reference's analysis script for the plasma lens decks (Examples/Tests/plasma_lens/analysis.py):
    the thick-lens solution x'' = -k^2 x, k^2 = e E' / (m gamma vz^2), through the four lenses and the drifts
    between them, against the simulated transverse position and momentum of the two particles.  Returns the
    relative errors (x, y, ux, uy) and the tolerances (position, velocity)."""
    from scipy.constants import c, e, m_e
    p = sim.particles(sid)
    i0 = int(np.argmax(np.abs(p[0]))), int(np.argmax(np.abs(p[1])))
    zz_sim = [p[2][i0[0]], p[2][i0[1]]]
    if gamma_boost > 1.0:
        uz_boost = math.sqrt(gamma_boost * gamma_boost - 1.0) * c
        t = sim.istep * sim.dt
        zz_sim = [gamma_boost * z + uz_boost * t for z in zz_sim]
    period = 0.5
    starts = [0.1, 0.11, 0.12, 0.13]
    lengths = [0.001, 0.0011, 0.0012, 0.0013] if short else [0.1, 0.11, 0.12, 0.13]
    strengths = [6.e7, 8.e7, 6.e7, 2.e7] if short else [6.e5, 8.e5, 6.e5, 2.e5]
    pos, u, zz = [0.05, 0.04], [0.0, 0.0], 0.05
    uz = 0.5 * c
    gamma = math.sqrt(uz ** 2 / c ** 2 + 1.0)
    vz = uz / gamma
    for i in range(4):
        z_lens = i * period + starts[i]
        kb0 = math.sqrt(e / (m_e * gamma * vz ** 2) * strengths[i])
        for a in range(2):
            v = u[a] / gamma
            x = pos[a] + (z_lens - zz) / vz * v
            x1 = x * math.cos(kb0 * lengths[i]) + (v / vz) / kb0 * math.sin(kb0 * lengths[i])
            v1 = vz * (-kb0 * x * math.sin(kb0 * lengths[i]) + (v / vz) * math.cos(kb0 * lengths[i]))
            pos[a], u[a] = x1, gamma * v1
        zz = z_lens + lengths[i]
    errs = []
    for a in range(2):
        x = pos[a] + (zz_sim[a] - zz) / vz * (u[a] / gamma)
        errs.append(abs((x - p[a][i0[a]]) / x))
    for a in range(2):
        errs.append(abs((u[a] - p[4 + a][i0[a]]) / u[a]))
    return errs, ((0.023, 0.003) if short else (0.02, 0.002))


@pytest.mark.parametrize("deck,boost,short", [("plasma_lens_3d.inputs", 1.0, False), ("plasma_lens_short_3d.inputs", 1.0, True),
                                              ("plasma_lens_boosted_3d.inputs", 2.0, False)])
def test_plasma_lens_orbits_follow_the_thick_lens_solution(lib, deck, boost, short):
    sim = WarpXSim.from_inputs(lib, os.path.join(DECKS, deck))
    sim.evolve(sim.max_step)
    errs, (ptol, vtol) = lens_orbit_errors(sim, gamma_boost=boost, short=short)
    print(deck, errs)
    assert errs[0] < ptol and errs[1] < ptol and errs[2] < vtol and errs[3] < vtol, errs
    sim.close()


def check_boosted_injection(sim):
    """tests/decks/boosted_injection_3d.inputs after its 30 steps: every electron carries u_z = -gamma beta c and the
    weight gamma n dV / ppc (the boosted branch of AddPlasma), the plasma entered through the right edge of a window
    that moved with c while the plasma streamed against it (their closing speed fixes how many layers entered), and
    the layers injected step by step join into one regular lattice: the injection front drifts with the plasma
    (UpdateInjectionPosition), so there is neither a gap nor a doubly filled layer between two injections."""
    c = 299792458.0
    gamma = 3.0
    beta = math.sqrt(1.0 - 1.0 / gamma ** 2)
    p = sim.particles(0)
    n = p.shape[1]
    assert n > 0 and n % 64 == 0                          # whole layers of 8 x 8 cells x 1 x 1 points
    assert np.allclose(p[6], -gamma * beta * c, rtol=1e-12) and np.max(np.abs(p[4:6])) < 1e-12 * c   # self-fields of 1e6 m^-3
    convert = 1.0 / (gamma * (1.0 - beta * 1.0))          # ConvertLabParamsToBoost with the window moving at c
    dz = 16e-6 * convert / 32
    dv = 2e-6 * 2e-6 * dz
    assert np.allclose(p[3], gamma * 1e6 * dv / 2, rtol=1e-12)
    layers = np.unique(np.round(p[2] / (dz / 2), 6))
    assert len(layers) * 64 == n
    assert np.allclose(np.diff(layers), 1.0, atol=1e-5)   # one lattice of spacing dz / 2 across all injections
    # the plasma edge: z0_lab = gamma (z + beta c t) >= 1 um; the left-most layer sits within one lattice spacing of it
    t = sim.istep * sim.dt
    edge = 1e-6 / gamma - beta * c * t
    assert 0.0 <= layers.min() * (dz / 2) - edge < dz / 2 * 1.001
    # ... and the right-most one within a cell of the window's right edge, which moved by c t in whole cells
    right = math.floor(c * t / dz) * dz
    assert 0.0 < right - layers.max() * (dz / 2) <= dz * 1.001


def check_radiation_reaction(sim):
    """The gate of Examples/Tests/radiation_reaction/analysis.py on tests/decks/radiation_reaction_3d.inputs after its 64
    steps: an electron moving along B keeps its Lorentz factor; one gyrating perpendicular to B loses energy as
    gamma(t) = coth(t / tau_c - C), tau_c = 1 / (omega_c^2 t0), t0 = 2 r_e / (3 c), C = -1/2 ln((gamma0 + 1) / (gamma0 - 1))
    (the Landau-Lifshitz solution in a constant field), for 50, 200 and 1000 m_e c, electrons and a positron: 5 %."""
    c, m_e, q_e, r_e = 299792458.0, 9.1093837015e-31, 1.602176634e-19, 2.81794e-15
    b_val = 300 * m_e * 2.0 * math.pi * c / q_e / 1.0e-6
    omega_c = q_e * b_val / m_e
    tau_c = 1.0 / omega_c / omega_c / ((2.0 / 3.0) * r_e / c)
    t = sim.istep * sim.dt
    start = {"ele_para0": (1000.0, True), "ele_perp0": (50.0, False), "ele_perp1": (200.0, False),
             "ele_perp2": (1000.0, False), "pos_perp2": (1000.0, False)}
    for name, (p0, parallel) in start.items():
        u = sim.particles(sim.species_names.index(name))[4:7, 0] / c
        g_end = math.sqrt(1.0 + float(np.dot(u, u)))
        g0 = math.sqrt(1.0 + p0 * p0)
        want = g0 if parallel else 1.0 / math.tanh(t / tau_c + 0.5 * math.log((g0 + 1.0) / (g0 - 1.0)))
        assert abs(g_end - want) / want < 0.05, (name, g_end, want)
        if not parallel:
            assert g_end < 0.999 * g0            # it did radiate


def test_radiation_reaction_analysis(lib):
    sim = WarpXSim.from_inputs(lib, os.path.join(DECKS, "radiation_reaction_3d.inputs"))
    sim.evolve(sim.max_step)
    check_radiation_reaction(sim)
    sim.close()


def check_particle_walls(sim, pos_tol=1e-15):
    """The gate of Examples/Tests/boundaries/analysis.py on tests/decks/particle_walls_3d.inputs after its 8 steps: one of
    the three particles heading for the absorbing walls is left; the two heading for the reflecting walls sit at the
    mirror image of their free flight with the velocity reversed, the two crossing the periodic faces at the wrapped
    position with the velocity unchanged (positions to 1e-15 on the CPU kernels, which keep the reference's operation
    order; the device's reciprocal square root may differ in the last bit per step)."""
    c = 299792458.0
    t = sim.istep * sim.dt
    names = sim.species_names
    refl, absb, peri = (sim.particles(names.index(n)) for n in
                        ("reflecting_particles", "absorbing_particles", "periodic_particles"))
    assert absb.shape[1] == 1 and refl.shape[1] == 2 and peri.shape[1] == 2
    for x0, u0 in ((-0.9, -0.9), (0.91, 0.91)):
        v0 = u0 / math.sqrt(1.0 + u0 * u0) * c
        xa = x0 + v0 * t
        xa = 2.0 * -1.0 - xa if xa < -1.0 else (2.0 * 1.0 - xa if xa > 1.0 else xa)
        i = int(np.argmin(np.abs(refl[0] - xa)))
        assert abs((refl[0][i] - xa) / xa) < pos_tol and refl[4][i] == -u0 * c
    for z0, u0 in ((-0.94, -0.94), (0.95, 0.95)):
        v0 = u0 / math.sqrt(1.0 + u0 * u0) * c
        za = z0 + v0 * t
        za = za + 2.0 if za < -1.0 else (za - 2.0 if za > 1.0 else za)
        i = int(np.argmin(np.abs(peri[2] - za)))
        assert abs((peri[2][i] - za) / za) < pos_tol and peri[6][i] == u0 * c


def test_particle_walls_analysis(lib):
    sim = WarpXSim.from_inputs(lib, os.path.join(DECKS, "particle_walls_3d.inputs"))
    sim.evolve(sim.max_step)
    check_particle_walls(sim)
    sim.close()


def check_boosted_laser(sim):
    """tests/decks/boosted_laser_3d.inputs after its 130 steps: the plane pulse the boosted antenna emitted travels
    along +z with the Lorentz-transformed amplitude E gamma (1 - beta) and wavelength lambda gamma (1 + beta) of the
    lab-frame laser the deck describes, polarised along y, the same on every transverse line."""
    g = 2.0
    b = math.sqrt(1.0 - 1.0 / g ** 2)
    ey = sim.field_valid("Ey")
    peak = np.abs(ey).max()
    assert abs(peak / (1e12 * g * (1.0 - b)) - 1.0) < 0.02, peak
    assert np.abs(sim.field_valid("Ex")).max() < 1e-6 * peak and np.abs(sim.field_valid("Ez")).max() < 1e-6 * peak
    assert np.abs(ey - ey[3:4, 3:4, :]).max() < 1e-6 * peak
    line = ey[3, 3, :]
    dz = 30e-6 / (g * (1.0 - b)) / 512
    k = np.fft.rfftfreq(len(line), d=dz)
    kp = k[np.argmax(np.abs(np.fft.rfft(line)))]
    assert abs(1.0 / kp / (0.8e-6 * g * (1.0 + b)) - 1.0) < 0.03, 1.0 / kp          # one FFT bin is 2.7 %
    # B of a wave along +z polarised along y: Bx = -Ey / c
    bx = sim.field_valid("Bx")
    assert abs(np.abs(bx).max() * 299792458.0 / peak - 1.0) < 0.05     # sampled half a cell and half a step apart


def test_boosted_frame_laser_antenna(lib):
    sim = WarpXSim.from_inputs(lib, os.path.join(DECKS, "boosted_laser_3d.inputs"))
    sim.evolve(sim.max_step)
    check_boosted_laser(sim)
    sim.close()


def test_boosted_frame_laser_wakefield_deck(lib):
    """tests/decks/laser_wakefield_boosted_3d.inputs (BASELINE config 5 in small: gamma = 5, window at c, CKC, Vay,
    order 3, antenna + continuously injected plasma): every piece of the boosted-frame row in one run.  Every electron
    carries gamma n dV, those the pulse has not reached still stream with u_z = -gamma beta c; the pulse has the transformed
    amplitude; the wake drives a longitudinal field and accelerates electrons forward."""
    c = 299792458.0
    g = 5.0
    b = math.sqrt(1.0 - 1.0 / g ** 2)
    sim = WarpXSim.from_inputs(lib, os.path.join(DECKS, "laser_wakefield_boosted_3d.inputs"))
    sim.evolve(sim.max_step)
    p = sim.particles(0)
    convert = 1.0 / (g * (1.0 - b))
    dv = (60e-6 / 16) ** 2 * (16e-6 * convert / 256)
    assert np.allclose(p[3], g * 2e23 * dv, rtol=1e-12)
    fresh = np.abs(p[6] / c + g * b) < 1e-9                  # untouched since their injection
    assert fresh.sum() > 100 and p.shape[1] > 20000          # the layers injected last; the pulse fills most of the window
    assert np.all(np.abs(p[4][fresh]) < 1e-9 * c) and np.all(np.abs(p[5][fresh]) < 1e-9 * c)
    ey, ez = sim.field_valid("Ey"), sim.field_valid("Ez")
    assert 0.5 < np.abs(ey).max() / (16e12 * g * (1.0 - b)) < 1.3     # focusing + coarse transverse grid
    assert np.abs(ez).max() > 0.02 * np.abs(ey).max()                 # the wake
    assert p[6].max() > 0                                             # electrons moving with the pulse
    sim.close()


def test_boosted_frame_injection_through_a_moving_window(lib):
    sim = WarpXSim.from_inputs(lib, os.path.join(DECKS, "boosted_injection_3d.inputs"))
    sim.evolve(sim.max_step)
    check_boosted_injection(sim)
    sim.close()


def test_evolve_in_pieces_without_the_synchronisation_in_between(lib):
    """wxa_sim_set_synchronize_at_end(0): three calls of wxa_sim_evolve + wxa_sim_synchronize are the steps of one long
    call (no PushP(+dt/2) / PushP(-dt/2) pair at the seams, which would round differently)."""
    deck = os.path.join(DECKS, "langmuir_multi_3d.inputs")
    ov = ["max_step = 9", "amr.n_cell = 16 16 16"]
    whole = WarpXSim.from_inputs(lib, deck, overrides=ov)
    whole.evolve(9)
    pieces = WarpXSim.from_inputs(lib, deck, overrides=ov)
    pieces.set_synchronize_at_end(False)
    for n in (2, 4, 3):
        pieces.evolve(n)
    half = pieces.particles(0)[4].copy()                  # momenta still at the half step
    pieces.synchronize()
    pieces.synchronize()                                  # a second call is a no-op
    assert not np.array_equal(half, pieces.particles(0)[4])
    # to round-off, not bit for bit: the CPU kernels' threaded deposition adds in no fixed order, and the final half
    # push gathers before the last periodic wrap in the one case and after it in the other
    for sid in range(len(whole.species_names)):
        a, b = whole.particles(sid), pieces.particles(sid)
        assert np.allclose(a[:3], b[:3], rtol=0, atol=1e-13 * 40e-6) and np.array_equal(a[3], b[3])
        assert np.allclose(a[4:], b[4:], rtol=0, atol=1e-12 * np.abs(a[4:]).max())
    for name in ("Ex", "Ey", "Ez", "jx", "jy", "jz"):          # (B of a Langmuir oscillation is round-off residue)
        fa, fb = whole.field_valid(name), pieces.field_valid(name)
        assert np.abs(fa - fb).max() <= 1e-12 * np.abs(fa).max(), name
    # with the default every call synchronises: same physics, different rounding at the seams
    seams = WarpXSim.from_inputs(lib, deck, overrides=ov)
    for n in (2, 4, 3):
        seams.evolve(n)
    a, b = whole.particles(0)[4], seams.particles(0)[4]
    assert np.allclose(a, b, rtol=0, atol=1e-11 * np.abs(a).max())
    for sim in (whole, pieces, seams):
        sim.close()


def test_constant_external_grid_fields(lib, tmp_path):
    """warpx.B_ext_grid_init_style = constant with non-zero values (WarpXInitData.cpp:940-960): the value at every
    point, guards included; a uniform B stays what it is under the Yee update."""
    deck = tmp_path / "inputs"
    deck.write_text("max_step = 3\namr.n_cell = 8 8 8\namr.max_level = 0\ngeometry.dims = 3\n"
                    "geometry.prob_lo = -1 -1 -1\ngeometry.prob_hi = 1 1 1\n"
                    "boundary.field_lo = periodic periodic periodic\nboundary.field_hi = periodic periodic periodic\n"
                    "algo.particle_shape = 1\nwarpx.B_ext_grid_init_style = constant\n"
                    "warpx.B_external_grid = 0. 0.5 2*0.125\nwarpx.E_ext_grid_init_style = constant\n"
                    "warpx.E_external_grid = 3.e3 0. 0.\n")
    sim = WarpXSim.from_inputs(lib, str(deck))
    assert np.all(sim.field("Bx") == 0.0) and np.all(sim.field("By") == 0.5) and np.all(sim.field("Bz") == 0.25)
    assert np.all(sim.field("Ex") == 3.e3)
    sim.evolve(sim.max_step)
    assert np.all(sim.field_valid("By") == 0.5) and np.all(sim.field_valid("Ex") == 3.e3)
    sim.close()


def test_a_reference_deck_outside_the_path_is_refused(lib):
    if not os.path.isdir(REFERENCE):
        pytest.skip("the reference checkout is not on this machine")
    with pytest.raises(_capi.WxaError) as e:   # PSATD, collocated grid
        WarpXSim.from_inputs(lib, os.path.join(REFERENCE, "Examples/Tests/langmuir/inputs_test_3d_langmuir_multi_psatd_nodal"))
    assert "maxwell_solver" in str(e.value) or "grid_type" in str(e.value)


def test_uniform_plasma_deck_with_random_momenta(lib):
    """The headline workload's own deck (Examples/Physics_applications/uniform_plasma): gaussian momenta come
    from a random stream that no other program reproduces, so the reference's golden
    file is matched exactly where it is deterministic (weight) and statistically elsewhere."""
    if not os.path.isdir(REFERENCE):
        pytest.skip("the reference checkout is not on this machine")
    gold = json.load(open(os.path.join(REFERENCE, "Regression/Checksum/benchmarks_json/test_3d_uniform_plasma.json")))
    deck = os.path.join(REFERENCE, "Examples/Physics_applications/uniform_plasma/inputs_test_3d_uniform_plasma")
    sim = WarpXSim.from_inputs(lib, deck)
    assert sim.max_step == 10
    sim.evolve(sim.max_step)
    got = sim.checksum()
    e, ge = got["electrons"], gold["electrons"]
    assert abs(e["particle_weight"] - ge["particle_weight"]) / ge["particle_weight"] < 1e-12
    for ax in "xyz":   # 131072 particles: sums of |.| fluctuate by a few 1e-3
        assert abs(e["particle_momentum_" + ax] - ge["particle_momentum_" + ax]) / ge["particle_momentum_" + ax] < 0.02
        assert abs(e["particle_position_" + ax] - ge["particle_position_" + ax]) / ge["particle_position_" + ax] < 0.01
    for name in ("Ex", "Ey", "Ez", "jx", "jy", "jz", "rho"):   # noise-driven fields: same scale
        assert 0.5 < got["lev=0"][name] / gold["lev=0"][name] < 2.0, name
    again = WarpXSim.from_inputs(lib, deck)
    again.evolve(again.max_step)
    assert again.checksum() == got          # the stream is seeded: a rerun is identical
    sim.close()
    again.close()


def test_headline_workload_deck_in_small(lib):
    """tests/decks/uniform_plasma_3d.inputs (BASELINE configs[1]) with the grid overridden to 16^3, like a
    command-line override of the reference: builds, steps, conserves the particle count and the weight."""
    sim = WarpXSim.from_inputs(lib, os.path.join(DECKS, "uniform_plasma_3d.inputs"),
                               overrides=['amr.n_cell = 16 16 16', "max_step = 3"])
    sim.evolve(sim.max_step)
    got = sim.checksum()
    assert got["lev=0"]["part_per_cell"] == 8 * 16 ** 3
    assert got["electrons"]["particle_weight"] == pytest.approx(1e25 * (40e-6) ** 3, rel=1e-12)
    assert got["lev=0"]["jx"] > 0 and got["lev=0"]["Ex"] > 0
    sim.close()
