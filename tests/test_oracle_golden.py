"""Pins the CPU oracle against the reference's own golden vectors for this path
(SURVEY.md 8(c)): the 3-D Langmuir multi-species checksum benchmark and the analytic
Langmuir field test.  CPU only."""
import json
import os

import numpy as np
import pytest

from warpx_amd import _capi, plasma
from warpx_amd.containers import view_to_numpy
from warpx_amd.sim import WarpXSim, field_energy, particle_moments

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def langmuir_run(oracle):
    """Examples/Tests/langmuir/inputs_base_3d run for max_step = 40 on the oracle."""
    n_cell = (64, 64, 64)
    el, lo, hi = plasma.langmuir_3d(n_cell, sign=+1.0)
    po, _, _ = plasma.langmuir_3d(n_cell, sign=-1.0)
    sim = WarpXSim(oracle, n_cell, lo, hi, nox=1, galerkin=1,
                   particle_pusher=_capi.PUSHER_BORIS,
                   current_deposition=_capi.DEPOSIT_ESIRKEPOV, use_filter=0, cfl=1.0)
    e = sim.add_species(-plasma.Q_E, plasma.M_E, el)
    p = sim.add_species(+plasma.Q_E, plasma.M_E, po)
    sim.evolve(40)
    return sim, e, p


def _cc_abs_sum(oracle, sim, name):
    import ctypes as C
    v = sim.field_view(name)
    return oracle.cell_centered_abs_sum(C.byref(v))


def test_dt_matches_reference_script(langmuir_run):
    # Examples/Tests/langmuir/analysis_3d.py:205 quotes dt = 1.203645751e-15
    sim, _, _ = langmuir_run
    assert abs(sim.dt - 1.203645751e-15) / 1.203645751e-15 < 1e-9


def test_golden_checksums(oracle, langmuir_run):
    sim, e, p = langmuir_run
    _check_langmuir_golden(oracle, sim, e, p, "langmuir_multi_3d_checksums.json")


def test_nodal_golden_checksums_pin_direct_deposition(oracle):
    """Examples/Tests/langmuir/inputs_test_3d_langmuir_multi_nodal (direct current deposition, collocated grid,
    same shape factors in all directions for the gather): the only 3-D FDTD golden file of the reference that
    runs doDepositionShapeN.  The collocated grid and its centred-difference solver exist in the CPU
    restatement for this pin only; the deposition and gather routines are the ones the Yee runs use (they
    take the staggering of the arrays they are handed, as the reference's do)."""
    n_cell = (64, 64, 64)
    el, lo, hi = plasma.langmuir_3d(n_cell, sign=+1.0)
    po, _, _ = plasma.langmuir_3d(n_cell, sign=-1.0)
    sim = WarpXSim(oracle, n_cell, lo, hi, nox=1, galerkin=1, particle_pusher=_capi.PUSHER_BORIS,
                   current_deposition=_capi.DEPOSIT_DIRECT, use_filter=0, cfl=1.0,
                   grid_type=_capi.GRID_COLLOCATED)
    e = sim.add_species(-plasma.Q_E, plasma.M_E, el)
    p = sim.add_species(+plasma.Q_E, plasma.M_E, po)
    sim.evolve(40)
    assert tuple(sim.field_view("Ex").stag) == (1, 1, 1) and tuple(sim.field_view("jz").stag) == (1, 1, 1)
    _check_langmuir_golden(oracle, sim, e, p, "langmuir_multi_nodal_3d_checksums.json")


def picmi_langmuir_sim(lib, device=None):
    """inputs_test_3d_langmuir_multi_picmi.py as the reference resolves it (see the golden file's _source)."""
    parts, lo, hi = plasma.half_domain_beam()
    sim = WarpXSim(lib, (64, 64, 64), lo, hi, nox=1, particle_pusher=_capi.PUSHER_BORIS,
                   current_deposition=_capi.DEPOSIT_DIRECT, use_filter=1, cfl=1.0, sort_interval=4)
    assert sim.cfg.galerkin == 0
    return sim, sim.add_species(-plasma.Q_E, plasma.M_E, parts)


def check_picmi_langmuir_golden(oracle, sim, e, cc_abs_sum=None):
    gold = json.load(open(os.path.join(HERE, "golden", "langmuir_multi_picmi_3d_checksums.json")))
    cc = cc_abs_sum or (lambda name: _cc_abs_sum(oracle, sim, name))
    m = particle_moments(sim, e)
    got = {"lev=0": {"Ex": cc("Ex"), "jx": cc("jx")},
           "electrons": {"particle_momentum_x": m["abs_momentum"][0], "particle_position_x": m["abs_position"][0],
                         "particle_position_y": m["abs_position"][1], "particle_position_z": m["abs_position"][2],
                         "particle_weight": m["weight"]}}
    for group, vals in got.items():
        for key, val in vals.items():
            want = gold["checksums"][group][key]
            rel = abs(val - want) / abs(want)
            print(f"{group}.{key}: got {val:.16e} want {want:.16e} rel {rel:.2e}")
            assert rel < gold["rtol"], (group, key, val, want)


@pytest.mark.parametrize("which", ["oracle", "host_layer"])
def test_picmi_golden_checksums_pin_direct_deposition_on_the_yee_grid(oracle, which):
    """Examples/Tests/langmuir/inputs_test_3d_langmuir_multi_picmi.py: direct deposition on the staggered grid
    with the bilinear filter, 8 particles per cell, a plasma edge inside the domain -- the hot path's
    `direct` variant exactly as the reference runs it (gather without Galerkin shapes)."""
    if which == "oracle":
        lib = oracle
    else:
        from tests.oracle_lib import load_host_cpu
        lib = load_host_cpu()
    sim, e = picmi_langmuir_sim(lib)
    sim.evolve(40)
    check_picmi_langmuir_golden(oracle, sim, e)


def _check_langmuir_golden(oracle, sim, e, p, golden_file):
    gold = json.load(open(os.path.join(HERE, "golden", golden_file)))
    rtol = gold["rtol"]
    ref = gold["checksums"]
    got = {}
    for name in ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz"):
        got[name] = _cc_abs_sum(oracle, sim, name)
    sim.compute_rho()
    got["rho"] = _cc_abs_sum(oracle, sim, "rho")
    report = []
    for name, val in got.items():
        want = ref["lev=0"][name]
        report.append((name, val, want, abs(val - want) / abs(want)))
    me = particle_moments(sim, e)
    mp = particle_moments(sim, p)
    pe, pp = ref["electrons"], ref["positrons"]
    report += [
        ("e.px", me["abs_momentum"][0], pe["particle_momentum_x"], None),
        ("e.x", me["abs_position"][0], pe["particle_position_x"], None),
        ("e.y", me["abs_position"][1], pe["particle_position_y"], None),
        ("e.z", me["abs_position"][2], pe["particle_position_z"], None),
        ("e.w", me["weight"], pe["particle_weight"], None),
        ("p.pz", mp["abs_momentum"][2], pp["particle_momentum_z"], None),
        ("p.x", mp["abs_position"][0], pp["particle_position_x"], None),
        ("p.y", mp["abs_position"][1], pp["particle_position_y"], None),
        ("p.z", mp["abs_position"][2], pp["particle_position_z"], None),
    ]
    bad = []
    for name, val, want, _ in report:
        err = abs(val - want) / abs(want)
        print(f"{name:6s} got {val:.16e} want {want:.16e} rel {err:.2e}")
        if not np.isclose(val, want, rtol=rtol, atol=1e-40):
            bad.append((name, val, want, err))
    assert not bad, bad
    assert ref["lev=0"]["part_per_cell"] == 2 * 64 ** 3


def test_langmuir_analytic_field(langmuir_run):
    # Examples/Tests/langmuir/analysis_3d.py:124-131,159-164: max-norm relative error < 5 %
    sim, _, _ = langmuir_run
    t = 40 * sim.dt
    n_cell = (64, 64, 64)
    Eth = plasma.langmuir_analytic_E(n_cell, 40e-6, 2e24, 0.01, t)
    worst = 0.0
    for name, th in zip(("Ex", "Ey", "Ez"), Eth):
        v = sim.field_view(name)
        a = view_to_numpy(v)
        g = v.ng
        a = a[g[0]: v.n[0] - g[0], g[1]: v.n[1] - g[1], g[2]: v.n[2] - g[2]]
        # cell-centre (average the nodal directions), as the plotfile writer does
        for d in range(3):
            if v.stag[d]:
                sl0 = [slice(None)] * 3
                sl1 = [slice(None)] * 3
                sl0[d] = slice(0, -1)
                sl1[d] = slice(1, None)
                a = 0.5 * (a[tuple(sl0)] + a[tuple(sl1)])
        worst = max(worst, np.max(np.abs(a - th)) / np.max(np.abs(th)))
    print("langmuir max-norm rel error", worst)
    assert worst < 5e-2


def test_energy_is_sane(langmuir_run):
    sim, e, p = langmuir_run
    ee, eb = field_energy(sim)
    ke = particle_moments(sim, e)["ekin"] + particle_moments(sim, p)["ekin"]
    assert ee > 0 and eb >= 0 and ke > 0


def test_particle_pusher_golden(oracle):
    """Examples/Tests/particle_pusher/inputs_test_3d_particle_pusher (one positron, E = -v x B from constant external
    particle fields, Higuera-Cary pusher, 10^4 steps) on the oracle stepper: every checksum of the reference's
    test_3d_particle_pusher.json digit for digit -- including x and px, which are pure round-off residue (1e-14 of y and
    py): the restatement keeps the reference's operation order in the pusher, the position update and the gather."""
    from tests import helpers as H
    gold = json.load(open(os.path.join(HERE, "golden", "particle_pusher_3d_checksums.json")))["checksums"]["positron"]
    Lh = 2.077023075927835e+07
    sim = WarpXSim(oracle, (8, 8, 8), (-Lh,) * 3, (Lh,) * 3, nox=1, particle_pusher=_capi.PUSHER_HC)
    c = plasma.C_LIGHT
    one = [np.array([v]) for v in (0.0, 0.0, 0.0, 0.0, 0.0, 19.974984355438178 * c, 0.0)]
    sid = sim.add_species(1.0, 1.0, one)
    oracle.sim_set_external_particle_fields(sim._h, sid, H.d3((-2.994174829214179e+08, 0.0, 0.0)), H.d3((0.0, 0.0, 1.0)))
    sim.evolve(10000)
    p = sim.particles(sid)[:, 0]
    assert abs(p[0]) == gold["particle_position_x"] and abs(p[1]) == gold["particle_position_y"] and p[2] == 0.0
    assert abs(1.0 * p[4]) == gold["particle_momentum_x"] and abs(1.0 * p[5]) == gold["particle_momentum_y"]
    # the analysis script's gate: the orbit stays straight (|x| < 1e-3 m after 10^4 steps; Boris drifts by 2321 m)
    assert abs(p[0]) < 1e-3
    sim.close()
