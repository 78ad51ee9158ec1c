"""Step-level parity: the host layer's WarpX::Evolve on the HIP kernels against the
independent CPU oracle stepper, same seeded inputs.  Gate (BASELINE.json north_star):
field energies and particle moments within 1e-10 relative, fp64."""
import json
import os

import numpy as np
import pytest

from tests import helpers as H
from warpx_amd import _capi, plasma
from warpx_amd.containers import FieldArray
from warpx_amd.sim import WarpXSim, field_energy, particle_moments

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
RTOL = 1e-10


def _run(lib, n_cell, species, steps, **kw):
    L = 40e-6
    sim = WarpXSim(lib, n_cell, (-L / 2,) * 3, (L / 2,) * 3, **kw)
    ids = [sim.add_species(q, m, parts) for q, m, parts in species]
    sim.evolve(steps)
    return sim, ids


def _metrics(sim, ids):
    ee, eb = field_energy(sim)
    out = {"E_energy": ee, "B_energy": eb}
    for i in ids:
        m = particle_moments(sim, i)
        out[f"ekin{i}"] = m["ekin"]
        out[f"abs_p{i}"] = m["abs_momentum"]
        out[f"abs_x{i}"] = m["abs_position"]
    return out


def _compare(a, b, rtol=RTOL):
    bad = []
    for k in a:
        va, vb = np.atleast_1d(a[k]), np.atleast_1d(b[k])
        rel = np.max(np.abs(va - vb) / np.maximum(np.abs(vb), 1e-300))
        print(f"{k:10s} rel diff {rel:.2e}")
        if rel > rtol:
            bad.append((k, rel))
    assert not bad, bad


@pytest.mark.parametrize("order,depos,pusher,filt", [
    (1, _capi.DEPOSIT_ESIRKEPOV, _capi.PUSHER_BORIS, 1),   # BASELINE config 1 (uniform_plasma defaults)
    (3, _capi.DEPOSIT_ESIRKEPOV, _capi.PUSHER_BORIS, 1),   # config 2 shape
    (3, _capi.DEPOSIT_DIRECT, _capi.PUSHER_VAY, 0),
    (2, _capi.DEPOSIT_ESIRKEPOV, _capi.PUSHER_VAY, 0),
    (4, _capi.DEPOSIT_ESIRKEPOV, _capi.PUSHER_BORIS, 1),   # order 4 (ShapeFactors.H:67-77, 138-149): on the LDS tiles since round 6
    (4, _capi.DEPOSIT_DIRECT, _capi.PUSHER_BORIS, 0),
])
def test_uniform_plasma_parity(oracle, product, order, depos, pusher, filt):
    n_cell = (32, 32, 32)
    L = 40e-6
    parts = plasma.uniform_plasma(n_cell, (-L / 2,) * 3, (L / 2,) * 3, (1, 1, 2), 1e25, 0.01, seed=12345)
    species = [(-plasma.Q_E, plasma.M_E, parts)]
    kw = dict(nox=order, galerkin=1, particle_pusher=pusher, current_deposition=depos, use_filter=filt,
              sort_interval=4)
    so, io = _run(oracle, n_cell, species, 10, **kw)
    sg, ig = _run(product, n_cell, species, 10, **kw)
    assert abs(so.dt - sg.dt) == 0.0
    _compare(_metrics(sg, ig), _metrics(so, io))
    # fields point by point (valid region), relative to the field scale
    for name in ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz"):
        a, b = sg.field_valid(name), so.field_valid(name)
        assert np.max(np.abs(a - b)) <= 1e-9 * np.max(np.abs(b)), name


@pytest.mark.parametrize("filt", [1, 0])
def test_baseline_config_1_exactly(oracle, product, filt):
    """BASELINE.json configs[0] / SURVEY.md 8(d) C1 itself on the HIP path: 64^3 cells, L = 40 um, one electron per cell
    at the cell centre, n = 1e25 m^-3, u_th = 0.01 c (seed 12345), order 1, Esirkepov, Boris, Yee, cfl 1, the bilinear
    filter on (the reference's default) and off, 10 steps -- the case bench.py's cpu_baseline.serial leg times on the
    oracle.  Gates: 1e-10 on energies and moments, 1e-9 point-wise."""
    n_cell = (64, 64, 64)
    L = 40e-6
    parts = plasma.uniform_plasma(n_cell, (-L / 2,) * 3, (L / 2,) * 3, (1, 1, 1), 1e25, 0.01, seed=12345)
    assert len(parts[0]) == 64 ** 3
    dx = L / 64
    assert np.allclose((np.asarray(parts[0]) + L / 2) / dx % 1.0, 0.5)    # at the cell centre
    species = [(-plasma.Q_E, plasma.M_E, parts)]
    kw = dict(nox=1, galerkin=1, particle_pusher=_capi.PUSHER_BORIS, current_deposition=_capi.DEPOSIT_ESIRKEPOV,
              use_filter=filt, cfl=1.0)
    so, io = _run(oracle, n_cell, species, 10, **kw)
    sg, ig = _run(product, n_cell, species, 10, **kw)
    assert abs(so.dt - sg.dt) == 0.0
    _compare(_metrics(sg, ig), _metrics(so, io))
    for name in ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz"):
        a, b = sg.field_valid(name), so.field_valid(name)
        assert np.max(np.abs(a - b)) <= 1e-9 * np.max(np.abs(b)), name


@pytest.mark.parametrize("interval,merged", [(1, 1), (2, 1), (3, 1), (2, 0)])
def test_sort_cycle_in_the_push_parity(oracle, product, interval, merged, monkeypatch):
    """The periodic cell sort folded into the push, as one special push per cycle (round 6: the sort step's push scatters
    with the last record and counts the next one, keys `interval` free-flight steps ahead; interval 1 -- a sort every step --
    records in the step after the first classic sort instead of never starting) and as a counting and a scattering push
    (round 5, WXA_SORT_MERGED=0): a warm plasma so that particles change cell and tile between sorts, ten steps = several
    cycles, against the oracle stepper at the parity gate."""
    monkeypatch.setenv("WXA_SORT_MERGED", str(merged))
    n_cell = (32, 32, 32)
    L = 40e-6
    parts = plasma.uniform_plasma(n_cell, (-L / 2,) * 3, (L / 2,) * 3, (2, 1, 2), 1e25, 0.1, seed=4321)
    species = [(-plasma.Q_E, plasma.M_E, parts)]
    kw = dict(nox=3, galerkin=1, particle_pusher=_capi.PUSHER_BORIS, current_deposition=_capi.DEPOSIT_ESIRKEPOV, use_filter=1)
    so, io = _run(oracle, n_cell, species, 10, sort_interval=4, **kw)
    sg, ig = _run(product, n_cell, species, 10, sort_interval=interval, **kw)
    _compare(_metrics(sg, ig), _metrics(so, io))
    for name in ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz"):
        a, b = sg.field_valid(name), so.field_valid(name)
        assert np.max(np.abs(a - b)) <= 1e-9 * np.max(np.abs(b)), name


def test_uniform_plasma_parity_in_the_benchmark_regime(oracle, product):
    """HIP path against the oracle stepper where bench.py runs: 8 particles per cell at random positions (Poisson
    cell occupancy: odd runs, unmergeable pairs, tile tails), u_th = 0.01 c with the thermalised crossing rate from the
    first step, order 3, Esirkepov, Boris, filter on, cell sort every 3rd step, 24 steps = 8 sorts, so that the tile
    kernels see stale sorts, deferred crossing particles and stragglers end to end (schedule: WarpXEvolve.cpp:354-455).
    128^3 cells = 1.7e7 particles: ~20 s of oracle time on the GPU box's host cores."""
    n = int(os.environ.get("WXA_BENCH_REGIME_N", "32" if H.HIP_ON_CPU else "128"))
    steps = 12 if H.HIP_ON_CPU else 24
    n_cell = (n, n, n)
    L = 40e-6
    rng = np.random.default_rng(2024)
    npart = 8 * n ** 3
    parts = [(-L / 2 + L * rng.random(npart)) for _ in range(3)]
    parts.append(np.full(npart, 1e25 * (L / n) ** 3 / 8.0))
    parts += [0.01 * plasma.C_LIGHT * rng.standard_normal(npart) for _ in range(3)]
    species = [(-plasma.Q_E, plasma.M_E, parts)]
    kw = dict(nox=3, galerkin=1, particle_pusher=_capi.PUSHER_BORIS, current_deposition=_capi.DEPOSIT_ESIRKEPOV,
              use_filter=1, sort_interval=3)
    sg, ig = _run(product, n_cell, species, steps, **kw)
    mg = _metrics(sg, ig)
    fields_g = {name: sg.field_valid(name) for name in ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz")}
    assert int(sg.particle_view(ig[0]).np) == npart
    sg.close()
    so, io = _run(oracle, n_cell, species, steps, **kw)
    _compare(mg, _metrics(so, io))
    for name, a in fields_g.items():
        b = so.field_valid(name)
        err = np.max(np.abs(a - b)) / np.max(np.abs(b))
        print(f"{name} max point-wise diff / max|field| {err:.2e}")
        assert err <= 1e-9, name


def test_uniform_plasma_fp32_deposition_tiles(oracle, product):
    """A thermal plasma stepped with the ds_add_f32 deposition tiles against the fp64 oracle stepper: energies and
    particle moments at the reference's single-precision regression tolerance, 2e-6
    (Examples/analysis_default_regression.py:18)."""
    n_cell = (32, 32, 32)
    L = 40e-6
    parts = plasma.uniform_plasma(n_cell, (-L / 2,) * 3, (L / 2,) * 3, (2, 2, 2), 1e25, 0.01, seed=4321)
    kw = dict(nox=3, galerkin=1, particle_pusher=_capi.PUSHER_BORIS, current_deposition=_capi.DEPOSIT_ESIRKEPOV,
              use_filter=1, sort_interval=3)
    sg = WarpXSim(product, n_cell, (-L / 2,) * 3, (L / 2,) * 3, **kw)
    ig = [sg.add_species(-plasma.Q_E, plasma.M_E, parts)]
    sg.set_deposit_accumulator(ig[0], _capi.ACC_FP32)
    sg.evolve(12)
    so, io = _run(oracle, n_cell, [(-plasma.Q_E, plasma.M_E, parts)], 12, **kw)
    _compare(_metrics(sg, ig), _metrics(so, io), rtol=2e-6)


def test_langmuir_golden_on_gpu(oracle, product):
    """The reference's own golden checksums (64^3 Langmuir, 40 steps) reproduced by the HIP path."""
    import ctypes as C
    n_cell = (64, 64, 64)
    el, lo, hi = plasma.langmuir_3d(n_cell, sign=+1.0)
    po, _, _ = plasma.langmuir_3d(n_cell, sign=-1.0)
    sim = WarpXSim(product, n_cell, lo, hi, nox=1, galerkin=1, use_filter=0, sort_interval=4)
    e = sim.add_species(-plasma.Q_E, plasma.M_E, el)
    p = sim.add_species(+plasma.Q_E, plasma.M_E, po)
    sim.evolve(40)
    gold = json.load(open(os.path.join(HERE, "golden", "langmuir_multi_3d_checksums.json")))
    ref, rtol = gold["checksums"], gold["rtol"]
    for name in ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz"):
        v = sim.field_view(name)
        host = FieldArray(n_cell, tuple(v.stag), tuple(v.ng), "cpu")
        host.from_numpy(sim.field(name))
        got = oracle.cell_centered_abs_sum(C.byref(host.view))   # the checksum reducer only
        want = ref["lev=0"][name]
        print(name, got, want, abs(got - want) / want)
        assert np.isclose(got, want, rtol=rtol, atol=1e-40), name
    me, mp = particle_moments(sim, e), particle_moments(sim, p)
    assert np.isclose(me["abs_momentum"][0], ref["electrons"]["particle_momentum_x"], rtol=rtol)
    assert np.isclose(mp["abs_momentum"][2], ref["positrons"]["particle_momentum_z"], rtol=rtol)
    for d, k in enumerate(("particle_position_x", "particle_position_y", "particle_position_z")):
        assert np.isclose(me["abs_position"][d], ref["electrons"][k], rtol=rtol)
    # analytic field, < 5 % (Examples/Tests/langmuir/analysis_3d.py:159-164)
    Eth = plasma.langmuir_analytic_E(n_cell, 40e-6, 2e24, 0.01, 40 * sim.dt)
    ex = sim.field_valid("Ex")
    exc = 0.25 * (ex[:, :-1, :-1] + ex[:, 1:, :-1] + ex[:, :-1, 1:] + ex[:, 1:, 1:])
    assert np.max(np.abs(exc - Eth[0])) / np.max(np.abs(Eth[0])) < 5e-2


def _metrics_f64(sim, ids):
    """The reductions of _metrics in float64, species by species and array by array (numpy sums pairwise: ~1e-15 at
    1e8 terms, five orders below the gate) -- the extended-precision copies of particle_moments would take 40 GB at
    the headline sizes."""
    ee, eb = field_energy(sim)
    out = {"E_energy": ee, "B_energy": eb}
    c2 = plasma.C_LIGHT ** 2
    for i in ids:
        _, m = sim.species[i]
        p = sim.particles(i)
        u2 = p[4] * p[4] + p[5] * p[5] + p[6] * p[6]
        out[f"ekin{i}"] = float(np.sum(p[3] * (m * u2 / (1.0 + np.sqrt(1.0 + u2 / c2)))))
        del u2
        out[f"abs_p{i}"] = np.array([float(np.sum(np.abs(m * p[c]))) for c in (4, 5, 6)])
        out[f"abs_x{i}"] = np.array([float(np.sum(np.abs(p[c]))) for c in (0, 1, 2)])
        out[f"np{i}"] = float(p.shape[1])
        del p
    return out


def _full_size_parity(oracle, product, n, species, steps, kw):
    """HIP path, then the oracle stepper, on the same host arrays: energies / moments at 1e-10, E, B, J point-wise at
    1e-9 of each field's scale (the gates of every step test in this file, at the headline size)."""
    import time
    n_cell = (n, n, n)
    names = ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz")
    t0 = time.perf_counter()
    sg, ig = _run(product, n_cell, species, steps, **kw)
    mg = _metrics_f64(sg, ig)
    fields_g = {name: sg.field_valid(name) for name in names}
    dt_g = sg.dt
    sg.close()
    t1 = time.perf_counter()
    so, io = _run(oracle, n_cell, species, steps, **kw)
    t2 = time.perf_counter()
    print(f"HIP path {t1 - t0:.1f} s (incl. upload, download), oracle stepper {t2 - t1:.1f} s for {steps} steps")
    assert abs(so.dt - dt_g) == 0.0
    _compare(mg, _metrics_f64(so, io))
    e_scale = max(np.max(np.abs(so.field_valid(c))) for c in ("Ex", "Ey", "Ez"))
    for name, a in fields_g.items():
        b = so.field_valid(name)
        # B of an electrostatic wave is round-off residue: its scale is that of E / c, not its own maximum
        scale = max(np.max(np.abs(b)), e_scale / plasma.C_LIGHT) if name.startswith("B") else np.max(np.abs(b))
        err = np.max(np.abs(a - b)) / scale
        print(f"{name} max point-wise diff / field scale {err:.2e}")
        assert err <= 1e-9, name
    so.close()


M_PROTON = 1.67262192369e-27   # CODATA 2018, the reference's m_p
FULL_SIZE = pytest.mark.skipif(os.environ.get("WXA_HIP_ON_CPU") == "1" and "WXA_FULL_SIZE_N" not in os.environ,
                               reason="full size: needs the GPU (WXA_FULL_SIZE_N=<n> runs it in small on the CPU model)")


@FULL_SIZE
def test_uniform_plasma_256_against_the_oracle(oracle, product):
    """BASELINE.json config 2 ITSELF -- 256^3 cells, 8 particles per cell, order 3, Esirkepov, Boris, bilinear filter,
    cell sort every 3rd step, thermal start (u_th = 0.01 c, Poisson cell occupancy) -- HIP path against the independent
    oracle stepper: 6 steps = 2 sorts (schedule: WarpXEvolve.cpp:354-455).  What only this size exercises: 32 768
    tiles, 1.3e8-entry index arithmetic, the windowed sort scatter at scale.  1.34e8 particles: ~1 min of oracle time
    on the GPU box's host cores, ~25 GB of host memory."""
    n = int(os.environ.get("WXA_FULL_SIZE_N", "256"))
    L = 40e-6
    rng = np.random.default_rng(256)
    counts = rng.poisson(8.0, n ** 3)
    cell = np.repeat(np.arange(n ** 3, dtype=np.int64), counts)   # cell-ordered: the oracle streams through its fields
    npart = cell.size
    dx = L / n
    parts = []
    for idx in (cell % n, (cell // n) % n, cell // (n * n)):
        parts.append(-L / 2 + (idx + rng.random(npart)) * dx)
    del cell, idx
    parts.append(np.full(npart, 1e25 * dx ** 3 / 8.0))
    parts += [0.01 * plasma.C_LIGHT * rng.standard_normal(npart) for _ in range(3)]
    kw = dict(nox=3, galerkin=1, particle_pusher=_capi.PUSHER_BORIS, current_deposition=_capi.DEPOSIT_ESIRKEPOV,
              use_filter=1, sort_interval=3)
    _full_size_parity(oracle, product, n, [(-plasma.Q_E, plasma.M_E, parts)], 6, kw)


@FULL_SIZE
def test_direct_vay_ckc_256_against_the_oracle(oracle, product):
    """The other branches of the hot path at the headline size, in one run: 256^3 cells, 8 particles per cell (Poisson
    occupancy, thermal), order 2, direct deposition with the gather the reference pairs it with (no Galerkin shapes,
    Source/WarpX.cpp:1208-1214), Vay pusher, CKC solver (its two extra guard exchanges per step), filter on, sort every 2nd
    step, 4 steps = 2 sorts -- HIP path against the oracle stepper.  (CartesianCKCAlgorithm.H:129-272, CurrentDeposition.H:
    48-335, UpdateMomentumVay.H.)"""
    n = int(os.environ.get("WXA_FULL_SIZE_N", "256"))
    L = 40e-6
    rng = np.random.default_rng(257)
    counts = rng.poisson(8.0, n ** 3)
    cell = np.repeat(np.arange(n ** 3, dtype=np.int64), counts)
    npart = cell.size
    dx = L / n
    parts = []
    for idx in (cell % n, (cell // n) % n, cell // (n * n)):
        parts.append(-L / 2 + (idx + rng.random(npart)) * dx)
    del cell, idx
    parts.append(np.full(npart, 1e25 * dx ** 3 / 8.0))
    parts += [0.01 * plasma.C_LIGHT * rng.standard_normal(npart) for _ in range(3)]
    kw = dict(nox=2, galerkin=0, particle_pusher=_capi.PUSHER_VAY, current_deposition=_capi.DEPOSIT_DIRECT,
              use_filter=1, sort_interval=2, maxwell_solver=_capi.SOLVER_CKC)
    _full_size_parity(oracle, product, n, [(-plasma.Q_E, plasma.M_E, parts)], 4, kw)


@FULL_SIZE
def test_langmuir_256_against_the_oracle(oracle, product):
    """BASELINE.json config 3 at full size against the oracle stepper: Langmuir wave, 256^3 cells, e- / e+, 8 particles
    per cell on the regular lattice, order 3, Esirkepov, no filter (Examples/Tests/langmuir/inputs_base_3d scaled up),
    6 steps with a sort every 4th.  2.7e8 particles."""
    n = int(os.environ.get("WXA_FULL_SIZE_N", "256"))
    n_cell = (n, n, n)
    el, lo, hi = plasma.langmuir_3d(n_cell, ppc=(2, 2, 2), sign=+1.0)
    po, _, _ = plasma.langmuir_3d(n_cell, ppc=(2, 2, 2), sign=-1.0)
    kw = dict(nox=3, galerkin=1, particle_pusher=_capi.PUSHER_BORIS, current_deposition=_capi.DEPOSIT_ESIRKEPOV,
              use_filter=0, sort_interval=4)
    _full_size_parity(oracle, product, n, [(-plasma.Q_E, plasma.M_E, el), (+plasma.Q_E, plasma.M_E, po)], 6, kw)


@FULL_SIZE
def test_langmuir_256_electrons_and_protons_against_the_oracle(oracle, product):
    """BASELINE.json config 3 as it is worded -- "2 species e+i": the same Langmuir wave with protons (at rest, m_p) as the
    second species instead of positrons, 256^3 cells, 8 particles per cell and species, order 3, Esirkepov, 6 steps with a
    sort every 4th, HIP path against the oracle stepper (same gates as the e- / e+ case above)."""
    n = int(os.environ.get("WXA_FULL_SIZE_N", "256"))
    n_cell = (n, n, n)
    el, lo, hi = plasma.langmuir_3d(n_cell, ppc=(2, 2, 2), sign=+1.0)
    io, _, _ = plasma.langmuir_3d(n_cell, ppc=(2, 2, 2), sign=-1.0)
    for c in (4, 5, 6):
        io[c][:] = 0.0
    kw = dict(nox=3, galerkin=1, particle_pusher=_capi.PUSHER_BORIS, current_deposition=_capi.DEPOSIT_ESIRKEPOV,
              use_filter=0, sort_interval=4)
    _full_size_parity(oracle, product, n, [(-plasma.Q_E, plasma.M_E, el), (+plasma.Q_E, M_PROTON, io)], 6, kw)


def test_back_transformed_snapshot_against_the_reference_golden_file_on_the_hip_path(product):
    """tests/decks/laser_wakefield_btd_3d.inputs (the reference's 3-D boosted wakefield deck with back-transformed
    diagnostics: gamma = 10, CKC, Vay, order 3, NCI corrector, moving window, PEC along z, Gaussian antenna, continuous
    injection, Gaussian beam, max_step from warpx.zmax_plasma_to_compute_max_step) on the HIP path: lab-frame snapshot 3
    against the reference's golden file at the reference's own 1e-9 (round 5; measured on the MI355X: fields 2.8e-10,
    back-transformed electrons 5.7e-10 -- profiles/round5/README.md; until then E and B 1.5e-5, jz and rho 1.3e-3)."""
    import json
    from tests.helpers import btd_snapshot_checksum, compare_btd_with_golden
    from warpx_amd.sim import WarpXSim
    here = os.path.dirname(os.path.abspath(__file__))
    gold = json.load(open(os.path.join(here, "golden", "laser_acceleration_btd_3d_checksums.json")))
    sim = WarpXSim.from_inputs(product, os.path.join(here, "decks", "laser_wakefield_btd_3d.inputs"))
    assert sim.max_step == 84
    sim.evolve(sim.max_step)
    got = btd_snapshot_checksum(sim, 3, ("electrons", "ions", "beam"), (plasma.M_E, M_PROTON, plasma.M_E))
    worst = compare_btd_with_golden(got, gold)
    print("worst relative deviation per group", worst)
    sim.close()


@pytest.mark.skipif(os.environ.get("WXA_HIP_ON_CPU") == "1", reason="full size: needs the GPU")
def test_langmuir_256_two_species_full_size(product):
    """BASELINE.json config 3 at full size (256^3, e-/e+, 8 ppc, Esirkepov, order 3, 40 steps) through
    size-independent properties -- the CPU oracle cannot run 2.7e8 particles in test time:
    (i) the reference's analytic gate (Examples/Tests/langmuir/analysis_3d.py: max-norm error < 5 %),
    (ii) Gauss' law div E = rho/eps0 kept to round-off (1e-10 of max rho/eps0) by the charge-conserving deposition,
    (iii) total energy conserved to 1e-3 over the run, (iv) particle count and weights unchanged."""
    import ctypes as C
    import torch
    from warpx_amd.containers import STAG, FieldArray, ParticleArrays, grid_geom
    n = 256
    n_cell = (n, n, n)
    L = 40e-6
    lo, hi = (-L / 2,) * 3, (L / 2,) * 3
    dev = "cuda"

    def species(sign):
        # plasma.langmuir_3d evaluated on the device (same formulas)
        k = 2.0 * 2.0 * np.pi / L
        wp = np.sqrt(2.0 * 2e24 * plasma.Q_E ** 2 / (plasma.EP0 * plasma.M_E))
        a = sign * 0.01 * k / (wp / plasma.C_LIGHT)
        dx = L / n
        ppc = 2
        cell = torch.arange(n ** 3, device=dev)
        ip = torch.arange(ppc ** 3, device=dev)
        f64 = torch.float64
        r = [(0.5 + (ip // (ppc * ppc)).to(f64)) / ppc, (0.5 + ((ip % (ppc * ppc)) // ppc).to(f64)) / ppc,
             (0.5 + (ip % ppc).to(f64)) / ppc]
        idx = [cell % n, (cell // n) % n, cell // (n * n)]
        out = torch.empty((7, n ** 3 * ppc ** 3), dtype=f64, device=dev)
        for d in range(3):
            out[d] = (lo[d] + (idx[d].to(f64)[:, None] + r[d][None, :]) * dx).reshape(-1)
        out[3] = 2e24 * dx ** 3 / ppc ** 3
        sx, cx = torch.sin(k * out[0]), torch.cos(k * out[0])
        sy, cy = torch.sin(k * out[1]), torch.cos(k * out[1])
        sz, cz = torch.sin(k * out[2]), torch.cos(k * out[2])
        out[4] = a * sx * cy * cz * plasma.C_LIGHT
        out[5] = a * cx * sy * cz * plasma.C_LIGHT
        out[6] = a * cx * cy * sz * plasma.C_LIGHT
        pa = ParticleArrays(out.shape[1], dev)
        pa.data = out
        return pa

    sim = WarpXSim(product, n_cell, lo, hi, nox=3, galerkin=1, use_filter=0, sort_interval=4)
    ids = [sim.add_species(-plasma.Q_E, plasma.M_E, species(+1.0)),
           sim.add_species(+plasma.Q_E, plasma.M_E, species(-1.0))]
    torch.cuda.empty_cache()

    def total_energy():
        e = 0.0
        dV = (L / n) ** 3
        for name, c in (("Ex", plasma.EP0), ("Ey", plasma.EP0), ("Ez", plasma.EP0),
                        ("Bx", 1 / plasma.MU0), ("By", 1 / plasma.MU0), ("Bz", 1 / plasma.MU0)):
            a = sim.field_valid(name)
            sl = tuple(slice(0, n) for _ in range(3))   # unique points of the periodic grid
            e += 0.5 * c * dV * float(np.sum(a[sl].astype(np.longdouble) ** 2))
        for i in ids:
            v = sim.particle_view(i)
            npart = int(v.np)
            u2 = torch.zeros(npart, dtype=torch.float64, device=dev)
            from warpx_amd.distributed import _as_tensor
            w = _as_tensor(v.w, 8 * npart, True).view(torch.float64)
            for ptr in (v.ux, v.uy, v.uz):
                c_ = _as_tensor(ptr, 8 * npart, True).view(torch.float64)
                u2 += c_ * c_
            gamma = torch.sqrt(1.0 + u2 / plasma.C_LIGHT ** 2)
            e += float(torch.sum(w * plasma.M_E * u2 / (1.0 + gamma)))
        return e

    e0 = total_energy()
    sim.evolve(40)
    e1 = total_energy()
    assert abs(e1 - e0) / e0 < 1e-3, (e0, e1)
    for i in ids:
        assert int(sim.particle_view(i).np) == n ** 3 * 8
    # (i) analytic field
    Eth = plasma.langmuir_analytic_E(n_cell, L, 2e24, 0.01, 40 * sim.dt)
    ex = sim.field_valid("Ex")
    exc = 0.25 * (ex[:, :-1, :-1] + ex[:, 1:, :-1] + ex[:, :-1, 1:] + ex[:, 1:, 1:])
    err = np.max(np.abs(exc - Eth[0])) / np.max(np.abs(Eth[0]))
    print("langmuir 256^3 max-norm rel error", err)
    assert err < 5e-2
    # (ii) Gauss' law on the nodes: div E - rho/eps0 = 0 to round-off
    ng = 5
    rho = FieldArray(n_cell, STAG["rho"], (ng,) * 3, dev)
    g = grid_geom(lo, (L / n,) * 3, (0, 0, 0), (ng,) * 3)
    for i, q in zip(ids, (-plasma.Q_E, plasma.Q_E)):
        v = sim.particle_view(i)
        product.deposit_charge(C.byref(v), C.byref(rho.view), C.byref(g), q, 3, None)
    product.sum_boundary_periodic(C.byref(rho.view), (C.c_int * 3)(ng, ng, ng), (C.c_int * 3)(1, 1, 1), None)
    product.device_synchronize()
    r = rho.valid()[:n, :n, :n]
    ey, ez = sim.field_valid("Ey"), sim.field_valid("Ez")
    dx = L / n
    # node (i,j,k): (Ex(i,j,k) - Ex(i-1,j,k))/dx + ... with periodic wrap of the cell-centred direction
    div = ((ex[:n, :n, :n] - np.roll(ex[:n, :n, :n], 1, axis=0)) + (ey[:n, :n, :n] - np.roll(ey[:n, :n, :n], 1, axis=1))
           + (ez[:n, :n, :n] - np.roll(ez[:n, :n, :n], 1, axis=2))) / dx
    resid = np.max(np.abs(div - r / plasma.EP0)) / np.max(np.abs(r / plasma.EP0))
    print("gauss residual", resid)
    assert resid < 1e-10   # 6.0e-12 on the MI355X (profiles/round3): atomic summation order of rho and J, nothing else


def test_pec_field_golden_on_gpu(oracle, product):
    """Examples/Tests/pec/inputs_test_3d_pec_field on the HIP path: the reference's golden checksums at the
    reference's tolerance, its analysis (standing wave of twice the amplitude, Ey = 0 on the walls), and the
    CPU stepper bit for bit (stencils and boundary kernels keep the reference's operation order)."""
    import ctypes as C

    from tests import pec_case
    sim = pec_case.make_sim(product)
    sim.evolve(pec_case.MAX_STEP)
    gold = json.load(open(os.path.join(HERE, "golden", "pec_field_3d_checksums.json")))
    ref = pec_case.make_sim(oracle)
    ref.evolve(pec_case.MAX_STEP)
    for name in ("Ex", "Ey", "Ez", "Bx", "By", "Bz"):     # valid points (the corner guards differ by design, see
        assert np.array_equal(sim.field_valid(name), ref.field_valid(name)), name   # tests/test_pec_golden.py)
    for name, want in gold["checksums"]["lev=0"].items():
        got = oracle.cell_centered_abs_sum(C.byref(ref.field_view(name)))   # == the GPU field, checked above
        assert abs(got - want) / want < gold["rtol"]
    ey = sim.field_valid("Ey")
    e_th = 2.0 * pec_case.EY_IN
    assert abs(ey.max() - e_th) / e_th < 0.01 and abs(ey.min() + e_th) / e_th < 0.01
    assert np.all(ey[:, :, 0] == 0.0) and np.all(ey[:, :, -1] == 0.0)


@H.FIRST_GPU_RUN
def test_pec_particle_golden_on_gpu(oracle, product):
    """Examples/Tests/pec/inputs_test_3d_pec_particle on the HIP path: two particles 0.004 cell from a PEC
    wall (image-charge fold of J, mirrored E/B guards in the gather, LDS tiles reaching behind the wall)
    against the reference's golden checksums (quantities that are not round-off residue, see
    tests/test_pec_golden.py) at the reference's tolerance, and against the CPU stepper at 1e-10."""
    import ctypes as C

    from tests import pec_case
    from tests.test_pec_golden import PINNED_FIELDS, PINNED_MOMENTS
    sim, e, p = pec_case.make_particle_sim(product)
    sim.evolve(pec_case.P_MAX_STEP)
    ref, re_, rp = pec_case.make_particle_sim(oracle)
    ref.evolve(pec_case.P_MAX_STEP)
    gold = json.load(open(os.path.join(HERE, "golden", "pec_particle_3d_checksums.json")))
    for name in PINNED_FIELDS:
        a, b = sim.field_valid(name), ref.field_valid(name)
        assert np.max(np.abs(a - b)) <= 1e-10 * np.max(np.abs(b)), name
        # the checksum reducer runs on the host: feed it the GPU field through the oracle run's array
        ref.set_field(name, sim.field(name))
        got = oracle.cell_centered_abs_sum(C.byref(ref.field_view(name)))
        want = gold["checksums"]["lev=0"][name]
        assert abs(got - want) / want < gold["rtol"], name
    for (s1, s2, species) in ((e, re_, "electron"), (p, rp, "proton")):
        m1 = particle_moments(sim, s1)
        for key in PINNED_MOMENTS:
            kind, ax = key.split("_")
            got = m1["abs_" + kind]["xyz".index(ax)]
            want = gold["checksums"][species]["particle_" + key]
            assert abs(got - want) / want < gold["rtol"], (species, key)


@H.FIRST_GPU_RUN
def test_particle_boundaries_golden_on_gpu(product):
    """Examples/Tests/boundaries/inputs_test_3d_particle_boundaries on the HIP path: the reference's golden
    particle checksums (reflecting, absorbing, periodic walls) at the reference's tolerance."""
    from tests import pec_case
    from tests.test_pec_golden import _boundaries_report, _check_boundaries
    sim, r, a, p = pec_case.make_boundaries_sim(product)
    sim.evolve(pec_case.B_MAX_STEP)
    _check_boundaries(_boundaries_report(sim, (r, a, p)))
    assert sim.particles(a).shape[1] == 1


@H.FIRST_GPU_RUN
def test_laser_acceleration_golden_on_gpu(oracle, product):
    """Examples/Physics_applications/laser_acceleration/inputs_test_3d_laser_acceleration on the HIP path (moving
    window at c, continuous injection, Gaussian antenna, PEC walls, filter, order 3): the reference's golden field,
    current, charge-density and particle checksums at the reference's tolerance."""
    from tests import pec_case
    from tests.test_pec_golden import check_lwfa_against_golden
    sim, e = pec_case.make_lwfa_sim(product)
    sim.evolve(pec_case.L_MAX_STEP)
    ref, _ = pec_case.make_lwfa_sim(oracle)            # only as the host-side array the checksum reducer reads
    got = {"lev=0": {}, "electrons": {}}
    import ctypes as C
    sim.compute_rho()
    ref.compute_rho()                                  # creates the oracle-side rho array that is overwritten below
    for name in ("Bx", "By", "Bz", "Ex", "Ey", "Ez", "jx", "jy", "jz", "rho"):
        ref.set_field(name, sim.field(name))
        got["lev=0"][name] = oracle.cell_centered_abs_sum(C.byref(ref.field_view(name)))
    m = particle_moments(sim, e)
    for i, ax in enumerate("xyz"):
        got["electrons"]["particle_momentum_" + ax] = m["abs_momentum"][i]
        got["electrons"]["particle_position_" + ax] = m["abs_position"][i]
    got["electrons"]["particle_weight"] = m["weight"]
    check_lwfa_against_golden(got)
    assert sim.particles(e).shape[1] == 69212


@H.FIRST_GPU_RUN
def test_picmi_langmuir_golden_on_gpu(oracle, product):
    """Examples/Tests/langmuir/inputs_test_3d_langmuir_multi_picmi.py on the HIP path: direct deposition on the
    Yee grid (LDS tiles), bilinear filter, gather without Galerkin shapes, 8 ppc, sorted every 4 steps: the
    reference's golden checksums at the reference's tolerance."""
    import ctypes as C

    from tests.test_oracle_golden import check_picmi_langmuir_golden, picmi_langmuir_sim
    sim, e = picmi_langmuir_sim(product)
    sim.evolve(40)
    ref, _ = picmi_langmuir_sim(oracle)                # host-side arrays for the checksum reducer

    def cc(name):
        ref.set_field(name, sim.field(name))
        return oracle.cell_centered_abs_sum(C.byref(ref.field_view(name)))
    check_picmi_langmuir_golden(oracle, sim, e, cc)


@H.FIRST_GPU_RUN
@pytest.mark.parametrize("deck,golden,skip", [
    ("langmuir_multi_3d.inputs", "langmuir_multi_3d_checksums.json", ()),
    ("langmuir_beam_direct_3d.inputs", "langmuir_multi_picmi_3d_checksums.json", ()),
    ("pec_standing_wave_3d.inputs", "pec_field_3d_checksums.json", ()),
    ("pec_two_particles_3d.inputs", "pec_particle_3d_checksums.json",
     ("By", "jx", "jz", "particle_momentum_z", "particle_position_z")),
    ("particle_walls_3d.inputs", "particle_boundaries_3d_checksums.json", ()),
    ("laser_wakefield_3d.inputs", "laser_acceleration_3d_checksums.json", ()),
    ("laser_injection_3d.inputs", "laser_injection_3d_checksums.json", ()),
    # Higuera-Cary, constant external fields, 10^4 steps: x and px are pure round-off residue (1e-14 of y and py),
    # reproduced digit for digit by the CPU kernels but not by contracted / rsqrt device arithmetic
    ("particle_pusher_3d.inputs", "particle_pusher_3d_checksums.json", ("particle_momentum_x", "particle_position_x")),
    ("radiation_reaction_3d.inputs", "radiation_reaction_3d_checksums.json", ()),
    ("plasma_lens_3d.inputs", "plasma_lens_3d_checksums.json", ()),
    ("plasma_lens_short_3d.inputs", "plasma_lens_short_3d_checksums.json", ()),
    ("plasma_lens_boosted_3d.inputs", "plasma_lens_boosted_3d_checksums.json", ()),
])
def test_decks_reach_the_reference_golden_checksums_on_gpu(product, deck, golden, skip):
    """tests/decks/*.inputs through wxa_sim_create_from_inputs, wxa_sim_evolve and wxa_sim_checksum_json on the HIP
    path: the reference's golden checksums at the reference's tolerance, with no oracle code in the loop."""
    from tests.test_inputs_cpu import compare_with_golden
    gold = json.load(open(os.path.join(HERE, "golden", golden)))
    sim = WarpXSim.from_inputs(product, os.path.join(HERE, "decks", deck))
    sim.evolve(sim.max_step)
    got = sim.checksum()
    compare_with_golden(got, gold["checksums"], gold["rtol"], skip)
    if deck.startswith("particle_pusher"):
        # the gate of the reference's analysis script (Examples/Tests/particle_pusher/analysis.py): the force-free orbit
        # stays straight with the Higuera-Cary pusher, |x| < 1e-3 m after 10^4 steps (Boris drifts by 2321 m)
        assert got["positron"]["particle_position_x"] < 1e-3
    if deck.startswith("radiation_reaction"):
        from tests.test_inputs_cpu import check_radiation_reaction   # Examples/Tests/radiation_reaction/analysis.py
        check_radiation_reaction(sim)
    if deck.startswith("particle_walls"):
        from tests.test_inputs_cpu import check_particle_walls   # the gate of Examples/Tests/boundaries/analysis.py
        check_particle_walls(sim, pos_tol=1e-13)
    if deck.startswith("plasma_lens"):
        # the gate of Examples/Tests/plasma_lens/analysis.py: the thick-lens orbit
        from tests.test_inputs_cpu import lens_orbit_errors
        short = "short" in deck
        errs, (ptol, vtol) = lens_orbit_errors(sim, gamma_boost=2.0 if "boosted" in deck else 1.0, short=short)
        assert errs[0] < ptol and errs[1] < ptol and errs[2] < vtol and errs[3] < vtol, errs
    sim.close()


def test_boosted_frame_injection_through_a_moving_window_on_gpu(product):
    """tests/decks/boosted_injection_3d.inputs on the HIP path: the boosted branch of wxa_add_plasma behind the moving
    window, checked by the lattice it must leave (tests/test_inputs_cpu.py::check_boosted_injection)."""
    from tests.test_inputs_cpu import check_boosted_injection
    sim = WarpXSim.from_inputs(product, os.path.join(HERE, "decks", "boosted_injection_3d.inputs"))
    sim.evolve(sim.max_step)
    check_boosted_injection(sim)
    sim.close()


@pytest.mark.parametrize("sort_behind_shift", [False, True])
def test_boosted_frame_laser_wakefield_deck_on_gpu(oracle, product, sort_behind_shift, monkeypatch):
    """tests/decks/laser_wakefield_boosted_3d.inputs (BASELINE config 5 in small, with the NCI corrector) on the HIP path
    against the INDEPENDENT oracle stepper set up call by call (tests/pec_case.make_boosted_lwfa_sim: its own window,
    injection front, antenna, walls, NCI filter, schedule): every regression checksum (fields, J, rho, particle sums)
    at the reference's 1e-9.  (Round 2 compared with the same host layer on the CPU kernels only.)
    sort_behind_shift: with the step's periodic sort also behind a window shift's sort (WXA_SORT_BEHIND_SHIFT=1; skipped by
    default since round 5) -- the order of the particles is all that differs."""
    if sort_behind_shift:
        monkeypatch.setenv("WXA_SORT_BEHIND_SHIFT", "1")
    from tests import pec_case
    from tests.test_inputs_cpu import compare_with_golden
    from tests.test_pec_golden import _lwfa_report
    deck = os.path.join(HERE, "decks", "laser_wakefield_boosted_3d.inputs")
    ref, e = pec_case.make_boosted_lwfa_sim(oracle)
    ref.evolve(pec_case.BOOST_MAX_STEP)
    want = _lwfa_report(oracle, ref, e)
    npart = ref.particles(e).shape[1]
    ref.close()
    sim = WarpXSim.from_inputs(product, deck)
    assert sim.max_step == pec_case.BOOST_MAX_STEP
    sim.evolve(sim.max_step)
    got = sim.checksum()
    assert got["lev=0"].pop("part_per_cell") == npart
    worst = compare_with_golden(got, want, 1e-9)
    print("boosted wakefield deck, HIP path vs oracle stepper: worst relative deviation", worst)
    sim.close()


def test_back_transformed_fields_on_gpu(oracle, product):
    """Lab-frame snapshots (wxa_sim_add_btd: BTDiagnostics.cpp, BackTransformFunctor.cpp, BackTransformParticleFunctor.cpp) of
    config 5 in small from the HIP path against the oracle stepper's own back-transformation: three snapshots, 50 steps, every
    field component and every particle row at 1e-9 of its scale (tests/test_btd_cpu.py has the host layer on the CPU kernels and the lab-frame physics)."""
    from tests import pec_case
    out, parts = [], []
    for lib in (product, oracle):
        sim, e = pec_case.make_boosted_lwfa_sim(lib)
        sim.add_btd(3, 12 * sim.dt * pec_case.BOOST_GAMMA, buffer_size=32, write_species=True)
        sim.evolve(50)
        out.append(([sim.btd_info(i) for i in range(3)],
                    [{c: sim.btd_snapshot(i, c) for c in WarpXSim.BTD_COMPONENTS} for i in range(3)]))
        parts.append([sim.btd_particles(i, e) for i in range(3)])
        sim.close()
    (ig, dg), (io, do) = out
    for i in range(3):   # the back-transformed electrons (wxa_btd_select_particles): same set, any order
        a, b = (q[:, np.lexsort((q[2], np.round(q[1] / 1e-10), np.round(q[0] / 1e-10)))] for q in (parts[0][i], parts[1][i]))
        assert a.shape == b.shape and (i > 0 or a.shape[1] > 100)
        for row in range(7):
            assert np.max(np.abs(a[row] - b[row])) <= 1e-9 * max(np.max(np.abs(b[row])), 1e-300), (i, row)
    for i in range(3):
        assert ig[i]["n"] == io[i]["n"] and ig[i]["slices"] == io[i]["slices"] > 0
        for c in WarpXSim.BTD_COMPONENTS:
            scale = np.max(np.abs(do[i][c]))
            assert scale > 0 and np.max(np.abs(dg[i][c] - do[i][c])) <= 1e-9 * scale, (i, c)


def test_boosted_frame_laser_antenna_on_gpu(product):
    """tests/decks/boosted_laser_3d.inputs on the HIP path: the pulse arrives with the Lorentz-transformed amplitude and
    wavelength (tests/test_inputs_cpu.py::check_boosted_laser)."""
    from tests.test_inputs_cpu import check_boosted_laser
    sim = WarpXSim.from_inputs(product, os.path.join(HERE, "decks", "boosted_laser_3d.inputs"))
    sim.evolve(sim.max_step)
    check_boosted_laser(sim)
    sim.close()


@H.FIRST_GPU_RUN
@pytest.mark.parametrize("order,filt", [(3, 1), (1, 0)])
def test_ckc_uniform_plasma_parity(oracle, product, order, filt):
    """algo.maxwell_solver = ckc on the HIP path (dt = min(dx)/c, wxa_evolve_b_ckc, every field exchange of the
    reference's schedule) against the oracle stepper: the same 1e-10 gate as the Yee runs."""
    n_cell = (16, 16, 16)
    L = 40e-6
    parts = plasma.uniform_plasma(n_cell, (-L / 2,) * 3, (L / 2,) * 3, (1, 1, 2), 1e25, 0.01, seed=12345)
    species = [(-plasma.Q_E, plasma.M_E, parts)]
    kw = dict(nox=order, use_filter=filt, sort_interval=4, maxwell_solver=_capi.SOLVER_CKC, cfl=0.98)
    so, io = _run(oracle, n_cell, species, 10, **kw)
    sg, ig = _run(product, n_cell, species, 10, **kw)
    assert abs(so.dt - sg.dt) == 0.0
    _compare(_metrics(sg, ig), _metrics(so, io))
    for name in ("Ex", "Ey", "Ez", "Bx", "By", "Bz"):
        a, b = sg.field_valid(name), so.field_valid(name)
        assert np.max(np.abs(a - b)) <= 1e-9 * np.max(np.abs(b)), name


def test_nci_corrector_step_parity(oracle, product):
    """particles.use_fdtd_nci_corr = 1 on the HIP path (E and B filtered along z with the Godfrey stencil before the
    gather, four more guard cells in z: applyNCIFilter, PhysicalParticleContainer.cpp:2097-2172) against the oracle
    stepper: a plasma drifting relativistically along z, CKC + Vay as in the reference's boosted example, order 3,
    bilinear filter, 10 steps with two sorts -- the 1e-10 gate of the other step tests, fields point-wise at 1e-9."""
    n_cell = (16, 16, 32)
    L3 = (8e-6, 8e-6, 16e-6)
    lo, hi = tuple(-v for v in L3), L3
    parts = plasma.uniform_plasma(n_cell, lo, hi, (1, 1, 2), 1e25, 0.02, seed=78)
    parts[6] = parts[6] + 2.0 * plasma.C_LIGHT
    kw = dict(nox=3, galerkin=1, particle_pusher=_capi.PUSHER_VAY, use_filter=1, sort_interval=4,
              maxwell_solver=_capi.SOLVER_CKC, cfl=0.98, use_fdtd_nci_corr=1)
    out = []
    for lib in (oracle, product):
        sim = WarpXSim(lib, n_cell, lo, hi, **kw)
        ids = [sim.add_species(-plasma.Q_E, plasma.M_E, [p.copy() for p in parts])]
        sim.evolve(10)
        out.append((sim, ids))
    (so, io), (sg, ig) = out
    assert tuple(sg.field_view("Ex").ng) == (4, 4, 8)
    _compare(_metrics(sg, ig), _metrics(so, io))
    for name in ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz"):
        a, b = sg.field_valid(name), so.field_valid(name)
        assert np.max(np.abs(a - b)) <= 1e-9 * np.max(np.abs(b)), name


def test_plotfile_from_the_hip_path(product, tmp_path):
    """wxa_sim_write_plotfile on the HIP path: the AMReX plotfile (FlushFormatPlotfile's output) of the Langmuir deck after
    40 steps, parsed back from the files, carries the reference's golden checksums at the reference's tolerance."""
    from tests.test_inputs_cpu import compare_with_golden
    from tests.test_plotfile_cpu import checksum_of, read_plotfile
    gold = json.load(open(os.path.join(HERE, "golden", "langmuir_multi_3d_checksums.json")))
    sim = WarpXSim.from_inputs(product, os.path.join(HERE, "decks", "langmuir_multi_3d.inputs"))
    sim.evolve(sim.max_step)
    plt = str(tmp_path / "plt")
    sim.write_plotfile(plt)
    sim.close()
    got = checksum_of(read_plotfile(plt))
    gold_cs = {g: {k: v for k, v in vals.items() if k != "part_per_cell"} for g, vals in gold["checksums"].items()}
    compare_with_golden(got, gold_cs, gold["rtol"])


def test_reduced_diags_on_the_device(oracle, product, tmp_path):
    """The parity gate's own quantities from the device: FieldEnergy / ParticleEnergy / ParticleMomentum / ParticleNumber
    of the host layer (wxa_reduce_field, wxa_reduce_particles; ReducedDiags.hpp) over a two-species run on the HIP path --
    every row of the files against the rows the oracle stepper writes with its own formulas, and the last row against
    numpy on the arrays copied back (what every other test of this module compares)."""
    n_cell = (32, 24, 24)
    L = 40e-6
    a = plasma.uniform_plasma(n_cell, (-L / 2,) * 3, (L / 2,) * 3, (1, 1, 2), 1e25, 0.02, seed=21)
    b = plasma.uniform_plasma(n_cell, (-L / 2,) * 3, (L / 2,) * 3, (2, 1, 1), 1e25, 0.001, seed=22)
    b[6] = b[6] + 0.02 * plasma.C_LIGHT
    species = [(-plasma.Q_E, plasma.M_E, a), (plasma.Q_E, 1836.0 * plasma.M_E, b)]
    rows = {}
    for lib, sub in ((product, "hip"), (oracle, "oracle")):
        path = str(tmp_path / sub) + "/"
        os.makedirs(path, exist_ok=True)
        sim = WarpXSim(lib, n_cell, (-L / 2,) * 3, (L / 2,) * 3, nox=3, use_filter=1, sort_interval=3)
        ids = [sim.add_species(q, m, parts) for q, m, parts in species]
        for name, kind in (("EF", "FieldEnergy"), ("EP", "ParticleEnergy"), ("PP", "ParticleMomentum"), ("NP", "ParticleNumber")):
            sim.add_reduced_diag(name, kind, "2", path)
        sim.evolve(8)
        rows[sub] = {n: np.atleast_2d(np.genfromtxt(path + n + ".txt")) for n in ("EF", "EP", "PP", "NP")}
        if lib is product:
            ee, eb = field_energy(sim)
            mom = [particle_moments(sim, i) for i in ids]
            assert np.allclose(rows[sub]["EF"][-1, 2:], [ee + eb, ee, eb], rtol=1e-12)
            assert np.allclose(rows[sub]["EP"][-1, 3:5], [m["ekin"] for m in mom], rtol=1e-12)
            assert np.isclose(rows[sub]["PP"][-1, 10], mom[1]["momentum"][2], rtol=1e-12)
            assert list(rows[sub]["NP"][-1, 3:5]) == [a[0].size, b[0].size]
        sim.close()
    for n in ("EF", "EP", "PP", "NP"):
        x, y = rows["hip"][n], rows["oracle"][n]
        assert x.shape == y.shape and np.array_equal(x[:, :2], y[:, :2]), n
        assert list(x[:, 0]) == [0, 2, 4, 6, 8]
        scale = np.max(np.abs(y[:, 2:]), axis=0)
        if n == "PP":   # thermal sums of +- terms: against the largest component of their group (sums | means)
            for g in (slice(0, 9), slice(9, 18)):
                scale[g] = np.maximum(scale[g], 1e-4 * np.max(scale[g]))
        worst = np.max(np.abs(x[:, 2:] - y[:, 2:]) / np.maximum(scale, 1e-300))
        print(f"{n}: worst relative deviation {worst:.2e}")
        assert worst <= RTOL, n


def test_full_diagnostics_of_a_deck_on_the_hip_path(product, tmp_path):
    """diagnostics.diags_names + warpx.reduced_diags_names of a deck on the HIP path: the step loop leaves the reference's
    output tree (plotfiles at the intervals and at max_step, diags/reducedfiles/*.txt); the last plotfile, read back,
    carries the library's own checksum of the final state and the last FieldEnergy row its field energy."""
    from tests.test_plotfile_cpu import checksum_of, read_plotfile
    deck = os.path.join(HERE, "decks", "langmuir_multi_3d.inputs")
    prefix = str(tmp_path / "diags" / "diag1")
    over = ["warpx_amd.write_diagnostics=1", "my_constants.nx=32", "max_step=9", "algo.particle_shape=3", "diag1.intervals=4",
            f"diag1.file_prefix={prefix}", "diag1.fields_to_plot=Ex Ey Ez Bx By Bz jx jy jz rho",
            "warpx.reduced_diags_names=EF EP", "EF.type=FieldEnergy", "EF.intervals=3", f"EF.path={tmp_path}/diags/reducedfiles/",
            "EP.type=ParticleEnergy", "EP.intervals=9", f"EP.path={tmp_path}/diags/reducedfiles/"]
    sim = WarpXSim.from_inputs(product, deck, overrides=over)
    sim.evolve(sim.max_step)
    direct = sim.checksum()
    sim.close()
    assert sorted(os.listdir(tmp_path / "diags")) == ["diag1000000", "diag1000004", "diag1000008", "diag1000009", "reducedfiles"]
    pf = read_plotfile(prefix + "000009")
    assert pf["step"] == 9 and sorted(pf["species"]) == ["electrons", "positrons"]
    got = checksum_of(pf)
    for group, vals in got.items():
        for k, v in vals.items():
            # rho is deposited with atomics on the GPU: two evaluations differ in the last digits
            assert abs(v - direct[group][k]) <= (1e-10 if k == "rho" else 1e-12) * max(abs(direct[group][k]), 1e-300), (group, k)
    ef = np.atleast_2d(np.genfromtxt(str(tmp_path / "diags" / "reducedfiles" / "EF.txt")))
    ep = np.atleast_2d(np.genfromtxt(str(tmp_path / "diags" / "reducedfiles" / "EP.txt")))
    assert list(ef[:, 0]) == [0, 3, 6, 9] and list(ep[:, 0]) == [0, 9]
    # energy of the cell-centred fields of the plotfile against the row of the staggered ones: the same to the accuracy of
    # the averaging (the reference's own analysis_reduced_diags_impl.py compares the two at 1e-3 .. 1e-2)
    dV = (40e-6 / 32) ** 3
    e2 = sum(np.sum(pf["fields"][n] ** 2) for n in ("Ex", "Ey", "Ez"))
    assert abs(0.5 * plasma.EP0 * e2 * dV - ef[-1, 3]) < 0.1 * ef[-1, 3]
    assert ep[-1, 2] > 0 and ep[0, 2] > ep[-1, 2]      # the oscillation has handed energy to the field
