"""Step-level parity: the host layer's WarpX::Evolve on the HIP kernels against the
independent CPU oracle stepper, same seeded inputs.  Gate (BASELINE.json north_star):
field energies and particle moments within 1e-10 relative, fp64."""
import json
import os

import numpy as np
import pytest

from warpx_amd import _capi, plasma
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


def test_langmuir_golden_on_gpu(oracle, product):
    """The reference's own golden checksums (64^3 Langmuir, 40 steps) reproduced by the HIP path."""
    import ctypes as C
    from warpx_amd.containers import FieldArray
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
