"""The product's C++17 host layer (class WarpX schedule, MultiFabRegister, containers,
BrickComm) executed on CPU against the oracle kernels (tests/host_cpu) and compared with the
independent oracle stepper: pins the host logic without a GPU."""
import numpy as np
import pytest

from tests.oracle_lib import load_host_cpu
from warpx_amd import _capi, plasma
from warpx_amd.sim import WarpXSim, field_energy, particle_moments

L = 40e-6


@pytest.fixture(scope="module")
def host_cpu():
    return load_host_cpu()


def _species(n_cell, ppc=(1, 1, 2), seed=3):
    return plasma.uniform_plasma(n_cell, (-L / 2,) * 3, (L / 2,) * 3, ppc, 1e25, 0.05, seed=seed)


@pytest.mark.parametrize("order,filt,sort,pusher,depos", [
    (1, 1, -1, _capi.PUSHER_BORIS, _capi.DEPOSIT_ESIRKEPOV), (3, 1, 2, _capi.PUSHER_BORIS, _capi.DEPOSIT_ESIRKEPOV),
    (2, 0, 1, _capi.PUSHER_BORIS, _capi.DEPOSIT_ESIRKEPOV), (3, 1, 3, _capi.PUSHER_VAY, _capi.DEPOSIT_DIRECT),
    (1, 0, 4, _capi.PUSHER_VAY, _capi.DEPOSIT_ESIRKEPOV), (2, 1, -1, _capi.PUSHER_BORIS, _capi.DEPOSIT_DIRECT),
    (4, 1, 3, _capi.PUSHER_BORIS, _capi.DEPOSIT_ESIRKEPOV)])
def test_single_brick_schedule_matches_oracle(oracle, host_cpu, order, filt, sort, pusher, depos):
    """The product's schedule (no exchange inside the field solve: guard layer of B computed, solver-depth fill of E
    dropped) against the oracle stepper, which keeps the reference's five fills."""
    n_cell = (16, 12, 12) if order < 4 else (16, 16, 16)   # a periodic direction must hold twice the guard depth
    parts = _species(n_cell)
    res = []
    for lib in (host_cpu, oracle):
        sim = WarpXSim(lib, n_cell, (-L / 2,) * 3, (L / 2,) * 3, nox=order, use_filter=filt, sort_interval=sort,
                       particle_pusher=pusher, current_deposition=depos)
        sid = sim.add_species(-plasma.Q_E, plasma.M_E, parts)
        sim.evolve(3)
        sim.evolve(2)   # a second Evolve call: de-synchronise again, same schedule as the reference
        sim.compute_rho()
        res.append((field_energy(sim), particle_moments(sim, sid),
                    {n: sim.field_valid(n) for n in ("Ex", "By", "jz", "rho")}))
        sim.close()
    (fa, ma, Fa), (fb, mb, Fb) = res
    assert np.allclose(fa, fb, rtol=1e-11)
    assert np.isclose(ma["ekin"], mb["ekin"], rtol=1e-12)
    assert np.allclose(ma["abs_momentum"], mb["abs_momentum"], rtol=1e-12)
    for n in Fa:
        assert np.max(np.abs(Fa[n] - Fb[n])) <= 1e-11 * np.max(np.abs(Fb[n])), n


@pytest.mark.parametrize("order,filt", [(1, 0), (3, 1)])
def test_ckc_schedule_matches_oracle(oracle, host_cpu, order, filt):
    """algo.maxwell_solver = ckc: dt = min(dx)/c, the extended update of B, and the solver-depth fill of E that the
    Yee schedule drops (the CKC update of B reads guard points of E) -- against the oracle stepper."""
    n_cell = (16, 12, 14)
    parts = _species(n_cell)
    res = []
    for lib in (host_cpu, oracle):
        sim = WarpXSim(lib, n_cell, (-L / 2,) * 3, (L / 2, L / 2 * 1.1, L / 2 * 0.9), nox=order, use_filter=filt,
                       sort_interval=2, maxwell_solver=_capi.SOLVER_CKC, cfl=0.95)
        sid = sim.add_species(-plasma.Q_E, plasma.M_E, parts)
        sim.evolve(5)
        res.append((sim.dt, field_energy(sim), particle_moments(sim, sid),
                    {n: sim.field_valid(n) for n in ("Ex", "By", "Bz", "jz")}))
        sim.close()
    (da, fa, ma, Fa), (db, fb, mb, Fb) = res
    assert da == db
    assert np.allclose(fa, fb, rtol=1e-11)
    assert np.isclose(ma["ekin"], mb["ekin"], rtol=1e-12)
    for n in Fa:
        assert np.max(np.abs(Fa[n] - Fb[n])) <= 1e-11 * np.max(np.abs(Fb[n])), n


def test_error_conventions(host_cpu):
    with pytest.raises(_capi.WxaError):   # particle shape out of range
        WarpXSim(host_cpu, (8, 8, 8), (-L / 2,) * 3, (L / 2,) * 3, nox=5)
    with pytest.raises(_capi.WxaError):   # more than one brick needs a comm
        WarpXSim(host_cpu, (8, 8, 16), (-L / 2,) * 3, (L / 2,) * 3, nbricks=(1, 1, 2), coord=(0, 0, 0))
    with pytest.raises(_capi.WxaError):   # cells do not divide into bricks
        WarpXSim(host_cpu, (8, 8, 9), (-L / 2,) * 3, (L / 2,) * 3, nbricks=(1, 1, 2), coord=(0, 0, 1))
