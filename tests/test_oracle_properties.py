"""Analytic properties of the path checked on the CPU oracle (no GPU): these pin what the
reference's end-to-end checksums do not ('not pinned by any reference test')."""
import ctypes as C

import numpy as np
import pytest

from tests import helpers as H
from warpx_amd import _capi, plasma
from warpx_amd.containers import STAG, FieldArray, ParticleArrays, field_triplet

NCELL = (12, 10, 8)


@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_esirkepov_continuity(oracle, order):
    assert H.continuity_residual(oracle, "cpu", order, NCELL) < 1e-11


def test_div_b_preserved(oracle):
    """EvolveB keeps the discrete div B at round-off (curl of a gradient-free update)."""
    ng = 2
    E = H.random_fields(("Ex", "Ey", "Ez"), NCELL, ng, 1, scale=1e9)
    B = [FieldArray(NCELL, STAG[n], (ng,) * 3) for n in ("Bx", "By", "Bz")]
    per = H.i3((1, 1, 1))
    for f in E:
        oracle.sync_nodal_periodic(C.byref(f.view), per, None)
        oracle.fill_boundary_periodic(C.byref(f.view), H.i3((ng,) * 3), per, None)
    _, dx = H.geom_for(NCELL, ng)
    dt = H.yee_dt(dx)
    oracle.evolve_b(field_triplet(E), field_triplet(B), dt, H.d3(1.0 / dx), None)
    bx, by, bz = (f.valid() for f in B)
    div = ((bx[1:, :, :] - bx[:-1, :, :]) / dx[0] + (by[:, 1:, :] - by[:, :-1, :]) / dx[1]
           + (bz[:, :, 1:] - bz[:, :, :-1]) / dx[2])
    scale = np.max(np.abs(bx)) / dx[0]
    assert np.max(np.abs(div)) < 1e-12 * scale


@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_gather_of_linear_field_is_exact(oracle, order):
    """B-spline interpolation reproduces a linear field; galerkin off so every component
    uses the full order along every direction."""
    ng = order + 1
    g, dx = H.geom_for(NCELL, ng)
    coef = np.array([3.0, -2.0, 5.0])
    names = ("Ex", "Ey", "Ez", "Bx", "By", "Bz")
    fields = []
    for n in names:
        f = FieldArray(NCELL, STAG[n], (ng,) * 3)
        idx = [(-H.LX / 2 + (np.arange(f.n[d]) + f.lo[d] + (0.0 if f.stag[d] else 0.5)) * dx[d]) / H.LX
               for d in range(3)]
        X, Y, Z = np.meshgrid(*idx, indexing="ij")
        f.from_numpy(1.0 + coef[0] * X + coef[1] * Y + coef[2] * Z)
        fields.append(f)
    parts = H.random_particles(500, NCELL, 5, u_scale=0.0, margin=1.0)
    parts[4][:] = 0; parts[5][:] = 0; parts[6][:] = 0
    p = ParticleArrays.from_numpy(parts)
    # with u = 0 and B irrelevant to the E kick: u_new = q dt/m * E(x_p) for Boris
    dt = H.yee_dt(dx)
    q, m = plasma.Q_E, plasma.M_E
    oracle.push_p(C.byref(p.view), field_triplet(fields[:3]), field_triplet(fields[3:]), C.byref(g), q, m, dt,
                  order, 0, _capi.PUSHER_BORIS, None)
    a = p.to_numpy()
    want = 1.0 + (coef[0] * a[0] + coef[1] * a[1] + coef[2] * a[2]) / H.LX
    # small B rotation: |t| = q dt B / 2m ~ 1e-4 -> compare the dominant E kick to 1e-3
    for c in range(3):
        got = a[4 + c] / (q * dt / m)
        assert np.max(np.abs(got - want)) < 2e-3 * np.max(np.abs(want))


def test_direct_vs_esirkepov_total_current(oracle):
    """Both schemes deposit the same total current sum_cells J dV = sum_p q w v (order 2)."""
    order = 2
    _, ng_depos, ng_j = H.guard_depths(order)
    g, dx = H.geom_for(NCELL, ng_depos)
    dt = H.yee_dt(dx)
    parts = H.random_particles(2000, NCELL, 7, u_scale=0.5, margin=1.0)
    p = ParticleArrays.from_numpy(parts)
    tot = {}
    for algo in (_capi.DEPOSIT_ESIRKEPOV, _capi.DEPOSIT_DIRECT):
        J = [FieldArray(NCELL, STAG[n], (ng_j,) * 3) for n in ("jx", "jy", "jz")]
        oracle.deposit_current(C.byref(p.view), field_triplet(J), C.byref(g), plasma.Q_E, dt, -0.5 * dt, order,
                               algo, None, None)
        tot[algo] = np.array([f.to_numpy().sum() for f in J]) * dx[0] * dx[1] * dx[2]
    a = np.array(parts)
    gam = np.sqrt(1 + (a[4] ** 2 + a[5] ** 2 + a[6] ** 2) / plasma.C_LIGHT ** 2)
    want = np.array([np.sum(plasma.Q_E * a[3] * a[4 + c] / gam) for c in range(3)])
    for algo in tot:
        assert np.allclose(tot[algo], want, rtol=1e-10)


def test_vay_force_free_orbit(oracle):
    """Examples/Tests/particle_pusher: a positron in E = -v x B keeps its velocity with the Vay
    pusher (|x_perp| stays ~1e-4 after 10^4 steps, analysis.py:18-22) but not with Boris."""
    ncell = (4, 4, 4)
    ng = 2
    g, dx = H.geom_for(ncell, ng)
    names = ("Ex", "Ey", "Ez", "Bx", "By", "Bz")
    gamma = 20.0
    v = plasma.C_LIGHT * np.sqrt(1 - 1 / gamma ** 2)
    Bz = 1.0
    vals = {"Ex": 0.0, "Ey": v * Bz, "Ez": 0.0, "Bx": 0.0, "By": 0.0, "Bz": Bz}  # E = -v x B, v along x
    fields = []
    for n in names:
        f = FieldArray(ncell, STAG[n], (ng,) * 3)
        f.from_numpy(np.full(f.n, vals[n]))
        fields.append(f)
    dt = 1e-13
    out = {}
    for pusher in (_capi.PUSHER_VAY, _capi.PUSHER_BORIS):
        p = ParticleArrays.from_numpy([np.zeros(1), np.zeros(1), np.zeros(1), np.ones(1),
                                       np.array([gamma * v]), np.zeros(1), np.zeros(1)])
        drift = 0.0
        for _ in range(2000):
            oracle.push_p(C.byref(p.view), field_triplet(fields[:3]), field_triplet(fields[3:]), C.byref(g),
                          plasma.Q_E, plasma.M_E, dt, 1, 1, pusher, None)
            a = p.to_numpy()
            gam = np.sqrt(1 + (a[4] ** 2 + a[5] ** 2 + a[6] ** 2) / plasma.C_LIGHT ** 2)
            drift += float(a[5, 0] / gam[0]) * dt
        out[pusher] = abs(drift)
    assert out[_capi.PUSHER_VAY] < 1e-9
    assert out[_capi.PUSHER_BORIS] > 1e3 * max(out[_capi.PUSHER_VAY], 1e-15)


# ---- algo.maxwell_solver = ckc (no deterministic 3-D golden file of the reference uses it) ----

def test_ckc_coefficients_cubic_cells(oracle):
    """Cowan et al., PRST-AB 16, 041303 (2013): cubic cells give alpha = 7/12, beta = 1/12, gamma = 1/48 (in units of
    1/dx), and alpha + 4 beta + 4 gamma = 1 (a field uniform in the transverse plane sees the plain difference)."""
    dx = 0.3e-6
    cx, cy, cz = ((C.c_double * 5)() for _ in range(3))
    oracle.ckc_stencil_coefficients(H.d3((dx, dx, dx)), cx, cy, cz)
    for c in (cx, cy, cz):
        assert np.allclose(np.array(c[:]) * dx, [1.0, 7.0 / 12.0, 1.0 / 12.0, 1.0 / 12.0, 1.0 / 48.0], rtol=1e-15)
    assert oracle.ckc_max_dt(H.d3((dx, 2 * dx, 3 * dx))) == dx / plasma.C_LIGHT
    # anisotropic cells: the transverse sums still add up to 1/dx_d
    oracle.ckc_stencil_coefficients(H.d3((dx, 1.5 * dx, 0.8 * dx)), cx, cy, cz)
    for c in (cx, cy, cz):
        assert np.isclose(c[1] + 2 * c[2] + 2 * c[3] + 4 * c[4], c[0], rtol=1e-14)


@pytest.mark.parametrize("axis", [0, 2])
def test_ckc_vacuum_pulse_travels_one_cell_per_step(oracle, axis):
    """The reason the solver exists: at cfl = 1 (c dt = dx) a plane wave along a grid axis has no numerical
    dispersion.  A pulse started on the discrete mode of the leapfrog (B = the average of its two neighbours of E / c)
    is translated by exactly one cell per step; the Yee solver on the same grid (c dt = dx / sqrt 3) disperses it."""
    from warpx_amd.sim import WarpXSim
    n = 24
    n_cell = [4, 4, 4]
    n_cell[axis] = n
    Lbox = [4 * 1e-6, 4 * 1e-6, 4 * 1e-6]
    Lbox[axis] = n * 1e-6
    lo, hi = tuple(-0.5 * v for v in Lbox), tuple(0.5 * v for v in Lbox)
    k = np.arange(n)
    f = 1e9 * np.exp(-((k - 7.0) / 2.5) ** 2) * np.cos(2 * np.pi * k / 6.0)          # E at the nodes along the axis
    steps = 9
    res = {}
    for solver in (_capi.SOLVER_CKC, _capi.SOLVER_YEE):
        sim = WarpXSim(oracle, n_cell, lo, hi, nox=1, use_filter=0, maxwell_solver=solver)
        # wave along +axis: (E, B) = (Ey, -Bx) along z, (Ez, -By) along x  [E x B parallel to the axis]
        en, bn = ("Ey", "Bx") if axis == 2 else ("Ez", "By")
        sign = -1.0
        E, B = sim.field(en), sim.field(bn)
        ge, gb = sim.field_view(en).ng[axis], sim.field_view(bn).ng[axis]
        idx_e = (np.arange(E.shape[axis]) - ge) % n          # node index of every point along the axis (periodic)
        idx_b = (np.arange(B.shape[axis]) - gb) % n          # cell index
        shape = [1, 1, 1]
        shape[axis] = -1
        E[...] = f[idx_e].reshape(shape)
        B[...] = sign * (0.5 * (f[idx_b] + f[(idx_b + 1) % n]) / plasma.C_LIGHT).reshape(shape)
        sim.set_field(en, E)
        sim.set_field(bn, B)
        sim.evolve(steps)
        out = sim.field_valid(en)
        line = np.moveaxis(out, axis, 0)[:n, 1, 1]
        res[solver] = np.max(np.abs(line - np.roll(f, steps))) / np.max(np.abs(f))
        if solver == _capi.SOLVER_CKC:
            assert np.isclose(sim.dt, 1e-6 / plasma.C_LIGHT, rtol=1e-15)
        sim.close()
    assert res[_capi.SOLVER_CKC] < 1e-12, res
    assert res[_capi.SOLVER_YEE] > 1e-2, res          # same grid, Yee: the pulse lags and disperses
