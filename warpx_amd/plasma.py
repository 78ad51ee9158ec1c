"""Synthetic initial conditions for the hot-path benchmarks and parity tests.

The reference's plasma injector is out of scope; what it
produces for the configs of BASELINE.json is restated here:
 - positions: regular lattice, InjectorPositionRegular
   (Source/Initialization/InjectorPosition.H:67-110): particle i_part of a cell with
   ppc = (nx,ny,nz) sits at ((0.5+ix)/nx, (0.5+iy)/ny, (0.5+iz)/nz) of the cell,
   pos = corner + (iv + r)*dx (Source/Particles/PhysicalParticleContainer.cpp:1133);
 - weight = density * dV / ppc (Source/Initialization/PlasmaInjector / AddPlasmaUtilities.H:73-77,
   PhysicalParticleContainer.cpp:1275-1276);
 - u = gamma*beta from the momentum function, then u *= c (:1271-1273).
AMReX's RNG stream cannot be reproduced, so thermal momenta come from a seeded
numpy Generator and are fed identically to the oracle and to the GPU.
"""
from __future__ import annotations

import numpy as np

# Source/ablastr/constant.H:41-50
C_LIGHT = 299792458.0
EP0 = 8.8541878128e-12
MU0 = 1.25663706212e-06
Q_E = 1.602176634e-19
M_E = 9.1093837015e-31
M_P = 1.67262192369e-27


def lattice_positions(n_cell, prob_lo, prob_hi, ppc, box_lo=(0, 0, 0), box_n=None):
    """Regular lattice over the cells [box_lo, box_lo+box_n) of the global grid.
    Order: cell-major (i fastest), then i_part (InjectorPositionRegular's decomposition
    ix = i_part // (ny*nz), iy = (i_part % (ny*nz)) // nz, iz = i_part % nz)."""
    n_cell = np.asarray(n_cell)
    prob_lo = np.asarray(prob_lo, dtype=np.float64)
    prob_hi = np.asarray(prob_hi, dtype=np.float64)
    dx = (prob_hi - prob_lo) / n_cell
    if box_n is None:
        box_n = n_cell
    nx, ny, nz = (int(v) for v in ppc)
    nppc = nx * ny * nz
    ip = np.arange(nppc)
    rx = (0.5 + ip // (ny * nz)) / nx
    ry = (0.5 + (ip % (ny * nz)) // nz) / ny
    rz = (0.5 + ip % nz) / nz
    ci = np.arange(box_lo[0], box_lo[0] + box_n[0], dtype=np.float64)
    cj = np.arange(box_lo[1], box_lo[1] + box_n[1], dtype=np.float64)
    ck = np.arange(box_lo[2], box_lo[2] + box_n[2], dtype=np.float64)
    K, J, I = np.meshgrid(ck, cj, ci, indexing="ij")   # i fastest when flattened
    I = I.reshape(-1, 1)
    J = J.reshape(-1, 1)
    K = K.reshape(-1, 1)
    x = prob_lo[0] + (I + rx[None, :]) * dx[0]
    y = prob_lo[1] + (J + ry[None, :]) * dx[1]
    z = prob_lo[2] + (K + rz[None, :]) * dx[2]
    return x.reshape(-1), y.reshape(-1), z.reshape(-1), dx, nppc


def uniform_plasma(n_cell, prob_lo, prob_hi, ppc, density, u_th, seed=12345,
                   box_lo=(0, 0, 0), box_n=None):
    """Examples/Physics_applications/uniform_plasma/inputs_base_3d: constant density,
    gaussian momentum with ux_th = uy_th = uz_th = u_th (in units of c)."""
    x, y, z, dx, nppc = lattice_positions(n_cell, prob_lo, prob_hi, ppc, box_lo, box_n)
    n = x.shape[0]
    w = np.full(n, density * dx[0] * dx[1] * dx[2] / nppc)
    rng = np.random.Generator(np.random.PCG64(seed))
    u = rng.standard_normal((3, n)) * u_th * C_LIGHT
    return [x, y, z, w, u[0], u[1], u[2]]


def langmuir_3d(n_cell=(64, 64, 64), lx=40.e-6, n0=2.e24, epsilon=0.01, ppc=(1, 1, 1), sign=+1.0):
    """Examples/Tests/langmuir/inputs_base_3d:2-12,52-89: momentum
    sign * epsilon * k/kp * sin(kx) cos(ky) cos(kz) (and cyclic), kp from 2*n0."""
    prob_lo = (-lx / 2.0,) * 3
    prob_hi = (lx / 2.0,) * 3
    wp = np.sqrt(2.0 * n0 * Q_E ** 2 / (EP0 * M_E))
    kp = wp / C_LIGHT
    k = 2.0 * 2.0 * np.pi / lx
    x, y, z, dx, nppc = lattice_positions(n_cell, prob_lo, prob_hi, ppc)
    a = sign * epsilon * k / kp
    ux = a * np.sin(k * x) * np.cos(k * y) * np.cos(k * z)
    uy = a * np.cos(k * x) * np.sin(k * y) * np.cos(k * z)
    uz = a * np.cos(k * x) * np.cos(k * y) * np.sin(k * z)
    w = np.full(x.shape[0], n0 * dx[0] * dx[1] * dx[2] / nppc)
    return [x, y, z, w, ux * C_LIGHT, uy * C_LIGHT, uz * C_LIGHT], prob_lo, prob_hi


def langmuir_analytic_E(n_cell, lx, n0, epsilon, t):
    """Analytic E at cell centres, Examples/Tests/langmuir/analysis_3d.py:46-94:
    amplitude = epsilon*(m_e c^2 k)/e * sin(wp t) with wp from the total density 2*n0."""
    wp = np.sqrt(2.0 * n0 * Q_E ** 2 / (EP0 * M_E))
    k = 2.0 * 2.0 * np.pi / lx
    dx = lx / np.asarray(n_cell)
    xs = [-lx / 2 + (np.arange(n_cell[d]) + 0.5) * dx[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*xs, indexing="ij")
    E0 = epsilon * M_E * C_LIGHT ** 2 * k / Q_E * np.sin(wp * t)
    Ex = E0 * np.sin(k * X) * np.cos(k * Y) * np.cos(k * Z)
    Ey = E0 * np.cos(k * X) * np.sin(k * Y) * np.cos(k * Z)
    Ez = E0 * np.cos(k * X) * np.cos(k * Y) * np.sin(k * Z)
    return Ex, Ey, Ez


def half_domain_beam(n_cell=(64, 64, 64), lx=40.e-6, density=1.e25, ppc=(2, 2, 2), u_x=0.1):
    """Examples/Tests/langmuir/inputs_test_3d_langmuir_multi_picmi.py: cold electrons filling x < 0
    (UniformDistribution with upper_bound = [0, None, None]) with u = (u_x c, 0, 0), GriddedLayout."""
    prob_lo = (-lx / 2.0,) * 3
    prob_hi = (lx / 2.0,) * 3
    x, y, z, dx, nppc = lattice_positions(n_cell, prob_lo, prob_hi, ppc)
    keep = x < 0.0
    x, y, z = x[keep], y[keep], z[keep]
    n = x.shape[0]
    w = np.full(n, density * dx[0] * dx[1] * dx[2] / nppc)
    return [x, y, z, w, np.full(n, u_x * C_LIGHT), np.zeros(n), np.zeros(n)], prob_lo, prob_hi
