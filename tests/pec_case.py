"""Examples/Tests/pec/inputs_test_3d_pec_field of the reference, restated for WarpXSim: a pulse of Ey/Bx
between two PEC walls in z (periodic in x, y), no particles; after 125 steps the reflected pulse
interferes constructively with itself (standing wave of twice the amplitude, Ey = 0 on the walls)."""
import numpy as np

from warpx_amd import _capi
from warpx_amd.sim import WarpXSim

C_LIGHT = 299792458.0
N_CELL = (32, 32, 256)
PROB_LO, PROB_HI = (-8e-6, -8e-6, -4e-6), (8e-6, 8e-6, 4e-6)
Z1, Z2, WAVELENGTH, EY_IN = -2e-6, 2e-6, 1e-6, 1e5
MAX_STEP = 125


def make_sim(lib):
    sim = WarpXSim(lib, N_CELL, PROB_LO, PROB_HI, nox=1, galerkin=1, use_filter=0, cfl=0.9,
                   field_boundary_lo=(_capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PEC),
                   field_boundary_hi=(_capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PEC))
    dz = (PROB_HI[2] - PROB_LO[2]) / N_CELL[2]

    def z_of(name):
        v = sim.field_view(name)
        k = np.arange(v.n[2]) + v.lo[2]
        # nodal direction: z = lo + k dz; cell-centred: z = lo + (k + 1/2) dz
        return PROB_LO[2] + (k + (0.0 if v.stag[2] else 0.5)) * dz, tuple(v.n)

    # warpx.Ey_external_grid_function = 1e5 sin(2 pi z / wavelength) (z < z2) (z > z1)
    z, n = z_of("Ey")
    prof = EY_IN * np.sin(2 * np.pi * z / WAVELENGTH) * (z < Z2) * (z > Z1)
    sim.set_field("Ey", np.broadcast_to(prof[None, None, :], n).copy())
    # warpx.Bx_external_grid_function = -1e5 sin(2 pi z / wavelength) / clight (z < z2) (z > z1)
    z, n = z_of("Bx")
    prof = -EY_IN * np.sin(2 * np.pi * z / WAVELENGTH) / C_LIGHT * (z < Z2) * (z > Z1)
    sim.set_field("Bx", np.broadcast_to(prof[None, None, :], n).copy())
    return sim


# ---- Examples/Tests/pec/inputs_test_3d_pec_particle ---------------------------------------------
# Two unit-weight particles 0.004 cell from a PEC wall in x (periodic in y, z), order 3, Vay pusher,
# filter on: an "electron" at rest (with the proton mass) and a proton moving along the wall with
# u_y = -2 c.  Their order-3 stencils reach behind the wall, so every step exercises the image-charge
# fold of J (ApplyReflectiveBoundarytoJfield) and the mirrored E/B guard cells in the gather.
P_N_CELL = (128, 64, 64)
P_PROB_LO, P_PROB_HI = (-32e-6,) * 3, (32e-6,) * 3
P_MAX_STEP = 20
Q_E, M_P = 1.602176634e-19, 1.67262192369e-27


def make_particle_sim(lib):
    sim = WarpXSim(lib, P_N_CELL, P_PROB_LO, P_PROB_HI, nox=3, galerkin=1, particle_pusher=_capi.PUSHER_VAY,
                   current_deposition=_capi.DEPOSIT_ESIRKEPOV, use_filter=1, cfl=0.9, sort_interval=4,
                   field_boundary_lo=(_capi.BOUNDARY_PEC, _capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PERIODIC),
                   field_boundary_hi=(_capi.BOUNDARY_PEC, _capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PERIODIC))
    one = lambda v: np.array([v], dtype=np.float64)
    pos = [one(31.998e-6), one(0.0), one(0.0)]
    electron = sim.add_species(-Q_E, M_P, pos + [one(1.0), one(0.0), one(0.0), one(0.0)])
    proton = sim.add_species(+Q_E, M_P, pos + [one(1.0), one(0.0), one(-2.0 * C_LIGHT), one(0.0)])
    return sim, electron, proton


# ---- Examples/Tests/boundaries/inputs_test_3d_particle_boundaries -----------------------------------
# Chargeless particles flying into reflecting (x), absorbing (y) and periodic (z) walls: pure kinematics of
# WarpXParticleContainer::ApplyBoundaryConditions + the periodic wrap.
B_N_CELL = (16, 16, 16)
B_PROB_LO, B_PROB_HI = (-1.0,) * 3, (1.0,) * 3
B_MAX_STEP = 8
M_E = 9.1093837015e-31


def make_boundaries_sim(lib):
    P, A, R = _capi.PBOUNDARY_PERIODIC, _capi.PBOUNDARY_ABSORBING, _capi.PBOUNDARY_REFLECTING
    sim = WarpXSim(lib, B_N_CELL, B_PROB_LO, B_PROB_HI, nox=1, galerkin=1, use_filter=1, cfl=1.0, sort_interval=4,
                   field_boundary_lo=(_capi.BOUNDARY_PEC, _capi.BOUNDARY_PEC, _capi.BOUNDARY_PERIODIC),
                   field_boundary_hi=(_capi.BOUNDARY_PEC, _capi.BOUNDARY_PEC, _capi.BOUNDARY_PERIODIC),
                   particle_boundary_lo=(R, A, P), particle_boundary_hi=(R, A, P))

    def species(pos, u):
        n = len(pos)
        cols = [np.array([p[d] for p in pos], dtype=np.float64) for d in range(3)]
        cols += [np.ones(n)]
        cols += [C_LIGHT * np.array([v[d] for v in u], dtype=np.float64) for d in range(3)]
        return sim.add_species(0.0, M_E, cols)

    refl = species([(-0.9, 0, 0), (0.91, 0, 0)], [(-0.9, 0, 0), (0.91, 0, 0)])
    absb = species([(0, -0.92, 0), (0, 0.93, 0), (0, 0, 0)], [(0, -0.92, 0), (0, 0.93, 0), (0, 0, 0)])
    peri = species([(0, 0, -0.94), (0, 0, 0.95)], [(0, 0, -0.94), (0, 0, 0.95)])
    return sim, refl, absb, peri


# ---- Examples/Physics_applications/laser_acceleration/inputs_test_3d_laser_acceleration ----------------
# Laser-wakefield stage in a window moving at c: PEC walls in z, Gaussian laser antenna, electrons at rest
# injected continuously as the window advances (order 3, filter, cfl 1, 100 steps).
L_N_CELL = (32, 32, 256)
L_PROB_LO, L_PROB_HI = (-30e-6, -30e-6, -56e-6), (30e-6, 30e-6, 12e-6)
L_MAX_STEP = 100
M_E_ = 9.1093837015e-31


def make_lwfa_sim(lib):
    """inputs_test_3d_laser_acceleration: moving window at c along z, electrons injected continuously, Gaussian antenna."""
    import ctypes as C
    sim = WarpXSim(lib, L_N_CELL, L_PROB_LO, L_PROB_HI, nox=3, galerkin=1, use_filter=1, cfl=1.0, sort_interval=4,
                   field_boundary_lo=(_capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PEC),
                   field_boundary_hi=(_capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PEC))
    mw = _capi.MovingWindow(dir=2, v=1.0)
    lib.sim_set_moving_window(sim._h, C.byref(mw))
    empty = [np.zeros(0) for _ in range(7)]
    electrons = sim.add_species(-Q_E, M_E_, empty)
    inj = _capi.PlasmaInjector()
    inj.density = 2e23
    for d in range(3):
        inj.ppc[d] = 1
    big = 1e300
    inj.lo[0], inj.hi[0] = -20e-6, 20e-6
    inj.lo[1], inj.hi[1] = -20e-6, 20e-6
    inj.lo[2], inj.hi[2] = 0.0, big
    lib.sim_set_injection(sim._h, electrons, C.byref(inj), 1, 1)
    la = _capi.LaserAntenna()
    for d, v in enumerate((0.0, 0.0, 9e-6)):
        la.position[d] = v
    for d, v in enumerate((0.0, 0.0, 1.0)):
        la.direction[d] = v
    for d, v in enumerate((0.0, 1.0, 0.0)):
        la.polarization[d] = v
    la.e_max, la.wavelength = 16e12, 0.8e-6
    la.waist, la.duration, la.t_peak, la.focal_distance = 5e-6, 15e-15, 30e-15, 100e-6
    lib.sim_add_laser(sim._h, C.byref(la))
    return sim, electrons


# ---- BASELINE config 5 in small, by hand: tests/decks/laser_wakefield_boosted_3d.inputs for the oracle stepper ----------
BOOST_GAMMA = 5.0
BOOST_MAX_STEP = 120


def make_boosted_lwfa_sim(lib):
    """The boosted laser-wakefield deck (gamma = 5, window at c, PEC z, CKC, Vay, order 3, bilinear filter, NCI
    corrector, Gaussian antenna, electrons injected continuously) set up call by call -- for the oracle stepper, which
    has no inputs reader.  The z bounds go through ConvertLabParamsToBoost (Source/Utils/WarpXUtil.cpp:180-262):
    divided by gamma (1 - beta beta_window), beta_window = moving_window_v."""
    import ctypes as C
    beta = np.sqrt(1.0 - 1.0 / BOOST_GAMMA ** 2)
    convert_factor = 1.0 / (BOOST_GAMMA * (1.0 - beta * 1.0))
    lo = (-30e-6, -30e-6, -14e-6 * convert_factor)
    hi = (30e-6, 30e-6, 2e-6 * convert_factor)
    sim = WarpXSim(lib, (16, 16, 256), lo, hi, nox=3, galerkin=1, particle_pusher=_capi.PUSHER_VAY, use_filter=1, cfl=1.0,
                   sort_interval=4, maxwell_solver=_capi.SOLVER_CKC, use_fdtd_nci_corr=1,
                   field_boundary_lo=(_capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PEC),
                   field_boundary_hi=(_capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PERIODIC, _capi.BOUNDARY_PEC),
                   gamma_boost=BOOST_GAMMA)
    mw = _capi.MovingWindow(dir=2, v=1.0)
    lib.sim_set_moving_window(sim._h, C.byref(mw))
    electrons = sim.add_species(-Q_E, M_E_, [np.zeros(0) for _ in range(7)])
    inj = _capi.PlasmaInjector()
    inj.density = 2e23
    for d in range(3):
        inj.ppc[d] = 1
    inj.lo[0], inj.hi[0] = -20e-6, 20e-6
    inj.lo[1], inj.hi[1] = -20e-6, 20e-6
    inj.lo[2], inj.hi[2] = 0.0, 1e300
    lib.sim_set_injection(sim._h, electrons, C.byref(inj), 1, 1)
    la = _capi.LaserAntenna()
    for d, v in enumerate((0.0, 0.0, -1e-6)):
        la.position[d] = v
    for d, v in enumerate((0.0, 0.0, 1.0)):
        la.direction[d] = v
    for d, v in enumerate((0.0, 1.0, 0.0)):
        la.polarization[d] = v
    la.e_max, la.wavelength = 16e12, 0.8e-6
    la.waist, la.duration, la.t_peak, la.focal_distance = 12e-6, 15e-15, 30e-15, 100e-6
    lib.sim_add_laser(sim._h, C.byref(la))
    return sim, electrons
