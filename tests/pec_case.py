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
