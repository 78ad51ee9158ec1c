"""Python handle on the step-level C API (`<prefix>sim_*`, include/warpx_amd.h):
the host layer's re-statement of WarpX::Evolve / OneStep_nosub
(Source/Evolve/WarpXEvolve.cpp:94-347,354-455).

`WarpXSim(lib, ...)` works with any library that exports the step-level symbols; the
product passes `load_product()` (HIP, device pointers).  The test-suite also binds the
CPU oracle through the same class -- the product never does.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from .containers import ParticleArrays, particles_to_numpy, view_to_numpy


class WarpXSim:
    def __init__(self, lib: _capi.CLib, n_cell, prob_lo, prob_hi, nox=1, galerkin=None,
                 particle_pusher=_capi.PUSHER_BORIS, current_deposition=_capi.DEPOSIT_ESIRKEPOV,
                 use_filter=0, cfl=1.0, sort_interval=-1, nbricks=(1, 1, 1), coord=(0, 0, 0),
                 comm: _capi.Comm | None = None, field_boundary_lo=(0, 0, 0), field_boundary_hi=(0, 0, 0),
                 particle_boundary_lo=(0, 0, 0), particle_boundary_hi=(0, 0, 0),
                 grid_type=_capi.GRID_STAGGERED, overlap_halo=0, maxwell_solver=_capi.SOLVER_YEE,
                 use_fdtd_nci_corr=0, gamma_boost=0.0):
        self.lib = lib
        self.on_device = lib.prefix == "wxa_"
        cfg = _capi.SimConfig()
        for d in range(3):
            cfg.n_cell[d] = int(n_cell[d])
            cfg.prob_lo[d] = float(prob_lo[d])
            cfg.prob_hi[d] = float(prob_hi[d])
            cfg.nbricks[d] = int(nbricks[d])
            cfg.coord[d] = int(coord[d])
            cfg.field_boundary_lo[d] = int(field_boundary_lo[d])   # _capi.BOUNDARY_PERIODIC / BOUNDARY_PEC
            cfg.field_boundary_hi[d] = int(field_boundary_hi[d])
            cfg.particle_boundary_lo[d] = int(particle_boundary_lo[d])   # _capi.PBOUNDARY_*
            cfg.particle_boundary_hi[d] = int(particle_boundary_hi[d])
        cfg.cfl = float(cfl)
        cfg.nox = int(nox)
        if galerkin is None:
            # the reference's resolution of WarpX::galerkin_interpolation (Source/WarpX.cpp:967,1208-1214): the same
            # shape factors in all directions on a collocated grid and with direct deposition + an EM solver
            galerkin = 0 if (int(current_deposition) == _capi.DEPOSIT_DIRECT or
                             int(grid_type) == _capi.GRID_COLLOCATED) else 1
        cfg.galerkin = int(galerkin)
        cfg.particle_pusher = int(particle_pusher)
        cfg.current_deposition = int(current_deposition)
        cfg.use_filter = int(use_filter)
        cfg.sort_interval = int(sort_interval)
        cfg.grid_type = int(grid_type)   # collocated: CPU restatement only
        cfg.maxwell_solver = int(maxwell_solver)   # algo.maxwell_solver: SOLVER_YEE or SOLVER_CKC
        cfg.overlap_halo = int(overlap_halo)
        cfg.gamma_boost = float(gamma_boost)   # warpx.gamma_boost (boost along z); prob_lo / prob_hi are boosted-frame values
        cfg.use_fdtd_nci_corr = int(use_fdtd_nci_corr)   # particles.use_fdtd_nci_corr: Godfrey filter on E, B before the gather
        self.cfg = cfg
        self._comm = comm  # keep the callbacks alive
        self._h = C.c_void_p()
        lib.sim_create(C.byref(cfg), C.byref(comm) if comm is not None else None, C.byref(self._h))
        self.species = []
        self.dx = [(float(prob_hi[d]) - float(prob_lo[d])) / int(n_cell[d]) for d in range(3)]

    @classmethod
    def from_inputs(cls, lib: _capi.CLib, inputs_path, overrides=(), nbricks=None, coord=None,
                    comm: _capi.Comm | None = None, diagnostics: bool = False):
        """The simulation a WarpX inputs file describes (wxa_sim_create_from_inputs): `overrides` are
        "name=value" strings like the reference's command line; raises WxaError naming any parameter that is
        outside this library's path.  diagnostics=False (or an override warpx_amd.write_diagnostics=0): the plotfiles
        and reduced-diagnostics files the deck asks for are not written (the C entry point's default is to write them,
        as the reference does: python -m warpx_amd.run)."""
        overrides = ([] if diagnostics else ["warpx_amd.write_diagnostics=0"]) + list(overrides)
        self = cls.__new__(cls)
        self.lib = lib
        self.on_device = lib.prefix == "wxa_"
        self.cfg = None
        self._comm = comm
        self._h = C.c_void_p()
        ov = (C.c_char_p * max(len(overrides), 1))(*[o.encode() for o in overrides])
        nb = (C.c_int32 * 3)(*nbricks) if nbricks is not None else None
        co = (C.c_int32 * 3)(*coord) if coord is not None else None
        lib.sim_create_from_inputs(str(inputs_path).encode(), len(overrides), ov,
                                   C.byref(comm) if comm is not None else None, nb, co, C.byref(self._h))
        self.max_step = lib.sim_max_step(self._h)
        self.species_names = [lib.sim_species_name(self._h, i).decode() for i in range(lib.sim_num_species(self._h))]
        self.species = [None] * len(self.species_names)
        self.dx = None
        return self

    def write_plotfile(self, path: str):
        """AMReX plotfile of the current state (wxa_sim_write_plotfile): the reference's FlushFormatPlotfile output."""
        self.lib.sim_write_plotfile(self._h, str(path).encode())

    BTD_COMPONENTS = ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz", "rho")

    def add_btd(self, num_snapshots: int, dt_snapshots_lab: float, buffer_size: int = 256, write_species: bool = True):
        """<diag>.diag_type = BackTransformed (wxa_sim_add_btd): lab-frame snapshots every dt_snapshots_lab -- the fields,
        and with write_species the particles of every species."""
        self.lib.sim_add_btd(self._h, int(num_snapshots), float(dt_snapshots_lab), int(buffer_size), 1 if write_species else 0)

    def btd_particles(self, i: int, sid: int) -> np.ndarray:
        """(7, n) lab-frame x, y, z, w, ux, uy, uz of the particles of species `sid` that snapshot i has met so far."""
        n = C.c_int64()
        self.lib.sim_btd_num_particles(self._h, int(i), int(sid), C.byref(n))
        out = np.zeros((7, n.value), dtype=np.float64)
        if n.value:
            self.lib.sim_btd_particles(self._h, int(i), int(sid), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def btd_info(self, i: int) -> dict:
        n = (C.c_int32 * 3)()
        z = (C.c_double * 2)()
        t = C.c_double()
        filled, full = C.c_int32(), C.c_int32()
        self.lib.sim_btd_info(self._h, int(i), n, z, C.byref(t), C.byref(filled), C.byref(full))
        return {"n": tuple(n), "z_lab": tuple(z), "t_lab": t.value, "slices": filled.value, "full": bool(full.value)}

    def btd_set_flush(self, file_prefix: str, file_min_digits: int = 6):
        """Write the lab-frame snapshots to <file_prefix><i>/ buffer by buffer instead of keeping them (wxa_sim_btd_set_flush)."""
        self.lib.sim_btd_set_flush(self._h, str(file_prefix).encode(), int(file_min_digits))

    def btd_flush(self):
        """The forced flush after the last step (wxa_sim_btd_flush)."""
        self.lib.sim_btd_flush(self._h)

    def btd_box(self, i: int):
        """(lo, hi), inclusive: this brick's share of snapshot i in the snapshot's (x, y, k_lab) index space."""
        lo, hi = (C.c_int32 * 3)(), (C.c_int32 * 3)()
        self.lib.sim_btd_box(self._h, int(i), lo, hi)
        return tuple(lo), tuple(hi)

    def btd_snapshot(self, i: int, name: str) -> np.ndarray:
        """Component `name` of lab-frame snapshot i, indexed [i, j, k] (zeros where no slice has arrived yet)."""
        n = self.btd_info(i)["n"]
        out = np.zeros(n[0] * n[1] * n[2], dtype=np.float64)
        self.lib.sim_btd_data(self._h, int(i), self.BTD_COMPONENTS.index(name), out.ctypes.data_as(C.POINTER(C.c_double)))
        return np.ascontiguousarray(out.reshape(n[2], n[1], n[0]).transpose(2, 1, 0))

    def btd_write_plotfile(self, i: int, path: str):
        """Lab-frame snapshot i as an AMReX plotfile (wxa_sim_btd_write_plotfile)."""
        self.lib.sim_btd_write_plotfile(self._h, int(i), str(path).encode())

    def add_full_diag(self, name: str, intervals: str, file_prefix: str | None = None, file_min_digits: int = 6,
                      fields=None, write_species: bool = True, dump_last_timestep: bool = True):
        """diagnostics.diags_names += name with diag_type = Full, format = plotfile (wxa_sim_add_full_diag): plotfiles
        <file_prefix><step> at the steps of `intervals`; fields: names out of Ex .. jz, rho (None: Ex .. jz)."""
        self.lib.sim_add_full_diag(self._h, name.encode(), str(intervals).encode(),
                                   None if file_prefix is None else str(file_prefix).encode(), int(file_min_digits),
                                   None if fields is None else " ".join(fields).encode(), 1 if write_species else 0,
                                   1 if dump_last_timestep else 0)

    def flush_diags_last_timestep(self):
        """The forced flush after the last step of a run without a deck's max_step (wxa_sim_flush_diags_last_timestep)."""
        self.lib.sim_flush_diags_last_timestep(self._h)

    def add_reduced_diag(self, name: str, rd_type: str, intervals: str = "1", path: str | None = None):
        """warpx.reduced_diags_names += name (wxa_sim_add_reduced_diag): rd_type in FieldEnergy, ParticleEnergy,
        ParticleMomentum, ParticleNumber; rows go to <path><name>.txt (path ends with '/'; None: kept in memory only)."""
        self.lib.sim_add_reduced_diag(self._h, name.encode(), rd_type.encode(), str(intervals).encode(),
                                      None if path is None else str(path).encode())

    def reduced_diag(self, name: str, compute_now: bool = False) -> np.ndarray:
        """The data columns (after step and time) of the last row of diagnostic `name`; compute_now: of the current state."""
        n = C.c_int32()
        self.lib.sim_reduced_diag_data(self._h, name.encode(), 1 if compute_now else 0, None, 0, C.byref(n))
        out = np.zeros(n.value, dtype=np.float64)
        if n.value:
            self.lib.sim_reduced_diag_data(self._h, name.encode(), 0, out.ctypes.data_as(C.POINTER(C.c_double)),
                                           n.value, C.byref(n))
        return out

    def checksum(self) -> dict:
        """The reference's regression checksum of the current state (wxa_sim_checksum_json), this brick's share."""
        import json
        # every call recomputes the sums (rho is deposited with atomics: its last digits, and with them the length of the
        # text, can differ from call to call), so the buffer is taken with a margin and the call repeated if it was short
        cap = 4096
        while True:
            buf = C.create_string_buffer(cap)
            n = self.lib.sim_checksum_json(self._h, buf, cap)
            if n < 0:
                raise _capi.WxaError("sim_checksum_json failed")
            if n < cap:
                return json.loads(buf.value.decode())
            cap = n + 256

    def close(self):
        if self._h:
            self.lib.sim_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- species ------------------------------------------------------------
    def add_species(self, charge, mass, particles):
        """`particles`: ParticleArrays living where the library expects them
        (device for the product, host for the oracle) or a list of 7 numpy arrays."""
        if not isinstance(particles, ParticleArrays):
            particles = ParticleArrays.from_numpy(particles, self.lib.memory)
        sid = C.c_int32(-1)
        v = particles.view
        if str(self.lib.memory).startswith("cuda"):   # whatever stream filled `particles` (torch's): done before the library's reads it
            import torch
            torch.cuda.synchronize()
        self.lib.sim_add_species(self._h, float(charge), float(mass), C.byref(v), C.byref(sid))
        self.species.append((float(charge), float(mass)))
        return sid.value

    def set_external_particle_fields(self, sid: int, E, B):
        """particles.E_external_particle / B_external_particle (constant) for species `sid`."""
        self.lib.sim_set_external_particle_fields(self._h, int(sid), (C.c_double * 3)(*map(float, E)),
                                                  (C.c_double * 3)(*map(float, B)))

    def set_deposit_accumulator(self, sid: int, acc: int):
        """_capi.ACC_FP64 (default) / _capi.ACC_FP32: accumulator of the LDS-tile Esirkepov deposition of species `sid`."""
        self.lib.sim_set_deposit_accumulator(self._h, int(sid), int(acc))

    def set_radiation_reaction(self, sid: int, on=True):
        """<species>.do_classical_radiation_reaction: Boris + radiation reaction for species `sid`."""
        self.lib.sim_set_radiation_reaction(self._h, int(sid), 1 if on else 0)

    # ---- stepping -----------------------------------------------------------
    def set_synchronize_at_end(self, on: bool):
        """evolve() leaves the momenta at the half step (on = False): consecutive calls are the steps of one long run."""
        self.lib.sim_set_synchronize_at_end(self._h, 1 if on else 0)

    def synchronize(self):
        """WarpX::Synchronize: the momenta to the time of the positions, if they are not there already."""
        self.lib.sim_synchronize(self._h)

    def set_safe_guard_cells(self, on: bool = True):
        """warpx.safe_guard_cells: all allocated guard cells in every exchange, every exchange of the reference's schedule."""
        self.lib.sim_set_safe_guard_cells(self._h, 1 if on else 0)

    def set_single_precision_comms(self, on: bool = True):
        """warpx.do_single_precision_comms: float on the wire of the guard exchanges between bricks."""
        self.lib.sim_set_single_precision_comms(self._h, 1 if on else 0)

    def evolve(self, numsteps: int):
        self.lib.sim_evolve(self._h, int(numsteps))

    @property
    def halo_overlap(self) -> bool:
        """Whether overlap_halo took effect (split direction, all-periodic, bricks thick enough)."""
        return bool(self.lib.sim_halo_overlap(self._h))

    @property
    def dt(self) -> float:
        return self.lib.sim_dt(self._h)

    @property
    def istep(self) -> int:
        return self.lib.sim_istep(self._h)

    # ---- data access --------------------------------------------------------
    def _d2h(self):
        return self.lib.copy_to_host if self.on_device else None

    def compute_rho(self):
        """RhoFunctor: total charge density at the current positions -> field "rho"."""
        self.lib.sim_compute_rho(self._h)

    def field_view(self, name: str) -> _capi.FieldView:
        v = _capi.FieldView()
        self.lib.sim_get_field(self._h, name.encode(), C.byref(v))
        return v

    def field(self, name: str) -> np.ndarray:
        """Dense [i,j,k] host copy including guards."""
        return view_to_numpy(self.field_view(name), self._d2h())

    def set_field(self, name: str, values: np.ndarray):
        """Overwrites a field (guards included) with a dense [i,j,k] array shaped like `field(name)`:
        the stand-in for warpx.E/B_ext_grid_init_style = parse_*_ext_grid_function."""
        v = self.field_view(name)
        n = tuple(v.n)
        assert values.shape == n, (values.shape, n)
        buf = np.zeros((n[2], n[1], int(v.jstride)), dtype=np.float64)
        buf[:, :, : n[0]] = np.transpose(values, (2, 1, 0))
        buf = np.ascontiguousarray(buf).reshape(-1)[: int(v.kstride) * n[2]]
        if self.on_device:
            self.lib.copy_to_device(v.p, buf.ctypes.data, 8 * buf.size)
        else:
            C.memmove(v.p, buf.ctypes.data, 8 * buf.size)

    def field_valid(self, name: str) -> np.ndarray:
        v = self.field_view(name)
        a = view_to_numpy(v, self._d2h())
        g = v.ng
        return a[g[0]: v.n[0] - g[0], g[1]: v.n[1] - g[1], g[2]: v.n[2] - g[2]]

    def particle_view(self, sid: int) -> _capi.ParticleView:
        v = _capi.ParticleView()
        self.lib.sim_get_particles(self._h, int(sid), C.byref(v))
        return v

    def particles(self, sid: int) -> np.ndarray:
        """(7, np) host copy in PIdx order x,y,z,w,ux,uy,uz."""
        return particles_to_numpy(self.particle_view(sid), self._d2h())

    def enable_timers(self, on=True):
        self.lib.sim_enable_timers(self._h, 1 if on else 0)

    def dry_comm(self, reps=10):
        """Only the step's neighbour exchanges on the run's own arrays (wxa_sim_dry_comm): ms per call of FillBoundary E+B,
        SumBoundary J, Redistribute, and the three in a row.  Collective over the run's ranks."""
        ms = (C.c_double * 4)()
        self.lib.sim_dry_comm(self._h, int(reps), ms)
        return {"FillBoundaryEB": ms[0], "SumBoundaryJ": ms[1], "Redistribute": ms[2], "all_three": ms[3]}

    def timers(self, reset=False):
        ms = (C.c_double * 8)()
        cnt = (C.c_int64 * 8)()
        self.lib.sim_get_timers(self._h, ms, cnt, 1 if reset else 0)
        names = ["GatherAndPush", "CurrentDeposition", "SyncCurrent", "EvolveB", "EvolveE",
                 "FillBoundary", "Redistribute", "other"]
        return {n: (ms[i], cnt[i]) for i, n in enumerate(names)}


# ---- order-independent reductions that define the parity metric -------------------
# (formulas: Source/Diagnostics/ReducedDiags/FieldEnergy.cpp:81-157,
#  ParticleEnergy.cpp:95-200 + Source/Particles/Algorithms/KineticEnergy.H:31-45,
#  ParticleMomentum.cpp; evaluated on host copies in extended precision so that both
#  sides of a comparison are reduced the same way)

def field_energy(sim: WarpXSim):
    from .plasma import EP0, MU0
    dV = sim.dx[0] * sim.dx[1] * sim.dx[2]

    def sumsq(name):
        v = sim.field_view(name)
        a = view_to_numpy(v, sim._d2h())
        g = v.ng
        # unique points: drop the duplicated high-edge nodal point (norm2 with periodicity)
        sl = tuple(slice(g[d], v.n[d] - g[d] - v.stag[d]) for d in range(3))
        x = a[sl].astype(np.longdouble)
        return float(np.sum(x * x))

    Es = sumsq("Ex") + sumsq("Ey") + sumsq("Ez")
    Bs = sumsq("Bx") + sumsq("By") + sumsq("Bz")
    return 0.5 * Es * EP0 * dV, 0.5 * Bs / MU0 * dV


def particle_moments(sim: WarpXSim, sid: int):
    from .plasma import C_LIGHT
    q, m = sim.species[sid]
    p = sim.particles(sid).astype(np.longdouble)
    w, ux, uy, uz = p[3], p[4], p[5], p[6]
    u2 = ux * ux + uy * uy + uz * uz
    gamma = np.sqrt(1.0 + u2 / (np.longdouble(C_LIGHT) ** 2))
    ekin = float(np.sum(w * (m * u2 / (1.0 + gamma))))
    mom = [float(np.sum(w * m * c)) for c in (ux, uy, uz)]
    absmom = [float(np.sum(np.abs(m * c))) for c in (ux, uy, uz)]
    abspos = [float(np.sum(np.abs(p[i]))) for i in range(3)]
    return {"ekin": ekin, "momentum": mom, "abs_momentum": absmom, "abs_position": abspos,
            "weight": float(np.sum(w))}
