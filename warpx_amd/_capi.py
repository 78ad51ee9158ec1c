"""ctypes binding of the C-ABI declared in include/warpx_amd.h.

The binder is generic over (shared library, symbol prefix) so the test-suite can
bind the CPU oracle's `orc_*` entry points with the very same
signatures; the product only ever loads `libwarpx_amd.so` (`wxa_*`) and raises
if it is missing -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_HERE, "libwarpx_amd.so")


class FieldView(C.Structure):
    _fields_ = [
        ("p", C.c_void_p),
        ("lo", C.c_int32 * 3),
        ("n", C.c_int32 * 3),
        ("ng", C.c_int32 * 3),
        ("stag", C.c_int32 * 3),
        ("jstride", C.c_int64),
        ("kstride", C.c_int64),
    ]


class ParticleView(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("w", C.c_void_p),
        ("ux", C.c_void_p), ("uy", C.c_void_p), ("uz", C.c_void_p),
        ("idcpu", C.c_void_p),
        ("np", C.c_int64),
    ]


class GridGeom(C.Structure):
    _fields_ = [
        ("xyzmin", C.c_double * 3),
        ("dinv", C.c_double * 3),
        ("lo", C.c_int32 * 3),
    ]


GRID_STAGGERED, GRID_COLLOCATED = 0, 1
SOLVER_YEE, SOLVER_CKC = 0, 1
PART_INTERIOR, PART_REST = 1, 2


class SimConfig(C.Structure):
    _fields_ = [
        ("n_cell", C.c_int32 * 3),
        ("prob_lo", C.c_double * 3),
        ("prob_hi", C.c_double * 3),
        ("cfl", C.c_double),
        ("nox", C.c_int32),
        ("galerkin", C.c_int32),
        ("particle_pusher", C.c_int32),
        ("current_deposition", C.c_int32),
        ("use_filter", C.c_int32),
        ("sort_interval", C.c_int32),
        ("nbricks", C.c_int32 * 3),
        ("coord", C.c_int32 * 3),
        ("field_boundary_lo", C.c_int32 * 3),
        ("field_boundary_hi", C.c_int32 * 3),
        ("particle_boundary_lo", C.c_int32 * 3),
        ("particle_boundary_hi", C.c_int32 * 3),
        ("overlap_halo", C.c_int32),
        ("grid_type", C.c_int32),
        ("maxwell_solver", C.c_int32),
        ("gamma_boost", C.c_double),
        ("use_fdtd_nci_corr", C.c_int32),
    ]


class MovingWindow(C.Structure):
    _fields_ = [("dir", C.c_int32), ("v", C.c_double)]


class PlasmaInjector(C.Structure):
    _fields_ = [("density", C.c_double), ("ppc", C.c_int32 * 3), ("lo", C.c_double * 3), ("hi", C.c_double * 3),
                ("gamma_boost", C.c_double), ("t", C.c_double)]


class RepeatedPlasmaLens(C.Structure):
    _fields_ = [("n_lenses", C.c_int32), ("period", C.c_double), ("starts", C.POINTER(C.c_double)),
                ("lengths", C.POINTER(C.c_double)), ("strengths_E", C.POINTER(C.c_double)),
                ("strengths_B", C.POINTER(C.c_double)), ("gamma_boost", C.c_double), ("dt", C.c_double)]

    @classmethod
    def make(cls, period, starts, lengths, strengths_E, strengths_B, gamma_boost, dt):
        n = len(starts)
        arrays = [(C.c_double * n)(*map(float, a)) for a in (starts, lengths, strengths_E, strengths_B)]
        lens = cls(n, float(period), *[C.cast(a, C.POINTER(C.c_double)) for a in arrays], float(gamma_boost), float(dt))
        lens._keep = arrays
        return lens


class InjectedMomentum(C.Structure):
    _fields_ = [("u_mean", C.c_double * 3), ("u_th", C.c_double * 3), ("seed", C.c_uint64), ("origin", C.c_double * 3)]


class LaserAntenna(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("direction", C.c_double * 3), ("polarization", C.c_double * 3),
                ("e_max", C.c_double), ("wavelength", C.c_double), ("waist", C.c_double), ("duration", C.c_double),
                ("t_peak", C.c_double), ("focal_distance", C.c_double)]


class LaserPushParams(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("p_X", C.c_double * 3), ("p_Y", C.c_double * 3),
                ("mobility", C.c_double), ("e_max", C.c_double), ("wavelength", C.c_double), ("waist", C.c_double),
                ("duration", C.c_double), ("t_peak", C.c_double), ("focal_distance", C.c_double),
                ("nvec", C.c_double * 3), ("gamma_boost", C.c_double)]


EXCHANGE_FN = C.CFUNCTYPE(
    C.c_int, C.c_void_p, C.c_int,
    C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
    C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
    C.c_void_p)
EXCHANGE_COUNTS_FN = C.CFUNCTYPE(
    C.c_int, C.c_void_p, C.c_int,
    C.POINTER(C.c_int32), C.POINTER(C.c_int64),
    C.POINTER(C.c_int32), C.POINTER(C.c_int64))


class Comm(C.Structure):
    _fields_ = [
        ("ctx", C.c_void_p),
        ("rank", C.c_int32),
        ("nranks", C.c_int32),
        ("exchange", EXCHANGE_FN),
        ("exchange_counts", EXCHANGE_COUNTS_FN),
    ]


class RcclStats(C.Structure):
    _fields_ = [("n_exchanges", C.c_int64), ("n_messages", C.c_int64), ("bytes_sent", C.c_int64),
                ("n_count_exchanges", C.c_int64), ("timed_exchanges", C.c_int64), ("timed_ms", C.c_double)]


RCCL_LOOPBACK, RCCL_TIMING = 1, 2
PUSHER_BORIS, PUSHER_VAY, PUSHER_HC, PUSHER_BORIS_RR = 0, 1, 2, 3
DEPOSIT_ESIRKEPOV, DEPOSIT_DIRECT = 0, 1
BOUNDARY_PERIODIC, BOUNDARY_PEC = 0, 1
PBOUNDARY_DEFAULT, PBOUNDARY_ABSORBING, PBOUNDARY_REFLECTING, PBOUNDARY_PERIODIC = 0, 1, 2, 3

_FV3 = FieldView * 3
_D3 = C.c_double * 3
_I3 = C.c_int * 3
_I32_3 = C.c_int32 * 3
_PFV = C.POINTER(FieldView)
_PPV = C.POINTER(ParticleView)
_PGG = C.POINTER(GridGeom)

# name -> (restype, argtypes); shared by the product (wxa_) and the oracle (orc_)
_KERNEL_SIGS = {
    "evolve_b": (C.c_int, [_FV3, _FV3, C.c_double, _D3, C.c_void_p]),
    "evolve_e": (C.c_int, [_FV3, _FV3, _FV3, C.c_double, _D3, C.c_void_p]),
    "gather_push": (C.c_int, [_PPV, _FV3, _FV3, _PGG, C.c_double, C.c_double, C.c_double,
                              C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "push_p": (C.c_int, [_PPV, _FV3, _FV3, _PGG, C.c_double, C.c_double, C.c_double,
                         C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "deposit_current": (C.c_int, [_PPV, _FV3, _PGG, C.c_double, C.c_double, C.c_double,
                                  C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "deposit_charge": (C.c_int, [_PPV, _PFV, _PGG, C.c_double, C.c_int, C.c_void_p]),
    "btd_select_particles": (C.c_int, [_PPV, C.POINTER(C.c_void_p), C.c_double, C.c_double, C.c_double, C.c_double,
                                       C.c_double, C.c_double, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
    "reduce_field": (C.c_int, [_PFV, _I32_3, _I32_3, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "reduce_particles": (C.c_int, [_PPV, C.c_double, C.c_int32, C.c_double * 6, C.c_void_p]),
    "enforce_periodic": (C.c_int, [_PPV, _D3, _D3, _I3, C.c_void_p]),
    "apply_pec_e": (C.c_int, [_FV3, _I32_3, _I32_3, _I32_3, _I32_3, _I32_3, C.c_void_p]),
    "apply_pec_b": (C.c_int, [_FV3, _I32_3, _I32_3, _I32_3, _I32_3, _I32_3, C.c_void_p]),
    "apply_pec_j": (C.c_int, [_FV3, _I32_3, _I32_3, _I32_3, _I32_3, C.c_void_p]),
    "apply_particle_boundaries": (C.c_int, [_PPV, _D3, _D3, _I32_3, _I32_3, C.POINTER(C.c_int64), C.c_void_p,
                                            C.c_void_p]),
    "evolve_b_guard_layer": (C.c_int, [_FV3, _FV3, C.c_double, _D3, _I32_3, C.c_void_p]),
    "evolve_b_ckc": (C.c_int, [_FV3, _FV3, C.c_double, C.c_double * 5, C.c_double * 5, C.c_double * 5, C.c_void_p]),
    "ckc_stencil_coefficients": (None, [_D3, C.c_double * 5, C.c_double * 5, C.c_double * 5]),
    "ckc_max_dt": (C.c_double, [_D3]),
    "apply_pec_rho": (C.c_int, [_PFV, _I32_3, _I32_3, _I32_3, _I32_3, C.c_void_p]),
    "shift_field_window": (C.c_int, [_PFV, C.c_void_p, C.c_int32, C.c_int32, _I3, C.c_void_p]),
    "laser_push": (C.c_int, [_PPV, C.POINTER(LaserPushParams), C.c_double, C.c_double, C.c_void_p]),
    "filter_bilinear": (C.c_int, [_PFV, _PFV, C.c_void_p]),
    "filter_stencil": (C.c_int, [_PFV, _PFV, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.c_int32,
                                 C.POINTER(C.c_double), C.c_int32, C.c_void_p]),
    "nci_godfrey_stencil": (C.c_int, [C.c_double, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "fill_boundary_periodic": (C.c_int, [_PFV, _I3, _I3, C.c_void_p]),
    "sum_boundary_periodic": (C.c_int, [_PFV, _I3, _I3, C.c_void_p]),
    "sync_nodal_periodic": (C.c_int, [_PFV, _I3, C.c_void_p]),
    "field_set_zero": (C.c_int, [_PFV, C.c_void_p]),
    "version": (C.c_char_p, []),
}

# the input-deck front end of the host layer (product and its CPU test build; not in the oracle)
_INPUTS_SIGS = {
    # the host layer's own entry points (not in the oracle stepper): the product, the CPU build of the host layer
    "sim_dry_comm": (C.c_int, [C.c_void_p, C.c_int32, C.c_double * 4]),
    "sim_create_from_inputs": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(Comm),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "sim_max_step": (C.c_int32, [C.c_void_p]),
    "sim_num_species": (C.c_int32, [C.c_void_p]),
    "sim_species_name": (C.c_char_p, [C.c_void_p, C.c_int32]),
    "sim_checksum_json": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
    "sim_write_plotfile": (C.c_int, [C.c_void_p, C.c_char_p]),
    "sim_add_full_diag": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32,
                                   C.c_int32]),
    "sim_flush_diags_last_timestep": (C.c_int, [C.c_void_p]),
    "sim_btd_write_plotfile": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p]),
    "sim_btd_set_flush": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "sim_btd_flush": (C.c_int, [C.c_void_p]),
    "sim_btd_box": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sim_set_synchronize_at_end": (C.c_int, [C.c_void_p, C.c_int32]),
    "sim_synchronize": (C.c_int, [C.c_void_p]),
    "sim_set_safe_guard_cells": (C.c_int, [C.c_void_p, C.c_int32]),
    "sim_set_single_precision_comms": (C.c_int, [C.c_void_p, C.c_int32]),
    "parser_eval": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                              C.POINTER(C.c_double)]),
    "last_error": (C.c_char_p, []),
}

_SIM_SIGS = {
    "sim_create": (C.c_int, [C.POINTER(SimConfig), C.c_void_p, C.POINTER(C.c_void_p)]),
    "sim_destroy": (None, [C.c_void_p]),
    "sim_add_species": (C.c_int, [C.c_void_p, C.c_double, C.c_double, _PPV, C.POINTER(C.c_int32)]),
    "sim_evolve": (C.c_int, [C.c_void_p, C.c_int32]),
    "sim_dt": (C.c_double, [C.c_void_p]),
    "sim_istep": (C.c_int64, [C.c_void_p]),
    "sim_get_field": (C.c_int, [C.c_void_p, C.c_char_p, _PFV]),
    "sim_get_particles": (C.c_int, [C.c_void_p, C.c_int32, _PPV]),
    "sim_get_timers": (C.c_int, [C.c_void_p, C.c_double * 8, C.c_int64 * 8, C.c_int]),
    "sim_enable_timers": (C.c_int, [C.c_void_p, C.c_int]),
    # moving window / continuous injection / laser antenna
    "sim_compute_rho": (C.c_int, [C.c_void_p]),
    "sim_halo_overlap": (C.c_int32, [C.c_void_p]),
    "sim_set_moving_window": (C.c_int, [C.c_void_p, C.POINTER(MovingWindow)]),
    "sim_set_injection": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(PlasmaInjector), C.c_int, C.c_int]),
    "sim_set_external_particle_fields": (C.c_int, [C.c_void_p, C.c_int32, _D3, _D3]),
    "sim_set_radiation_reaction": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "sim_add_laser": (C.c_int, [C.c_void_p, C.POINTER(LaserAntenna)]),
    "sim_add_btd": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_int32]),
    "sim_btd_num_particles": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "sim_btd_particles": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "sim_btd_info": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double),
                              C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sim_btd_data": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "sim_add_reduced_diag": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
    "sim_reduced_diag_data": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_double), C.c_int32,
                                       C.POINTER(C.c_int32)]),
}

# product-only entry points
_PRODUCT_SIGS = {
    "fill_boundary_periodic_multi": (C.c_int, [_PFV, C.c_int32, _I3, _I3, C.c_void_p]),
    "sum_boundary_periodic_multi": (C.c_int, [_PFV, C.c_int32, _I3, _I3, C.c_void_p]),
    "workspace_set_external_particle_fields": (C.c_int, [C.c_void_p, _D3, _D3]),
    "workspace_set_repeated_plasma_lens": (C.c_int, [C.c_void_p, C.POINTER(RepeatedPlasmaLens)]),
    "workspace_set_time": (C.c_int, [C.c_void_p, C.c_double]),
    "workspace_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "workspace_destroy": (None, [C.c_void_p]),
    "last_error": (C.c_char_p, []),
    "sort_particles_by_cell": (C.c_int, [_PPV, _PPV, _D3, _D3, _I32_3, _I32_3, C.c_void_p, C.c_void_p]),
    "gather_push_ws": (C.c_int, [_PPV, _FV3, _FV3, _PGG, C.c_double, C.c_double, C.c_double,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "add_plasma": (C.c_int, [_PPV, C.POINTER(PlasmaInjector), _D3, _I32_3, _D3, _D3, _D3,
                             C.POINTER(InjectedMomentum), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]),
    "gather_push_part": (C.c_int, [_PPV, _FV3, _FV3, _PGG, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "partition_particles": (C.c_int, [_PPV, _PPV, C.c_int, C.c_double, C.c_double,
                                      C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]),
    "wrap_and_classify": (C.c_int, [_PPV, C.c_int64, C.c_int64, _D3, _D3, _I3, _D3, _D3, _I3, C.c_void_p,
                                    C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]),
    "wrap_and_classify_dest": (C.c_int, [_PPV, C.c_int64, C.c_int64, _D3, _D3, _I3, _D3, _D3, _I3, C.c_void_p,
                                         C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]),
    "pack_leavers": (C.c_int, [_PPV, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, _D3, _D3,
                               C.c_void_p]),
    "sort_live_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    "push_sort_begin": (C.c_int, [C.c_void_p, C.c_int32, _PPV, _PPV, _D3, _D3, _I32_3, _I32_3, _I32_3, C.c_int32, C.c_double, C.c_void_p]),
    "push_sort_end": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]),
    "push_sort_pending": (C.c_int32, [C.c_void_p, _PPV]),
    "pack_box": (C.c_int, [_PFV, _I32_3, _I32_3, C.c_void_p, C.c_void_p]),
    "unpack_box": (C.c_int, [_PFV, _I32_3, _I32_3, C.c_void_p, C.c_int, C.c_void_p]),
    "pack_box_f32": (C.c_int, [_PFV, _I32_3, _I32_3, C.c_void_p, C.c_void_p]),
    "unpack_box_f32": (C.c_int, [_PFV, _I32_3, _I32_3, C.c_void_p, C.c_int, C.c_void_p]),
    "field_set_zero_multi": (C.c_int, [_PFV, C.c_int32, C.c_void_p]),
    "copy_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "copy_to_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "device_synchronize": (C.c_int, []),
    "enforce_periodic_sorted": (C.c_int, [_PPV, _D3, _D3, _I3, C.c_void_p, C.c_int32, C.c_void_p]),
    "workspace_set_deposit_accumulator": (C.c_int, [C.c_void_p, C.c_int32]),
    "workspace_set_streaming_plasma": (C.c_int, [C.c_void_p, C.c_int32]),
    "sim_set_deposit_accumulator": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
}
ACC_FP64, ACC_FP32 = 0, 1
PUSH_SORT_COUNT, PUSH_SORT_SCATTER = 1, 2

# the library's own transport (rccl_comm.hip): product only
_TRANSPORT_SIGS = {
    "rccl_unique_id": (C.c_int, [C.c_char * 128]),
    "rccl_comm_create": (C.c_int, [C.c_char * 128, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Comm)]),
    "rccl_comm_destroy": (None, [C.POINTER(Comm)]),
    "rccl_comm_stats": (C.c_int, [C.POINTER(Comm), C.POINTER(RcclStats), C.c_int32]),
    "rccl_comm_set_timing": (C.c_int, [C.POINTER(Comm), C.c_int32]),
}

# oracle-only entry points (diagnostic formulas that define the parity metric)
_ORACLE_SIGS = {
    "sum_sq_unique": (C.c_double, [_PFV]),
    "field_energy": (None, [_FV3, _FV3, _D3, C.c_double * 3]),
    "particle_energy": (C.c_double, [_PPV, C.c_double]),
    "particle_momentum": (None, [_PPV, C.c_double, C.c_double * 3]),
    "cell_centered_abs_sum": (C.c_double, [_PFV]),
    "abs_sum": (C.c_double, [C.c_void_p, C.c_int64, C.c_double]),

    "pack_box_f32": (C.c_int, [_PFV, _I32_3, _I32_3, C.c_void_p, C.c_void_p]),
    "unpack_box_f32": (C.c_int, [_PFV, _I32_3, _I32_3, C.c_void_p, C.c_int, C.c_void_p]),
    "num_threads": (C.c_int, []),
    "set_num_threads": (C.c_int, [C.c_int]),
    "add_plasma": (C.c_int, [_PPV, C.POINTER(PlasmaInjector), _D3, _I32_3, _D3, _D3, _D3,
                             C.POINTER(InjectedMomentum), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]),
    # CPU counterparts of the Redistribute entry points (host-layer tests, parity tests)
    "sort_particles_by_cell": (C.c_int, [_PPV, _PPV, _D3, _D3, _I32_3, _I32_3, C.c_void_p, C.c_void_p]),
    "wrap_and_classify": (C.c_int, [_PPV, C.c_int64, C.c_int64, _D3, _D3, _I3, _D3, _D3, _I3, C.c_void_p,
                                    C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]),
    "wrap_and_classify_dest": (C.c_int, [_PPV, C.c_int64, C.c_int64, _D3, _D3, _I3, _D3, _D3, _I3, C.c_void_p,
                                         C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]),
    "pack_leavers": (C.c_int, [_PPV, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, _D3, _D3,
                               C.c_void_p]),
}


class WxaError(RuntimeError):
    pass


# int-returning entry points whose result is a value, not a status
_RETURNS_A_VALUE = {"sim_max_step", "sim_num_species", "num_threads", "set_num_threads", "sim_halo_overlap", "push_sort_pending"}


class CLib:
    """A loaded C library whose symbols `<prefix><name>` follow include/warpx_amd.h."""

    def __init__(self, path: str, prefix: str, extra_sigs: dict | None = None, kernels: bool = True,
                 memory: str | None = None):
        # where the library expects the buffers it is handed: device memory for the product
        self.memory = memory or ("cuda" if prefix == "wxa_" else "cpu")
        if not os.path.exists(path):
            raise WxaError(
                f"{path} not found: build it first (python -c 'import __graft_entry__ as g; g.build()')")
        self.path = path
        self.prefix = prefix
        self._dll = C.CDLL(path, mode=C.RTLD_GLOBAL if prefix == "wxa_" else C.RTLD_LOCAL)
        sigs = dict(_KERNEL_SIGS) if kernels else {}
        sigs.update(_SIM_SIGS)
        if extra_sigs:
            sigs.update(extra_sigs)
        self.names = []
        for name, (res, args) in sigs.items():
            fn = getattr(self._dll, prefix + name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
            setattr(self, "_" + name, fn)
            self.names.append(prefix + name)

    def __getattr__(self, name):
        # checked call wrappers: lib.evolve_b(...) raises on a negative status
        raw = object.__getattribute__(self, "_" + name)
        if raw.restype is C.c_int and name not in _RETURNS_A_VALUE:
            def checked(*a):
                rc = raw(*a)
                if rc != 0:
                    msg = ""
                    if hasattr(self, "_last_error"):
                        try:
                            msg = (self._last_error() or b"").decode()
                        except Exception:
                            pass
                    raise WxaError(f"{self.prefix}{name} failed with status {rc} {msg}")
                return rc
            return checked
        return raw


_product = None


def load_product() -> CLib:
    """The HIP library. Fails loudly when it has not been built (no CPU fallback)."""
    global _product
    if _product is None:
        # WXA_PRODUCT_LIB: another build of the same library (kernel experiments, scripts/)
        _product = CLib(os.environ.get("WXA_PRODUCT_LIB", PRODUCT_LIB), "wxa_",
                       {**_PRODUCT_SIGS, **_INPUTS_SIGS, **_TRANSPORT_SIGS})
    return _product


def declared_symbols(header_path: str) -> list[str]:
    """Every `wxa_*` function name declared in include/warpx_amd.h (for the symbol-export test)."""
    import re
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wxa_[a-z0-9_]+)\s*\(", text)))
