// Kernel/memory backend table used by the C++17 host layer.
//
// The host layer (MultiFab / MultiFabRegister / WarpXParticleContainer /
// FiniteDifferenceSolver / WarpX shims and the step schedule) is plain C++ and talks to
// the device only through this table, whose kernel entries have exactly the C-ABI
// signatures of include/warpx_amd.h.  The product library fills it with the HIP
// kernels (warpx_host.hip).  The test-suite compiles the same host sources against the
// CPU oracle's entry points to exercise the multi-brick exchange logic over gloo on a
// machine without a GPU (tests/host_cpu/); that build is never shipped.
#ifndef WXA_HOST_BACKEND_HPP_
#define WXA_HOST_BACKEND_HPP_

#include <cstddef>
#include <cstdint>

#include "../../../include/warpx_amd.h"

namespace wxa::host {

struct Backend {
    const char* name;
    // ---- kernels (C-ABI signatures) ----
    int (*evolve_b)(const wxa_field_view*, const wxa_field_view*, double, const double*, void*);
    int (*evolve_e)(const wxa_field_view*, const wxa_field_view*, const wxa_field_view*, double,
                    const double*, void*);
    // first guard layer of B from the guards already present (instead of FillBoundaryB after the update)
    int (*evolve_b_guard_layer)(const wxa_field_view*, const wxa_field_view*, double, const double*, const int32_t* grow,
                                void*);
    // algo.maxwell_solver = ckc: stencil coefficients, time-step limit, update of B (wxa_evolve_b_ckc)
    void (*ckc_stencil_coefficients)(const double* cell_size, double* cx, double* cy, double* cz);
    double (*ckc_max_dt)(const double* cell_size);
    int (*evolve_b_ckc)(const wxa_field_view*, const wxa_field_view*, double, const double* cx, const double* cy,
                        const double* cz, void*);
    // constant external fields of the container that owns workspace `ws` (wxa_workspace_set_external_particle_fields)
    int (*ws_set_external_eb)(void* ws, const double* E, const double* B);
    // optional: repeated plasma lens of that container and the time its fields are evaluated at
    // (wxa_workspace_set_repeated_plasma_lens, wxa_workspace_set_time)
    int (*ws_set_repeated_plasma_lens)(void* ws, const wxa_repeated_plasma_lens*) = nullptr;
    int (*ws_set_time)(void* ws, double t) = nullptr;
    // optional: accumulator type of the LDS-tile deposition of that container (wxa_workspace_set_deposit_accumulator)
    int (*ws_set_deposit_accumulator)(void* ws, int32_t acc) = nullptr;
    // optional: the container's plasma streams through the grid (wxa_workspace_set_streaming_plasma)
    int (*ws_set_streaming_plasma)(void* ws, int32_t on) = nullptr;
    // gather + push; move != 0 -> PushPX, move == 0 -> PushP; ws = the container's workspace
    int (*gather_push)(const wxa_particle_view*, const wxa_field_view*, const wxa_field_view*,
                       const wxa_grid_geom*, double, double, double, int, int, int, int move, void* ws, void*);
    // PushPX on the interior tiles / on the rest (wxa_gather_push_part)
    int (*gather_push_part)(const wxa_particle_view*, const wxa_field_view*, const wxa_field_view*,
                            const wxa_grid_geom*, double, double, double, int, int, int, void* ws, int part, void*);
    int (*deposit_current)(const wxa_particle_view*, const wxa_field_view*, const wxa_grid_geom*, double,
                           double, double, int, int, void* ws, void*);
    // PhysicalParticleContainer::AddPlasma on the device (constant density; at rest, constant or gaussian momentum)
    int (*add_plasma)(const wxa_particle_view* dst, const wxa_plasma_injector*, const double* corner, const int32_t* ncells,
                      const double* dx, const double* brick_lo, const double* brick_hi, const wxa_injected_momentum* momentum,
                      int64_t* n_added,
                      void* ws, void*);
    // diagnostics: doChargeDepositionShapeN
    int (*deposit_charge)(const wxa_particle_view*, const wxa_field_view*, const wxa_grid_geom*, double, int, void*);
    // PEC field boundary (wxa_apply_pec_e / wxa_apply_pec_b)
    int (*apply_pec_e)(const wxa_field_view*, const int32_t* dom_lo, const int32_t* dom_hi, const int32_t* pec_lo,
                       const int32_t* pec_hi, const int32_t* ng, void*);
    int (*apply_pec_b)(const wxa_field_view*, const int32_t* dom_lo, const int32_t* dom_hi, const int32_t* pec_lo,
                       const int32_t* pec_hi, const int32_t* ng, void*);
    int (*apply_pec_j)(const wxa_field_view*, const int32_t* dom_lo, const int32_t* dom_hi, const int32_t* pec_lo,
                       const int32_t* pec_hi, void*);
    int (*apply_pec_rho)(const wxa_field_view*, const int32_t* dom_lo, const int32_t* dom_hi, const int32_t* pec_lo,
                         const int32_t* pec_hi, void*);
    // moving window field shift (scratch: a second array of the same shape) and the laser antenna push
    int (*shift_field_window)(const wxa_field_view*, double* tmp, int32_t dir, int32_t num_shift, const int* periodic,
                              void*);
    int (*laser_push)(const wxa_particle_view*, const wxa_laser_push_params*, double t, double dt, void*);
    int (*filter_bilinear)(const wxa_field_view*, const wxa_field_view*, void*);
    // Filter::DoFilter with any half stencils (the NCI corrector: lengths 1, 1, 5); optional
    // BackTransformParticleFunctor's selection + transform (wxa_btd_select_particles); optional
    int (*btd_select_particles)(const wxa_particle_view*, const double* const old6[6], double z_boost, double z_boost_old,
                                double t_boost, double dt, double t_lab, double gamma_boost, double* out, int64_t capacity,
                                int64_t* n_selected, void* stream) = nullptr;
    // the reductions of the reduced diagnostics (wxa_reduce_field, wxa_reduce_particles); optional
    int (*reduce_field)(const wxa_field_view*, const int32_t* lo, const int32_t* hi, double* sum_sq, double* max_abs,
                        void* stream) = nullptr;
    int (*reduce_particles)(const wxa_particle_view*, double mass, int32_t photon, double* out6, void* stream) = nullptr;
    int (*filter_stencil)(const wxa_field_view*, const wxa_field_view*, const double* s0, int32_t n0, const double* s1,
                          int32_t n1, const double* s2, int32_t n2, void*) = nullptr;
    int (*fill_boundary_periodic)(const wxa_field_view*, const int*, const int*, void*);
    int (*sync_nodal_periodic)(const wxa_field_view*, const int*, void*);
    int (*sum_boundary_periodic)(const wxa_field_view*, const int*, const int*, void*);
    // optional (may be null: the callers loop over the fields): several fields per launch
    int (*fill_boundary_periodic_multi)(const wxa_field_view*, int32_t, const int*, const int*, void*) = nullptr;
    int (*sum_boundary_periodic_multi)(const wxa_field_view*, int32_t, const int*, const int*, void*) = nullptr;
    int (*pack_box)(const wxa_field_view*, const int32_t*, const int32_t*, double*, void*);
    int (*unpack_box)(const wxa_field_view*, const int32_t*, const int32_t*, const double*, int, void*);
    // optional: the same with float on the wire (warpx.do_single_precision_comms)
    int (*pack_box_f32)(const wxa_field_view*, const int32_t*, const int32_t*, float*, void*) = nullptr;
    int (*unpack_box_f32)(const wxa_field_view*, const int32_t*, const int32_t*, const float*, int, void*) = nullptr;
    int (*field_set_zero)(const wxa_field_view*, void*);
    int (*field_set_zero_multi)(const wxa_field_view*, int32_t, void*) = nullptr;   // optional: several fields per launch
    int (*enforce_periodic)(const wxa_particle_view*, const double*, const double*, const int*, void*);
    // optional: the wrap restricted to the face tiles of the last sort (ws, steps since that sort)
    int (*enforce_periodic_sorted)(const wxa_particle_view*, const double*, const double*, const int*, void* ws, int32_t,
                                   void*) = nullptr;
    int (*sort_particles_by_cell)(const wxa_particle_view*, const wxa_particle_view*, const double*,
                                  const double*, const int32_t*, const int32_t*, void* ws, void*);
    // Stable 3-way partition of a tile along `dim` for Redistribute: dst = [stay | to-minus | to-plus]
    // for positions in [lo,hi) / < lo / >= hi; counts (host, 3 entries) valid on return.
    int (*partition_particles)(const wxa_particle_view*, const wxa_particle_view*, int dim, double lo,
                               double hi, int64_t* counts, void* ws, void*);
    // Redistribute without moving the tile (include/warpx_amd.h): wrap + leaver lists, pack + retire,
    // and the live count of the last sort
    int (*wrap_and_classify)(const wxa_particle_view*, int64_t first, int64_t count, const double* prob_lo,
                             const double* prob_hi, const int* periodic, const double* brick_lo,
                             const double* brick_hi, const int* split, int32_t* lists, int64_t capacity,
                             int64_t* counts, void* ws, void*);
    int (*wrap_and_classify_dest)(const wxa_particle_view*, int64_t first, int64_t count, const double* prob_lo,
                                  const double* prob_hi, const int* periodic, const double* brick_lo,
                                  const double* brick_hi, const int* split, int32_t* lists, int64_t capacity,
                                  int64_t* counts27, void* ws, void*);
    int (*pack_leavers)(const wxa_particle_view*, const int32_t* list, int64_t n, void* msg, int64_t row_len,
                        int64_t offset, int retire, const double* brick_lo, const double* brick_hi, void*);
    int (*sort_live_count)(void* ws, int64_t* n, void*);
    // optional: the cell sort folded into PushPX (wxa_push_sort_begin / _end / _pending, include/warpx_amd.h); without
    // them the container sorts with sort_particles_by_cell
    int (*push_sort_begin)(void* ws, int32_t mode, const wxa_particle_view* p, const wxa_particle_view* dst, const double* plo,
                           const double* dinv, const int32_t* cell_lo, const int32_t* ncell, const int32_t* wrap,
                           int32_t check_retired, double predict_dt, void*) = nullptr;
    int (*push_sort_end)(void* ws, int32_t read_live, int64_t* live, int64_t* appended, void*) = nullptr;
    int (*push_sort_pending)(const void* ws, const wxa_particle_view* p) = nullptr;
    // WarpXParticleContainer::ApplyBoundaryConditions (reflecting / absorbing walls); *n_lost valid on return
    int (*apply_particle_boundaries)(const wxa_particle_view*, const double* prob_lo, const double* prob_hi,
                                     const int32_t* bc_lo, const int32_t* bc_hi, int64_t* n_lost, void* ws, void*);
    // ---- workspace ----
    int (*workspace_create)(void** ws);
    void (*workspace_destroy)(void* ws);
    // ---- memory / streams ----
    void* (*dmalloc)(size_t bytes);
    void (*dfree)(void* p);
    int (*memset_async)(void* p, int value, size_t bytes, void* stream);
    int (*memcpy_async)(void* dst, const void* src, size_t bytes, void* stream);  // device <-> device
    int (*memcpy_h2d)(void* dst, const void* src, size_t bytes);
    int (*memcpy_d2h)(void* dst, const void* src, size_t bytes);
    int (*stream_sync)(void* stream);
    // a second stream and cross-stream ordering for the overlapped halo exchange (may be null: no overlap)
    void* (*stream_create)();
    void (*stream_destroy)(void* stream);
    void (*stream_wait_event)(void* stream, void* ev);
    // ---- timing (may be null) ----
    void* (*event_create)();
    void (*event_destroy)(void* ev);
    void (*event_record)(void* ev, void* stream);
    float (*event_elapsed_ms)(void* start, void* stop);  // synchronises on stop
};

}  // namespace wxa::host
#endif
