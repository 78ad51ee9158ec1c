// Product instantiation of the host layer: HIP backend table + the wxa_sim_* C API.
#include <hip/hip_runtime.h>

#include "../common.hpp"
#include "WarpXInputs.hpp"

namespace {

using wxa::host::Backend;

void* hip_dmalloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) {
        wxa::set_last_error("hipMalloc of %zu bytes failed", bytes);
        return nullptr;
    }
    return p;
}
void hip_dfree(void* p) { if (p) (void)hipFree(p); }
int hip_memset_async(void* p, int v, size_t n, void* st) {
    return hipMemsetAsync(p, v, n, (hipStream_t)st) == hipSuccess ? 0 : WXA_ERR_HIP;
}
int hip_memcpy_async(void* d, const void* s, size_t n, void* st) {
    if (n == 0) return 0;
    return hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, (hipStream_t)st) == hipSuccess ? 0 : WXA_ERR_HIP;
}
int hip_memcpy_h2d(void* d, const void* s, size_t n) {
    return hipMemcpy(d, s, n, hipMemcpyHostToDevice) == hipSuccess ? 0 : WXA_ERR_HIP;
}
int hip_memcpy_d2h(void* d, const void* s, size_t n) {
    return hipMemcpy(d, s, n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : WXA_ERR_HIP;
}
int hip_stream_sync(void* st) { return hipStreamSynchronize((hipStream_t)st) == hipSuccess ? 0 : WXA_ERR_HIP; }
void* hip_event_create() {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void* hip_stream_create() {
    hipStream_t st = nullptr;   // non-blocking: no implicit ordering with the null stream the kernels run on
    return hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess ? (void*)st : nullptr;
}
void hip_stream_destroy(void* st) { if (st) (void)hipStreamDestroy((hipStream_t)st); }
void hip_stream_wait_event(void* st, void* e) { (void)hipStreamWaitEvent((hipStream_t)st, (hipEvent_t)e, 0); }
void hip_event_destroy(void* e) { if (e) (void)hipEventDestroy((hipEvent_t)e); }
void hip_event_record(void* e, void* st) { (void)hipEventRecord((hipEvent_t)e, (hipStream_t)st); }
float hip_event_elapsed(void* a, void* b) {
    float ms = 0.f;
    (void)hipEventSynchronize((hipEvent_t)b);
    (void)hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b);
    return ms;
}

int ws_create(void** ws) { return wxa_workspace_create(reinterpret_cast<wxa_workspace**>(ws)); }
void ws_destroy(void* ws) { wxa_workspace_destroy(static_cast<wxa_workspace*>(ws)); }
int k_deposit(const wxa_particle_view* p, const wxa_field_view* J, const wxa_grid_geom* g, double q, double dt,
              double rel, int order, int algo, void* ws, void* st) {
    return wxa_deposit_current(p, J, g, q, dt, rel, order, algo, static_cast<wxa_workspace*>(ws), st);
}
int k_sort(const wxa_particle_view* s, const wxa_particle_view* d, const double* plo, const double* dinv,
           const int32_t* lo, const int32_t* nc, void* ws, void* st) {
    return wxa_sort_particles_by_cell(s, d, plo, dinv, lo, nc, static_cast<wxa_workspace*>(ws), st);
}
int k_partition(const wxa_particle_view* s, const wxa_particle_view* d, int dim, double lo, double hi,
                int64_t* counts, void* ws, void* st) {
    return wxa_partition_particles(s, d, dim, lo, hi, counts, static_cast<wxa_workspace*>(ws), st);
}

const Backend* hip_backend() {
    static const Backend be = [] {
        Backend b{};
        b.name = "hip-gfx950";
        b.evolve_b = [](const wxa_field_view* E, const wxa_field_view* B, double dt, const double* di, void* st) -> int {
            return wxa_evolve_b(E, B, dt, di, st); };
        b.evolve_e = [](const wxa_field_view* E, const wxa_field_view* B, const wxa_field_view* J, double dt,
                        const double* di, void* st) -> int { return wxa_evolve_e(E, B, J, dt, di, st); };
        b.ws_set_external_eb = [](void* ws, const double* E, const double* B) -> int {
            return wxa_workspace_set_external_particle_fields(static_cast<wxa_workspace*>(ws), E, B); };
        b.ws_set_repeated_plasma_lens = [](void* ws, const wxa_repeated_plasma_lens* lens) -> int {
            return wxa_workspace_set_repeated_plasma_lens(static_cast<wxa_workspace*>(ws), lens); };
        b.ws_set_time = [](void* ws, double t) -> int { return wxa_workspace_set_time(static_cast<wxa_workspace*>(ws), t); };
        b.ws_set_deposit_accumulator = [](void* ws, int32_t acc) -> int {
            return wxa_workspace_set_deposit_accumulator(static_cast<wxa_workspace*>(ws), acc); };
        b.ws_set_streaming_plasma = [](void* ws, int32_t on) -> int {
            return wxa_workspace_set_streaming_plasma(static_cast<wxa_workspace*>(ws), on); };
        b.ckc_stencil_coefficients = wxa_ckc_stencil_coefficients;
        b.ckc_max_dt = wxa_ckc_max_dt;
        b.evolve_b_ckc = [](const wxa_field_view* E, const wxa_field_view* B, double dt, const double* cx, const double* cy,
                            const double* cz, void* st) -> int { return wxa_evolve_b_ckc(E, B, dt, cx, cy, cz, st); };
        b.gather_push = [](const wxa_particle_view* p, const wxa_field_view* E, const wxa_field_view* B,
                           const wxa_grid_geom* g, double q, double m, double dt, int o, int ga, int pu, int move,
                           void* ws, void* st) -> int {
            return wxa_gather_push_ws(p, E, B, g, q, m, dt, o, ga, pu, move, static_cast<wxa_workspace*>(ws), st); };
        b.gather_push_part = [](const wxa_particle_view* p, const wxa_field_view* E, const wxa_field_view* B,
                                const wxa_grid_geom* g, double q, double m, double dt, int o, int ga, int pu, void* ws,
                                int part, void* st) -> int {
            return wxa_gather_push_part(p, E, B, g, q, m, dt, o, ga, pu, static_cast<wxa_workspace*>(ws), part, st); };
        b.add_plasma = [](const wxa_particle_view* dst, const wxa_plasma_injector* inj, const double* corner,
                          const int32_t* nc, const double* dx, const double* blo, const double* bhi,
                          const wxa_injected_momentum* u, int64_t* n, void* ws, void* st) -> int {
            return wxa_add_plasma(dst, inj, corner, nc, dx, blo, bhi, u, n, static_cast<wxa_workspace*>(ws), st); };
        b.deposit_current = k_deposit;
        b.filter_bilinear = [](const wxa_field_view* s, const wxa_field_view* d, void* st) -> int {
            return wxa_filter_bilinear(s, d, st); };
        b.btd_select_particles = [](const wxa_particle_view* p, const double* const o[6], double zb, double zbo, double tb,
                                    double dt, double tl, double g, double* out, int64_t cap, int64_t* n, void* st) -> int {
            return wxa_btd_select_particles(p, o, zb, zbo, tb, dt, tl, g, out, cap, n, st); };
        b.reduce_field = [](const wxa_field_view* f, const int32_t* lo, const int32_t* hi, double* ss, double* mx,
                            void* st) -> int { return wxa_reduce_field(f, lo, hi, ss, mx, st); };
        b.reduce_particles = [](const wxa_particle_view* p, double m, int32_t photon, double* out, void* st) -> int {
            return wxa_reduce_particles(p, m, photon, out, st); };
        b.filter_stencil = [](const wxa_field_view* s, const wxa_field_view* d, const double* s0, int32_t n0,
                              const double* s1, int32_t n1, const double* s2, int32_t n2, void* st) -> int {
            return wxa_filter_stencil(s, d, s0, n0, s1, n1, s2, n2, st); };
        b.fill_boundary_periodic = [](const wxa_field_view* f, const int* ng, const int* per, void* st) -> int {
            return wxa_fill_boundary_periodic(f, ng, per, st); };
        b.sync_nodal_periodic = [](const wxa_field_view* f, const int* per, void* st) -> int {
            return wxa_sync_nodal_periodic(f, per, st); };
        b.sum_boundary_periodic = [](const wxa_field_view* f, const int* ng, const int* per, void* st) -> int {
            return wxa_sum_boundary_periodic(f, ng, per, st); };
        b.fill_boundary_periodic_multi = [](const wxa_field_view* f, int32_t nf, const int* ng, const int* per, void* st) -> int {
            return wxa_fill_boundary_periodic_multi(f, nf, ng, per, st); };
        b.sum_boundary_periodic_multi = [](const wxa_field_view* f, int32_t nf, const int* ng, const int* per, void* st) -> int {
            return wxa_sum_boundary_periodic_multi(f, nf, ng, per, st); };
        b.pack_box = [](const wxa_field_view* f, const int32_t* lo, const int32_t* hi, double* buf, void* st) -> int {
            return wxa_pack_box(f, lo, hi, buf, st); };
        b.unpack_box = [](const wxa_field_view* f, const int32_t* lo, const int32_t* hi, const double* buf, int mode,
                          void* st) -> int { return wxa_unpack_box(f, lo, hi, buf, mode, st); };
        b.pack_box_f32 = [](const wxa_field_view* f, const int32_t* lo, const int32_t* hi, float* buf, void* st) -> int {
            return wxa_pack_box_f32(f, lo, hi, buf, st); };
        b.unpack_box_f32 = [](const wxa_field_view* f, const int32_t* lo, const int32_t* hi, const float* buf, int mode,
                              void* st) -> int { return wxa_unpack_box_f32(f, lo, hi, buf, mode, st); };
        b.field_set_zero = [](const wxa_field_view* f, void* st) -> int { return wxa_field_set_zero(f, st); };
        b.field_set_zero_multi = [](const wxa_field_view* f, int32_t nf, void* st) -> int { return wxa_field_set_zero_multi(f, nf, st); };
        b.enforce_periodic = [](const wxa_particle_view* p, const double* lo, const double* hi, const int* per,
                                void* st) -> int { return wxa_enforce_periodic(p, lo, hi, per, st); };
        b.enforce_periodic_sorted = [](const wxa_particle_view* p, const double* lo, const double* hi, const int* per,
                                       void* ws, int32_t steps, void* st) -> int {
            return wxa_enforce_periodic_sorted(p, lo, hi, per, static_cast<wxa_workspace*>(ws), steps, st); };
        b.sort_particles_by_cell = k_sort;
        b.partition_particles = k_partition;
        b.wrap_and_classify = [](const wxa_particle_view* p, int64_t first, int64_t count, const double* plo,
                                 const double* phi, const int* per, const double* blo, const double* bhi,
                                 const int* split, int32_t* lists, int64_t cap, int64_t* counts, void* ws,
                                 void* st) -> int {
            return wxa_wrap_and_classify(p, first, count, plo, phi, per, blo, bhi, split, lists, cap, counts,
                                         static_cast<wxa_workspace*>(ws), st); };
        b.wrap_and_classify_dest = [](const wxa_particle_view* p, int64_t first, int64_t count, const double* plo,
                                      const double* phi, const int* per, const double* blo, const double* bhi,
                                      const int* split, int32_t* lists, int64_t cap, int64_t* counts, void* ws,
                                      void* st) -> int {
            return wxa_wrap_and_classify_dest(p, first, count, plo, phi, per, blo, bhi, split, lists, cap, counts,
                                              static_cast<wxa_workspace*>(ws), st); };
        b.pack_leavers = [](const wxa_particle_view* p, const int32_t* list, int64_t n, void* msg, int64_t row_len,
                            int64_t offset, int retire, const double* blo, const double* bhi, void* st) -> int {
            return wxa_pack_leavers(p, list, n, msg, row_len, offset, retire, blo, bhi, st); };
        b.apply_pec_e = [](const wxa_field_view* E, const int32_t* dlo, const int32_t* dhi, const int32_t* plo,
                           const int32_t* phi, const int32_t* ng, void* st) -> int {
            return wxa_apply_pec_e(E, dlo, dhi, plo, phi, ng, st); };
        b.apply_pec_b = [](const wxa_field_view* B, const int32_t* dlo, const int32_t* dhi, const int32_t* plo,
                           const int32_t* phi, const int32_t* ng, void* st) -> int {
            return wxa_apply_pec_b(B, dlo, dhi, plo, phi, ng, st); };
        b.apply_pec_j = [](const wxa_field_view* J, const int32_t* dlo, const int32_t* dhi, const int32_t* plo,
                           const int32_t* phi, void* st) -> int { return wxa_apply_pec_j(J, dlo, dhi, plo, phi, st); };
        b.apply_pec_rho = [](const wxa_field_view* r, const int32_t* dlo, const int32_t* dhi, const int32_t* plo,
                             const int32_t* phi, void* st) -> int { return wxa_apply_pec_rho(r, dlo, dhi, plo, phi, st); };
        b.deposit_charge = [](const wxa_particle_view* p, const wxa_field_view* r, const wxa_grid_geom* g, double q,
                              int order, void* st) -> int { return wxa_deposit_charge(p, r, g, q, order, st); };
        b.shift_field_window = [](const wxa_field_view* f, double* tmp, int32_t dir, int32_t n, const int* per,
                                  void* st) -> int { return wxa_shift_field_window(f, tmp, dir, n, per, st); };
        b.laser_push = [](const wxa_particle_view* p, const wxa_laser_push_params* par, double t, double dt,
                          void* st) -> int { return wxa_laser_push(p, par, t, dt, st); };
        b.apply_particle_boundaries = [](const wxa_particle_view* p, const double* plo, const double* phi,
                                         const int32_t* blo, const int32_t* bhi, int64_t* n_lost, void* ws,
                                         void* st) -> int {
            return wxa_apply_particle_boundaries(p, plo, phi, blo, bhi, n_lost, static_cast<wxa_workspace*>(ws), st); };
        b.sort_live_count = [](void* ws, int64_t* n, void* st) -> int {
            return wxa_sort_live_count(static_cast<wxa_workspace*>(ws), n, st); };
        b.push_sort_begin = [](void* ws, int32_t mode, const wxa_particle_view* p, const wxa_particle_view* dst,
                               const double* plo, const double* dinv, const int32_t* lo, const int32_t* nc,
                               const int32_t* wrap, int32_t check_retired, double predict_dt, void* st) -> int {
            return wxa_push_sort_begin(static_cast<wxa_workspace*>(ws), mode, p, dst, plo, dinv, lo, nc, wrap, check_retired,
                                       predict_dt, st); };
        b.push_sort_end = [](void* ws, int32_t read_live, int64_t* live, int64_t* appended, void* st) -> int {
            return wxa_push_sort_end(static_cast<wxa_workspace*>(ws), read_live, live, appended, st); };
        b.push_sort_pending = [](const void* ws, const wxa_particle_view* p) -> int {
            return wxa_push_sort_pending(static_cast<const wxa_workspace*>(ws), p); };
        b.workspace_create = ws_create;
        b.workspace_destroy = ws_destroy;
        b.dmalloc = hip_dmalloc;
        b.dfree = hip_dfree;
        b.memset_async = hip_memset_async;
        b.memcpy_async = hip_memcpy_async;
        b.memcpy_h2d = hip_memcpy_h2d;
        b.memcpy_d2h = hip_memcpy_d2h;
        b.stream_sync = hip_stream_sync;
        b.stream_create = hip_stream_create;
        b.stream_destroy = hip_stream_destroy;
        b.stream_wait_event = hip_stream_wait_event;
        b.evolve_b_guard_layer = [](const wxa_field_view* E, const wxa_field_view* B, double dt, const double* dinv,
                                    const int32_t* grow, void* st) -> int {
            return wxa_evolve_b_guard_layer(E, B, dt, dinv, grow, st); };
        b.event_create = hip_event_create;
        b.event_destroy = hip_event_destroy;
        b.event_record = hip_event_record;
        b.event_elapsed_ms = hip_event_elapsed;
        return b;
    }();
    return &be;
}

void set_err(const char* msg) { wxa::set_last_error("%s", msg); }

}  // namespace

// NCIGodfreyFilter::ComputeStencils through the host layer's class (host function)
extern "C" wxa_status wxa_nci_godfrey_stencil(double cdtodz, int32_t nodal_gather, int32_t coeff_set, double stencil_z[5]) {
    if (!stencil_z || (coeff_set != WXA_NCI_EX_EY_BZ && coeff_set != WXA_NCI_BX_BY_EZ)) {
        wxa::set_last_error("wxa_nci_godfrey_stencil: bad argument");
        return WXA_ERR_INVALID_ARG;
    }
    wxa::host::NCIGodfreyFilter f(coeff_set == WXA_NCI_EX_EY_BZ ? wxa::host::godfrey_coeff_set::Ex_Ey_Bz
                                                                  : wxa::host::godfrey_coeff_set::Bx_By_Ez,
                                  cdtodz, nodal_gather != 0);
    f.ComputeStencils();
    for (int i = 0; i < 5; ++i) stencil_z[i] = f.stencil_z[i];
    return WXA_OK;
}

struct wxa_sim {};
WXA_SIM_CAPI(wxa_, wxa_status, wxa_sim, hip_backend, set_err)
WXA_INPUTS_CAPI(wxa_, wxa_status, wxa_sim, hip_backend, set_err)
