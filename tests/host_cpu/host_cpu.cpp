// TEST INFRASTRUCTURE ONLY.  Compiles the product's C++17 host layer
// (warpx_amd/csrc/host/*.hpp: MultiFab/MultiFabRegister/WarpXParticleContainer/
// FiniteDifferenceSolver/BrickComm/WarpX) against the CPU oracle's entry points, so that the
// multi-brick exchange and migration logic can be exercised with torch.distributed/gloo on a
// machine without a GPU.  Exports `hst_sim_*`.  Never linked into libwarpx_amd.so.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../warpx_amd/csrc/host/WarpXInputs.hpp"

extern "C" {
int orc_evolve_b(const wxa_field_view*, const wxa_field_view*, double, const double*, void*);
int orc_evolve_e(const wxa_field_view*, const wxa_field_view*, const wxa_field_view*, double, const double*, void*);
void orc_ckc_stencil_coefficients(const double*, double*, double*, double*);
double orc_ckc_max_dt(const double*);
int orc_evolve_b_ckc(const wxa_field_view*, const wxa_field_view*, double, const double*, const double*, const double*,
                     void*);
int orc_gather_push(const wxa_particle_view*, const wxa_field_view*, const wxa_field_view*, const wxa_grid_geom*,
                    double, double, double, int, int, int, void*);
int orc_push_p(const wxa_particle_view*, const wxa_field_view*, const wxa_field_view*, const wxa_grid_geom*,
               double, double, double, int, int, int, void*);
int orc_gather_push_lens(const wxa_particle_view*, const wxa_field_view*, const wxa_field_view*, const wxa_grid_geom*,
                         double, double, double, int, int, int, int, const double*, const wxa_repeated_plasma_lens*, double);
int orc_deposit_current(const wxa_particle_view*, const wxa_field_view*, const wxa_grid_geom*, double, double,
                        double, int, int, void*, void*);
int orc_filter_bilinear(const wxa_field_view*, const wxa_field_view*, void*);
int orc_btd_select_particles(const wxa_particle_view*, const double* const[6], double, double, double, double, double, double,
                             double*, int64_t, int64_t*, void*);
int orc_reduce_field(const wxa_field_view*, const int32_t*, const int32_t*, double*, double*, void*);
int orc_reduce_particles(const wxa_particle_view*, double, int32_t, double*, void*);
int orc_filter_stencil(const wxa_field_view*, const wxa_field_view*, const double*, int32_t, const double*, int32_t,
                       const double*, int32_t, void*);
int orc_fill_boundary_periodic(const wxa_field_view*, const int*, const int*, void*);
int orc_sync_nodal_periodic(const wxa_field_view*, const int*, void*);
int orc_sum_boundary_periodic(const wxa_field_view*, const int*, const int*, void*);
int orc_pack_box(const wxa_field_view*, const int32_t*, const int32_t*, double*, void*);
int orc_unpack_box(const wxa_field_view*, const int32_t*, const int32_t*, const double*, int, void*);
int orc_pack_box_f32(const wxa_field_view*, const int32_t*, const int32_t*, float*, void*);
int orc_unpack_box_f32(const wxa_field_view*, const int32_t*, const int32_t*, const float*, int, void*);
int orc_field_set_zero(const wxa_field_view*, void*);
int orc_enforce_periodic(const wxa_particle_view*, const double*, const double*, const int*, void*);
int orc_sort_particles_by_cell(const wxa_particle_view*, const wxa_particle_view*, const double*, const double*,
                               const int32_t*, const int32_t*, void*, void*);
int orc_partition_particles(const wxa_particle_view*, const wxa_particle_view*, int, double, double, int64_t*,
                            void*, void*);
int orc_wrap_and_classify(const wxa_particle_view*, int64_t, int64_t, const double*, const double*, const int*,
                          const double*, const double*, const int*, int32_t*, int64_t, int64_t*, void*, void*);
int orc_wrap_and_classify_dest(const wxa_particle_view*, int64_t, int64_t, const double*, const double*, const int*,
                               const double*, const double*, const int*, int32_t*, int64_t, int64_t*, void*, void*);
int orc_pack_leavers(const wxa_particle_view*, const int32_t*, int64_t, void*, int64_t, int64_t, int, const double*,
                     const double*, void*);
int orc_sort_live_count(void*, int64_t*, void*);
int orc_apply_pec_e(const wxa_field_view*, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const int32_t*,
                    void*);
int orc_apply_pec_b(const wxa_field_view*, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const int32_t*,
                    void*);
int orc_evolve_b_guard_layer(const wxa_field_view*, const wxa_field_view*, double, const double*, const int32_t*, void*);
int orc_add_plasma(const wxa_particle_view*, const wxa_plasma_injector*, const double*, const int32_t*, const double*,
                   const double*, const double*, const wxa_injected_momentum*, int64_t*, void*, void*);
int orc_apply_pec_j(const wxa_field_view*, const int32_t*, const int32_t*, const int32_t*, const int32_t*, void*);
int orc_apply_pec_rho(const wxa_field_view*, const int32_t*, const int32_t*, const int32_t*, const int32_t*, void*);
int orc_deposit_charge(const wxa_particle_view*, const wxa_field_view*, const wxa_grid_geom*, double, int, void*);
int orc_shift_field_window(const wxa_field_view*, double*, int32_t, int32_t, const int*, void*);
int orc_laser_push(const wxa_particle_view*, const wxa_laser_push_params*, double, double, void*);
int orc_apply_particle_boundaries(const wxa_particle_view*, const double*, const double*, const int32_t*, const int32_t*,
                                  int64_t*, void*, void*);
}

namespace {

using wxa::host::Backend;

// the container's workspace on this backend: the live count of the last sort (orc_sort_particles_by_cell writes it
// to the first 8 bytes) followed by the constant external fields
struct CpuWorkspace {
    int64_t live;
    double ext[6];
    wxa_repeated_plasma_lens lens;       // arrays point into lens_tab
    std::vector<double>* lens_tab;
    double time;
};
int ws_create(void** ws) {
    auto* w = static_cast<CpuWorkspace*>(std::calloc(1, sizeof(CpuWorkspace)));
    w->lens_tab = new std::vector<double>();
    w->lens.gamma_boost = 1.0;
    *ws = w;
    return 0;
}
int ws_set_ext(void* ws, const double* E, const double* B) {
    double* e = static_cast<CpuWorkspace*>(ws)->ext;
    for (int d = 0; d < 3; ++d) { e[d] = E[d]; e[3 + d] = B[d]; }
    return 0;
}
int ws_set_lens(void* ws, const wxa_repeated_plasma_lens* lens) {
    auto* w = static_cast<CpuWorkspace*>(ws);
    const int n = lens->n_lenses;
    w->lens_tab->assign((size_t)4 * n, 0.0);
    for (int i = 0; i < n; ++i) {
        (*w->lens_tab)[i] = lens->starts[i]; (*w->lens_tab)[n + i] = lens->lengths[i];
        (*w->lens_tab)[2 * n + i] = lens->strengths_E[i]; (*w->lens_tab)[3 * n + i] = lens->strengths_B[i];
    }
    w->lens = *lens;
    const double* t = w->lens_tab->data();
    w->lens.starts = t; w->lens.lengths = t + n; w->lens.strengths_E = t + 2 * n; w->lens.strengths_B = t + 3 * n;
    return 0;
}
int ws_set_time(void* ws, double t) { static_cast<CpuWorkspace*>(ws)->time = t; return 0; }
int ws_gather_push(const wxa_particle_view* p, const wxa_field_view* E, const wxa_field_view* B, const wxa_grid_geom* g,
                   double q, double m, double dt, int o, int ga, int pu, int move, void* ws) {
    auto* w = static_cast<CpuWorkspace*>(ws);
    return orc_gather_push_lens(p, E, B, g, q, m, dt, o, ga, pu, move, w->ext, &w->lens, w->time);
}
void ws_destroy(void* ws) { delete static_cast<CpuWorkspace*>(ws)->lens_tab; std::free(ws); }
void* h_malloc(size_t n) { return std::malloc(n ? n : 8); }
void h_free(void* p) { std::free(p); }
int h_memset(void* p, int v, size_t n, void*) { std::memset(p, v, n); return 0; }
int h_memcpy(void* d, const void* s, size_t n, void*) { if (n) std::memmove(d, s, n); return 0; }
int h_memcpy2(void* d, const void* s, size_t n) { if (n) std::memmove(d, s, n); return 0; }
int h_sync(void*) { return 0; }

const Backend* cpu_backend() {
    static const Backend be = [] {
        Backend b{};
        b.name = "cpu-oracle (tests only)";
        b.evolve_b = orc_evolve_b; b.evolve_e = orc_evolve_e;
        b.ckc_stencil_coefficients = orc_ckc_stencil_coefficients; b.ckc_max_dt = orc_ckc_max_dt;
        b.evolve_b_ckc = orc_evolve_b_ckc;
        b.gather_push = [](const wxa_particle_view* p, const wxa_field_view* E, const wxa_field_view* B,
                           const wxa_grid_geom* g, double q, double m, double dt, int o, int ga, int pu, int move,
                           void* ws, void*) -> int {
            return ws_gather_push(p, E, B, g, q, m, dt, o, ga, pu, move, ws); };
        b.ws_set_external_eb = ws_set_ext;
        b.ws_set_repeated_plasma_lens = ws_set_lens;
        b.ws_set_time = ws_set_time;
        // no tiles on this backend: the interior part is empty, the rest is everything (a valid split)
        b.gather_push_part = [](const wxa_particle_view* p, const wxa_field_view* E, const wxa_field_view* B,
                                const wxa_grid_geom* g, double q, double m, double dt, int o, int ga, int pu, void* ws,
                                int part, void*) -> int {
            return part == WXA_PART_INTERIOR ? 0 : ws_gather_push(p, E, B, g, q, m, dt, o, ga, pu, 1, ws); };
        b.add_plasma = orc_add_plasma;
        b.deposit_current = orc_deposit_current;
        b.filter_bilinear = orc_filter_bilinear;
        b.filter_stencil = orc_filter_stencil;
        b.btd_select_particles = orc_btd_select_particles;
        b.reduce_field = orc_reduce_field;
        b.reduce_particles = orc_reduce_particles;
        b.fill_boundary_periodic = orc_fill_boundary_periodic;
        b.sync_nodal_periodic = orc_sync_nodal_periodic;
        b.sum_boundary_periodic = orc_sum_boundary_periodic;
        b.pack_box = orc_pack_box; b.unpack_box = orc_unpack_box;
        b.pack_box_f32 = orc_pack_box_f32; b.unpack_box_f32 = orc_unpack_box_f32;
        b.field_set_zero = orc_field_set_zero;
        b.enforce_periodic = orc_enforce_periodic;
        b.sort_particles_by_cell = orc_sort_particles_by_cell;
        b.partition_particles = orc_partition_particles;
        b.wrap_and_classify = orc_wrap_and_classify;
        b.wrap_and_classify_dest = orc_wrap_and_classify_dest;
        b.pack_leavers = orc_pack_leavers;
        b.sort_live_count = orc_sort_live_count;
        b.apply_pec_e = orc_apply_pec_e;
        b.apply_pec_b = orc_apply_pec_b;
        b.apply_pec_j = orc_apply_pec_j;
        b.evolve_b_guard_layer = orc_evolve_b_guard_layer;
        // streams are not a thing here: the "second stream" is a tag, ordering is program order
        b.stream_create = []() -> void* { static int tag; return &tag; };
        b.stream_destroy = [](void*) {};
        b.stream_wait_event = [](void*, void*) {};
        b.apply_pec_rho = orc_apply_pec_rho;
        b.deposit_charge = orc_deposit_charge;
        b.apply_particle_boundaries = orc_apply_particle_boundaries;
        b.shift_field_window = orc_shift_field_window;
        b.laser_push = orc_laser_push;
        b.workspace_create = ws_create; b.workspace_destroy = ws_destroy;
        b.dmalloc = h_malloc; b.dfree = h_free;
        b.memset_async = h_memset; b.memcpy_async = h_memcpy;
        b.memcpy_h2d = h_memcpy2; b.memcpy_d2h = h_memcpy2;
        b.stream_sync = h_sync;
        return b;
    }();
    return &be;
}

std::string g_last_error;
void set_err(const char* msg) { g_last_error = msg; std::fprintf(stderr, "[host_cpu] %s\n", msg); }

}  // namespace

struct hst_sim {};
extern "C" const char* hst_last_error(void) { return g_last_error.c_str(); }
WXA_SIM_CAPI(hst_, int, hst_sim, cpu_backend, set_err)
WXA_INPUTS_CAPI(hst_, int, hst_sim, cpu_backend, set_err)
