// Step-level C API (`<prefix>sim_*`, include/warpx_amd.h) over class WarpX.
// WXA_SIM_CAPI(prefix, backend_getter) instantiates the entry points; the product uses
// prefix wxa_ with the HIP backend (warpx_host.hip).
#ifndef WXA_HOST_SIM_CAPI_HPP_
#define WXA_HOST_SIM_CAPI_HPP_

#include <cstring>

#include "WarpX.hpp"

namespace wxa::host {

struct SimHandle {
    std::unique_ptr<WarpX> warpx;
    std::string error;
    // set when the simulation was built from an inputs file (WarpXInputs.hpp)
    int max_step = -1;
    std::vector<std::string> species_names;
    std::shared_ptr<void> multi_diags;   // MultiDiagnostics of FullDiagnostics.hpp (declared after the plotfile writer)
};

inline int sim_create(const Backend* be, const wxa_sim_config* cfg, const wxa_comm* comm, SimHandle** out,
                      std::string& err) {
    if (!cfg || !out) return WXA_ERR_INVALID_ARG;
    try {
        auto h = std::make_unique<SimHandle>();
        h->warpx = std::make_unique<WarpX>(be, *cfg, comm);
        *out = h.release();
        return WXA_OK;
    } catch (const std::exception& e) {
        err = e.what();
        return WXA_ERR_INVALID_ARG;
    }
}

inline int sim_add_species(SimHandle* h, double charge, double mass, const wxa_particle_view* init, int32_t* id) {
    if (!h || !init || init->np < 0) return WXA_ERR_INVALID_ARG;
    try {
        WarpX& w = *h->warpx;
        if (charge != 0.0 && w.any_reflecting_wall())
            throw std::runtime_error("charged species with a reflecting particle boundary: the current fold at the "
                                     "walls has the image-charge sign of absorbing walls only");
        const int sid = w.GetPartContainer().AddSpecies(charge, mass);
        w.HookBTDSpecies();
        ParticleTile& t = w.GetPartContainer().GetParticleContainer(sid).tile();
        t.resize(init->np);
        const Backend* be = w.context().be;
        const double* src[7] = {init->x, init->y, init->z, init->w, init->ux, init->uy, init->uz};
        if (init->np > 0) {
            for (int c = 0; c < 7; ++c)
                be->memcpy_async(t.comp(c), src[c], sizeof(double) * (size_t)init->np, w.context().stream);
            if (init->idcpu)
                be->memcpy_async(t.idcpu(), init->idcpu, sizeof(uint64_t) * (size_t)init->np, w.context().stream);
            else
                be->memset_async(t.idcpu(), 0, sizeof(uint64_t) * (size_t)init->np, w.context().stream);
            be->stream_sync(w.context().stream);
            // start from a sorted tile so that the very first gather/deposit use the LDS-tile kernels
            if (w.sort_intervals > 0) w.GetPartContainer().GetParticleContainer(sid).SortParticlesByBin(amrex::IntVect(1));
        }
        if (id) *id = sid;
        return WXA_OK;
    } catch (const std::exception& e) {
        h->error = e.what();
        return WXA_ERR_INVALID_ARG;
    }
}

inline int sim_set_moving_window(SimHandle* h, const wxa_moving_window* mw) {
    if (!h || !mw) return WXA_ERR_INVALID_ARG;
    try {
        h->warpx->SetMovingWindow(mw->dir, mw->v);
        return WXA_OK;
    } catch (const std::exception& e) {
        h->error = e.what();
        return WXA_ERR_INVALID_ARG;
    }
}

// particles.E_external_particle / B_external_particle (constant) for one species
inline int sim_set_external_particle_fields(SimHandle* h, int32_t id, const double* E, const double* B) {
    if (!h || !E || !B || id < 0 || id >= h->warpx->GetPartContainer().nSpecies()) return WXA_ERR_INVALID_ARG;
    try {
        h->warpx->GetPartContainer().GetParticleContainer(id).SetExternalParticleFields(E, B);
        return WXA_OK;
    } catch (const std::exception& e) {
        h->error = e.what();
        return WXA_ERR_INVALID_ARG;
    }
}

inline int sim_set_deposit_accumulator(SimHandle* h, int32_t id, int32_t acc) {
    if (!h || id < 0 || id >= h->warpx->GetPartContainer().nSpecies()) return WXA_ERR_INVALID_ARG;
    try {
        h->warpx->GetPartContainer().GetParticleContainer(id).SetDepositAccumulator(acc);
        return WXA_OK;
    } catch (const std::exception& e) {
        h->error = e.what();
        return WXA_ERR_INVALID_ARG;
    }
}

inline int sim_set_radiation_reaction(SimHandle* h, int32_t id, int32_t on) {
    if (!h || id < 0 || id >= h->warpx->GetPartContainer().nSpecies()) return WXA_ERR_INVALID_ARG;
    h->warpx->GetPartContainer().GetParticleContainer(id).SetRadiationReaction(on != 0);
    return WXA_OK;
}

// <species>.injection_style = NUniformPerCell (...): add_initial fills the current domain now
// (PhysicalParticleContainer::InitData -> AddParticles -> AddPlasma), continuous keeps injecting behind a moving window
inline int sim_set_injection(SimHandle* h, int32_t id, const wxa_plasma_injector* inj, int add_initial, int continuous) {
    if (!h || !inj || id < 0 || id >= h->warpx->GetPartContainer().nSpecies()) return WXA_ERR_INVALID_ARG;
    try {
        WarpX& w = *h->warpx;
        auto* pc = dynamic_cast<PhysicalParticleContainer*>(&w.GetPartContainer().GetParticleContainer(id));
        if (!pc) return WXA_ERR_INVALID_ARG;
        pc->SetPlasmaInjector(*inj, continuous != 0);
        if (add_initial) {
            pc->AddPlasma(w.context().prob_lo.data(), w.context().prob_hi.data());
            if (w.sort_intervals > 0 && pc->TotalNumberOfParticles() > 0) pc->SortParticlesByBin(amrex::IntVect(1));
        }
        return WXA_OK;
    } catch (const std::exception& e) {
        h->error = e.what();
        return WXA_ERR_INVALID_ARG;
    }
}

inline int sim_add_laser(SimHandle* h, const wxa_laser_antenna* la) {
    if (!h || !la) return WXA_ERR_INVALID_ARG;
    try {
        h->warpx->GetPartContainer().AddLaser(*la);
        return WXA_OK;
    } catch (const std::exception& e) {
        h->error = e.what();
        return WXA_ERR_INVALID_ARG;
    }
}

inline int sim_evolve(SimHandle* h, int32_t numsteps) {
    if (!h || numsteps < 0) return WXA_ERR_INVALID_ARG;
    try {
        h->warpx->Evolve(numsteps);
        return WXA_OK;
    } catch (const std::exception& e) {
        h->error = e.what();
        return WXA_ERR_HIP;
    }
}

inline int sim_get_field(SimHandle* h, const char* name, wxa_field_view* out) {
    using warpx::fields::FieldType;
    using ablastr::fields::Direction;
    if (!h || !name || !out) return WXA_ERR_INVALID_ARG;
    if (std::strcmp(name, "rho") == 0) {                       // valid after sim_compute_rho
        if (!h->warpx->rho()) return WXA_ERR_INVALID_ARG;
        *out = h->warpx->rho()->view();
        return WXA_OK;
    }
    if (std::strlen(name) != 2) return WXA_ERR_INVALID_ARG;
    const char* comps = "xyz";
    const char* cp = std::strchr(comps, name[1]);
    if (!cp) return WXA_ERR_INVALID_ARG;
    const int d = (int)(cp - comps);
    FieldType ft;
    if (name[0] == 'E') ft = FieldType::Efield_fp;
    else if (name[0] == 'B') ft = FieldType::Bfield_fp;
    else if (name[0] == 'j') ft = FieldType::current_fp;
    else return WXA_ERR_INVALID_ARG;
    *out = h->warpx->fields().get(ft, Direction{d}, 0)->view();
    return WXA_OK;
}

inline int sim_compute_rho(SimHandle* h) {
    if (!h) return WXA_ERR_INVALID_ARG;
    try {
        h->warpx->ComputeRho();
        return WXA_OK;
    } catch (const std::exception& e) {
        h->error = e.what();
        return WXA_ERR_HIP;
    }
}

inline int sim_add_btd(SimHandle* h, int32_t num_snapshots, double dt_snapshots_lab, int32_t buffer_size,
                       int32_t write_species) {
    if (!h) return WXA_ERR_INVALID_ARG;
    try {
        h->warpx->AddBTDiagnostics(num_snapshots, dt_snapshots_lab, buffer_size, write_species != 0);
        return WXA_OK;
    } catch (const std::exception& e) {
        h->error = e.what();
        return WXA_ERR_INVALID_ARG;
    }
}

// n: cells of snapshot i (x, y, z); z_lab: its lab-frame extent along z; t_lab; filled: slices received; full: closed
inline int sim_btd_info(SimHandle* h, int32_t i, int32_t n[3], double z_lab[2], double* t_lab, int32_t* filled,
                        int32_t* full) {
    if (!h || !h->warpx->btd() || i < 0 || i >= h->warpx->btd()->num_snapshots()) return WXA_ERR_INVALID_ARG;
    const auto& s = h->warpx->btd()->snapshot(i);
    if (n) for (int d = 0; d < 3; ++d) n[d] = s.n[d];
    if (z_lab) { z_lab[0] = s.zlo_lab; z_lab[1] = s.zhi_lab; }
    if (t_lab) *t_lab = s.t_lab;
    if (filled) *filled = s.counter;
    if (full) *full = s.full;
    return WXA_OK;
}

// component comp (Ex Ey Ez Bx By Bz jx jy jz rho) of snapshot i into out[k][j][i] (host memory, n[0] n[1] n[2] doubles)
inline int sim_btd_data(SimHandle* h, int32_t i, int32_t comp, double* out) {
    if (!h || !out || !h->warpx->btd() || i < 0 || i >= h->warpx->btd()->num_snapshots() || comp < 0 ||
        comp >= BTDiagnostics::NCOMP)
        return WXA_ERR_INVALID_ARG;
    const auto& s = h->warpx->btd()->snapshot(i);
    const size_t n = (size_t)s.n[0] * s.n[1] * s.n[2];
    if (s.data.size() != (size_t)BTDiagnostics::NCOMP * n) return WXA_ERR_INVALID_ARG;   // flushed to disk, not kept (wxa_sim_btd_set_flush)
    std::memcpy(out, s.data.data() + (size_t)comp * n, sizeof(double) * n);
    return WXA_OK;
}

// back-transformed particles of species `id` in snapshot i: their number, and the rows x y z w ux uy uz (lab frame) into
// out[7][n] (host memory)
inline int sim_btd_num_particles(SimHandle* h, int32_t i, int32_t id, int64_t* n) {
    if (!h || !n || !h->warpx->btd() || i < 0 || i >= h->warpx->btd()->num_snapshots() || id < 0) return WXA_ERR_INVALID_ARG;
    const auto& s = h->warpx->btd()->snapshot(i);
    *n = id < (int32_t)s.particles.size() ? (int64_t)s.particles[(size_t)id][0].size() : 0;
    return WXA_OK;
}
inline int sim_btd_particles(SimHandle* h, int32_t i, int32_t id, double* out) {
    int64_t n = 0;
    const int rc = sim_btd_num_particles(h, i, id, &n);
    if (rc != WXA_OK || !out) return rc != WXA_OK ? rc : WXA_ERR_INVALID_ARG;
    if (n == 0) return WXA_OK;
    const auto& rows = h->warpx->btd()->snapshot(i).particles[(size_t)id];
    for (int c = 0; c < 7; ++c) std::memcpy(out + (size_t)c * (size_t)n, rows[(size_t)c].data(), sizeof(double) * (size_t)n);
    return WXA_OK;
}

// warpx.reduced_diags_names (ReducedDiags.hpp)
inline int sim_add_reduced_diag(SimHandle* h, const char* name, const char* type, const char* intervals, const char* path) {
    if (!h || !name || !*name || !type) return WXA_ERR_INVALID_ARG;
    try {
        h->warpx->reduced_diags.Add(name, type, intervals ? intervals : "", path);
        h->warpx->reduced_diags.SetSpeciesNames(h->species_names);
        return WXA_OK;
    } catch (const std::exception& e) {
        h->error = e.what();
        return WXA_ERR_INVALID_ARG;
    }
}
inline int sim_reduced_diag_data(SimHandle* h, const char* name, int32_t compute_now, double* out, int32_t capacity,
                                 int32_t* n) {
    if (!h || !name || !n || capacity < 0 || (capacity > 0 && !out)) return WXA_ERR_INVALID_ARG;
    try {
        const std::vector<double>& d = h->warpx->reduced_diags.Data(*h->warpx, name, compute_now != 0);
        *n = (int32_t)d.size();
        for (int32_t i = 0; i < std::min(*n, capacity); ++i) out[i] = d[(size_t)i];
        return WXA_OK;
    } catch (const std::exception& e) {
        h->error = e.what();
        return WXA_ERR_INVALID_ARG;
    }
}

inline int sim_get_particles(SimHandle* h, int32_t id, wxa_particle_view* out) {
    if (!h || !out || id < 0 || id >= h->warpx->GetPartContainer().nSpecies()) return WXA_ERR_INVALID_ARG;
    *out = h->warpx->GetPartContainer().GetParticleContainer(id).tile().view();
    return WXA_OK;
}

inline int sim_dry_comm(SimHandle* h, int32_t reps, double ms[4]) {
    if (!h || !ms) return WXA_ERR_INVALID_ARG;
    try {
        h->warpx->DryComm(reps, ms);
        return WXA_OK;
    } catch (const std::exception& e) {
        h->error = e.what();
        return WXA_ERR_HIP;
    }
}

inline int sim_get_timers(SimHandle* h, double ms[8], int64_t counts[8], int reset) {
    if (!h) return WXA_ERR_INVALID_ARG;
    WarpXContext& c = h->warpx->context();
    c.resolve_timers();
    for (int i = 0; i < 8; ++i) { ms[i] = c.ms[i]; counts[i] = c.counts[i]; }
    if (reset) for (int i = 0; i < 8; ++i) { c.ms[i] = 0; c.counts[i] = 0; }
    return WXA_OK;
}

}  // namespace wxa::host

#define WXA_SIM_CAPI(PFX, RET, SIMTYPE, BACKEND_GETTER, SET_ERROR)                                          \
    extern "C" {                                                                                       \
    RET PFX##sim_create(const wxa_sim_config* cfg, const wxa_comm* comm, SIMTYPE** out) {              \
        std::string err;                                                                               \
        wxa::host::SimHandle* h = nullptr;                                                             \
        int rc = wxa::host::sim_create(BACKEND_GETTER(), cfg, comm, &h, err);                          \
        if (rc != 0) { SET_ERROR(err.c_str()); return (RET)rc; }                                            \
        *out = reinterpret_cast<SIMTYPE*>(h);                                                          \
        return (RET)0;                                                                                   \
    }                                                                                                  \
    void PFX##sim_destroy(SIMTYPE* s) { delete reinterpret_cast<wxa::host::SimHandle*>(s); }           \
    RET PFX##sim_add_species(SIMTYPE* s, double q, double m, const wxa_particle_view* init, int32_t* id) { \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        int rc = wxa::host::sim_add_species(h, q, m, init, id);                                        \
        if (rc != 0 && h) SET_ERROR(h->error.c_str());                                                 \
        return (RET)rc;                                                                                     \
    }                                                                                                  \
    RET PFX##sim_evolve(SIMTYPE* s, int32_t n) {                                                       \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        int rc = wxa::host::sim_evolve(h, n);                                                          \
        if (rc != 0 && h) SET_ERROR(h->error.c_str());                                                 \
        return (RET)rc;                                                                                     \
    }                                                                                                  \
    double PFX##sim_dt(const SIMTYPE* s) {                                                             \
        return reinterpret_cast<const wxa::host::SimHandle*>(s)->warpx->getdt(0);                      \
    }                                                                                                  \
    int64_t PFX##sim_istep(const SIMTYPE* s) {                                                         \
        return reinterpret_cast<const wxa::host::SimHandle*>(s)->warpx->getistep();                    \
    }                                                                                                  \
    RET PFX##sim_get_field(SIMTYPE* s, const char* name, wxa_field_view* out) {                        \
        return (RET)wxa::host::sim_get_field(reinterpret_cast<wxa::host::SimHandle*>(s), name, out);        \
    }                                                                                                  \
    int32_t PFX##sim_halo_overlap(const SIMTYPE* s) {                                                  \
        return s && reinterpret_cast<const wxa::host::SimHandle*>(s)->warpx->halo_overlap() ? 1 : 0;     \
    }                                                                                                  \
    RET PFX##sim_compute_rho(SIMTYPE* s) {                                                             \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        int rc = wxa::host::sim_compute_rho(h);                                                        \
        if (rc != 0 && h) SET_ERROR(h->error.c_str());                                                 \
        return (RET)rc;                                                                                \
    }                                                                                                  \
    RET PFX##sim_set_moving_window(SIMTYPE* s, const wxa_moving_window* mw) {                          \
        return (RET)wxa::host::sim_set_moving_window(reinterpret_cast<wxa::host::SimHandle*>(s), mw);    \
    }                                                                                                  \
    RET PFX##sim_set_injection(SIMTYPE* s, int32_t id, const wxa_plasma_injector* inj, int add_initial, \
                               int continuous) {                                                       \
        return (RET)wxa::host::sim_set_injection(reinterpret_cast<wxa::host::SimHandle*>(s), id, inj,    \
                                                 add_initial, continuous);                             \
    }                                                                                                  \
    RET PFX##sim_set_external_particle_fields(SIMTYPE* s, int32_t id, const double E[3], const double B[3]) { \
        return (RET)wxa::host::sim_set_external_particle_fields(reinterpret_cast<wxa::host::SimHandle*>(s), id, E, B); \
    }                                                                                                  \
    RET PFX##sim_set_deposit_accumulator(SIMTYPE* s, int32_t id, int32_t acc) {                        \
        return (RET)wxa::host::sim_set_deposit_accumulator(reinterpret_cast<wxa::host::SimHandle*>(s), id, acc); \
    }                                                                                                  \
    RET PFX##sim_set_radiation_reaction(SIMTYPE* s, int32_t id, int32_t on) {                          \
        return (RET)wxa::host::sim_set_radiation_reaction(reinterpret_cast<wxa::host::SimHandle*>(s), id, on); \
    }                                                                                                  \
    RET PFX##sim_add_laser(SIMTYPE* s, const wxa_laser_antenna* la) {                                  \
        return (RET)wxa::host::sim_add_laser(reinterpret_cast<wxa::host::SimHandle*>(s), la);            \
    }                                                                                                  \
    RET PFX##sim_add_btd(SIMTYPE* s, int32_t num_snapshots, double dt_snapshots_lab, int32_t buffer_size, \
                         int32_t write_species) {                                                      \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        int rc = wxa::host::sim_add_btd(h, num_snapshots, dt_snapshots_lab, buffer_size, write_species); \
        if (rc != 0 && h) SET_ERROR(h->error.c_str());                                                 \
        return (RET)rc;                                                                                \
    }                                                                                                  \
    RET PFX##sim_btd_info(SIMTYPE* s, int32_t i, int32_t n[3], double z_lab[2], double* t_lab, int32_t* filled, \
                          int32_t* full) {                                                             \
        return (RET)wxa::host::sim_btd_info(reinterpret_cast<wxa::host::SimHandle*>(s), i, n, z_lab, t_lab, filled, full); \
    }                                                                                                  \
    RET PFX##sim_btd_data(SIMTYPE* s, int32_t i, int32_t comp, double* out) {                          \
        return (RET)wxa::host::sim_btd_data(reinterpret_cast<wxa::host::SimHandle*>(s), i, comp, out); \
    }                                                                                                  \
    RET PFX##sim_btd_num_particles(SIMTYPE* s, int32_t i, int32_t id, int64_t* n) {                    \
        return (RET)wxa::host::sim_btd_num_particles(reinterpret_cast<wxa::host::SimHandle*>(s), i, id, n); \
    }                                                                                                  \
    RET PFX##sim_btd_particles(SIMTYPE* s, int32_t i, int32_t id, double* out) {                       \
        return (RET)wxa::host::sim_btd_particles(reinterpret_cast<wxa::host::SimHandle*>(s), i, id, out); \
    }                                                                                                  \
    RET PFX##sim_add_reduced_diag(SIMTYPE* s, const char* name, const char* type, const char* intervals, \
                                  const char* path) {                                                  \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        int rc = wxa::host::sim_add_reduced_diag(h, name, type, intervals, path);                      \
        if (rc != 0 && h) SET_ERROR(h->error.c_str());                                                 \
        return (RET)rc;                                                                                \
    }                                                                                                  \
    RET PFX##sim_reduced_diag_data(SIMTYPE* s, const char* name, int32_t compute_now, double* out,     \
                                   int32_t capacity, int32_t* n) {                                     \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        int rc = wxa::host::sim_reduced_diag_data(h, name, compute_now, out, capacity, n);             \
        if (rc != 0 && h) SET_ERROR(h->error.c_str());                                                 \
        return (RET)rc;                                                                                \
    }                                                                                                  \
    RET PFX##sim_get_particles(SIMTYPE* s, int32_t id, wxa_particle_view* out) {                       \
        return (RET)wxa::host::sim_get_particles(reinterpret_cast<wxa::host::SimHandle*>(s), id, out);      \
    }                                                                                                  \
    RET PFX##sim_get_timers(SIMTYPE* s, double ms[8], int64_t counts[8], int reset) {                  \
        return (RET)wxa::host::sim_get_timers(reinterpret_cast<wxa::host::SimHandle*>(s), ms, counts, reset); \
    }                                                                                                  \
    RET PFX##sim_dry_comm(SIMTYPE* s, int32_t reps, double ms[4]) {                                    \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        int rc = wxa::host::sim_dry_comm(h, reps, ms);                                                 \
        if (rc != 0 && h) SET_ERROR(h->error.c_str());                                                 \
        return (RET)rc;                                                                                \
    }                                                                                                  \
    RET PFX##sim_enable_timers(SIMTYPE* s, int enable) {                                               \
        if (!s) return (RET)-1;                                                                            \
        reinterpret_cast<wxa::host::SimHandle*>(s)->warpx->context().timers_on = enable != 0;          \
        return (RET)0;                                                                                   \
    }                                                                                                  \
    }
#endif
