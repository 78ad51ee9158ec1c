// Particle containers of the host layer: the operator surface that
// Source/Evolve/WarpXEvolve.cpp drives on the explicit FDTD branch
// (WarpXParticleContainer / PhysicalParticleContainer / MultiParticleContainer;
// Source/Particles/WarpXParticleContainer.H:111-509,
// Source/Particles/PhysicalParticleContainer.cpp:1812-2095,2368-2516,2549-2786,
// Source/Particles/MultiParticleContainer.cpp:460-500,615-654).
// One brick per process: a container holds one pure-SoA tile on the device.
#ifndef WXA_HOST_PARTICLES_HPP_
#define WXA_HOST_PARTICLES_HPP_

#include <climits>
#include <cmath>
#include <cstdio>
#include <functional>
#include <limits>
#include <cstdlib>

#include "BrickComm.hpp"
#include "BTDiagnostics.hpp"

namespace wxa::host {

// What the reference reads from `WarpX::` statics inside the particle operators
// (WarpXParticleContainer.cpp:481-816, PhysicalParticleContainer.cpp:2603-2604,2656).
struct WarpXContext {
    const Backend* be = nullptr;
    int nox = 1;
    bool galerkin_interpolation = true;
    ParticlePusherAlgo particle_pusher_algo = ParticlePusherAlgo::Boris;
    CurrentDepositionAlgo current_deposition_algo = CurrentDepositionAlgo::Esirkepov;
    std::array<double, 3> prob_lo{}, prob_hi{}, dx{}, dinv{};
    amrex::Box brick_box;              // this brick's cells in global index space
    std::array<double, 3> brick_plo{}, brick_phi{};
    amrex::IntVect ng_alloc_EB, ng_depos_J;
    void* stream = nullptr;
    bool sort_now = false;             // this step re-sorts the tiles (sort_intervals)
    bool count_now = false;            // the next step does: this step's push records the sort keys (sort_in_push)
    bool sort_intervals_on = false;    // warpx.sort_intervals > 0
    bool skip_sort_behind_a_window_shift = true;   // WXA_SORT_BEHIND_SHIFT=1 switches the skipping off (A/B runs)
    bool sort_in_push = false;         // the periodic sorts are folded into PushPX (wxa_push_sort_begin, include/warpx_amd.h)
    int32_t sort_wrap[3] = {0, 0, 0};  // directions along which this brick is its own periodic neighbour
    double sort_predict_dt = 0.0;      // the recorded keys are those of the positions one free-flight step ahead (0: of the positions)
    // One special push per sort cycle (round 6): the push of a sort step SCATTERs with the record of the previous sort step
    // and COUNTs the record of the next one -- keys of the positions sort_intervals free-flight steps ahead -- in one pass;
    // the pushes between two sort steps are plain.  (Separate COUNT and SCATTER pushes made every push of an interval-2
    // cycle a special one.)  A run starts, and restarts after anything that drops the record (a classic sort, a window
    // shift's sort), with a COUNT alone in the first push that finds no record.
    bool sort_merged = false;
    int32_t sort_interval_steps = 0;   // warpx.sort_intervals
    int32_t steps_to_next_sort = 0;    // pushes behind this one up to and including the next sort step's
    double step_dt = 0.0;
    // boundary.particle_lo/hi resolved to WXA_PBOUNDARY_PERIODIC / _ABSORBING / _REFLECTING
    int32_t particle_bc_lo[3] = {WXA_PBOUNDARY_PERIODIC, WXA_PBOUNDARY_PERIODIC, WXA_PBOUNDARY_PERIODIC};
    int32_t particle_bc_hi[3] = {WXA_PBOUNDARY_PERIODIC, WXA_PBOUNDARY_PERIODIC, WXA_PBOUNDARY_PERIODIC};
    bool any_particle_wall = false;
    // particles.use_fdtd_nci_corr: E and B filtered along z (Godfrey stencil) into these before every species' gather
    // (PhysicalParticleContainer::applyNCIFilter, PhysicalParticleContainer.cpp:2097-2172); owned by WarpX
    bool use_fdtd_nci_corr = false;
    amrex::MultiFab* nci_E[3] = {nullptr, nullptr, nullptr};
    amrex::MultiFab* nci_B[3] = {nullptr, nullptr, nullptr};
    double nci_stencil_exeybz[5] = {0.5, 0, 0, 0, 0}, nci_stencil_bxbyez[5] = {0.5, 0, 0, 0, 0};
    // warpx.gamma_boost / beta_boost (boost along z, WarpXUtil.cpp:114-141) and warpx.gett_new(0)
    double gamma_boost = 1.0, beta_boost = 0.0;
    double t_new = 0.0;
    // <diag>.diag_type = BackTransformed with species output: owned by WarpX (BTDiagnostics.hpp)
    BTDiagnostics* btd = nullptr;
    // per-phase device timers, named after the reference's profiler regions
    bool timers_on = false;
    double ms[8] = {0};
    int64_t counts[8] = {0};
    struct PendingInterval { int id; void* e0; void* e1; };
    std::vector<PendingInterval> pending;
    void resolve_timers() {
        for (auto& p : pending) {
            const double t = be->event_elapsed_ms(p.e0, p.e1);   // synchronises on e1
            if (std::getenv("WXA_TIMER_TRACE")) std::fprintf(stderr, "[wxa timer] phase %d: %.3f ms\n", p.id, t);
            ms[p.id] += t;
            counts[p.id] += 1;
            be->event_destroy(p.e0); be->event_destroy(p.e1);
        }
        pending.clear();
    }

    // WarpX::LowerCorner(box.grow(ng)) + lbound (Source/WarpX.cpp:2851-2875)
    wxa_grid_geom geom(const amrex::IntVect& ng) const {
        wxa_grid_geom g{};
        for (int d = 0; d < 3; ++d) {
            const int lo = brick_box.lo[d] - ng[d];
            g.lo[d] = lo;
            g.xyzmin[d] = prob_lo[d] + (double)lo * dx[d];
            g.dinv[d] = dinv[d];
        }
        return g;
    }
};

enum Phase { kGatherAndPush = 0, kCurrentDeposition = 1, kSyncCurrent = 2, kEvolveB = 3, kEvolveE = 4,
             kFillBoundary = 5, kRedistribute = 6, kOther = 7 };

// RAII region timer (HIP events on the kernels' stream); mirrors WARPX_PROFILE regions
// (SURVEY.md section 5).  The event pairs are only recorded here and resolved later
// (WarpXContext::resolve_timers), so that timing never drains the queue: an interval then
// starts when the previous kernel finishes on the device, free of host launch latency.
struct PhaseTimer {
    WarpXContext* c; int id; void* e0 = nullptr; void* e1 = nullptr;
    PhaseTimer(WarpXContext* ctx, int phase) : c(ctx), id(phase) {
        if (c->timers_on && c->be->event_create) {
            e0 = c->be->event_create(); e1 = c->be->event_create();
            c->be->event_record(e0, c->stream);
        }
    }
    ~PhaseTimer() {
        if (e0) {
            c->be->event_record(e1, c->stream);
            c->pending.push_back({id, e0, e1});
        }
    }
};

inline void check(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + " returned status " + std::to_string(rc));
}

// Pure-SoA tile on the device, PIdx order (NamedComponentParticleContainer.H:23-40)
class ParticleTile {
public:
    explicit ParticleTile(const Backend* be) : m_be(be) {}
    ~ParticleTile() { release(); }
    ParticleTile(const ParticleTile&) = delete;
    ParticleTile& operator=(const ParticleTile&) = delete;

    int64_t numParticles() const { return m_np; }
    int64_t capacity() const { return m_cap; }
    void reserve(int64_t n) {
        if (n <= m_cap) return;
        const int64_t cap = n + n / 16 + 1024;
        double* nd = static_cast<double*>(m_be->dmalloc(sizeof(double) * 7 * (size_t)cap));
        uint64_t* ni = static_cast<uint64_t*>(m_be->dmalloc(sizeof(uint64_t) * (size_t)cap));
        if (!nd || !ni) throw std::runtime_error("ParticleTile: allocation failed");
        if (m_np > 0) {
            for (int c = 0; c < 7; ++c)
                m_be->memcpy_async(nd + c * cap, m_data + c * m_cap, sizeof(double) * (size_t)m_np, nullptr);
            m_be->memcpy_async(ni, m_id, sizeof(uint64_t) * (size_t)m_np, nullptr);
            m_be->stream_sync(nullptr);
        }
        release();
        m_data = nd; m_id = ni; m_cap = cap;
    }
    void resize(int64_t n) { reserve(n); m_np = n; }
    double* comp(int c) const { return m_data + (int64_t)c * m_cap; }
    uint64_t* idcpu() const { return m_id; }
    wxa_particle_view view(int64_t offset = 0, int64_t count = -1) const {
        wxa_particle_view v{};
        v.x = comp(0) + offset; v.y = comp(1) + offset; v.z = comp(2) + offset; v.w = comp(3) + offset;
        v.ux = comp(4) + offset; v.uy = comp(5) + offset; v.uz = comp(6) + offset;
        v.idcpu = m_id ? m_id + offset : nullptr;
        v.np = count < 0 ? m_np - offset : count;
        return v;
    }
    void swap(ParticleTile& o) {
        std::swap(m_data, o.m_data); std::swap(m_id, o.m_id); std::swap(m_cap, o.m_cap); std::swap(m_np, o.m_np);
    }

private:
    void release() {
        if (m_data) m_be->dfree(m_data);
        if (m_id) m_be->dfree(m_id);
        m_data = nullptr; m_id = nullptr; m_cap = 0;
    }
    const Backend* m_be;
    double* m_data = nullptr;
    uint64_t* m_id = nullptr;
    int64_t m_cap = 0, m_np = 0;
};

class WarpXParticleContainer {
public:
    WarpXParticleContainer(WarpXContext* ctx, double charge, double mass)
        : m_ctx(ctx), m_tile(ctx->be), m_spare(ctx->be), charge(charge), mass(mass) {
        m_sendbuf.be = ctx->be; m_recvbuf.be = ctx->be; m_lists.be = ctx->be;
        for (auto& b : m_arrival_lists) b.be = ctx->be;
        check(ctx->be->workspace_create(&m_ws), "workspace_create");
    }
    virtual ~WarpXParticleContainer() { if (m_ws) m_ctx->be->workspace_destroy(m_ws); }
    // particles.E_external_particle / B_external_particle, *_ext_particle_init_style = constant
    // (m_E_external_particle / m_B_external_particle, PhysicalParticleContainer.cpp:2589-2596): they live with the
    // container's workspace, where the gather kernels find them
    // <species>.do_classical_radiation_reaction: doParticleMomentumPush takes the radiation-reaction branch first,
    // whatever algo.particle_pusher says (PushSelector.H:60-87)
    void SetRadiationReaction(bool on) { m_do_crr = on; }
    int pusher_algo() const { return m_do_crr ? WXA_PUSHER_BORIS_RR : (int)m_ctx->particle_pusher_algo; }
    // fp32 / fp64 accumulators of the LDS-tile current deposition of this container (include/warpx_amd.h, WXA_ACC_*)
    void SetDepositAccumulator(int32_t acc) {
        if (!m_ctx->be->ws_set_deposit_accumulator) throw std::runtime_error("deposit accumulator: not in this backend");
        check(m_ctx->be->ws_set_deposit_accumulator(m_ws, acc), "ws_set_deposit_accumulator");
    }
    // particles.*_ext_particle_init_style = repeated_plasma_lens (MultiParticleContainer.cpp:210-260): lens.gamma_boost
    // and lens.dt are filled in here from the run's own values
    void SetRepeatedPlasmaLens(wxa_repeated_plasma_lens lens, double level_dt) {
        if (!m_ctx->be->ws_set_repeated_plasma_lens || !m_ctx->be->ws_set_time)
            throw std::runtime_error("repeated plasma lens: not in this backend");
        lens.gamma_boost = m_ctx->gamma_boost;
        lens.dt = level_dt;
        check(m_ctx->be->ws_set_repeated_plasma_lens(m_ws, &lens), "ws_set_repeated_plasma_lens");
        m_time_dependent_ext = lens.n_lenses > 0;
    }
    // m_time of GetExternalEBField (GetExternalFields.cpp:43), before a push
    void stamp_external_time() {
        if (m_time_dependent_ext) check(m_ctx->be->ws_set_time(m_ws, m_ctx->t_new), "ws_set_time");
    }
    void SetExternalParticleFields(const double E[3], const double B[3]) {
        if (!m_ctx->be->ws_set_external_eb) throw std::runtime_error("external particle fields: not in this backend");
        check(m_ctx->be->ws_set_external_eb(m_ws, E, B), "ws_set_external_eb");
    }

    // Source/Particles/WarpXParticleContainer.H:150-154
    virtual void Evolve(ablastr::fields::MultiFabRegister& fields, int lev, const std::string& current_fp_string,
                        amrex::Real t, amrex::Real dt, DtType a_dt_type = DtType::Full,
                        bool skip_deposition = false, PushType push_type = PushType::Explicit) = 0;
    // :180-186
    virtual void PushP(int lev, amrex::Real dt, const amrex::MultiFab& Ex, const amrex::MultiFab& Ey,
                       const amrex::MultiFab& Ez, const amrex::MultiFab& Bx, const amrex::MultiFab& By,
                       const amrex::MultiFab& Bz) = 0;

    // Source/Particles/WarpXParticleContainer.cpp:352-827 (single tile = whole brick)
    void DepositCurrent(amrex::MultiFab* jx, amrex::MultiFab* jy, amrex::MultiFab* jz, amrex::Real dt,
                        amrex::Real relative_time) {
        if (m_tile.numParticles() == 0) return;
        const wxa_field_view J[3] = {jx->view(), jy->view(), jz->view()};
        const wxa_grid_geom g = m_ctx->geom(m_ctx->ng_depos_J);
        const wxa_particle_view p = m_tile.view();
        check(m_ctx->be->deposit_current(&p, J, &g, charge, dt, relative_time, m_ctx->nox,
                                         (int)m_ctx->current_deposition_algo, m_ws, m_ctx->stream),
              "deposit_current");
    }

    // WarpXParticleContainer::DepositCharge (Source/Particles/WarpXParticleContainer.cpp:1180-1295), the
    // box grown by the guards of rho; retired particles carry zero weight
    void DepositCharge(amrex::MultiFab* rho) {
        if (m_tile.numParticles() == 0) return;
        const wxa_grid_geom g = m_ctx->geom(rho->nGrowVect());
        const wxa_particle_view p = m_tile.view();
        check(m_ctx->be->deposit_charge(&p, &rho->view(), &g, charge, m_ctx->nox, m_ctx->stream), "deposit_charge");
    }

    // amrex SortParticlesByBin with bin = one cell (MultiParticleContainer.cpp:615-621).  Also
    // drops the particles retired by Redistribute and merges the arrivals appended since the
    // last sort into the cell order.
    void SortParticlesByBin(const amrex::IntVect& /*bin_size*/) {
        const int64_t np = m_tile.numParticles();
        if (np == 0) return;
        m_spare.resize(np);
        const wxa_particle_view src = m_tile.view(), dst = m_spare.view();
        int32_t lo[3], nc[3];
        for (int d = 0; d < 3; ++d) { lo[d] = m_ctx->brick_box.lo[d]; nc[d] = m_ctx->brick_box.length(d); }
        check(m_ctx->be->sort_particles_by_cell(&src, &dst, m_ctx->brick_plo.data(), m_ctx->dinv.data(), lo, nc,
                                                m_ws, m_ctx->stream),
              "sort_particles_by_cell");
        m_tile.swap(m_spare);
        m_steps_since_sort = 0;
        m_count_nretired = 0;   // (the backend has dropped the record of a COUNT, if there was one)
        if (m_nretired > 0) {
            int64_t live = np;
            check(m_ctx->be->sort_live_count(m_ws, &live, m_ctx->stream), "sort_live_count");
            if (live != np - m_nretired) throw std::runtime_error("SortParticlesByBin: retired-particle count mismatch");
            m_tile.resize(live);
            m_nretired = 0;
        }
    }

    // The periodic cell sort folded into the push (wxa_push_sort_begin / _end, include/warpx_amd.h; csrc/push_sort.hpp): the
    // push of the step before a sort step records every particle's cell key and rank (COUNT), the push of the sort
    // step writes the particles straight into the sorted tile (SCATTER) -- no pass of its own over the eight arrays.
    // Called before the first push of a step (PushInterior or Evolve), idempotent until FinishPushSort.
    void ArmPushSort() {
        const Backend* be = m_ctx->be;
        if (m_push_sort_mode != 0 || !m_ctx->sort_in_push || !be->push_sort_begin || m_tile.numParticles() == 0) return;
        // a species of the BackTransformed diagnostic is compared, index by index, with its copy from before the push
        if (m_ctx->btd != nullptr && btd_species_id >= 0) return;
        const wxa_particle_view p = m_tile.view();
        int32_t mode = 0;
        double predict_dt = m_ctx->sort_predict_dt;
        m_skip_classic_sort = false;
        if (m_ctx->sort_merged) {
            const bool pending = be->push_sort_pending(m_ws, &p) != 0;
            const bool predict = m_ctx->sort_predict_dt != 0.0;
            if (m_ctx->sort_now && pending) {
                mode = WXA_PUSH_SORT_SCATTER | WXA_PUSH_SORT_COUNT;   // this cycle's scatter, the next cycle's record
                predict_dt = predict ? m_ctx->sort_interval_steps * m_ctx->step_dt : 0.0;
            } else if (!pending && (!m_ctx->sort_now || (m_ctx->sort_interval_steps == 1 && m_steps_since_sort == 1))) {
                // no record: taken alone, for the next sort step.  (At an interval of 1 every step sorts: the step after a
                // classic sort records instead of sorting again -- the tile is one push old, which every kernel tolerates.)
                mode = WXA_PUSH_SORT_COUNT;
                predict_dt = predict ? (m_ctx->sort_now ? 1 : m_ctx->steps_to_next_sort) * m_ctx->step_dt : 0.0;
                m_skip_classic_sort = m_ctx->sort_now;
            }
        } else {
            if (m_ctx->sort_now && be->push_sort_pending(m_ws, &p)) mode |= WXA_PUSH_SORT_SCATTER;
            // a record is of no use when a sort of the classic kind follows this push (it replaces the order the record indexes)
            if (m_ctx->count_now && (!m_ctx->sort_now || mode != 0)) mode |= WXA_PUSH_SORT_COUNT;
        }
        if (mode == 0) return;
        wxa_particle_view dst{};
        if (mode & WXA_PUSH_SORT_SCATTER) {
            m_spare.resize(p.np);
            dst = m_spare.view();
        }
        int32_t lo[3], nc[3];
        for (int d = 0; d < 3; ++d) { lo[d] = m_ctx->brick_box.lo[d]; nc[d] = m_ctx->brick_box.length(d); }
        check(be->push_sort_begin(m_ws, mode, &p, &dst, m_ctx->brick_plo.data(), m_ctx->dinv.data(), lo, nc, m_ctx->sort_wrap,
                                  m_nretired > 0 ? 1 : 0, predict_dt, m_ctx->stream),
              "push_sort_begin");
        m_push_sort_mode = mode;
    }
    // after the last push of the step: the scan of a COUNT; after a SCATTER the sorted tile becomes the tile.
    // Returns true when this step's sort has been done by the push.
    bool FinishPushSort() {
        if (m_push_sort_mode == 0) return false;
        const int32_t mode = m_push_sort_mode;
        m_push_sort_mode = 0;
        // the record of a SCATTER was taken m_count_nretired retired particles into the cycle: they are dropped now
        const bool scatter = (mode & WXA_PUSH_SORT_SCATTER) != 0;
        int64_t live = 0, appended = 0;
        check(m_ctx->be->push_sort_end(m_ws, scatter && m_count_nretired > 0 ? 1 : 0, &live, &appended, m_ctx->stream),
              "push_sort_end");
        if (scatter) {
            const int64_t np = m_tile.numParticles();
            if (live + appended > np || np - (live + appended) != m_count_nretired)
                throw std::runtime_error("FinishPushSort: retired-particle count mismatch");
            m_tile.swap(m_spare);
            m_tile.resize(live + appended);
            m_nretired -= m_count_nretired;
            m_count_nretired = 0;
            m_steps_since_sort = 1;   // the order is the cell order of the positions before this push
        }
        if (mode & WXA_PUSH_SORT_COUNT) m_count_nretired = m_nretired;   // the retired ones the record has put behind the cells
        static const bool trace = std::getenv("WXA_PUSH_SORT_TRACE") != nullptr;
        if (trace)
            std::fprintf(stderr, "[push sort] mode %d: %lld particles, %lld cell-sorted + %lld appended, %lld retired in the tile\n",
                         (int)mode, (long long)m_tile.numParticles(), (long long)live, (long long)appended, (long long)m_nretired);
        return scatter;
    }

    // WarpXParticleContainer::ApplyBoundaryConditions (WarpXParticleContainer.cpp:1574-1660): reflecting and
    // absorbing walls; absorbed particles are retired in place and dropped by the next sort, like the
    // particles handed to a neighbour brick
    void ApplyBoundaryConditions() {
        if (!m_ctx->any_particle_wall || m_tile.numParticles() == 0) return;   // :1578 all periodic
        const wxa_particle_view p = m_tile.view();
        int64_t lost = 0;
        check(m_ctx->be->apply_particle_boundaries(&p, m_ctx->prob_lo.data(), m_ctx->prob_hi.data(), m_ctx->particle_bc_lo,
                                                   m_ctx->particle_bc_hi, &lost, m_ws, m_ctx->stream),
              "apply_particle_boundaries");
        m_nretired += lost;
        // without a periodic sort (warpx.sort_intervals <= 0) nothing else would ever drop the retired particles:
        // compact once they make up a quarter of the tile
        if (!m_ctx->sort_intervals_on && m_nretired > 1024 && 4 * m_nretired > m_tile.numParticles())
            SortParticlesByBin(amrex::IntVect(1));
    }

    // amrex ParticleContainer::Redistribute restricted to what the periodic brick decomposition
    // needs: periodic wrap, then hand the particles that left the brick to the +/- neighbour,
    // direction by direction (a particle moves < 1 cell per step, so corners take up to three
    // hops).  The tile is NOT re-bucketed: per step only ~1e-5 of a brick's particles cross a
    // face, so one scan lists them, they are packed and retired in place (weight 0, dropped by the
    // next sort) and arrivals are appended behind the sorted part -- the cell order that the
    // LDS-tile kernels rely on survives between sorts.
    void Redistribute(BrickComm& comm) {
        const Backend* be = m_ctx->be;
        // particles wrap only along the periodic directions (walls: ApplyBoundaryConditions)
        const int periodic[3] = {comm.periodic(0) ? 1 : 0, comm.periodic(1) ? 1 : 0, comm.periodic(2) ? 1 : 0};
        int split[3];
        bool any_split = false;
        for (int d = 0; d < 3; ++d) { split[d] = comm.self_periodic(d) ? 0 : 1; any_split = any_split || split[d]; }
        const int64_t np0 = m_tile.numParticles();
        if (!any_split) {
            if (np0 > 0) {
                const wxa_particle_view p = m_tile.view();
                if (be->enforce_periodic_sorted && m_steps_since_sort >= 0)
                    check(be->enforce_periodic_sorted(&p, m_ctx->prob_lo.data(), m_ctx->prob_hi.data(), periodic, m_ws,
                                                      m_steps_since_sort, m_ctx->stream),
                          "enforce_periodic_sorted");
                else
                    check(be->enforce_periodic(&p, m_ctx->prob_lo.data(), m_ctx->prob_hi.data(), periodic, m_ctx->stream),
                          "enforce_periodic");
            }
            if (m_steps_since_sort >= 0) ++m_steps_since_sort;
            return;
        }
        // One scan of the whole tile: periodic wrap + the leavers listed by DESTINATION brick (26 neighbours): a
        // particle moves less than a cell per step, so it is handed to its final brick in one message.  Per step: one
        // host read of the 27 counts, ONE count round with all neighbours, one grouped data exchange -- no second
        // classification of arrivals (edges and corners took up to three hops, each with its own blocking count
        // exchange, before).
        // per list; the faces of a 256^3 brick at |v| -> c send ~np / 150.  A thin brick of a boosted-frame run sends far
        // more through its low z face (the plasma streams at ~ -c and the window shift moves the brick by a cell: c dt / dz
        // of its thickness per step): a list that overflows is not fatal -- the counts are exact whatever the capacity,
        // so the lists grow to what was counted and the scan runs again (the wrap is idempotent, nothing has been packed
        // or retired yet); the capacity reached is kept for the following steps.
        if (m_be_release_lists) {   // decided by the previous call, whose packs have long read the lists
            m_be_release_lists = false;
            if (m_lists.p) { be->dfree(m_lists.p); m_lists.p = nullptr; m_lists.cap = 0; }
        }
        int64_t cap = std::max<int64_t>(m_list_cap, np0 / 64 + 4096);
        int32_t* lists = nullptr;
        int64_t cnt[27];
        for (int64_t& c : cnt) c = 0;
        for (int attempt = 0; np0 > 0; ++attempt) {
            m_lists.reserve(sizeof(int32_t) * 27 * (size_t)cap);
            lists = static_cast<int32_t*>(m_lists.p);
            const wxa_particle_view p = m_tile.view();
            check(be->wrap_and_classify_dest(&p, 0, np0, m_ctx->prob_lo.data(), m_ctx->prob_hi.data(), periodic,
                                             m_ctx->brick_plo.data(), m_ctx->brick_phi.data(), split, lists, cap, cnt,
                                             m_ws, m_ctx->stream),
                  "wrap_and_classify_dest");
            const int64_t most = *std::max_element(cnt, cnt + 27);
            if (most <= cap) break;
            if (attempt > 0) throw std::runtime_error("Redistribute: leaver list overflow after the lists were grown");
            cap = most + most / 8 + 64;
        }
        if (np0 == 0) { m_lists.reserve(sizeof(int32_t) * 27 * (size_t)cap); lists = static_cast<int32_t*>(m_lists.p); }
        m_list_cap = cap;
        {   // a capacity that one crowded step needed is given back after 32 quiet ones (the lists are 27 x capacity ints:
            // in a thin brick of a boosted-frame run more than the particles they list -- ADVICE round 4)
            const int64_t most = *std::max_element(cnt, cnt + 27);
            m_list_quiet_steps = 4 * (most + 64) < m_list_cap && m_list_cap > np0 / 64 + 4096 ? m_list_quiet_steps + 1 : 0;
            if (m_list_quiet_steps >= 32) {
                m_list_cap = std::max<int64_t>(2 * (most + 64), np0 / 64 + 4096);
                m_be_release_lists = true;   // at the start of the next call: this step's packs still read them
                m_list_quiet_steps = 0;
            }
        }
        // the distinct peers, in the same canonical order on both sides of every pair: ascending offset code on the
        // sender is descending code (the mirrored offset) on the receiver, so peers are ordered by rank instead
        struct Peer { int rank; int64_t nsend = 0, nrecv = 0; std::vector<int> codes; };
        std::vector<Peer> peers;
        for (int code = 0; code < 27; ++code) {
            if (code == 13) continue;
            const int o[3] = {code % 3 - 1, (code / 3) % 3 - 1, code / 9 - 1};
            bool reachable = true;
            for (int d = 0; d < 3; ++d) reachable = reachable && (o[d] == 0 || split[d]);
            if (!reachable) continue;
            const int r = comm.rank_at_offset(o);
            if (r < 0) {
                // beyond the domain boundary of a non-periodic direction: the particle walls have dealt with every such
                // particle before this point (ApplyBoundaryConditions); what is left is retired here, sent nowhere
                if (cnt[code] > 0) {
                    const wxa_particle_view p = m_tile.view();
                    m_sendbuf.reserve(64 * (size_t)cnt[code]);
                    check(be->pack_leavers(&p, lists + (int64_t)code * cap, cnt[code], m_sendbuf.p, cnt[code], 0,
                                           /*retire=*/1, m_ctx->brick_plo.data(), m_ctx->brick_phi.data(), m_ctx->stream),
                          "pack_leavers");
                    m_nretired += cnt[code];
                }
                continue;
            }
            auto it = std::find_if(peers.begin(), peers.end(), [&](const Peer& q) { return q.rank == r; });
            if (it == peers.end()) { peers.push_back(Peer{r}); it = peers.end() - 1; }
            it->codes.push_back(code);
            it->nsend += cnt[code];
        }
        std::sort(peers.begin(), peers.end(), [](const Peer& a, const Peer& b) { return a.rank < b.rank; });
        const int npeers = (int)peers.size();
        {
            std::vector<int32_t> pr(npeers);
            std::vector<int64_t> sv(npeers), rv(npeers, 0);
            for (int i = 0; i < npeers; ++i) { pr[i] = peers[i].rank; sv[i] = peers[i].nsend; }
            comm.exchange_counts_with(npeers, pr.data(), sv.data(), rv.data());
            for (int i = 0; i < npeers; ++i) peers[i].nrecv = rv[i];
        }
        int64_t nsend_tot = 0, nrecv_tot = 0;
        for (const Peer& q : peers) { nsend_tot += q.nsend; nrecv_tot += q.nrecv; }
        // staging: one message per peer = 8 SoA rows (7 reals + idcpu) of n entries
        m_sendbuf.reserve(64 * (size_t)std::max<int64_t>(nsend_tot, 1));
        m_recvbuf.reserve(64 * (size_t)std::max<int64_t>(nrecv_tot, 1));
        std::vector<void*> sb(npeers), rb(npeers);
        std::vector<int64_t> sbytes(npeers), rbytes(npeers);
        std::vector<int32_t> prank(npeers);
        {
            const wxa_particle_view p = m_tile.view();
            char* sp = static_cast<char*>(m_sendbuf.p);
            char* rp = static_cast<char*>(m_recvbuf.p);
            for (int i = 0; i < npeers; ++i) {
                const Peer& q = peers[i];
                int64_t off = 0;
                for (int code : q.codes) {
                    if (cnt[code] == 0) continue;
                    check(be->pack_leavers(&p, lists + (int64_t)code * cap, cnt[code], sp, q.nsend, off, /*retire=*/1,
                                           m_ctx->brick_plo.data(), m_ctx->brick_phi.data(), m_ctx->stream),
                          "pack_leavers");
                    off += cnt[code];
                }
                prank[i] = q.rank;
                sb[i] = sp; sbytes[i] = 64 * q.nsend; sp += 64 * q.nsend;
                rb[i] = rp; rbytes[i] = 64 * q.nrecv; rp += 64 * q.nrecv;
            }
        }
        m_nretired += nsend_tot;
        if (nsend_tot > 0 || nrecv_tot > 0)
            comm.exchange_with(npeers, prank.data(), sb.data(), sbytes.data(), rb.data(), rbytes.data(), m_ctx->stream);
        if (nrecv_tot > 0) {
            int64_t n0 = m_tile.numParticles();
            m_tile.resize(n0 + nrecv_tot);   // appends behind the sorted part (reallocation keeps the contents)
            for (int i = 0; i < npeers; ++i) {
                const int64_t n = peers[i].nrecv;
                if (n == 0) continue;
                const char* srcb = static_cast<const char*>(rb[i]);
                for (int c = 0; c < 7; ++c)
                    be->memcpy_async(m_tile.comp(c) + n0, srcb + 8 * (int64_t)c * n, 8 * (size_t)n, m_ctx->stream);
                be->memcpy_async(m_tile.idcpu() + n0, srcb + 8 * 7 * n, 8 * (size_t)n, m_ctx->stream);
                n0 += n;
            }
        }
        if (m_steps_since_sort >= 0) ++m_steps_since_sort;
        // No stream_sync here (until round 4 every step ended with one): everything above is ordered on the compute
        // stream -- the packs, the data exchange, the appends -- and the staging buffers belong to this container; the
        // host runs ahead and queues the next step's kernels.  The one host wait of Redistribute is the read of the 27
        // counts inside wrap_and_classify_dest, plus the count round on the transport's own stream.
    }

    // AddNParticles restricted to what injection needs: host columns x,y,z,w,ux,uy,uz appended behind the
    // tile (ids 0); like arrivals from a neighbour brick they are the tile's tail until the next sort
    void AppendFromHost(const std::vector<double> (&cols)[7]) {
        const int64_t n = (int64_t)cols[0].size();
        if (n == 0) return;
        const Backend* be = m_ctx->be;
        const int64_t n0 = m_tile.numParticles();
        m_tile.resize(n0 + n);
        for (int c = 0; c < 7; ++c) be->memcpy_h2d(m_tile.comp(c) + n0, cols[c].data(), sizeof(double) * (size_t)n);
        be->memset_async(m_tile.idcpu() + n0, 0, sizeof(uint64_t) * (size_t)n, m_ctx->stream);
        be->stream_sync(m_ctx->stream);
    }

    // see PhysicalParticleContainer::PushInterior; containers that gather nothing have nothing to do
    virtual void PushInterior(ablastr::fields::MultiFabRegister& /*fields*/, amrex::Real /*dt*/) {}

    // WarpXParticleContainer::doContinuousInjection / ContinuousInjection / m_current_injection_position
    virtual bool doContinuousInjection() const { return false; }
    virtual void ContinuousInjection(const double* /*box_lo*/, const double* /*box_hi*/) {}
    // InjectorMomentum::getBulkMomentum of the base injector along `dir`, in units of c
    // (evaluated at the current injection position along `dir`, 0 in the other directions: WarpXMovingWindow.cpp:78-97)
    virtual amrex::Real BulkMomentum(int /*dir*/) const { return 0.0; }
    amrex::Real m_current_injection_position = std::numeric_limits<amrex::Real>::quiet_NaN();   // unset

    // The tile without retired particles (compacts by sorting if Redistribute retired some since
    // the last sort): what diagnostics and callers outside the step loop should look at.
    ParticleTile& tile() {
        if (m_nretired > 0) SortParticlesByBin(amrex::IntVect(1));
        return m_tile;
    }
    amrex::Long TotalNumberOfParticles() const { return m_tile.numParticles() - m_nretired; }

protected:
    WarpXContext* m_ctx;
    ParticleTile m_tile, m_spare;
    DeviceBuffer m_sendbuf, m_recvbuf, m_lists, m_arrival_lists[3];
    int64_t m_list_cap = 0;   // entries per destination list that Redistribute has grown to (0: the default sizing)
    int32_t m_list_quiet_steps = 0;      // consecutive steps that used less than a quarter of it
    bool m_be_release_lists = false;     // the lists are freed at the start of the next Redistribute and reallocated smaller
    DeviceBuffer m_btd_old[6], m_btd_scratch;   // back-transformed diagnostics: attributes before the push, selection output
public:
    int btd_species_id = -1;                    // >= 0: this species is written by the BackTransformed diagnostic
protected:
    int64_t m_nretired = 0;            // retired by Redistribute since the last sort (still in the tile)
    int32_t m_push_sort_mode = 0;      // WXA_PUSH_SORT_* armed for this step's push (ArmPushSort)
    bool m_skip_classic_sort = false;  // merged mode, interval 1: this sort step records instead of sorting
    int64_t m_count_nretired = 0;      // retired particles in the tile when the last COUNT was taken
    int32_t m_steps_since_sort = -1;   // Redistribute calls since the last cell sort (-1: never sorted)
public:
    // The window shift behind the last push has sorted the tile (WarpX::MoveWindow): this step's periodic sort -- between
    // push and deposition, one push later -- would find the order that the folded sort delivers anyway (the cell order of
    // the positions before the push) and is skipped.  A window that moves every step sorted twice per step at
    // warpx.sort_intervals = 1: 8.4 of BASELINE config 5's 94 ms per step on one GPU (profiles/round5/README.md).
    bool m_sorted_by_window_shift = false;
protected:
    void* m_ws = nullptr;
    bool m_do_crr = false;
    bool m_time_dependent_ext = false;   // an external field on the particles that depends on the time (boosted lens)

public:
    amrex::ParticleReal charge, mass;
};

class PhysicalParticleContainer : public WarpXParticleContainer {
public:
    using WarpXParticleContainer::WarpXParticleContainer;

    // <species>.injection_style = NUniformPerCell, profile = constant, momentum at_rest
    void SetPlasmaInjector(const wxa_plasma_injector& inj, bool continuous) {
        if (inj.ppc[0] < 1 || inj.ppc[1] < 1 || inj.ppc[2] < 1 || !(inj.density >= 0.0))
            throw std::runtime_error("plasma injector: bad density or particles per cell");
        m_inj = inj;
        m_has_injector = true;
        m_do_continuous_injection = continuous;
    }
    bool doContinuousInjection() const override { return m_has_injector && m_do_continuous_injection; }
    // PhysicalParticleContainer::ContinuousInjection (PhysicalParticleContainer.cpp:2518-2528)
    void ContinuousInjection(const double* box_lo, const double* box_hi) override { AddPlasma(box_lo, box_hi); }
    amrex::Real BulkMomentum(int dir) const override {
        if (m_momentum_on_device) return m_device_momentum.u_mean[dir];
        if (!m_momentum) return 0.0;   // at rest
        // InjectorMomentumParser::getBulkMomentum(x, y, z): the parsed functions at the injection position
        double pos[3] = {0.0, 0.0, 0.0}, u[3] = {0.0, 0.0, 0.0};
        pos[dir] = m_current_injection_position;
        m_momentum(pos[0], pos[1], pos[2], u);
        return u[dir];
    }

    // PhysicalParticleContainer::AddPlasma (:924-1333) for one box per brick, lab frame, plasma at rest: the
    // cells of part_box that overlap this brick (find_overlap, Source/Particles/AddPlasmaUtilities.cpp:12-43),
    // positions from InjectorPositionRegular, weight = density * cell volume / particles per cell.
    // Generated on the host (a slab of one or two cell layers per step in a moving window) and appended.
    void AddPlasma(const double* part_lo, const double* part_hi) {
        if (!m_has_injector) return;
        m_inj.gamma_boost = m_ctx->gamma_boost;
        m_inj.t = m_ctx->t_new;
        // a plasma injected in a boosted frame streams through the grid at -beta c (most particles cross a cell per step):
        // its deposition takes the wide-frame body inside the tile loop (wxa_workspace_set_streaming_plasma).
        // WXA_STREAMING_PLASMA=0: the default kernel, for A/B timing
        if (m_ctx->gamma_boost > 1.0 && !m_streaming_set && m_ctx->be->ws_set_streaming_plasma) {
            const char* e = std::getenv("WXA_STREAMING_PLASMA");
            check(m_ctx->be->ws_set_streaming_plasma(m_ws, e && std::atoi(e) == 0 ? 0 : 1), "ws_set_streaming_plasma");
            m_streaming_set = true;
        }
        const wxa_plasma_injector& in = m_inj;
        double olo[3], ohi[3];
        int nov[3];
        for (int d = 0; d < 3; ++d) {
            const double tlo = m_ctx->brick_plo[d], thi = m_ctx->brick_phi[d], dx = m_ctx->dx[d];
            if (!(tlo <= part_hi[d])) return;
            olo[d] = part_lo[d] + std::max(std::floor((tlo - part_lo[d]) / dx), 0.0) * dx;
            if (!(thi >= part_lo[d])) return;
            ohi[d] = part_hi[d] - std::max(std::floor((part_hi[d] - thi) / dx), 0.0) * dx;
            nov[d] = (int)std::round((ohi[d] - olo[d]) / dx);
        }
        const int nppc = in.ppc[0] * in.ppc[1] * in.ppc[2];
        if (m_ctx->be->add_plasma && (!m_momentum || m_momentum_on_device)) {
            // on the device: no host arrays, no copy (a plane of a moving window at 256^2 x 8 ppc is 30 MB)
            const int64_t room = (int64_t)nov[0] * nov[1] * nov[2] * nppc;
            if (room == 0) return;
            const int64_t n0 = m_tile.numParticles();
            m_tile.resize(n0 + room);
            const wxa_particle_view dst = m_tile.view(n0, room);
            const int32_t nc[3] = {nov[0], nov[1], nov[2]};
            int64_t added = 0;
            check(m_ctx->be->add_plasma(&dst, &in, olo, nc, m_ctx->dx.data(), m_ctx->brick_plo.data(), m_ctx->brick_phi.data(),
                                        m_momentum_on_device ? &m_device_momentum : nullptr, &added, m_ws, m_ctx->stream),
                  "add_plasma");
            m_tile.resize(n0 + added);
            return;
        }
        if (m_ctx->gamma_boost > 1.0)
            throw std::runtime_error("AddPlasma: a momentum function evaluated on the host is lab-frame only; use at_rest, "
                                     "constant or gaussian momenta in a boosted frame");
        const double scale_fac = m_ctx->dx[0] * m_ctx->dx[1] * m_ctx->dx[2] / nppc;   // compute_scale_fac_volume
        auto inside = [&](double x, double y, double z) {   // InjectorPosition::insideBounds
            return x < in.hi[0] && x >= in.lo[0] && y < in.hi[1] && y >= in.lo[1] && z < in.hi[2] && z >= in.lo[2];
        };
        std::vector<double> cols[7];
        for (int k = 0; k < nov[2]; ++k)
            for (int j = 0; j < nov[1]; ++j)
                for (int i = 0; i < nov[0]; ++i) {
                    const int iv[3] = {i, j, k};
                    double lo[3], hi[3];
                    bool overlaps = true;   // InjectorPosition::overlapsWith
                    for (int d = 0; d < 3; ++d) {
                        lo[d] = olo[d] + (iv[d] + 0.0) * m_ctx->dx[d];
                        hi[d] = olo[d] + (iv[d] + 1.0) * m_ctx->dx[d];
                        overlaps = overlaps && !(lo[d] > in.hi[d] || hi[d] < in.lo[d]);
                    }
                    if (!overlaps) continue;
                    bool any = false;       // :1030-1048 a corner, edge midpoint or the centre has density
                    for (int a = 0; a < 27 && !any; ++a) {
                        const int t[3] = {a % 3, (a / 3) % 3, a / 9};
                        double q[3];
                        for (int d = 0; d < 3; ++d) q[d] = t[d] == 0 ? lo[d] : (t[d] == 1 ? (lo[d] + hi[d]) / 2. : hi[d]);
                        any = inside(q[0], q[1], q[2]) && in.density > 0;
                    }
                    if (!any) continue;
                    for (int ip = 0; ip < nppc; ++ip) {
                        // InjectorPositionRegular::getPositionUnitBox (Source/Initialization/InjectorPosition.H:74-92)
                        const int nx = in.ppc[0], ny = in.ppc[1], nz = in.ppc[2];
                        const int ix_part = ip / (ny * nz);
                        const int iz_part = (ip - ix_part * (ny * nz)) / ny;
                        const int iy_part = (ip - ix_part * (ny * nz)) - ny * iz_part;
                        const double r[3] = {(0.5 + ix_part) / nx, (0.5 + iy_part) / ny, (0.5 + iz_part) / nz};
                        double pos[3];
                        bool in_tile = true;   // tile_realbox.contains: strictly inside
                        for (int d = 0; d < 3; ++d) {
                            pos[d] = olo[d] + (iv[d] + r[d]) * m_ctx->dx[d];   // getCellCoords
                            in_tile = in_tile && pos[d] > m_ctx->brick_plo[d] && pos[d] < m_ctx->brick_phi[d];
                        }
                        if (!in_tile || !inside(pos[0], pos[1], pos[2])) continue;
                        for (int d = 0; d < 3; ++d) cols[d].push_back(pos[d]);
                        cols[3].push_back(in.density * scale_fac);
                        double u[3] = {0.0, 0.0, 0.0};                       // at_rest
                        if (m_momentum) m_momentum(pos[0], pos[1], pos[2], u);   // InjectorMomentum::getMomentum
                        for (int d = 0; d < 3; ++d) cols[4 + d].push_back(u[d] * 299'792'458.);   // :1271-1273
                    }
                }
        AppendFromHost(cols);
    }

    // <species>.momentum_distribution_type = constant | parse_momentum_function: u (in units of c) at a position
    // (InjectorMomentumConstant / InjectorMomentumParser, Source/Initialization/InjectorMomentum.H)
    void SetMomentumFunction(std::function<void(double, double, double, double*)> f) {   // evaluated on the host
        m_momentum = std::move(f);
        m_momentum_on_device = false;
    }
    // InjectorMomentumConstant (u_th = 0) / InjectorMomentumGaussian: drawn on the device (wxa_add_plasma)
    void SetGaussianMomentum(const double u_mean[3], const double u_th[3], uint64_t seed) {
        for (int d = 0; d < 3; ++d) {
            m_device_momentum.u_mean[d] = u_mean[d];
            m_device_momentum.u_th[d] = u_th[d];
            m_device_momentum.origin[d] = m_ctx->prob_lo[d];   // the lattice of this moment; a moving window shifts by whole cells
        }
        m_device_momentum.seed = seed;
        m_momentum = [](double, double, double, double*) { throw std::runtime_error("AddPlasma: no backend entry for the injection"); };
        m_momentum_on_device = true;
    }

private:
    wxa_plasma_injector m_inj{};
    bool m_has_injector = false, m_do_continuous_injection = false;
    std::function<void(double, double, double, double*)> m_momentum;
    bool m_interior_pushed = false;
    bool m_streaming_set = false;
    bool m_momentum_on_device = false;
    wxa_injected_momentum m_device_momentum{};

public:

    // Source/Particles/PhysicalParticleContainer.cpp:1812-2095: PushPX then DepositCurrent
    void Evolve(ablastr::fields::MultiFabRegister& fields, int lev, const std::string& current_fp_string,
                amrex::Real t_now, amrex::Real dt, DtType /*a_dt_type*/ = DtType::Full, bool skip_deposition = false,
                PushType push_type = PushType::Explicit) override {
        using warpx::fields::FieldType;
        if (push_type != PushType::Explicit) throw std::runtime_error("only the explicit push is supported");
        if (current_fp_string != "current_fp") throw std::runtime_error("unknown current field");
        auto E = fields.get_alldirs(FieldType::Efield_aux, lev);
        auto B = fields.get_alldirs(FieldType::Bfield_aux, lev);
        auto J = fields.get_alldirs(FieldType::current_fp, lev);
        const bool btd_particles = m_ctx->btd != nullptr && btd_species_id >= 0 && m_tile.numParticles() > 0;
        if (btd_particles) {   // CopyParticleAttribs (:2626-2629): x y z ux uy uz before the push
            const int64_t np = m_tile.numParticles();
            const int comp[6] = {0, 1, 2, 4, 5, 6};
            for (int c = 0; c < 6; ++c) {
                m_btd_old[c].be = m_ctx->be;
                m_btd_old[c].reserve(sizeof(double) * (size_t)np);
                m_ctx->be->memcpy_async(m_btd_old[c].p, m_tile.comp(comp[c]), sizeof(double) * (size_t)np, m_ctx->stream);
            }
        }
        ArmPushSort();
        {
            PhaseTimer t(m_ctx, kGatherAndPush);  // "PhysicalParticleContainer::Evolve::GatherAndPush"
            if (m_ctx->use_fdtd_nci_corr) {   // :1900-1911: filter E and B, gather from the filtered copies
                applyNCIFilter(E, B);
                PushPX(*m_ctx->nci_E[0], *m_ctx->nci_E[1], *m_ctx->nci_E[2], *m_ctx->nci_B[0], *m_ctx->nci_B[1],
                       *m_ctx->nci_B[2], dt);
            } else {
                PushPX(*E[0], *E[1], *E[2], *B[0], *B[1], *B[2], dt);
            }
        }
        if (btd_particles) {   // the particles that a lab-frame snapshot's plane met during this push (BTDiagnostics.hpp)
            const wxa_particle_view p = m_tile.view();
            const double* old6[6];
            for (int c = 0; c < 6; ++c) old6[c] = static_cast<const double*>(m_btd_old[c].p);
            m_btd_scratch.be = m_ctx->be;
            m_ctx->btd->PackParticles(*m_ctx, btd_species_id, p, old6, t_now + dt, dt, m_btd_scratch);
        }
        // Cell sort (amrex SortParticlesByBin, called by the reference from
        // HandleParticlesAtBoundaries, WarpXEvolve.cpp:575-580).  Sorting only permutes the
        // tile, so it is placed here, between push and deposition: the deposition then sees
        // positions that match the sort exactly (every stencil inside its LDS tile) and so does
        // the next step's gather.
        // With the sort folded into the push (ArmPushSort) the particles have been written into the sorted tile by the
        // push itself, in the cell order of their positions before it.
        {
            PhaseTimer t(m_ctx, kRedistribute);
            const bool sorted_by_the_push = FinishPushSort();
            const bool fresh = m_sorted_by_window_shift && m_ctx->skip_sort_behind_a_window_shift;
            m_sorted_by_window_shift = false;
            if (m_ctx->sort_now && !sorted_by_the_push && !fresh && !m_skip_classic_sort) SortParticlesByBin(amrex::IntVect(1));
            m_skip_classic_sort = false;
        }
        if (!skip_deposition) {
            PhaseTimer t(m_ctx, kCurrentDeposition);  // "...::DepositCurrent::CurrentDeposition"
            // :2029 relative_time = -0.5*dt: deposit at the half step
            DepositCurrent(J[0], J[1], J[2], dt, -0.5 * dt);
        }
    }

    // PhysicalParticleContainer::applyNCIFilter (:2097-2172): Ex, Ey, Bz with one Godfrey stencil, Bx, By, Ez with the
    // other, along z (NCIGodfreyFilter: stencil lengths 1, 1, 5).  The reference filters the tile box grown by the shape
    // order; here the whole allocation is filtered (zero padding beyond it, as Filter::DoFilter pads): the same values
    // wherever the gather reads.  Like the reference, once per species and step.
    void applyNCIFilter(const ablastr::fields::VectorField& E, const ablastr::fields::VectorField& B) {
        const Backend* be = m_ctx->be;
        if (!be->filter_stencil) throw std::runtime_error("particles.use_fdtd_nci_corr: not in this backend");
        const double half[1] = {0.5};
        auto run = [&](const amrex::MultiFab& src, amrex::MultiFab& dst, const double* sz) {
            check(be->filter_stencil(&src.view(), &dst.view(), half, 1, half, 1, sz, 5, m_ctx->stream), "filter_stencil");
        };
        run(*E[0], *m_ctx->nci_E[0], m_ctx->nci_stencil_exeybz);
        run(*E[2], *m_ctx->nci_E[2], m_ctx->nci_stencil_bxbyez);
        run(*B[1], *m_ctx->nci_B[1], m_ctx->nci_stencil_bxbyez);
        run(*E[1], *m_ctx->nci_E[1], m_ctx->nci_stencil_exeybz);
        run(*B[0], *m_ctx->nci_B[0], m_ctx->nci_stencil_bxbyez);
        run(*B[2], *m_ctx->nci_B[2], m_ctx->nci_stencil_exeybz);
    }

    // :2549-2786
    void PushPX(const amrex::MultiFab& Ex, const amrex::MultiFab& Ey, const amrex::MultiFab& Ez,
                const amrex::MultiFab& Bx, const amrex::MultiFab& By, const amrex::MultiFab& Bz, amrex::Real dt) {
        if (m_tile.numParticles() == 0) return;
        const wxa_field_view E[3] = {Ex.view(), Ey.view(), Ez.view()};
        const wxa_field_view B[3] = {Bx.view(), By.view(), Bz.view()};
        const wxa_grid_geom g = m_ctx->geom(m_ctx->ng_alloc_EB);
        const wxa_particle_view p = m_tile.view();
        stamp_external_time();
        if (m_interior_pushed) {   // PushInterior ran on these particles already: the rest
            m_interior_pushed = false;
            check(m_ctx->be->gather_push_part(&p, E, B, &g, charge, mass, dt, m_ctx->nox,
                                              m_ctx->galerkin_interpolation ? 1 : 0, pusher_algo(), m_ws,
                                              WXA_PART_REST, m_ctx->stream),
                  "gather_push_part");
            return;
        }
        check(m_ctx->be->gather_push(&p, E, B, &g, charge, mass, dt, m_ctx->nox, m_ctx->galerkin_interpolation ? 1 : 0,
                                     pusher_algo(), /*move=*/1, m_ws, m_ctx->stream),
              "gather_push");
    }
    // PushPX of the particles that read no guard point of E and B (the interior tiles of the last sort), issued
    // while the guard exchange of this step is still in flight; the PushPX of this step's Evolve then does the rest
    void PushInterior(ablastr::fields::MultiFabRegister& fields, amrex::Real dt) override {
        if (m_tile.numParticles() == 0 || !m_ctx->be->gather_push_part) return;
        using warpx::fields::FieldType;
        ArmPushSort();
        PhaseTimer t(m_ctx, kGatherAndPush);
        auto Ef = fields.get_alldirs(FieldType::Efield_aux, 0);
        auto Bf = fields.get_alldirs(FieldType::Bfield_aux, 0);
        const wxa_field_view E[3] = {Ef[0]->view(), Ef[1]->view(), Ef[2]->view()};
        const wxa_field_view B[3] = {Bf[0]->view(), Bf[1]->view(), Bf[2]->view()};
        const wxa_grid_geom g = m_ctx->geom(m_ctx->ng_alloc_EB);
        const wxa_particle_view p = m_tile.view();
        stamp_external_time();
        check(m_ctx->be->gather_push_part(&p, E, B, &g, charge, mass, dt, m_ctx->nox, m_ctx->galerkin_interpolation ? 1 : 0,
                                          pusher_algo(), m_ws, WXA_PART_INTERIOR, m_ctx->stream),
              "gather_push_part");
        m_interior_pushed = true;
    }

    // :2368-2516
    void PushP(int /*lev*/, amrex::Real dt, const amrex::MultiFab& Ex, const amrex::MultiFab& Ey,
               const amrex::MultiFab& Ez, const amrex::MultiFab& Bx, const amrex::MultiFab& By,
               const amrex::MultiFab& Bz) override {
        if (m_tile.numParticles() == 0) return;
        const wxa_field_view E[3] = {Ex.view(), Ey.view(), Ez.view()};
        const wxa_field_view B[3] = {Bx.view(), By.view(), Bz.view()};
        const wxa_grid_geom g = m_ctx->geom(m_ctx->ng_alloc_EB);
        const wxa_particle_view p = m_tile.view();
        stamp_external_time();
        check(m_ctx->be->gather_push(&p, E, B, &g, charge, mass, dt, m_ctx->nox, m_ctx->galerkin_interpolation ? 1 : 0,
                                     pusher_algo(), /*move=*/0, m_ws, m_ctx->stream),
              "push_p");
    }
};

// Source/Particles/LaserParticleContainer.{H,cpp}: the antenna, pairs of +-weight macro-particles on a plane,
// moved with the velocity that radiates the requested field and depositing current like any species
// (charge 1); lab frame, Gaussian profile
class LaserParticleContainer : public WarpXParticleContainer {
public:
    LaserParticleContainer(WarpXContext* ctx, const wxa_laser_antenna& la)
        : WarpXParticleContainer(ctx, /*charge=*/1.0, /*mass=*/std::numeric_limits<double>::max()), m_cfg(la) {   // :86-87
        if (!(la.e_max > 0.0) || !(la.wavelength > 0.0)) throw std::runtime_error("laser: e_max and wavelength must be > 0");
        double n = 0, p = 0;
        for (int d = 0; d < 3; ++d) { n += la.direction[d] * la.direction[d]; p += la.polarization[d] * la.polarization[d]; }
        const double sn = 1.0 / std::sqrt(n), sp = 1.0 / std::sqrt(p);                          // :197-214
        for (int d = 0; d < 3; ++d) { m_nvec[d] = la.direction[d] * sn; m_p_X[d] = la.polarization[d] * sp; m_position[d] = la.position[d]; }
        double dp = 0;
        for (int d = 0; d < 3; ++d) dp += m_nvec[d] * m_p_X[d];
        if (std::abs(dp) >= 1.0e-14) throw std::runtime_error("Laser plane vector is not perpendicular to the main polarization vector");
        if (ctx->gamma_boost > 1.0) {                                                            // :183-197
            if (m_nvec[0] * m_nvec[0] + m_nvec[1] * m_nvec[1] + (m_nvec[2] - 1.0) * (m_nvec[2] - 1.0) >= 1.e-12)
                throw std::runtime_error("The Lorentz boost should be in the same direction as the laser propagation");
            // the plane's position along the boost direction in the lab frame, and the antenna in the boosted frame
            m_Z0_lab = m_nvec[0] * m_position[0] + m_nvec[1] * m_position[1] + m_nvec[2] * m_position[2];
            const double Z0_boost = m_Z0_lab / ctx->gamma_boost;
            for (int d = 0; d < 3; ++d) m_position[d] += (Z0_boost - m_Z0_lab) * m_nvec[d];
        }
        m_p_Y[0] = m_nvec[1] * m_p_X[2] - m_nvec[2] * m_p_X[1];                                  // :222 CrossProduct
        m_p_Y[1] = m_nvec[2] * m_p_X[0] - m_nvec[0] * m_p_X[2];
        m_p_Y[2] = m_nvec[0] * m_p_X[1] - m_nvec[1] * m_p_X[0];
    }

    // InitData (:360-559), 3-D: one pair of particles per cell of the antenna plane inside the domain
    void InitData() {
        const auto& dx = m_ctx->dx;
        const double eps = dx[0] * 1e-50;                                                        // ComputeSpacing :727-762
        auto spacing = [&](const double u[3]) {
            return std::min(std::min(dx[0] / (std::abs(u[0]) + eps), dx[1] / (std::abs(u[1]) + eps)),
                            dx[2] / (std::abs(u[2]) + eps));
        };
        m_S_X = spacing(m_p_X);
        m_S_Y = spacing(m_p_Y);
        m_mobility = 0.05 / m_cfg.e_max;                                                         // ComputeWeightMobility :764-781
        m_weight = 8.8541878128e-12 / m_mobility;
        m_weight *= m_S_X * m_S_Y;
        m_mobility = m_mobility / m_ctx->gamma_boost;   // e_max is a lab-frame amplitude (:772-775)
        int plo[2] = {INT_MAX, INT_MAX}, phi[2] = {INT_MIN, INT_MIN};
        for (int c = 0; c < 8; ++c) {                                                           // :418-457
            const double pos[3] = {(c & 1) ? m_ctx->prob_hi[0] : m_ctx->prob_lo[0], (c & 2) ? m_ctx->prob_hi[1] : m_ctx->prob_lo[1],
                                   (c & 4) ? m_ctx->prob_hi[2] : m_ctx->prob_lo[2]};
            double X = 0, Y = 0;
            for (int d = 0; d < 3; ++d) { X += m_p_X[d] * (pos[d] - m_position[d]); Y += m_p_Y[d] * (pos[d] - m_position[d]); }
            const int i = (int)(X / m_S_X), j = (int)(Y / m_S_Y);
            plo[0] = std::min(plo[0], i); plo[1] = std::min(plo[1], j);
            phi[0] = std::max(phi[0], i); phi[1] = std::max(phi[1], j);
        }
        std::vector<double> cols[7];
        for (int j = plo[1]; j <= phi[1]; ++j)
            for (int i = plo[0]; i <= phi[0]; ++i) {
                double pos[3];
                bool inside = true;   // injection box (the domain) strictly contains the point, and it is this brick's
                for (int d = 0; d < 3; ++d) {
                    pos[d] = m_position[d] + (m_S_X * ((double)i + 0.5)) * m_p_X[d] + (m_S_Y * ((double)j + 0.5)) * m_p_Y[d];
                    inside = inside && pos[d] > m_ctx->prob_lo[d] && pos[d] < m_ctx->prob_hi[d] &&
                             pos[d] >= m_ctx->brick_plo[d] && pos[d] < m_ctx->brick_phi[d];
                }
                if (!inside) continue;
                for (int k = 0; k < 2; ++k) {
                    for (int d = 0; d < 3; ++d) cols[d].push_back(pos[d]);
                    cols[3].push_back(k == 0 ? m_weight : -m_weight);
                    for (int d = 4; d < 7; ++d) cols[d].push_back(0.0);
                }
            }
        AppendFromHost(cols);
        if (m_ctx->sort_intervals_on && m_tile.numParticles() > 0) SortParticlesByBin(amrex::IntVect(1));
    }

    // Evolve (:563-713): push the antenna particles with the field to emit at time t, deposit at t_{n+1/2}
    void Evolve(ablastr::fields::MultiFabRegister& fields, int lev, const std::string& current_fp_string, amrex::Real t,
                amrex::Real dt, DtType /*a_dt_type*/ = DtType::Full, bool skip_deposition = false,
                PushType push_type = PushType::Explicit) override {
        using warpx::fields::FieldType;
        if (push_type != PushType::Explicit) throw std::runtime_error("only the explicit push is supported");
        if (current_fp_string != "current_fp") throw std::runtime_error("unknown current field");
        if (m_tile.numParticles() == 0) return;
        wxa_laser_push_params par{};
        for (int d = 0; d < 3; ++d) { par.position[d] = m_position[d]; par.p_X[d] = m_p_X[d]; par.p_Y[d] = m_p_Y[d]; }
        par.mobility = m_mobility;
        par.e_max = m_cfg.e_max; par.wavelength = m_cfg.wavelength; par.waist = m_cfg.waist;
        par.duration = m_cfg.duration; par.t_peak = m_cfg.t_peak; par.focal_distance = m_cfg.focal_distance;
        for (int d = 0; d < 3; ++d) par.nvec[d] = m_nvec[d];
        par.gamma_boost = m_ctx->gamma_boost;
        amrex::Real t_lab = t;
        if (m_ctx->gamma_boost > 1.0)   // the field to emit is the lab-frame one at the antenna's lab-frame time (:574-579)
            t_lab = 1.0 / m_ctx->gamma_boost * t + m_ctx->beta_boost * m_Z0_lab / 299'792'458.;
        {
            PhaseTimer tm(m_ctx, kGatherAndPush);   // "LaserParticleContainer::Evolve::ParticlePush"
            const wxa_particle_view p = m_tile.view();
            check(m_ctx->be->laser_push(&p, &par, t_lab, dt, m_ctx->stream), "laser_push");
        }
        if (m_ctx->sort_now) SortParticlesByBin(amrex::IntVect(1));
        if (!skip_deposition) {
            PhaseTimer tm(m_ctx, kCurrentDeposition);
            auto J = fields.get_alldirs(FieldType::current_fp, lev);
            DepositCurrent(J[0], J[1], J[2], dt, -0.5 * dt);
        }
    }

    // :783-789 nothing to do
    void PushP(int, amrex::Real, const amrex::MultiFab&, const amrex::MultiFab&, const amrex::MultiFab&, const amrex::MultiFab&,
               const amrex::MultiFab&, const amrex::MultiFab&) override {}

private:
    wxa_laser_antenna m_cfg;
    double m_nvec[3], m_p_X[3], m_p_Y[3], m_position[3];
    double m_S_X = 0, m_S_Y = 0, m_mobility = 0, m_weight = 0;
    double m_Z0_lab = 0;
};

// Source/Particles/MultiParticleContainer.{H,cpp}
class MultiParticleContainer {
public:
    explicit MultiParticleContainer(WarpXContext* ctx) : m_ctx(ctx) {}

    int AddSpecies(double charge, double mass) {
        if (m_nlasers > 0) throw std::runtime_error("species must be added before the lasers (the antennas come last)");
        allcontainers.push_back(std::make_unique<PhysicalParticleContainer>(m_ctx, charge, mass));
        return (int)allcontainers.size() - 1;
    }
    // lasers.names: the antennas follow the species in allcontainers (MultiParticleContainer.cpp:95-105)
    int AddLaser(const wxa_laser_antenna& la) {
        auto pc = std::make_unique<LaserParticleContainer>(m_ctx, la);
        pc->InitData();
        allcontainers.push_back(std::move(pc));
        ++m_nlasers;
        return (int)allcontainers.size() - 1;
    }
    WarpXParticleContainer& GetParticleContainer(int i) { return *allcontainers.at(i); }
    int nSpecies() const { return (int)allcontainers.size() - m_nlasers; }
    int nContainers() const { return (int)allcontainers.size(); }

    // MultiParticleContainer.cpp:460-482: zero J once, then every species
    void Evolve(ablastr::fields::MultiFabRegister& fields, int lev, const std::string& current_fp_string,
                amrex::Real t, amrex::Real dt, DtType a_dt_type = DtType::Full, bool skip_deposition = false,
                PushType push_type = PushType::Explicit) {
        using warpx::fields::FieldType;
        if (!skip_deposition) {
            auto J = fields.get_alldirs(FieldType::current_fp, lev);
            if (m_ctx->be->field_set_zero_multi) {   // the three components in one launch
                const wxa_field_view v[3] = {J[0]->view(), J[1]->view(), J[2]->view()};
                check(m_ctx->be->field_set_zero_multi(v, 3, m_ctx->stream), "field_set_zero_multi");
            } else {
                for (int d = 0; d < 3; ++d) J[d]->setVal(0.0, m_ctx->stream);
            }
        }
        for (auto& pc : allcontainers)
            pc->Evolve(fields, lev, current_fp_string, t, dt, a_dt_type, skip_deposition, push_type);
    }
    // :492-500
    void PushP(int lev, amrex::Real dt, const amrex::MultiFab& Ex, const amrex::MultiFab& Ey,
               const amrex::MultiFab& Ez, const amrex::MultiFab& Bx, const amrex::MultiFab& By,
               const amrex::MultiFab& Bz) {
        PhaseTimer t(m_ctx, kOther);  // (de)synchronisation half-pushes, twice per Evolve call
        for (auto& pc : allcontainers) pc->PushP(lev, dt, Ex, Ey, Ez, Bx, By, Bz);
    }
    void PushInterior(ablastr::fields::MultiFabRegister& fields, amrex::Real dt) {
        for (auto& pc : allcontainers) pc->PushInterior(fields, dt);
    }
    // Source/Particles/MultiParticleContainer.cpp (ApplyBoundaryConditions over all species)
    void ApplyBoundaryConditions() {
        for (auto& pc : allcontainers) pc->ApplyBoundaryConditions();
    }
    // :651-654
    void RedistributeLocal(int /*num_ghost*/, BrickComm& comm) {
        for (auto& pc : allcontainers) pc->Redistribute(comm);
    }
    // :615-621
    void SortParticlesByBin(const amrex::IntVect& bin_size) {
        for (auto& pc : allcontainers) pc->SortParticlesByBin(bin_size);
    }

private:
    WarpXContext* m_ctx;
    std::vector<std::unique_ptr<WarpXParticleContainer>> allcontainers;
    int m_nlasers = 0;
};

}  // namespace wxa::host
#endif
