// Particle containers of the host layer: the operator surface that
// Source/Evolve/WarpXEvolve.cpp drives on the explicit FDTD branch
// (WarpXParticleContainer / PhysicalParticleContainer / MultiParticleContainer;
// Source/Particles/WarpXParticleContainer.H:111-509,
// Source/Particles/PhysicalParticleContainer.cpp:1812-2095,2368-2516,2549-2786,
// Source/Particles/MultiParticleContainer.cpp:460-500,615-654).
// One brick per process: a container holds one pure-SoA tile on the device.
#ifndef WXA_HOST_PARTICLES_HPP_
#define WXA_HOST_PARTICLES_HPP_

#include <cstdio>
#include <cstdlib>

#include "BrickComm.hpp"

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
    // boundary.particle_lo/hi resolved to WXA_PBOUNDARY_PERIODIC / _ABSORBING / _REFLECTING
    int32_t particle_bc_lo[3] = {WXA_PBOUNDARY_PERIODIC, WXA_PBOUNDARY_PERIODIC, WXA_PBOUNDARY_PERIODIC};
    int32_t particle_bc_hi[3] = {WXA_PBOUNDARY_PERIODIC, WXA_PBOUNDARY_PERIODIC, WXA_PBOUNDARY_PERIODIC};
    bool any_particle_wall = false;
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
        if (m_nretired > 0) {
            int64_t live = np;
            check(m_ctx->be->sort_live_count(m_ws, &live, m_ctx->stream), "sort_live_count");
            if (live != np - m_nretired) throw std::runtime_error("SortParticlesByBin: retired-particle count mismatch");
            m_tile.resize(live);
            m_nretired = 0;
        }
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
        const int none[3] = {0, 0, 0};
        int split[3];
        bool any_split = false;
        for (int d = 0; d < 3; ++d) { split[d] = comm.self_periodic(d) ? 0 : 1; any_split = any_split || split[d]; }
        const int64_t np0 = m_tile.numParticles();
        if (!any_split) {
            if (np0 > 0) {
                const wxa_particle_view p = m_tile.view();
                check(be->enforce_periodic(&p, m_ctx->prob_lo.data(), m_ctx->prob_hi.data(), periodic, m_ctx->stream),
                      "enforce_periodic");
            }
            return;
        }
        // one scan of the whole tile: wrap + six leaver lists (by first split direction)
        struct Segment { const int32_t* list; int64_t n; };
        std::vector<Segment> seg[6];
        const int64_t cap = np0 / 4 + 4096;
        m_lists.reserve(sizeof(int32_t) * 6 * (size_t)cap);
        int32_t* lists = static_cast<int32_t*>(m_lists.p);
        int64_t cnt[6] = {0, 0, 0, 0, 0, 0};
        if (np0 > 0) {
            const wxa_particle_view p = m_tile.view();
            check(be->wrap_and_classify(&p, 0, np0, m_ctx->prob_lo.data(), m_ctx->prob_hi.data(), periodic,
                                        m_ctx->brick_plo.data(), m_ctx->brick_phi.data(), split, lists, cap, cnt, m_ws,
                                        m_ctx->stream),
                  "wrap_and_classify");
            for (int c = 0; c < 6; ++c) {
                if (cnt[c] > cap) throw std::runtime_error("Redistribute: leaver list overflow");
                if (cnt[c] > 0) seg[c].push_back({lists + c * cap, cnt[c]});
            }
        }
        for (int d = 0; d < 3; ++d) {
            if (!split[d]) continue;
            int64_t nsend[2] = {0, 0};
            for (int s = 0; s < 2; ++s)
                for (const Segment& g : seg[2 * d + s]) nsend[s] += g.n;
            int64_t from_plus = 0, from_minus = 0;
            comm.exchange_counts(d, nsend[0], nsend[1], from_plus, from_minus);
            const int64_t nrecv = from_plus + from_minus;
            // staging: one message per peer = 8 SoA rows (7 reals + idcpu) of n entries
            m_sendbuf.reserve(64 * (size_t)std::max<int64_t>(nsend[0] + nsend[1], 1));
            m_recvbuf.reserve(64 * (size_t)std::max<int64_t>(nrecv, 1));
            char* msg[2] = {static_cast<char*>(m_sendbuf.p), static_cast<char*>(m_sendbuf.p) + 64 * nsend[0]};
            const wxa_particle_view p = m_tile.view();
            for (int s = 0; s < 2; ++s) {
                int64_t off = 0;
                for (const Segment& g : seg[2 * d + s]) {
                    check(be->pack_leavers(&p, g.list, g.n, msg[s], nsend[s], off, /*retire=*/1,
                                           m_ctx->brick_plo.data(), m_ctx->brick_phi.data(), m_ctx->stream),
                          "pack_leavers");
                    off += g.n;
                }
            }
            m_nretired += nsend[0] + nsend[1];
            char* rb = static_cast<char*>(m_recvbuf.p);
            char* rmsg_plus = rb;
            char* rmsg_minus = rb + 64 * from_plus;
            comm.exchange_raw(d, msg[0], 64 * nsend[0], msg[1], 64 * nsend[1], rmsg_plus, 64 * from_plus, rmsg_minus,
                              64 * from_minus, m_ctx->stream);
            if (nrecv == 0) continue;
            const int64_t n0 = m_tile.numParticles();
            m_tile.resize(n0 + nrecv);   // appends behind the sorted part (reallocation keeps the contents)
            auto unpack_msg = [&](const char* srcb, int64_t off, int64_t n) {
                for (int c = 0; c < 7; ++c)
                    be->memcpy_async(m_tile.comp(c) + off, srcb + 8 * (int64_t)c * n, 8 * (size_t)n, m_ctx->stream);
                be->memcpy_async(m_tile.idcpu() + off, srcb + 8 * 7 * n, 8 * (size_t)n, m_ctx->stream);
            };
            if (from_plus > 0) unpack_msg(rmsg_plus, n0, from_plus);
            if (from_minus > 0) unpack_msg(rmsg_minus, n0 + from_plus, from_minus);
            // arrivals may have to travel on along the remaining directions (edges and corners)
            int later[3] = {0, 0, 0};
            bool any_later = false;
            for (int e = d + 1; e < 3; ++e) { later[e] = split[e]; any_later = any_later || split[e]; }
            if (any_later) {
                DeviceBuffer& al = m_arrival_lists[d];
                al.reserve(sizeof(int32_t) * 6 * (size_t)nrecv);
                int32_t* alist = static_cast<int32_t*>(al.p);
                int64_t acnt[6];
                const wxa_particle_view pa = m_tile.view();
                check(be->wrap_and_classify(&pa, n0, nrecv, m_ctx->prob_lo.data(), m_ctx->prob_hi.data(), none,
                                            m_ctx->brick_plo.data(), m_ctx->brick_phi.data(), later, alist, nrecv, acnt,
                                            m_ws, m_ctx->stream),
                      "wrap_and_classify(arrivals)");
                for (int c = 0; c < 6; ++c)
                    if (acnt[c] > 0) seg[c].push_back({alist + c * nrecv, acnt[c]});
            }
        }
        be->stream_sync(m_ctx->stream);
    }

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
    int64_t m_nretired = 0;            // retired by Redistribute since the last sort (still in the tile)
    void* m_ws = nullptr;

public:
    amrex::ParticleReal charge, mass;
};

class PhysicalParticleContainer : public WarpXParticleContainer {
public:
    using WarpXParticleContainer::WarpXParticleContainer;

    // Source/Particles/PhysicalParticleContainer.cpp:1812-2095: PushPX then DepositCurrent
    void Evolve(ablastr::fields::MultiFabRegister& fields, int lev, const std::string& current_fp_string,
                amrex::Real /*t*/, amrex::Real dt, DtType /*a_dt_type*/ = DtType::Full, bool skip_deposition = false,
                PushType push_type = PushType::Explicit) override {
        using warpx::fields::FieldType;
        if (push_type != PushType::Explicit) throw std::runtime_error("only the explicit push is supported");
        if (current_fp_string != "current_fp") throw std::runtime_error("unknown current field");
        auto E = fields.get_alldirs(FieldType::Efield_aux, lev);
        auto B = fields.get_alldirs(FieldType::Bfield_aux, lev);
        auto J = fields.get_alldirs(FieldType::current_fp, lev);
        {
            PhaseTimer t(m_ctx, kGatherAndPush);  // "PhysicalParticleContainer::Evolve::GatherAndPush"
            PushPX(*E[0], *E[1], *E[2], *B[0], *B[1], *B[2], dt);
        }
        // Cell sort (amrex SortParticlesByBin, called by the reference from
        // HandleParticlesAtBoundaries, WarpXEvolve.cpp:575-580).  Sorting only permutes the
        // tile, so it is placed here, between push and deposition: the deposition then sees
        // positions that match the sort exactly (every stencil inside its LDS tile) and so does
        // the next step's gather.
        if (m_ctx->sort_now) {
            PhaseTimer t(m_ctx, kRedistribute);
            SortParticlesByBin(amrex::IntVect(1));
        }
        if (!skip_deposition) {
            PhaseTimer t(m_ctx, kCurrentDeposition);  // "...::DepositCurrent::CurrentDeposition"
            // :2029 relative_time = -0.5*dt: deposit at the half step
            DepositCurrent(J[0], J[1], J[2], dt, -0.5 * dt);
        }
    }

    // :2549-2786
    void PushPX(const amrex::MultiFab& Ex, const amrex::MultiFab& Ey, const amrex::MultiFab& Ez,
                const amrex::MultiFab& Bx, const amrex::MultiFab& By, const amrex::MultiFab& Bz, amrex::Real dt) {
        if (m_tile.numParticles() == 0) return;
        const wxa_field_view E[3] = {Ex.view(), Ey.view(), Ez.view()};
        const wxa_field_view B[3] = {Bx.view(), By.view(), Bz.view()};
        const wxa_grid_geom g = m_ctx->geom(m_ctx->ng_alloc_EB);
        const wxa_particle_view p = m_tile.view();
        check(m_ctx->be->gather_push(&p, E, B, &g, charge, mass, dt, m_ctx->nox, m_ctx->galerkin_interpolation ? 1 : 0,
                                     (int)m_ctx->particle_pusher_algo, /*move=*/1, m_ws, m_ctx->stream),
              "gather_push");
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
        check(m_ctx->be->gather_push(&p, E, B, &g, charge, mass, dt, m_ctx->nox, m_ctx->galerkin_interpolation ? 1 : 0,
                                     (int)m_ctx->particle_pusher_algo, /*move=*/0, m_ws, m_ctx->stream),
              "push_p");
    }
};

// Source/Particles/MultiParticleContainer.{H,cpp}
class MultiParticleContainer {
public:
    explicit MultiParticleContainer(WarpXContext* ctx) : m_ctx(ctx) {}

    int AddSpecies(double charge, double mass) {
        allcontainers.push_back(std::make_unique<PhysicalParticleContainer>(m_ctx, charge, mass));
        return (int)allcontainers.size() - 1;
    }
    WarpXParticleContainer& GetParticleContainer(int i) { return *allcontainers.at(i); }
    int nSpecies() const { return (int)allcontainers.size(); }

    // MultiParticleContainer.cpp:460-482: zero J once, then every species
    void Evolve(ablastr::fields::MultiFabRegister& fields, int lev, const std::string& current_fp_string,
                amrex::Real t, amrex::Real dt, DtType a_dt_type = DtType::Full, bool skip_deposition = false,
                PushType push_type = PushType::Explicit) {
        using warpx::fields::FieldType;
        if (!skip_deposition) {
            for (int d = 0; d < 3; ++d)
                fields.get(FieldType::current_fp, ablastr::fields::Direction{d}, lev)->setVal(0.0, m_ctx->stream);
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
};

}  // namespace wxa::host
#endif
