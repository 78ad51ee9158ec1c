// Back-transformed diagnostics: lab-frame snapshots (fields and particles) assembled slice by slice from a boosted-frame run.
// Restates Source/Diagnostics/BTDiagnostics.cpp, Source/Diagnostics/ComputeDiagFunctors/BackTransformFunctor.cpp and
// BackTransformParticleFunctor.cpp for one level, boost and window along z (one box per rank: a brick
// keeps its own x-y part of every snapshot, see ComputeAndPack):
//   DerivedInitData (:66-205), InitializeBufferData (:333-506): lab time, lab-frame extent and index box of snapshot i;
//   PrepareBufferData / UpdateBufferData (:755-798), GetZSliceInDomainFlag (:999-1018), k_index_zlab (:892-905):
//     where the snapshot's plane z_lab(t) sits in the boosted frame at this step and which lab-frame index it fills;
//   BackTransformFunctor::operator() (:49-149): the cell-centred fields (CellCenterFunctor: Ex .. Bz, jx .. jz, rho)
//     sliced at z_boost with linear interpolation between the two nearest cell centres (amrex::get_slice_data with
//     interpolate = true -- AMReX is not on disk: the rule is restated from its documented behaviour and from the
//     half-cell exclusion of GetZSliceInDomainFlag; pinned since round 4 by the reference's golden file
//     test_3d_laser_acceleration_btd.json, tests/golden/laser_acceleration_btd_3d_checksums.json), LorentzTransformZ
//     (:246-317), copy to index k_lab;
//   BackTransformParticleFunctor::operator() (BackTransformParticleFunctor.cpp:76-152): PackParticles below, the selection
//     and the transform on the device (host/btd_kernels.hip).
//   DefineFieldBufferMultiFab (:916-974), DoDump (:294-320), Flush (:1027-1137), MergeBuffersForPlotfile (:1146-1314):
//     with SetFlush a snapshot is assembled buffer_size slices at a time and every full buffer goes to disk as one more
//     grid of the snapshot's plotfile (flush below); without it a snapshot is kept whole in host memory.
// What is not here: mesh refinement, RZ, openPMD.
// Output stage, not on the step path: the fields are copied to the host (Plotfile.hpp does the same).
#ifndef WXA_HOST_BTDIAGNOSTICS_HPP_
#define WXA_HOST_BTDIAGNOSTICS_HPP_

#include <array>
#include <cmath>
#include <stdexcept>
#include <vector>

#include "amrex_shim.hpp"
#include "backend.hpp"
#include "BrickComm.hpp"
#include "PlotfileFormat.hpp"

namespace wxa::host {

class BTDiagnostics {
public:
    static constexpr int NCOMP = 10;   // Ex Ey Ez Bx By Bz jx jy jz rho (BackTransformFunctor.cpp:190-201)

    struct Snapshot {
        double t_lab = 0;                       // m_t_lab
        double zlo_lab = 0, zhi_lab = 0;        // m_snapshot_domain_lab along z
        int ksmall = 0, kbig = 0;               // m_snapshot_box along z
        double z_boost = 0, z_lab = 0;          // m_current_z_boost / m_current_z_lab
        int counter = 0, last_valid = 0, full = 0;
        int n[3] = {0, 0, 0};                   // cells of the snapshot array (x, y, z): this brick's x-y cells, all of z
        int ilo[2] = {0, 0};                    // ... and where its first cell sits in the snapshot's index box
        int glo[2] = {0, 0}, gn[2] = {0, 0};    // the snapshot's x-y box as a whole (all bricks)
        std::vector<double> data;               // [comp][k][j][i], zero until a slice arrives
        // back-transformed particles per species: rows x y z w ux uy uz (lab frame), in arrival order
        std::vector<std::array<std::vector<double>, 7>> particles;
        // the planes `data` holds: all of the snapshot, or (SetFlush) the buffer being filled
        int data_k0 = 0, data_nz = 0;
        // m_buffer_box along z, m_buffer_counter, m_buffer_k_index_hi; one PlotGrid per flushed buffer (m_buffer_flush_counter)
        int buffer_klo = 0, buffer_khi = -1, buffer_counter = 0, buffer_k_index_hi = 0;
        std::vector<PlotGrid> flushed;
        std::vector<std::vector<ParticleGrid>> flushed_particles;   // per species
    };

    BTDiagnostics(int num_snapshots, double dt_snapshots_lab, int buffer_size)
        : m_num(num_snapshots), m_dt_snap(dt_snapshots_lab), m_buffer_size(buffer_size) {
        if (num_snapshots < 1 || !(dt_snapshots_lab > 0.0) || buffer_size < 1)
            throw std::runtime_error("BackTransformed diagnostic: num_snapshots_lab >= 1, dt_snapshots_lab > 0, buffer_size >= 1");
    }

    // <diag>.file_prefix / file_min_digits: snapshot i is written to <prefix><i, min_digits>/ buffer by buffer, and only the
    // buffer being filled is kept in memory.  Before the first slice arrives.
    void SetFlush(const std::string& file_prefix, int file_min_digits) {
        for (const Snapshot& s : m_snap)
            if (s.counter > 0) throw std::runtime_error("BackTransformed diagnostic: SetFlush comes before the first slice");
        if (file_prefix.empty() || file_min_digits < 1) throw std::runtime_error("BackTransformed diagnostic: file_prefix / file_min_digits");
        m_flush_prefix = file_prefix;
        m_file_min_digits = file_min_digits;
        for (Snapshot& s : m_snap) { s.data.clear(); s.data.shrink_to_fit(); s.data_k0 = 0; s.data_nz = 0; }
    }
    bool flushing() const { return !m_flush_prefix.empty(); }
    std::string snapshot_path(int i) const { return numbered(m_flush_prefix, i, m_file_min_digits); }
    void SetSpeciesNames(const std::vector<std::string>& names) { m_species_names = names; }

    int num_snapshots() const { return m_num; }
    const Snapshot& snapshot(int i) const { return m_snap.at((size_t)i); }

    // c dt / (beta gamma): lab-frame distance between the slices of consecutive steps (:885-890)
    double dz_lab(double dt) const { return kC * dt * 1.0 / m_beta * 1.0 / m_gamma; }

    // InitializeBufferData for every snapshot; called once, when the diagnostic is added
    template <class WX>
    void Init(WX& wx) {
        const auto& ctx = wx.context();
        if (!(ctx.gamma_boost > 1.0)) throw std::runtime_error("BackTransformed diagnostic: needs warpx.gamma_boost > 1");
        if (wx.moving_window_dir != 2) throw std::runtime_error("BackTransformed diagnostic: boost and window along z");
        m_gamma = ctx.gamma_boost;
        m_beta = std::sqrt(1.0 - 1.0 / (m_gamma * m_gamma));
        m_mw_beta = wx.do_moving_window ? wx.moving_window_v / kC : 0.0;
        const double t_new = wx.gett_new();
        const double dt = wx.getdt();
        const double dzl = dz_lab(dt);
        const double bmw_v = (m_mw_beta - m_beta) / (1.0 - m_beta * m_mw_beta);   // :341-342
        m_snap.assign((size_t)m_num, Snapshot{});
        m_xbuf.be = ctx.be;
        for (int i = 0; i < m_num; ++i) {
            Snapshot& s = m_snap[(size_t)i];
            const double zmax_boost = ctx.prob_hi[2];
            s.t_lab = i * m_dt_snap + m_gamma * m_beta * zmax_boost / kC;              // :346-347
            // diag domain = the whole boosted-frame domain (m_lo / m_hi default), re-derived from the index box (:391-397)
            int lo[3], hi[3];
            double dlo[3], dhi[3];
            for (int d = 0; d < 3; ++d) {
                const double cs = ctx.dx[d];
                const int lo_index = (int)std::floor((ctx.prob_lo[d] - ctx.prob_lo[d]) / cs);
                lo[d] = std::max(0, lo_index);
                const int hi_index = (int)std::ceil((ctx.prob_hi[d] - ctx.prob_lo[d]) / cs);
                hi[d] = std::max(0, hi_index) - 1;
                if (hi[d] <= lo[d]) hi[d] = lo[d] + 1;
                dlo[d] = ctx.prob_lo[d] + lo[d] * cs;
                dhi[d] = ctx.prob_lo[d] + (hi[d] + 1) * cs;
            }
            // lab-frame extent along z (:401-404), literally: the reference multiplies this ratio of betas by the time
            // (a restart term; t_new is 0 when a diagnostic is set up at the start of a run)
            const double zmin_lab = (dlo[2] - bmw_v * t_new) * (1.0 - m_beta * m_mw_beta) * m_gamma;
            const double zmax_lab = (dhi[2] - bmw_v * t_new) * (1.0 - m_beta * m_mw_beta) * m_gamma;
            s.z_boost = z_boost_of(s.t_lab, t_new);
            s.z_lab = z_lab_of(s.t_lab, t_new);
            const int nz_lab = std::max(0, (int)std::floor((zmax_lab - zmin_lab) / dzl));          // :430-434
            const int nx_lab = std::max(0, (int)std::floor((dhi[0] - dlo[0]) / ctx.dx[0]));
            const int ny_lab = std::max(0, (int)std::floor((dhi[1] - dlo[1]) / ctx.dx[1]));
            const int max_buffers = (int)std::ceil((double)nz_lab / (double)m_buffer_size);        // :464-466
            const int nzs = max_buffers * m_buffer_size;                                            // :469
            s.zlo_lab = zmin_lab + wx.moving_window_v * s.t_lab;                                    // :472-475
            s.zhi_lab = zmax_lab + wx.moving_window_v * s.t_lab;
            s.zhi_lab = s.zhi_lab + 0.5 * dzl;                                                      // :478-480
            s.zlo_lab = s.zhi_lab - nzs * dzl;                                                      // :481-484
            const int kindex_hi = (int)std::floor((s.zhi_lab - (s.zlo_lab + 0.5 * dzl)) / dzl);     // :489-494
            s.kbig = kindex_hi;
            s.ksmall = kindex_hi - (nzs - 1);
            // this brick's share: its own cells in x and y (the reference keeps one buffer box per boosted-frame box)
            const int nxy_lab[2] = {nx_lab, ny_lab};
            for (int d = 0; d < 2; ++d) {
                s.glo[d] = lo[d];
                s.gn[d] = nxy_lab[d];
                s.ilo[d] = ctx.brick_box.lo[d];
                s.n[d] = std::max(0, std::min(ctx.brick_box.hi[d] + 1, lo[d] + nxy_lab[d]) - ctx.brick_box.lo[d]);
            }
            s.n[2] = nzs;
            s.buffer_k_index_hi = s.kbig;                                                            // :502-504
            if (!flushing()) {
                s.data_k0 = s.ksmall;
                s.data_nz = nzs;
                s.data.assign((size_t)NCOMP * (size_t)nzs * (size_t)s.n[1] * (size_t)s.n[0], 0.0);
            }
        }
    }

    // The particle half (BackTransformParticleFunctor::operator(), BackTransformParticleFunctor.cpp:76-152) for one species,
    // called by its container right after PushPX with the attributes saved before it (CopyParticleAttribs,
    // PhysicalParticleContainer.cpp:2626-2629): the particles that the snapshot's plane met during this step, in the lab
    // frame.  The reference does this in the BackTransformed pass of the diagnostics, after the step's field solve and
    // before the window moves (WarpXEvolve.cpp:241); here the tile is re-sorted between the push and the deposition, which
    // would separate the particles from their saved attributes -- same positions, momenta, plane and domain.
    template <class CTX>
    void PackParticles(const CTX& ctx, int species, const wxa_particle_view& p, const double* const old6[6], double t_new,
                       double dt, DeviceBuffer& scratch) {
        if (!ctx.be->btd_select_particles) throw std::runtime_error("BackTransformed diagnostic: particles not in this backend");
        for (Snapshot& s : m_snap) {
            if ((int)s.particles.size() <= species) s.particles.resize((size_t)species + 1);
            Snapshot now = s;
            now.z_boost = z_boost_of(s.t_lab, t_new);
            now.z_lab = z_lab_of(s.t_lab, t_new);
            if (!slice_in_domain(now, ctx) || s.full) continue;           // m_perform_backtransform (:166-176)
            const double zb_old = z_boost_of(s.t_lab, t_new - dt);        // m_old_z_boost: the plane at the previous step
            int64_t cap = p.np / 16 + 1024, n = 0;
            for (;;) {
                scratch.reserve(sizeof(double) * 7 * (size_t)cap);
                if (ctx.be->btd_select_particles(&p, old6, now.z_boost, zb_old, t_new, dt, s.t_lab, m_gamma,
                                                 static_cast<double*>(scratch.p), cap, &n, ctx.stream) != 0)
                    throw std::runtime_error("BackTransformed diagnostic: particle selection failed");
                if (n <= cap) break;
                cap = n;
            }
            if (n == 0) continue;
            std::vector<double> host((size_t)7 * (size_t)cap);
            if (ctx.be->memcpy_d2h(host.data(), scratch.p, sizeof(double) * host.size()) != 0)
                throw std::runtime_error("BackTransformed diagnostic: device copy failed");
            for (int c = 0; c < 7; ++c)
                s.particles[(size_t)species][(size_t)c].insert(s.particles[(size_t)species][(size_t)c].end(),
                                                               host.begin() + (std::ptrdiff_t)((size_t)c * cap),
                                                               host.begin() + (std::ptrdiff_t)((size_t)c * cap + (size_t)n));
        }
    }

    // Diagnostics::ComputeAndPack (Diagnostics.cpp:558-608) for this diagnostic, once per step: after the field solve and
    // the time update, BEFORE MoveWindow and the particle boundaries (WarpXEvolve.cpp:241, MultiDiagnostics.cpp:81-96)
    template <class WX>
    void ComputeAndPack(WX& wx) {
        const auto& ctx = wx.context();
        const double t_new = wx.gett_new();
        const double dzl = dz_lab(wx.getdt());
        // PrepareBufferData
        for (Snapshot& s : m_snap) {
            s.z_boost = z_boost_of(s.t_lab, t_new);
            s.z_lab = z_lab_of(s.t_lab, t_new);
        }
        bool any = false;
        std::vector<char> in_domain(m_snap.size(), 0);
        for (size_t i = 0; i < m_snap.size(); ++i) {
            in_domain[i] = slice_in_domain(m_snap[i], ctx) ? 1 : 0;
            any = any || (in_domain[i] && !m_snap[i].full);
        }
        // DefineFieldBufferMultiFab: an empty buffer starts below the last one as soon as the plane's index is in the snapshot
        for (Snapshot& s : m_snap) {
            const int k_lab = k_index_zlab(s, dzl);
            if (k_lab < s.ksmall || k_lab > s.kbig || s.buffer_counter != 0 || s.full) continue;
            s.buffer_khi = s.buffer_k_index_hi;
            s.buffer_klo = s.buffer_khi - m_buffer_size + 1;
            if (flushing()) {
                s.data_k0 = s.buffer_klo;
                s.data_nz = m_buffer_size;
                s.data.assign((size_t)NCOMP * (size_t)m_buffer_size * (size_t)s.n[1] * (size_t)s.n[0], 0.0);
            }
        }
        if (any) {
            // PrepareFieldDataForOutput: the ten cell-centred components of the whole (single) box, on the host
            using warpx::fields::FieldType;
            using ablastr::fields::Direction;
            int nc[3] = {0, 0, 0};
            std::vector<std::vector<double>> cc;
            const FieldType fts[3] = {FieldType::Efield_fp, FieldType::Bfield_fp, FieldType::current_fp};
            wx.sync_stream();
            for (const FieldType ft : fts)
                for (int d = 0; d < 3; ++d) cc.push_back(cell_centered(ctx.be, *wx.fields().get(ft, Direction{d}, 0), nc));
            cc.push_back(cell_centered(ctx.be, wx.ComputeRho(), nc));
            // bricks stacked along z: the interpolation across a brick face needs the neighbour's first / last plane of
            // cell-centred values (m_cell_centered_data has one guard cell, BTDiagnostics.cpp:516-527, filled by the
            // FillBoundary of PrepareFieldDataForOutput, :824-828)
            if (wx.comm().exchanges(2)) exchange_guard_planes(wx, cc, nc);
            for (size_t i = 0; i < m_snap.size(); ++i) {
                Snapshot& s = m_snap[i];
                if (!in_domain[i] || s.full) continue;                     // m_perform_backtransform (:152-164)
                const int k_lab = k_index_zlab(s, dzl);
                if (k_lab < s.ksmall || k_lab > s.kbig) continue;          // outside the snapshot's box: no buffer there
                back_transform_slice(s, k_lab, cc, nc, ctx);
            }
        }
        // UpdateBufferData, then what DoDump / Flush do to the flags (:294-320, :907-914)
        for (size_t i = 0; i < m_snap.size(); ++i) {
            Snapshot& s = m_snap[i];
            if (in_domain[i]) { ++s.counter; ++s.buffer_counter; }
            const int k_lab = k_index_zlab(s, dzl);
            if (k_lab == s.ksmall) s.last_valid = 1;
            // DoDump: the plane has reached the bottom of the buffer, or of the snapshot
            if (flushing() && !s.full && s.buffer_khi >= s.buffer_klo && s.buffer_counter > 0 &&
                (k_lab == s.buffer_klo || s.last_valid == 1))
                flush(wx, (int)i);
            if (s.last_valid == 1) s.full = 1;
        }
    }

    // The forced flush after the last step (Diagnostics::FilterComputePackFlushLastTimestep, DoDump's force_flush): the
    // buffers that have received slices since their last flush go to disk as they are
    template <class WX>
    void FlushLast(WX& wx) {
        if (!flushing()) return;
        for (size_t i = 0; i < m_snap.size(); ++i)
            if (!m_snap[i].full && m_snap[i].buffer_counter > 0 && m_snap[i].buffer_khi >= m_snap[i].buffer_klo) flush(wx, (int)i);
    }

private:
    static constexpr double kC = 299792458.0;

    // Flush + MergeBuffersForPlotfile for snapshot i: the buffer becomes grid number m_buffer_flush_counter of the
    // snapshot's plotfile (Level_0/Cell_D_<n>, <species>/Level_0/DATA_<n>), the headers are rewritten for the grids so
    // far (the reference interleaves the buffer's headers into the snapshot's, :1316-1434), the buffer is emptied and the
    // next one starts below it (:1127-1137).  This brick's share, like wxa_sim_btd_write_plotfile.
    template <class WX>
    void flush(WX& wx, int i) {
        {
            const int* nb = wx.comm().nbricks();
            if (nb[0] * nb[1] * nb[2] > 1) { flush_bricks(wx, i); return; }
        }
        Snapshot& s = m_snap[(size_t)i];
        const auto& ctx = wx.context();
        const std::string dir = snapshot_path(i);
        const int id = (int)s.flushed.size();
        if (id == 0) { make_dirs(dir); make_dir(dir + "/Level_0"); }
        PlotGrid g;
        g.lo[0] = s.ilo[0]; g.lo[1] = s.ilo[1]; g.lo[2] = s.buffer_klo;
        g.hi[0] = s.ilo[0] + s.n[0] - 1; g.hi[1] = s.ilo[1] + s.n[1] - 1; g.hi[2] = s.buffer_khi;
        g.fab_file = numbered("Cell_D_", id, 5);
        const size_t npts = (size_t)s.n[0] * s.n[1] * (size_t)s.data_nz;
        std::vector<const double*> comps;
        for (int c = 0; c < NCOMP; ++c) comps.push_back(s.data.data() + (size_t)c * npts);
        write_fab(dir, g, comps);
        s.flushed.push_back(g);
        // the particles the plane has met while this buffer was filled, momenta as m u (FlushFormatPlotfile.cpp:412-424)
        const int nspecies = (int)s.particles.size();   // empty without <diag>.write_species
        if ((int)s.flushed_particles.size() < nspecies) s.flushed_particles.resize((size_t)nspecies);
        for (int sp = 0; sp < nspecies; ++sp) {
            const std::string name = sp < (int)m_species_names.size() ? m_species_names[(size_t)sp] : "species" + std::to_string(sp);
            if (id == 0) { make_dir(dir + "/" + name); make_dir(dir + "/" + name + "/Level_0"); }
            auto& rows = s.particles[(size_t)sp];
            const size_t np = rows[0].size();
            const double mass = wx.GetPartContainer().GetParticleContainer(sp).mass;
            std::vector<double> rec(7 * np);
            for (size_t q = 0; q < np; ++q)
                for (int c = 0; c < 7; ++c) rec[7 * q + (size_t)c] = c >= 4 ? rows[(size_t)c][q] * mass : rows[(size_t)c][q];
            if (np) write_particle_records(dir, name, id, rec, np);   // no file for an empty grid (:1274, :1290)
            for (auto& r : rows) { r.clear(); r.shrink_to_fit(); }
            ParticleGrid pg;
            for (int d = 0; d < 3; ++d) { pg.lo[d] = g.lo[d]; pg.hi[d] = g.hi[d]; }
            pg.which = id;
            pg.count = (int64_t)np;
            s.flushed_particles[(size_t)sp].push_back(pg);
            write_species_headers(dir, name, s.flushed_particles[(size_t)sp]);
        }
        // the snapshot's Header covers the grids written so far (InterleaveBufferAndSnapshotHeader, :1316-1359)
        int dom_lo[3] = {g.lo[0], g.lo[1], g.lo[2]}, dom_hi[3] = {g.hi[0], g.hi[1], g.hi[2]};
        for (const PlotGrid& q : s.flushed) { dom_lo[2] = std::min(dom_lo[2], q.lo[2]); dom_hi[2] = std::max(dom_hi[2], q.hi[2]); }
        const double dzl = (s.zhi_lab - s.zlo_lab) / s.n[2];
        const double dx[3] = {ctx.dx[0], ctx.dx[1], dzl};
        const double rlo[3] = {ctx.prob_lo[0] + dom_lo[0] * dx[0], ctx.prob_lo[1] + dom_lo[1] * dx[1],
                               s.zlo_lab + (dom_lo[2] - s.ksmall) * dzl};
        const double rhi[3] = {ctx.prob_lo[0] + (dom_hi[0] + 1) * dx[0], ctx.prob_lo[1] + (dom_hi[1] + 1) * dx[1],
                               s.zlo_lab + (dom_hi[2] + 1 - s.ksmall) * dzl};
        static const char* comp_names[NCOMP] = {"Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz", "rho"};
        write_cell_headers(dir, std::vector<std::string>(comp_names, comp_names + NCOMP), s.flushed, dom_lo, dom_hi, rlo, rhi,
                           dx, s.t_lab, wx.getistep());
        s.buffer_counter = 0;
        s.buffer_k_index_hi = s.buffer_klo - 1;
        s.buffer_khi = s.buffer_klo - 1;   // no buffer until the next one is defined
        s.data.clear();
        s.data_nz = 0;
    }

    // The same on several bricks: ONE plotfile per snapshot, the buffer one grid of it, as on one brick.  Every brick holds
    // its x-y share of the buffer with the slices whose plane lay in its cells (zeros in the planes another brick of its
    // column filled); brick 0 collects the shares (GatherRealToRoot), adds them into the buffer's whole x-y box and writes
    // grid, particle file and headers.  The flush conditions are global (the plane's index, the slice counters), so the
    // bricks arrive here in the same step.  (Until round 5 every brick wrote a plotfile of its own, <prefix>brick<r>_<i>/,
    // and the snapshot of a run with bricks stacked along z was the sum of them.)
    template <class WX>
    void flush_bricks(WX& wx, int i) {
        Snapshot& s = m_snap[(size_t)i];
        const auto& ctx = wx.context();
        BrickComm& comm = wx.comm();
        const Backend* be = ctx.be;
        const int* nb = comm.nbricks();
        const int nranks = nb[0] * nb[1] * nb[2], me = comm.rank_of(comm.coord());
        const bool root = me == 0;
        const std::string dir = snapshot_path(i);
        const int id = (int)s.flushed.size();
        const int nz = s.data_nz;
        const int bc[2] = {ctx.brick_box.hi[0] - ctx.brick_box.lo[0] + 1, ctx.brick_box.hi[1] - ctx.brick_box.lo[1] + 1};
        // brick r's share of the snapshot's x-y box (the formula of the set-up above, for every brick)
        auto share_of = [&](int r, int lo[2], int n[2]) {
            const int c[2] = {r % nb[0], (r / nb[0]) % nb[1]};
            for (int d = 0; d < 2; ++d) {
                lo[d] = wx.m_dom_lo[d] + c[d] * bc[d];
                n[d] = std::max(0, std::min(lo[d] + bc[d], s.glo[d] + s.gn[d]) - lo[d]);
            }
        };
        {
            int lo[2], n[2];
            share_of(me, lo, n);
            if (lo[0] != s.ilo[0] || lo[1] != s.ilo[1] || n[0] != s.n[0] || n[1] != s.n[1])
                throw std::runtime_error("BackTransformed diagnostic: brick numbering mismatch");
        }
        wx.sync_stream();
        std::vector<int64_t> count((size_t)nranks);
        for (int r = 0; r < nranks; ++r) {
            int lo[2], n[2];
            share_of(r, lo, n);
            count[(size_t)r] = (int64_t)NCOMP * nz * n[1] * n[0];
        }
        std::vector<std::vector<double>> shares;
        GatherRealToRoot(comm, be, s.data.data(), count, shares, ctx.stream, &m_gather_staging, gather_piece());
        PlotGrid g;
        g.lo[0] = s.glo[0]; g.lo[1] = s.glo[1]; g.lo[2] = s.buffer_klo;
        g.hi[0] = s.glo[0] + s.gn[0] - 1; g.hi[1] = s.glo[1] + s.gn[1] - 1; g.hi[2] = s.buffer_khi;
        g.fab_file = numbered("Cell_D_", id, 5);
        if (root) {
            if (id == 0) { make_dirs(dir); make_dir(dir + "/Level_0"); }
            const size_t gpl = (size_t)s.gn[0] * s.gn[1], gpts = gpl * (size_t)nz;
            std::vector<double> whole((size_t)NCOMP * gpts, 0.0);
            for (int r = 0; r < nranks; ++r) {
                int lo[2], n[2];
                share_of(r, lo, n);
                const std::vector<double>& v = shares[(size_t)r];
                if (v.empty()) continue;
                const size_t spts = (size_t)n[0] * n[1] * (size_t)nz;
                for (int c = 0; c < NCOMP; ++c)
                    for (int k = 0; k < nz; ++k)
                        for (int j = 0; j < n[1]; ++j) {
                            const double* src = v.data() + (size_t)c * spts + ((size_t)k * n[1] + (size_t)j) * n[0];
                            double* dst = whole.data() + (size_t)c * gpts + (size_t)k * gpl +
                                          (size_t)(lo[1] - s.glo[1] + j) * s.gn[0] + (size_t)(lo[0] - s.glo[0]);
                            for (int ii = 0; ii < n[0]; ++ii) dst[ii] += src[ii];
                        }
            }
            std::vector<const double*> comps;
            for (int c = 0; c < NCOMP; ++c) comps.push_back(whole.data() + (size_t)c * gpts);
            write_fab(dir, g, comps);
        }
        s.flushed.push_back(g);   // (extrema on brick 0 only, which writes the headers)
        // the particles the plane has met while this buffer was filled: brick after brick into one file.  (A brick whose
        // species were empty so far has not met PackParticles: the bricks agree on the number of lists first.)
        int nspecies = (int)s.particles.size();
        {
            std::vector<double> ns((size_t)nranks, 0.0);
            ns[(size_t)me] = (double)nspecies;
            ReduceRealSum(comm, be, ns, ctx.stream);
            for (double v : ns) nspecies = std::max(nspecies, (int)v);
            if ((int)s.particles.size() < nspecies) s.particles.resize((size_t)nspecies);
        }
        if ((int)s.flushed_particles.size() < nspecies) s.flushed_particles.resize((size_t)nspecies);
        for (int sp = 0; sp < nspecies; ++sp) {
            const std::string name = sp < (int)m_species_names.size() ? m_species_names[(size_t)sp] : "species" + std::to_string(sp);
            auto& rows = s.particles[(size_t)sp];
            const size_t np = rows[0].size();
            const double mass = wx.GetPartContainer().GetParticleContainer(sp).mass;
            std::vector<double> rec(7 * np);
            for (size_t q = 0; q < np; ++q)
                for (int c = 0; c < 7; ++c) rec[7 * q + (size_t)c] = c >= 4 ? rows[(size_t)c][q] * mass : rows[(size_t)c][q];
            for (auto& r : rows) { r.clear(); r.shrink_to_fit(); }
            std::vector<double> nper((size_t)nranks, 0.0);
            nper[(size_t)me] = (double)np;
            ReduceRealSum(comm, be, nper, ctx.stream);
            std::vector<int64_t> pc((size_t)nranks);
            int64_t total = 0;
            for (int r = 0; r < nranks; ++r) { pc[(size_t)r] = 7 * (int64_t)nper[(size_t)r]; total += (int64_t)nper[(size_t)r]; }
            std::vector<std::vector<double>> recs;
            GatherRealToRoot(comm, be, rec.data(), pc, recs, ctx.stream, &m_gather_staging, gather_piece());
            ParticleGrid pg;
            for (int d = 0; d < 3; ++d) { pg.lo[d] = g.lo[d]; pg.hi[d] = g.hi[d]; }
            pg.which = id;
            pg.count = total;
            s.flushed_particles[(size_t)sp].push_back(pg);
            if (root) {
                if (id == 0) { make_dir(dir + "/" + name); make_dir(dir + "/" + name + "/Level_0"); }
                std::vector<double> all;
                all.reserve((size_t)(7 * total));
                for (const auto& v : recs) all.insert(all.end(), v.begin(), v.end());
                if (total) write_particle_records(dir, name, id, all, (size_t)total);   // no file for an empty grid (:1274, :1290)
                write_species_headers(dir, name, s.flushed_particles[(size_t)sp]);
            }
        }
        if (root) {
            int dom_lo[3] = {g.lo[0], g.lo[1], g.lo[2]}, dom_hi[3] = {g.hi[0], g.hi[1], g.hi[2]};
            for (const PlotGrid& q : s.flushed) { dom_lo[2] = std::min(dom_lo[2], q.lo[2]); dom_hi[2] = std::max(dom_hi[2], q.hi[2]); }
            const double dzl = (s.zhi_lab - s.zlo_lab) / s.n[2];
            const double dx[3] = {ctx.dx[0], ctx.dx[1], dzl};
            const double rlo[3] = {ctx.prob_lo[0] + dom_lo[0] * dx[0], ctx.prob_lo[1] + dom_lo[1] * dx[1],
                                   s.zlo_lab + (dom_lo[2] - s.ksmall) * dzl};
            const double rhi[3] = {ctx.prob_lo[0] + (dom_hi[0] + 1) * dx[0], ctx.prob_lo[1] + (dom_hi[1] + 1) * dx[1],
                                   s.zlo_lab + (dom_hi[2] + 1 - s.ksmall) * dzl};
            static const char* comp_names[NCOMP] = {"Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz", "rho"};
            write_cell_headers(dir, std::vector<std::string>(comp_names, comp_names + NCOMP), s.flushed, dom_lo, dom_hi, rlo, rhi,
                               dx, s.t_lab, wx.getistep());
        }
        s.buffer_counter = 0;
        s.buffer_k_index_hi = s.buffer_klo - 1;
        s.buffer_khi = s.buffer_klo - 1;   // no buffer until the next one is defined
        s.data.clear();
        s.data_nz = 0;
    }

    double z_boost_of(double t_lab, double t_boost) const { return (t_lab / m_gamma - t_boost) * kC / m_beta; }   // BTDiagnostics.H:276-280
    double z_lab_of(double t_lab, double t_boost) const { return (t_lab - t_boost / m_gamma) * kC / m_beta; }     // :285-289

    template <class CTX>
    bool slice_in_domain(const Snapshot& s, const CTX& ctx) const {   // GetZSliceInDomainFlag
        const double cs = ctx.dx[2];
        const bool out = (s.z_boost <= ctx.prob_lo[2] + 0.5 * cs) || (s.z_boost >= ctx.prob_hi[2] - 0.5 * cs) ||
                         (s.z_lab <= s.zlo_lab) || (s.z_lab >= s.zhi_lab);
        return !out;
    }
    int k_index_zlab(const Snapshot& s, double dzl) const {
        return (int)std::floor((s.z_lab - s.zlo_lab) / dzl) + s.ksmall;
    }

    // staggered component -> cell centres of the valid box (CellCenterFunctor.cpp:20-29 -> ablastr/coarsen/sample.H:47-99)
    static std::vector<double> cell_centered(const Backend* be, const amrex::MultiFab& mf, int ncell[3]) {
        const wxa_field_view& v = mf.view();
        std::vector<double> a((size_t)v.kstride * (size_t)v.n[2]);
        if (be->memcpy_d2h(a.data(), v.p, sizeof(double) * a.size()) != 0)
            throw std::runtime_error("BackTransformed diagnostic: device copy failed");
        int np[3];
        for (int d = 0; d < 3; ++d) {
            np[d] = 1 + v.stag[d];
            ncell[d] = v.n[d] - 2 * v.ng[d] - v.stag[d];
        }
        const double wx = 1.0 / np[0], wy = 1.0 / np[1], wz = 1.0 / np[2];
        std::vector<double> out((size_t)ncell[0] * ncell[1] * ncell[2]);
        size_t o = 0;
        for (int k = 0; k < ncell[2]; ++k)
            for (int j = 0; j < ncell[1]; ++j)
                for (int i = 0; i < ncell[0]; ++i) {
                    double c = 0.0;
                    for (int kr = 0; kr < np[2]; ++kr)
                        for (int jr = 0; jr < np[1]; ++jr)
                            for (int ir = 0; ir < np[0]; ++ir)
                                c += wx * wy * wz * a[(size_t)(i + ir + v.ng[0]) + (size_t)(j + jr + v.ng[1]) * v.jstride +
                                                     (size_t)(k + kr + v.ng[2]) * v.kstride];
                    out[o++] = c;
                }
        return out;
    }

    // the plane next to each z face of this brick, from the brick behind that face (zeros at a domain boundary)
    template <class WX>
    void exchange_guard_planes(WX& wx, const std::vector<std::vector<double>>& cc, const int nc[3]) {
        const auto& ctx = wx.context();
        const size_t plane = (size_t)nc[0] * nc[1], bytes = sizeof(double) * NCOMP * plane;
        std::vector<double> host(2 * NCOMP * plane);
        for (int c = 0; c < NCOMP; ++c) {
            std::copy(cc[(size_t)c].begin(), cc[(size_t)c].begin() + (std::ptrdiff_t)plane, host.begin() + (std::ptrdiff_t)(c * plane));
            std::copy(cc[(size_t)c].end() - (std::ptrdiff_t)plane, cc[(size_t)c].end(),
                      host.begin() + (std::ptrdiff_t)((NCOMP + c) * plane));
        }
        m_xbuf.reserve(4 * bytes);   // to minus | to plus | from plus | from minus
        char* base = static_cast<char*>(m_xbuf.p);
        if (ctx.be->memcpy_h2d(base, host.data(), 2 * bytes) != 0 ||
            ctx.be->memset_async(base + 2 * bytes, 0, 2 * bytes, ctx.stream) != 0)
            throw std::runtime_error("BackTransformed diagnostic: device copy failed");
        wx.comm().exchange_raw(2, base, (int64_t)bytes, base + bytes, (int64_t)bytes, base + 2 * bytes, (int64_t)bytes,
                               base + 3 * bytes, (int64_t)bytes, ctx.stream);
        wx.sync_stream();
        if (ctx.be->memcpy_d2h(host.data(), base + 2 * bytes, 2 * bytes) != 0)
            throw std::runtime_error("BackTransformed diagnostic: device copy failed");
        m_guard_hi.assign(host.begin(), host.begin() + (std::ptrdiff_t)(NCOMP * plane));
        m_guard_lo.assign(host.begin() + (std::ptrdiff_t)(NCOMP * plane), host.end());
    }

    template <class CTX>
    void back_transform_slice(Snapshot& s, int k_lab, const std::vector<std::vector<double>>& cc, const int nc[3],
                              const CTX& ctx) const {
        // get_slice_data(dir = z, coord = z_boost, interpolate): the cell that holds the coordinate and its neighbour on
        // the side of the coordinate, weighted linearly between the two cell centres
        const double cs = ctx.dx[2];
        const int kc = (int)std::floor((s.z_boost - ctx.prob_lo[2]) / cs);
        const double zc = ctx.prob_lo[2] + (kc + 0.5) * cs;
        int klo, khi;
        double w;
        if (s.z_boost >= zc) { klo = kc; khi = kc + 1; w = (s.z_boost - zc) / cs; }
        else { klo = kc - 1; khi = kc; w = (s.z_boost - (zc - cs)) / cs; }
        const int nz_domain = (int)std::lround((ctx.prob_hi[2] - ctx.prob_lo[2]) / cs);
        if (klo < 0 || khi >= nz_domain) return;   // excluded by GetZSliceInDomainFlag's half cell; kept as a guard
        // the brick that holds the plane's cell fills the slice (get_slice_data: the box that contains the coordinate)
        const int k0 = ctx.brick_box.lo[2];
        if (kc < k0 || kc >= k0 + nc[2]) return;
        const size_t plane = (size_t)nc[0] * nc[1];
        const bool guards = m_guard_lo.size() == (size_t)NCOMP * plane;
        if ((klo < k0 || khi >= k0 + nc[2]) && !guards) return;
        // component c on the global cell plane kg: this brick's, or the neighbour's plane behind a z face
        auto plane_of = [&](int c, int kg) -> const double* {
            if (kg < k0) return m_guard_lo.data() + (size_t)c * plane;
            if (kg >= k0 + nc[2]) return m_guard_hi.data() + (size_t)c * plane;
            return cc[(size_t)c].data() + (size_t)(kg - k0) * plane;
        };
        const double* plo[NCOMP];
        const double* phi[NCOMP];
        for (int c = 0; c < NCOMP; ++c) { plo[c] = plane_of(c, klo); phi[c] = plane_of(c, khi); }
        const size_t snap_plane = (size_t)s.n[0] * s.n[1];
        if (k_lab < s.data_k0 || k_lab >= s.data_k0 + s.data_nz) return;   // not in the planes held (a buffer not yet defined)
        const size_t kk = (size_t)(k_lab - s.data_k0);
        const double clight = kC, inv_clight = 1.0 / kC;
        for (int j = 0; j < s.n[1] && j < nc[1]; ++j)
            for (int i = 0; i < s.n[0] && i < nc[0]; ++i) {
                double v[NCOMP];
                const size_t at = (size_t)i + (size_t)j * nc[0];
                for (int c = 0; c < NCOMP; ++c)
                    v[c] = (1.0 - w) * plo[c][at] + w * phi[c][at];
                // LorentzTransformZ (BackTransformFunctor.cpp:289-313)
                const double ex_lab = m_gamma * (v[0] + m_beta * clight * v[4]);
                const double by_lab = m_gamma * (v[4] + m_beta * inv_clight * v[0]);
                v[0] = ex_lab; v[4] = by_lab;
                const double ey_lab = m_gamma * (v[1] - m_beta * clight * v[3]);
                const double bx_lab = m_gamma * (v[3] - m_beta * inv_clight * v[1]);
                v[1] = ey_lab; v[3] = bx_lab;
                const double j_lab = m_gamma * (v[8] + m_beta * clight * v[9]);
                const double rho_lab = m_gamma * (v[9] + m_beta * inv_clight * v[8]);
                v[8] = j_lab; v[9] = rho_lab;
                const size_t dst = (size_t)i + (size_t)j * s.n[0] + kk * snap_plane;
                for (int c = 0; c < NCOMP; ++c) s.data[(size_t)c * snap_plane * (size_t)s.data_nz + dst] = v[c];
            }
    }

    int m_num;
    double m_dt_snap;
    int m_buffer_size;
    double m_gamma = 1.0, m_beta = 0.0, m_mw_beta = 0.0;
    std::vector<Snapshot> m_snap;
    std::string m_flush_prefix;                  // SetFlush
    int m_file_min_digits = 6;
    std::vector<std::string> m_species_names;
    DeviceBuffer m_xbuf;                         // exchange_guard_planes
    // doubles per piece of a brick's share on its way to brick 0 (WXA_BTD_GATHER_PIECE: a test makes the pieces small)
    static int64_t gather_piece() {
        static const int64_t v = [] { const char* e = std::getenv("WXA_BTD_GATHER_PIECE"); return e && std::atoll(e) > 0 ? (int64_t)std::atoll(e) : (int64_t(8) << 20); }();
        return v;
    }
    DeviceBuffer m_gather_staging;   // flush_bricks: the pieces of the bricks' shares on their way to brick 0 (<= 64 MB, kept)
    std::vector<double> m_guard_lo, m_guard_hi;  // [comp][j][i] behind the low / high z face
};

}  // namespace wxa::host
#endif
