// class WarpX of the host layer: the per-step schedule of the explicit FDTD,
// single-level, periodic branch, with the reference's method names and call order
// (Source/Evolve/WarpXEvolve.cpp:94-347 Evolve, :354-455 OneStep_nosub,
//  :473-531 ExplicitFillBoundaryEBUpdateAux, :533-581 HandleParticlesAtBoundaries,
//  :583-652 SyncCurrentAndRho, :65-93 Synchronize; Source/FieldSolver/WarpXPushFieldsEM.cpp:877-1011;
//  Source/Parallelization/WarpXComm.cpp:699-827,1073-1240,1357-1424).
#ifndef WXA_HOST_WARPX_HPP_
#define WXA_HOST_WARPX_HPP_

#include <chrono>
#include <functional>

#include "BTDiagnostics.hpp"
#include "NCIGodfreyFilter.hpp"
#include "ReducedDiags.hpp"
#include "WarpXParticleContainer.hpp"

namespace wxa::host {

// Source/Parallelization/GuardCellManager.{H,cpp}: guard depths, 3-D FDTD subset of Init (:33-344)
struct guardCellManager {
    amrex::IntVect ng_alloc_EB, ng_alloc_J, ng_depos_J, ng_FieldSolver, ng_FieldGather, ng_UpdateAux;

    void Init(const amrex::Real dt, const std::array<amrex::Real, 3>& dx, const int nox, const bool use_filter,
              const amrex::IntVect& bilinear_filter_stencil_length, const bool do_fdtd_nci_corr = false,
              const int nci_corr_stencil = 0, const bool safe_guard_cells = false) {
        constexpr double c = 299'792'458.;
        for (int d = 0; d < 3; ++d) {
            const int ng_tmp = nox;                                   // :62-64 (no subcycling / MR)
            ng_alloc_EB[d] = (ng_tmp % 2) ? ng_tmp + 1 : ng_tmp;      // :83-85 always even
            if (d == 2 && do_fdtd_nci_corr) {                         // :87-89 more guard cells in z for the NCI filter
                const int ng = ng_tmp + nci_corr_stencil;
                ng_alloc_EB[d] = (ng % 2) ? ng + 1 : ng;
            }
            int ngJ = ng_tmp;                                         // :96-98
            ngJ += static_cast<int>(std::ceil(c * 0.5 * dt / dx[d])); // :161 half a step of motion
            ng_depos_J[d] = ngJ;                                      // :166
            ng_alloc_J[d] = ngJ + (use_filter ? bilinear_filter_stencil_length[d] - 1 : 0);  // :169-172
            ng_FieldSolver[d] = 1;                                    // :276-278 Yee: GetMaxGuardCell
            ng_FieldGather[d] = (nox + 1) / 2;                        // :314-316 (staggered, galerkin: +0)
            ng_UpdateAux[d] = 0;
        }
        {   // :315-325 ng_FieldGather_noNCI capped by the allocation, then the NCI filter's cells along z
            const amrex::IntVect no_nci(std::min(ng_FieldGather[0], nox % 2 ? nox + 1 : nox),
                                        std::min(ng_FieldGather[1], nox % 2 ? nox + 1 : nox),
                                        std::min(ng_FieldGather[2], nox % 2 ? nox + 1 : nox));
            ng_FieldGather = no_nci;
            if (do_fdtd_nci_corr) ng_FieldGather[2] += nci_corr_stencil;   // NCIGodfreyFilter::m_stencil_width
        }
        ng_FieldGather = amrex::min(ng_FieldGather, ng_alloc_EB);     // :333
        for (int d = 0; d < 3; ++d) ng_FieldGather[d] = std::max(ng_FieldGather[d], ng_FieldSolver[d]);   // :338
        if (safe_guard_cells) {   // :297-308 "Run in safe mode: exchange all allocated guard cells at each call of FillBoundary"
            ng_FieldSolver = ng_alloc_EB;
            ng_FieldGather = ng_alloc_EB;
            ng_UpdateAux = ng_alloc_EB;
        }
    }
};

// Source/FieldSolver/FiniteDifferenceSolver/FiniteDifferenceSolver.{H,cpp} (Yee, Cartesian)
class FiniteDifferenceSolver {
public:
    // FiniteDifferenceSolver::FiniteDifferenceSolver (FiniteDifferenceSolver.cpp:29-79): the stencil coefficients of
    // the chosen algorithm (ElectromagneticSolverAlgo::Yee or ::CKC)
    FiniteDifferenceSolver(WarpXContext* ctx, int fdtd_algo, const std::array<amrex::Real, 3>& cell_size)
        : m_ctx(ctx), m_fdtd_algo(fdtd_algo) {
        // CartesianYeeAlgorithm::InitializeStencilCoefficients (CartesianYeeAlgorithm.H:29-43)
        for (int d = 0; d < 3; ++d) m_stencil_coefs[d] = 1.0 / cell_size[d];
        if (m_fdtd_algo == WXA_SOLVER_CKC) {
            if (!ctx->be->evolve_b_ckc) throw std::runtime_error("algo.maxwell_solver = ckc: not in this backend");
            // CartesianCKCAlgorithm::InitializeStencilCoefficients (CartesianCKCAlgorithm.H:28-102)
            ctx->be->ckc_stencil_coefficients(cell_size.data(), m_ckc_x.data(), m_ckc_y.data(), m_ckc_z.data());
        } else if (m_fdtd_algo != WXA_SOLVER_YEE) {
            throw std::runtime_error("algo.maxwell_solver: yee or ckc");
        }
    }
    // FiniteDifferenceSolver.H:55-59
    void EvolveB(ablastr::fields::MultiFabRegister& fields, int lev, PatchType patch_type, amrex::Real dt) {
        using warpx::fields::FieldType;
        if (patch_type != PatchType::fine) throw std::runtime_error("single level: fine patch only");
        auto E = fields.get_alldirs(FieldType::Efield_fp, lev);
        auto B = fields.get_alldirs(FieldType::Bfield_fp, lev);
        const wxa_field_view Ev[3] = {E[0]->view(), E[1]->view(), E[2]->view()};
        const wxa_field_view Bv[3] = {B[0]->view(), B[1]->view(), B[2]->view()};
        if (m_fdtd_algo == WXA_SOLVER_CKC)   // EvolveB.cpp:102-105
            check(m_ctx->be->evolve_b_ckc(Ev, Bv, dt, m_ckc_x.data(), m_ckc_y.data(), m_ckc_z.data(), m_ctx->stream),
                  "evolve_b_ckc");
        else
            check(m_ctx->be->evolve_b(Ev, Bv, dt, m_stencil_coefs.data(), m_ctx->stream), "evolve_b");
    }
    // FiniteDifferenceSolver.H:61-66
    void EvolveE(ablastr::fields::MultiFabRegister& fields, int lev, PatchType patch_type,
                 const ablastr::fields::VectorField& Efield, amrex::Real dt) {
        using warpx::fields::FieldType;
        if (patch_type != PatchType::fine) throw std::runtime_error("single level: fine patch only");
        auto B = fields.get_alldirs(FieldType::Bfield_fp, lev);
        auto J = fields.get_alldirs(FieldType::current_fp, lev);
        const wxa_field_view Ev[3] = {Efield[0]->view(), Efield[1]->view(), Efield[2]->view()};
        const wxa_field_view Bv[3] = {B[0]->view(), B[1]->view(), B[2]->view()};
        const wxa_field_view Jv[3] = {J[0]->view(), J[1]->view(), J[2]->view()};
        check(m_ctx->be->evolve_e(Ev, Bv, Jv, dt, m_stencil_coefs.data(), m_ctx->stream), "evolve_e");
    }
    // the first guard layer of B from the guards already present (wxa_evolve_b_guard_layer)
    void EvolveBGuardLayer(ablastr::fields::MultiFabRegister& fields, int lev, amrex::Real dt, const int32_t grow[3]) {
        using warpx::fields::FieldType;
        auto E = fields.get_alldirs(FieldType::Efield_fp, lev);
        auto B = fields.get_alldirs(FieldType::Bfield_fp, lev);
        const wxa_field_view Ev[3] = {E[0]->view(), E[1]->view(), E[2]->view()};
        const wxa_field_view Bv[3] = {B[0]->view(), B[1]->view(), B[2]->view()};
        check(m_ctx->be->evolve_b_guard_layer(Ev, Bv, dt, m_stencil_coefs.data(), grow, m_ctx->stream), "evolve_b_guard_layer");
    }

private:
    WarpXContext* m_ctx;
    int m_fdtd_algo = WXA_SOLVER_YEE;
    std::array<amrex::Real, 3> m_stencil_coefs{};          // 1/dx: Yee, and the downward differences of CKC
    std::array<amrex::Real, 5> m_ckc_x{}, m_ckc_y{}, m_ckc_z{};
};

class WarpX {
public:
    static constexpr bool sync_nodal_points = true;  // Source/WarpX.H:1523

    WarpX(const Backend* be, const wxa_sim_config& cfg, const wxa_comm* comm)
        : m_be(be), m_cfg(cfg), m_fields(be), mypc(std::make_unique<MultiParticleContainer>(&m_ctx)) {
        using warpx::fields::FieldType;
        using ablastr::fields::Direction;
        if (cfg.nox < 1 || cfg.nox > 4) throw std::runtime_error("algo.particle_shape must be 1..4");
        // the switches of the environment are read once, here; the bricks of a run compare theirs before the first step
        m_env_no_guard_layer = std::getenv("WXA_NO_GUARD_LAYER") != nullptr;
        if (const char* e = std::getenv("WXA_PEC_RHO_FOLD_GUARD_COLUMNS")) m_env_fold_rho_guard_columns = std::atoi(e) != 0;
        m_ctx.be = be;
        m_ctx.nox = cfg.nox;
        m_ctx.galerkin_interpolation = cfg.galerkin != 0;
        m_ctx.particle_pusher_algo = (ParticlePusherAlgo)cfg.particle_pusher;
        m_ctx.current_deposition_algo = (CurrentDepositionAlgo)cfg.current_deposition;
        if (cfg.gamma_boost > 1.0) {   // ReadBoostedFrameParameters (WarpXUtil.cpp:114-141), boost along z
            m_ctx.gamma_boost = cfg.gamma_boost;
            m_ctx.beta_boost = std::sqrt(1.0 - 1.0 / std::pow(cfg.gamma_boost, 2.0));
        }
        amrex::IntVect blo, bhi;
        for (int d = 0; d < 3; ++d) {
            if (cfg.nbricks[d] < 1 || cfg.coord[d] < 0 || cfg.coord[d] >= cfg.nbricks[d])
                throw std::runtime_error("bad brick decomposition");
            if (cfg.n_cell[d] % cfg.nbricks[d] != 0) throw std::runtime_error("n_cell must divide evenly into bricks");
            m_ctx.prob_lo[d] = cfg.prob_lo[d];
            m_ctx.prob_hi[d] = cfg.prob_hi[d];
            m_ctx.dx[d] = (cfg.prob_hi[d] - cfg.prob_lo[d]) / cfg.n_cell[d];
            m_ctx.dinv[d] = 1.0 / m_ctx.dx[d];
            const int nb = cfg.n_cell[d] / cfg.nbricks[d];
            blo[d] = cfg.coord[d] * nb;
            bhi[d] = blo[d] + nb - 1;
            m_ctx.brick_plo[d] = cfg.prob_lo[d] + blo[d] * m_ctx.dx[d];
            m_ctx.brick_phi[d] = (cfg.coord[d] == cfg.nbricks[d] - 1) ? cfg.prob_hi[d]
                                                                       : cfg.prob_lo[d] + (bhi[d] + 1) * m_ctx.dx[d];
        }
        m_ctx.brick_box = amrex::Box(blo, bhi);
        if (cfg.grid_type != WXA_GRID_STAGGERED)   // warpx.grid_type: the kernels are written for the Yee grid
            throw std::runtime_error("warpx.grid_type: only the staggered (Yee) grid is on this path");
        ComputeDt();
        // warpx.use_filter: 1-pass bilinear, stencil length npass+1 = 2 (BilinearFilter.cpp:63-68)
        use_filter = cfg.use_filter != 0;
        guard_cells.Init(dt[0], m_ctx.dx, cfg.nox, use_filter, amrex::IntVect(2), cfg.use_fdtd_nci_corr != 0,
                         NCIGodfreyFilter::m_stencil_width);   // Source/WarpX.cpp:2019-2025
        m_ctx.ng_alloc_EB = guard_cells.ng_alloc_EB;
        m_ctx.ng_depos_J = guard_cells.ng_depos_J;
        for (int d = 0; d < 3; ++d)
            if (m_ctx.brick_box.length(d) < guard_cells.ng_alloc_J[d] + 1)
                throw std::runtime_error("brick thinner than the guard depth");
        m_comm = std::make_unique<BrickComm>(be, comm, cfg.nbricks, cfg.coord);
        // boundary.field_lo / field_hi (Source/WarpX.cpp, ReadBoundaryConditions): periodic or PEC
        int periodic[3];
        for (int d = 0; d < 3; ++d) {
            const int lo = cfg.field_boundary_lo[d], hi = cfg.field_boundary_hi[d];
            const bool ok = (lo == WXA_BOUNDARY_PERIODIC || lo == WXA_BOUNDARY_PEC) &&
                            (hi == WXA_BOUNDARY_PERIODIC || hi == WXA_BOUNDARY_PEC) &&
                            ((lo == WXA_BOUNDARY_PERIODIC) == (hi == WXA_BOUNDARY_PERIODIC));
            if (!ok) throw std::runtime_error("field boundary: a direction is periodic on both sides or on neither");
            periodic[d] = lo == WXA_BOUNDARY_PERIODIC;
            // a wall belongs to the bricks that touch it; the schedule (which exchanges are issued) follows the
            // domain's walls, the same on every brick
            m_pec_lo[d] = lo == WXA_BOUNDARY_PEC && cfg.coord[d] == 0;
            m_pec_hi[d] = hi == WXA_BOUNDARY_PEC && cfg.coord[d] == cfg.nbricks[d] - 1;
            m_any_pec = m_any_pec || lo == WXA_BOUNDARY_PEC || hi == WXA_BOUNDARY_PEC;
            m_pec_here = m_pec_here || m_pec_lo[d] || m_pec_hi[d];
            m_dom_lo[d] = 0;
            m_dom_hi[d] = cfg.n_cell[d] - 1;
        }
        m_comm->set_periodic(periodic);
        SetUpHaloOverlap(cfg.overlap_halo != 0);
        // boundary.particle_lo / particle_hi: default = periodic with a periodic field boundary, absorbing
        // otherwise; periodic particles need a periodic field boundary and vice versa
        for (int d = 0; d < 3; ++d)
            for (int side = 0; side < 2; ++side) {
                int32_t want = side == 0 ? cfg.particle_boundary_lo[d] : cfg.particle_boundary_hi[d];
                const bool fper = periodic[d] != 0;
                if (want == WXA_PBOUNDARY_DEFAULT) want = fper ? WXA_PBOUNDARY_PERIODIC : WXA_PBOUNDARY_ABSORBING;
                const bool ok = (want == WXA_PBOUNDARY_PERIODIC) == fper &&
                                (want == WXA_PBOUNDARY_PERIODIC || want == WXA_PBOUNDARY_ABSORBING ||
                                 want == WXA_PBOUNDARY_REFLECTING);
                if (!ok) throw std::runtime_error("particle boundary: periodic if and only if the field boundary is");
                (side == 0 ? m_ctx.particle_bc_lo[d] : m_ctx.particle_bc_hi[d]) = want;
                m_ctx.any_particle_wall = m_ctx.any_particle_wall || want != WXA_PBOUNDARY_PERIODIC;
                m_any_reflecting_wall = m_any_reflecting_wall || want == WXA_PBOUNDARY_REFLECTING;
            }

        // AllocLevelMFs (Source/WarpX.cpp:2078-2700): Yee nodal flags :2117-2125
        const amrex::IntVect Etype[3] = {{0, 1, 1}, {1, 0, 1}, {1, 1, 0}};
        const amrex::IntVect Btype[3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        for (int d = 0; d < 3; ++d) {
            m_fields.alloc_init(FieldType::Efield_fp, Direction{d}, 0, m_ctx.brick_box, Etype[d], guard_cells.ng_alloc_EB);
            m_fields.alloc_init(FieldType::Bfield_fp, Direction{d}, 0, m_ctx.brick_box, Btype[d], guard_cells.ng_alloc_EB);
            m_fields.alloc_init(FieldType::current_fp, Direction{d}, 0, m_ctx.brick_box, Etype[d], guard_cells.ng_alloc_J);
            // aux = alias of fp at level 0 (Source/WarpX.cpp:2489-2501)
            m_fields.alias_init(FieldType::Efield_aux, FieldType::Efield_fp, Direction{d}, 0);
            m_fields.alias_init(FieldType::Bfield_aux, FieldType::Bfield_fp, Direction{d}, 0);
        }
        if (use_filter)
            for (int d = 0; d < 3; ++d)
                m_filter_tmp[d] = std::make_unique<amrex::MultiFab>(be, m_ctx.brick_box, Etype[d], guard_cells.ng_alloc_J);
        for (int d = 0; d < 3; ++d)   // the device SumBoundary of a self-periodic direction needs this
            if (cfg.nbricks[d] == 1 && m_comm->periodic(d) &&
                m_ctx.brick_box.length(d) < 2 * guard_cells.ng_alloc_J[d] + 1)
                throw std::runtime_error("periodic direction shorter than twice the guard depth");
        if (cfg.use_fdtd_nci_corr) InitNCICorrector(Etype, Btype);
        {   // the exchanges of the time loop: E + B together at any guard depth, J at its guard sum
            std::vector<amrex::MultiFab*> eb, jj;
            for (int d = 0; d < 3; ++d) {
                eb.push_back(m_fields.get(FieldType::Efield_fp, Direction{d}, 0));
                eb.push_back(m_fields.get(FieldType::Bfield_fp, Direction{d}, 0));
                jj.push_back(m_fields.get(FieldType::current_fp, Direction{d}, 0));
            }
            m_comm->presize(eb, guard_cells.ng_alloc_EB);
            m_comm->presize(jj, guard_cells.ng_alloc_J);
        }
        m_fdtd_solver_fp = std::make_unique<FiniteDifferenceSolver>(&m_ctx, cfg.maxwell_solver, m_ctx.dx);
        sort_intervals = cfg.sort_interval;
        m_ctx.sort_intervals_on = sort_intervals > 0;
        // the periodic sorts folded into the push (WarpXParticleContainer::ArmPushSort); WXA_SORT_IN_PUSH=0: the sort as
        // passes of its own, for A/B timing and for bisecting
        const char* fold = std::getenv("WXA_SORT_IN_PUSH");
        m_ctx.sort_in_push = sort_intervals > 0 && !(fold && std::atoi(fold) == 0);
        if (const char* e = std::getenv("WXA_SORT_BEHIND_SHIFT")) m_ctx.skip_sort_behind_a_window_shift = std::atoi(e) == 0;
        for (int d = 0; d < 3; ++d) m_ctx.sort_wrap[d] = m_comm->periodic(d) && m_comm->self_periodic(d) ? 1 : 0;
        // the keys of the positions behind the scattering push (free flight over its time step): the deposition of the sort
        // step and the next gather meet a fresh sort.  WXA_SORT_PREDICT=0: the keys of the positions in front of it.
        const char* predict = std::getenv("WXA_SORT_PREDICT");
        m_ctx.sort_predict_dt = predict && std::atoi(predict) == 0 ? 0.0 : dt[0];
        // one special push per sort cycle (WarpXContext::sort_merged); WXA_SORT_MERGED=0: a counting and a scattering push
        const char* merged = std::getenv("WXA_SORT_MERGED");
        m_ctx.sort_merged = m_ctx.sort_in_push && !(merged && std::atoi(merged) == 0);
        m_ctx.sort_interval_steps = (int32_t)sort_intervals;
        m_ctx.step_dt = dt[0];
    }

    // WarpX::InitNCICorrector (Source/Initialization/WarpXInitData.cpp:858-890): the two Godfrey filters for
    // c dt / dz, Galerkin tables unless the gather uses the same shape in every direction; plus the six arrays that
    // receive the filtered fields (the reference's per-tile filtered_Ex ... filtered_Bz)
    void InitNCICorrector(const amrex::IntVect (&Etype)[3], const amrex::IntVect (&Btype)[3]) {
        constexpr double c = 299'792'458.;
        const double cdtodz = c * dt[0] / m_ctx.dx[2];
        const bool nodal_gather = !m_ctx.galerkin_interpolation;
        nci_godfrey_filter_exeybz = std::make_unique<NCIGodfreyFilter>(godfrey_coeff_set::Ex_Ey_Bz, cdtodz, nodal_gather);
        nci_godfrey_filter_bxbyez = std::make_unique<NCIGodfreyFilter>(godfrey_coeff_set::Bx_By_Ez, cdtodz, nodal_gather);
        nci_godfrey_filter_exeybz->ComputeStencils();
        nci_godfrey_filter_bxbyez->ComputeStencils();
        for (int i = 0; i < 5; ++i) {
            m_ctx.nci_stencil_exeybz[i] = nci_godfrey_filter_exeybz->stencil_z[i];
            m_ctx.nci_stencil_bxbyez[i] = nci_godfrey_filter_bxbyez->stencil_z[i];
        }
        for (int d = 0; d < 3; ++d) {
            m_nci_E[d] = std::make_unique<amrex::MultiFab>(m_be, m_ctx.brick_box, Etype[d], guard_cells.ng_alloc_EB);
            m_nci_B[d] = std::make_unique<amrex::MultiFab>(m_be, m_ctx.brick_box, Btype[d], guard_cells.ng_alloc_EB);
            m_ctx.nci_E[d] = m_nci_E[d].get();
            m_ctx.nci_B[d] = m_nci_B[d].get();
        }
        m_ctx.use_fdtd_nci_corr = true;
    }

    // Source/Evolve/WarpXComputeDt.cpp:41-102 with CartesianYeeAlgorithm::ComputeMaxDt (:48-56) or
    // CartesianCKCAlgorithm::ComputeMaxDt (CartesianCKCAlgorithm.H:107-120)
    void ComputeDt() {
        constexpr double c = 299'792'458.;
        const auto& dx = m_ctx.dx;
        amrex::Real deltat;
        if (m_cfg.maxwell_solver == WXA_SOLVER_CKC) {
            if (!m_be->ckc_max_dt) throw std::runtime_error("algo.maxwell_solver = ckc: not in this backend");
            deltat = m_cfg.cfl * m_be->ckc_max_dt(dx.data());
        } else {
            deltat = m_cfg.cfl * 1.0 / (std::sqrt(1.0 / (dx[0] * dx[0]) + 1.0 / (dx[1] * dx[1]) + 1.0 / (dx[2] * dx[2])) * c);
        }
        dt.assign(1, deltat);
    }

    // Source/Evolve/WarpXEvolve.cpp:94-347
    void Evolve(int numsteps) {
        const int numsteps_max = numsteps;
        if (!m_switches_verified) {   // collective (the count round): every brick of a run is in its first Evolve here
            m_switches_verified = true;
            m_comm->VerifySwitchesAgree((safe_guard_cells ? 1 : 0) | (m_env_no_guard_layer ? 2 : 0) |
                                        (m_env_fold_rho_guard_columns ? 4 : 0) | (m_overlap ? 8 : 0));
        }
        if (!m_reduced_diags_started) {   // WarpX::InitData (WarpXInitData.cpp:612-619): full and reduced diagnostics before the first iteration
            m_reduced_diags_started = true;
            if (diag_hook) diag_hook((int)istep - 1, kDiagFlush);
            reduced_diags.ComputeAndWrite(*this, (int)istep - 1);
        }
        for (int step = 0; step < numsteps_max; ++step) {
            if (diag_hook) diag_hook((int)istep, kDiagNewIteration);   // :118 multi_diags->NewIteration()
            // :142-145 if synchronized, push velocity backward one half step
            ExplicitFillBoundaryEBUpdateAux();
            // warpx.sort_intervals (Source/WarpX.cpp:1335): the sort itself runs inside
            // PhysicalParticleContainer::Evolve, between push and deposition
            m_ctx.sort_now = sort_intervals > 0 && (istep % sort_intervals == 0);
            m_ctx.count_now = sort_intervals > 0 && ((istep + 1) % sort_intervals == 0);   // the next step sorts
            m_ctx.steps_to_next_sort = sort_intervals > 0 ? (int32_t)(sort_intervals - istep % sort_intervals) : 0;
            // :157-166 ionization / collisions / QED: not on this path
            OneStep_nosub(cur_time);
            // :222-226 at the end of the last step, push p by 0.5*dt to synchronize
            if (step == numsteps_max - 1 && synchronize_at_end) Synchronize();
            ++istep;
            cur_time += dt[0];
            m_ctx.t_new = cur_time;
            // :241 multi_diags->FilterComputePackFlush(step, false, true): the BackTransformed diagnostics, and only they,
            // run BEFORE the window moves and before the particles meet the boundaries (MultiDiagnostics.cpp:81-96) --
            // they see this step's domain, J where it was deposited, and rho without the plasma injected behind the shift.
            // (Until round 4 this call sat with the other diagnostics below: found against the reference's own golden file
            // test_3d_laser_acceleration_btd.json, whose field sums were 17 ... 97 % away.)
            if (m_btd) m_btd->ComputeAndPack(*this);
            const bool move_j = is_synchronized;
            const int num_moved = MoveWindow(istep, move_j);     // :246 MoveWindow(step+1, move_j)
            HandleParticlesAtBoundaries(step, cur_time, num_moved);  // :256
            reduced_diags.ComputeAndWrite(*this, (int)istep - 1);   // :299-305 reduced_diags->ComputeDiags(step), WriteToFile(step)
            if (diag_hook) diag_hook((int)istep - 1, kDiagFlush);   // :306 multi_diags->FilterComputePackFlush(step): the Full diagnostics (FullDiagnostics.hpp)
        }
        m_be->stream_sync(m_ctx.stream);
        // :341-343 the forced flush of the last time step, once the run has reached the deck's max_step
        if (max_step >= 0 && istep == max_step) FlushDiagsLastTimestep();
    }
    // MultiDiagnostics::FilterComputePackFlushLastTimestep (MultiDiagnostics.cpp:98-107): every diagnostic that dumps its
    // last time step -- the Full ones through the hook, and the BackTransformed one's partly filled buffers (until round 4
    // only the former: a deck run lost the slices since the last full buffer)
    void FlushDiagsLastTimestep() {
        if (diag_hook) diag_hook((int)istep, kDiagLastTimestep);
        if (m_btd) m_btd->FlushLast(*this);
    }
    // <diag>.diag_type = Full: the plotfile writer sits above this class (FullDiagnostics.hpp installs the hook)
    static constexpr int kDiagNewIteration = 0, kDiagFlush = 1, kDiagLastTimestep = 2;
    std::function<void(int step, int what)> diag_hook;
    int64_t max_step = -1;   // the deck's max_step (-1: not known; the caller flushes the last time step itself)

    // <diag>.diag_type = BackTransformed with do_back_transformed_fields = 1 (BTDiagnostics.hpp): lab-frame snapshots
    // num_snapshots_lab, dt_snapshots_lab (= dz_snapshots_lab / c), buffer_size as in BTDiagnostics::ReadParameters (:206-292)
    // write_species (<diag>.write_species, default 1 in the reference): the species' particles too
    void AddBTDiagnostics(int num_snapshots, amrex::Real dt_snapshots_lab, int buffer_size, bool write_species) {
        m_btd = std::make_unique<BTDiagnostics>(num_snapshots, dt_snapshots_lab, buffer_size);
        m_btd->Init(*this);
        m_btd_write_species = write_species;
        HookBTDSpecies();
    }
    // species added after the diagnostic are picked up too (SetDoBackTransformedParticles, BTDiagnostics.cpp:124-130)
    void HookBTDSpecies() {
        if (!m_btd || !m_btd_write_species) return;
        m_ctx.btd = m_btd.get();
        for (int i = 0; i < mypc->nSpecies(); ++i) mypc->GetParticleContainer(i).btd_species_id = i;
    }
    BTDiagnostics* btd() const { return m_btd.get(); }

    // warpx.reduced_diags_names (ReducedDiags.hpp): FieldEnergy, ParticleEnergy, ParticleMomentum, ParticleNumber
    MultiReducedDiags reduced_diags;
    // The points of component f that this brick owns in a sum over the domain -- amrex's owner mask, as
    // MultiFab::norm2(0, periodicity) applies it (FieldEnergy.cpp:127-135): a nodal point on the face between two bricks,
    // or on the two ends of a periodic direction, is counted by the brick below it; the last node of a direction with
    // walls belongs to the topmost brick.  Index box [lo, hi).
    void owned_points(const wxa_field_view& f, int32_t lo[3], int32_t hi[3]) const {
        for (int d = 0; d < 3; ++d) {
            lo[d] = f.lo[d] + f.ng[d];
            hi[d] = f.lo[d] + f.n[d] - f.ng[d];
            const bool top_wall = !m_comm->periodic(d) && m_comm->coord()[d] == m_comm->nbricks()[d] - 1;
            if (f.stag[d] && !top_wall) hi[d] -= 1;
        }
    }
    amrex::Real getdt() const { return dt[0]; }
    void sync_stream() { m_be->stream_sync(m_ctx.stream); }

    // warpx.do_moving_window / moving_window_dir / moving_window_v (Source/WarpX.cpp:620-660): forward window, lab frame
    void SetMovingWindow(int dir, amrex::Real v_over_c) {
        if (dir < 0 || dir > 2 || !(v_over_c > 0.0)) throw std::runtime_error("moving window: direction 0..2, v > 0");
        if (m_comm->periodic(dir)) throw std::runtime_error("the moving window direction cannot be periodic");
        do_moving_window = true;
        moving_window_dir = dir;
        moving_window_v = v_over_c * 299'792'458.;
        moving_window_x = m_ctx.prob_lo[dir];                                       // :649
        // Behind every shift of the window the tiles are sorted afresh (the shift's sort stands for the periodic one), and
        // that sort drops the push's record: with one special push per cycle every cycle would start over with a COUNT
        // whose record is thrown away.  A counting and a scattering push as in round 5 then (BASELINE config 5 on one
        // GPU: 88 against 93 ms per step, profiles/round6/README.md) -- unless WXA_SORT_MERGED=1 asks for it.
        const char* merged = std::getenv("WXA_SORT_MERGED");
        if (!(merged && std::atoi(merged) != 0)) m_ctx.sort_merged = false;
    }

    // WarpX::MoveWindow (Source/Utils/WarpXMovingWindow.cpp:138-476): window position, whole-cell field shift,
    // new domain bounds, continuous injection into the cells that entered
    int MoveWindow(int /*step*/, bool move_j) {
        if (!do_moving_window) return 0;                                            // :151 moving_window_active
        using warpx::fields::FieldType;
        using ablastr::fields::Direction;
        // WarpX::InitData (:301): injection starts at the upper domain bound as it is when the container first
        // meets the window (containers may be added after SetMovingWindow)
        for (int i = 0; i < mypc->nContainers(); ++i) {
            WarpXParticleContainer& pc = mypc->GetParticleContainer(i);
            if (std::isnan(pc.m_current_injection_position)) pc.m_current_injection_position = m_ctx.prob_hi[moving_window_dir];
        }
        const amrex::Real c = 299'792'458.;
        moving_window_x += (moving_window_v - m_ctx.beta_boost * c) / (1 - moving_window_v * m_ctx.beta_boost / c) * dt[0];   // :157
        const int dir = moving_window_dir;
        UpdateInjectionPosition(dt[0]);                                             // :161
        const amrex::Real cdx = m_ctx.dx[dir];
        const int num_shift_base = static_cast<int>((moving_window_x - m_ctx.prob_lo[dir]) / cdx);   // :171
        if (num_shift_base == 0) return 0;
        m_ctx.prob_lo[dir] += num_shift_base * cdx;                                 // :181-186 ResetProbDomain
        m_ctx.prob_hi[dir] += num_shift_base * cdx;
        // the bricks move with the domain: brick c of the window direction covers its cells of the shifted index space
        m_ctx.brick_plo[dir] = m_ctx.prob_lo[dir] + (m_ctx.brick_box.lo[dir] - m_dom_lo[dir]) * cdx;
        m_ctx.brick_phi[dir] = m_ctx.prob_lo[dir] + (m_ctx.brick_box.lo[dir] - m_dom_lo[dir] + m_ctx.brick_box.length(dir)) * cdx;
        for (int dim = 0; dim < 3; ++dim) {                                         // :222-246
            shiftMF(*m_fields.get(FieldType::Bfield_fp, Direction{dim}, 0), num_shift_base, dir);
            shiftMF(*m_fields.get(FieldType::Efield_fp, Direction{dim}, 0), num_shift_base, dir);
            if (move_j) shiftMF(*m_fields.get(FieldType::current_fp, Direction{dim}, 0), num_shift_base, dir);
        }
        for (int i = 0; i < mypc->nContainers(); ++i) {                             // :388-437
            WarpXParticleContainer& pc = mypc->GetParticleContainer(i);
            if (!pc.doContinuousInjection()) continue;
            const amrex::Real new_injection_position =
                pc.m_current_injection_position + std::floor((m_ctx.prob_hi[dir] - pc.m_current_injection_position) / cdx) * cdx;
            double blo[3] = {m_ctx.prob_lo[0], m_ctx.prob_lo[1], m_ctx.prob_lo[2]};
            double bhi[3] = {m_ctx.prob_hi[0], m_ctx.prob_hi[1], m_ctx.prob_hi[2]};
            blo[dir] = pc.m_current_injection_position;
            bhi[dir] = new_injection_position;
            if (bhi[dir] > blo[dir] && pc.m_current_injection_position != new_injection_position) {
                pc.ContinuousInjection(blo, bhi);
                pc.m_current_injection_position = new_injection_position;
            }
        }
        // every particle's cell index along the window moved with the domain: the tile-major order of the
        // last sort no longer lines up with the tiles, so sort now instead of at the next interval
        if (sort_intervals > 0) {
            mypc->SortParticlesByBin(amrex::IntVect(1));
            for (int i = 0; i < mypc->nContainers(); ++i) mypc->GetParticleContainer(i).m_sorted_by_window_shift = true;
        }
        return num_shift_base;
    }

    // WarpX::UpdateInjectionPosition (Source/Utils/WarpXMovingWindow.cpp:60-136): the plasma drifts with its bulk
    // velocity -- in a boosted frame even a plasma at rest in the lab does -- and the injection front follows it
    void UpdateInjectionPosition(amrex::Real a_dt) {
        const amrex::Real c = 299'792'458.;
        const int dir = moving_window_dir;
        for (int i = 0; i < mypc->nContainers(); ++i) {
            WarpXParticleContainer& pc = mypc->GetParticleContainer(i);
            if (!pc.doContinuousInjection()) continue;
            const amrex::Real u_bulk = pc.BulkMomentum(dir);                         // getBulkMomentum, in units of c
            amrex::Real v_shift = c * u_bulk / std::sqrt(1.0 + u_bulk * u_bulk);
            if (m_ctx.gamma_boost > 1.0) {
                v_shift = (v_shift - c * m_ctx.beta_boost) / (1.0 - v_shift * m_ctx.beta_boost / c);
                v_shift *= (dir == 2) ? 1.0 : 0.0;                                   // boost_direction[dir]
            }
            pc.m_current_injection_position += v_shift * a_dt;
        }
    }

    // WarpX::shiftMF (:478-648), zero external field.  Across the bricks of the window direction the cells that enter a
    // brick come from its upper neighbour: the reference fills the guards of its temporary along the window
    // (FillBoundary(tmpmf, ng_mw, periodicity), :503-513) and shifts over them; here the field's own guards are filled
    // to the shift's depth first and the device routine is told to keep them (WXA_WINDOW_KEEP_GUARDS) -- only the
    // topmost brick, where the domain ends, lets zeros in.
    void shiftMF(amrex::MultiFab& mf, int num_shift, int dir) {
        const wxa_field_view& v = mf.view();
        if (num_shift > v.ng[dir]) throw std::runtime_error("shiftMF: shift exceeds the guard depth");   // :491
        m_shift_tmp.be = m_be;
        m_shift_tmp.reserve(sizeof(double) * (size_t)v.kstride * (size_t)v.n[2]);
        int periodic[3] = {m_comm->periodic(0) ? 1 : 0, m_comm->periodic(1) ? 1 : 0, m_comm->periodic(2) ? 1 : 0};
        if (m_comm->exchanges(dir)) {
            amrex::IntVect ng(0);
            ng[dir] = num_shift;
            m_comm->FillBoundary(mf, ng, false, m_ctx.stream);
            if (m_comm->has_neighbor(dir, 1)) periodic[dir] = WXA_WINDOW_KEEP_GUARDS;
        }
        if (m_be->shift_field_window(&v, static_cast<double*>(m_shift_tmp.p), dir, num_shift, periodic, m_ctx.stream) != 0)
            throw std::runtime_error("shift_field_window failed");
    }

    // :354-455
    void OneStep_nosub(amrex::Real a_cur_time) {
        PushParticlesandDeposit(a_cur_time);                     // :366
        SyncCurrentAndRho();                                     // :373
        // :416-419 EvolveF/G: no-ops
        EvolveB(0.5 * dt[0], DtType::FirstHalf);                 // :421
        // :422 FillBoundaryB(ng_FieldSolver, sync): only when the guard layer was not computed above (PEC walls)
        if (!m_grown_b) FillBoundaryB(guard_cells.ng_FieldSolver, WarpX::sync_nodal_points);
        if (m_overlap) order_streams(m_ctx.stream, 1, m_comm_stream);   // J's guard sum has arrived
        EvolveE(dt[0]);                                          // :426
        // :433 FillBoundaryE(ng_FieldSolver, sync) is not issued: the Yee update of B reads no guard point of E,
        // the copies of a shared nodal plane are computed from bit-identical operands (so the sync changes
        // nothing), and FillBoundaryE/B(ng_FieldGather) refills every guard before the next reader (the gather).
        // Fields bit for bit the same (tests/test_multibrick_cpu.py).  The CKC update of B does read guard points of
        // E (the transverse neighbours of its extended differences): there the exchange is issued.
        if (m_cfg.maxwell_solver == WXA_SOLVER_CKC || safe_guard_cells)
            FillBoundaryE(guard_cells.ng_FieldSolver, WarpX::sync_nodal_points);
        EvolveB(0.5 * dt[0], DtType::SecondHalf);                // :437
        // :447-451 "E and B are up-to-date in the domain, but all guard cells are outdated."
        if (safe_guard_cells) FillBoundaryB(guard_cells.ng_alloc_EB);
    }

    // warpx.safe_guard_cells (Source/WarpX.cpp:625; GuardCellManager.cpp:297-308; WarpXComm.cpp:759,824;
    // WarpXEvolve.cpp:449-451): every FillBoundary exchanges all allocated guard cells, every exchange of the reference's
    // schedule is issued (none replaced by the guard-layer update, none dropped), B's guards are refilled at the end of
    // the step and J's after the guard sum.  A debugging mode: the valid points are the same bit for bit
    // (tests/test_multibrick_cpu.py::test_safe_guard_cells).  Before the first step.
    void SetSafeGuardCells(bool on) {
        if (istep != 0 && on != safe_guard_cells) throw std::runtime_error("warpx.safe_guard_cells: before the first step");
        safe_guard_cells = on;
        guard_cells.Init(dt[0], m_ctx.dx, m_cfg.nox, use_filter, amrex::IntVect(2), m_cfg.use_fdtd_nci_corr != 0,
                         NCIGodfreyFilter::m_stencil_width, safe_guard_cells);
        if (on && m_overlap) throw std::runtime_error("warpx.safe_guard_cells with overlap_halo: the overlapped schedule relies on the "
                                                      "guard-layer update the safe mode switches off");
        if (on) m_grown_b = false;
    }
    // warpx.do_single_precision_comms (Source/WarpX.cpp:614): float on the wire between bricks (BrickComm)
    void SetSinglePrecisionComms(bool on) { m_comm->set_single_precision_comms(on); }
    bool do_single_precision_comms() const { return m_comm->single_precision_comms(); }

    // ---- exchanges of the field solve ---------------------------------------------------------
    // All-periodic runs issue none: the first half update of B is followed by the same update of the first guard layer, from the
    // guards the gather-depth fill of the step start left (E and B have not changed since), so EvolveE finds the
    // guard points it reads without FillBoundaryB -- the neighbour computes the same numbers from the same
    // operands, bit for bit.  With overlap_halo the two field exchanges of a step travel on a second stream:
    //   main:  push of the interior tiles | push of the rest, deposit, filter J | B (valid + guard layer) | E, B'
    //   comm:  FillBoundary E+B (gather)  |                                     | SumBoundaryJ            |
    // (the tiles that touch no face of the brick read no guard point; EvolveE is the first reader of the summed J)
    void SetUpHaloOverlap(bool want) {
        // a wall's boundary kernel owns the guards behind it; WXA_NO_GUARD_LAYER=1 brings the exchange back (debugging)
        // (the guard-layer kernel is the Yee update: with CKC the reference's exchange stays)
        m_grown_b = !m_any_pec && m_be->evolve_b_guard_layer != nullptr && m_cfg.maxwell_solver == WXA_SOLVER_YEE &&
                    !m_env_no_guard_layer;
        m_overlap = false;
        bool any_split = false;
        for (int d = 0; d < 3; ++d) any_split = any_split || !m_comm->self_periodic(d);
        // PushInterior relies on the particles of the interior tiles of the last sort reading no guard point while the
        // fill is in flight: true while the drift since that sort stays below a tile minus the stencil reach, i.e. for
        // short sort intervals only (a particle moves < 1 cell per step)
        if (want && !(m_cfg.sort_interval > 0 && m_cfg.sort_interval <= 4))
            throw std::runtime_error("overlap_halo needs 0 < warpx.sort_intervals <= 4 (interior tiles must stay clear of the guards)");
        // with the NCI corrector the gather reads filtered copies that need the guards along z filled first
        if (!want || !m_grown_b || !any_split || !m_be->stream_create || m_cfg.use_fdtd_nci_corr) return;
        m_comm_stream = m_be->stream_create();
        if (!m_comm_stream) return;
        if (m_be->event_create)
            for (auto& e : m_halo_events) e = m_be->event_create();
        m_overlap = true;
    }
    bool halo_overlap() const { return m_overlap; }
    // waiting_stream continues after what recorded_stream holds now (a backend without streams runs in program order)
    void order_streams(void* waiting_stream, int ev, void* recorded_stream) {
        if (!m_halo_events[ev]) return;
        m_be->event_record(m_halo_events[ev], recorded_stream);
        m_be->stream_wait_event(waiting_stream, m_halo_events[ev]);
    }

    // :1101-1180
    void PushParticlesandDeposit(amrex::Real a_cur_time, bool skip_current = false,
                                 PushType push_type = PushType::Explicit) {
        if (m_eb_fill_in_flight) {
            mypc->PushInterior(m_fields, dt[0]);               // reads no guard point
            order_streams(m_ctx.stream, 3, m_comm_stream);     // the guards of E and B have arrived
            m_eb_fill_in_flight = false;
        }
        mypc->Evolve(m_fields, 0, "current_fp", a_cur_time, dt[0], DtType::Full, skip_current, push_type);
    }

    // :583-652 -> SyncCurrent (WarpXComm.cpp:1073-1240), single level.  The reference loops
    // filter + SumBoundary per component (:1233-1237); here the three components are filtered
    // first and then summed together so that their guard slabs share one message per neighbour.
    void SyncCurrentAndRho() {
        PhaseTimer t(&m_ctx, kSyncCurrent);  // "WarpX::SyncCurrent()"
        using warpx::fields::FieldType;
        auto J = m_fields.get_alldirs(FieldType::current_fp, 0);
        if (use_filter)
            for (int idim = 0; idim < 3; ++idim) ApplyFilterJ(J, 0, idim);
        if (m_overlap) {   // the sum travels on the exchange stream; OneStep_nosub waits for it before EvolveE
            order_streams(m_comm_stream, 0, m_ctx.stream);
            SumBoundaryJ(J, 0, m_comm_stream);
            return;        // no PEC in this mode: nothing to reflect
        }
        SumBoundaryJ(J, 0);
        // :625-640 reflect the current density over PEC boundaries
        ApplyJfieldBoundary(0, J[0], J[1], J[2], PatchType::fine);
    }

    // Source/BoundaryConditions/WarpXFieldBoundaries.cpp:175-188 -> PEC::ApplyReflectiveBoundarytoJfield
    // (WarpX_PEC.cpp:713-900) with absorbing particle boundaries next to the PEC walls
    void ApplyJfieldBoundary(int /*lev*/, amrex::MultiFab* Jx, amrex::MultiFab* Jy, amrex::MultiFab* Jz,
                             PatchType /*patch_type*/) {
        if (!m_pec_here) return;
        const wxa_field_view Jv[3] = {Jx->view(), Jy->view(), Jz->view()};
        if (m_be->apply_pec_j(Jv, m_dom_lo, m_dom_hi, m_pec_lo, m_pec_hi, m_ctx.stream) != 0)
            throw std::runtime_error("apply_pec_j failed");
    }

    // RhoFunctor::operator() (Source/Diagnostics/ComputeDiagFunctors/RhoFunctor.cpp:42-61):
    // MultiParticleContainer::GetChargeDensity (all species and antennas at their current positions; each
    // species' rho is mirrored over the PEC walls right after its deposition, WarpXParticleContainer.cpp:
    // 1285-1290 -- linear, so once on the total) + ApplyFilterandSumBoundaryRho (WarpXComm.cpp:1426-1470).
    // Diagnostics only: the arrays are created at the first call.
    amrex::MultiFab& ComputeRho() {
        if (!m_rho) {
            amrex::IntVect ng_rho;                                                  // GuardCellManager.cpp:62-172
            for (int d = 0; d < 3; ++d)
                ng_rho[d] = m_ctx.nox + 1 + static_cast<int>(std::ceil(299'792'458. * dt[0] / m_ctx.dx[d]));
            m_rho = std::make_unique<amrex::MultiFab>(m_be, m_ctx.brick_box, amrex::IntVect(1), ng_rho);
            if (use_filter) m_rho_tmp = std::make_unique<amrex::MultiFab>(m_be, m_ctx.brick_box, amrex::IntVect(1), ng_rho);
        }
        amrex::MultiFab& rho = *m_rho;
        rho.setVal(0.0, m_ctx.stream);
        for (int i = 0; i < mypc->nContainers(); ++i) mypc->GetParticleContainer(i).DepositCharge(&rho);
        if (m_pec_here) ApplyRhofieldBoundary(rho);
        if (use_filter) {
            check(m_be->filter_bilinear(&rho.view(), &m_rho_tmp->view(), m_ctx.stream), "filter_bilinear");
            rho.swap_storage(*m_rho_tmp);
        }
        m_comm->SumBoundary(rho, rho.nGrowVect(), /*refresh_guards=*/false, m_ctx.stream);
        m_be->stream_sync(m_ctx.stream);
        return rho;
    }
    amrex::MultiFab* rho() { return m_rho.get(); }

    // WarpX::ApplyRhofieldBoundary (WarpXFieldBoundaries.cpp:161-173) -> PEC::ApplyReflectiveBoundarytoRhofield
    // (WarpX_PEC.cpp:628-710) as WarpXParticleContainer::DepositCharge calls it (WarpXParticleContainer.cpp:1276-1283):
    // BEFORE the guard sum, over the VALID points of the box (:697-698).  The charge a particle next to a wall leaves in the
    // guard columns of a wall-free direction (towards a periodic image or a neighbour box) therefore keeps its deposit behind
    // the wall and gets no image: the guard sum then adds those columns to the valid ones as they are.  The backend's
    // apply_pec_rho folds those columns too (what a fold after the sum would give); to give the reference's numbers the
    // columns are saved before it and put back after it.  (Found in round 5 as the origin of the 1.3e-3 by which rho and jz
    // of the reference's back-transformed golden file were missed: in that deck the plasma ends one cell from the periodic
    // faces and streams through the lower wall, profiles/round5/README.md.)  WXA_PEC_RHO_FOLD_GUARD_COLUMNS=1: as before.
    void ApplyRhofieldBoundary(amrex::MultiFab& rho) {
        const wxa_field_view& v = rho.view();
        const bool fold_columns = m_env_fold_rho_guard_columns;   // WXA_PEC_RHO_FOLD_GUARD_COLUMNS, read at construction
        struct Slab { int32_t lo[3], hi[3]; size_t at; };
        std::vector<Slab> slabs;
        size_t total = 0;
        if (!fold_columns) {
            for (int d = 0; d < 3; ++d) {
                if (m_pec_lo[d] || m_pec_hi[d] || v.ng[d] == 0) continue;   // (the backend grows its loop along wall-free directions)
                for (int side = 0; side < 2; ++side) {
                    // (the faces every box layout of the reference has: the domain's.  Behind a face between two bricks of
                    // this run the columns are folded, so that the result does not depend on the brick layout.)
                    if (!m_comm->domain_face(d, side)) continue;
                    Slab sl;
                    for (int e = 0; e < 3; ++e) { sl.lo[e] = v.lo[e]; sl.hi[e] = v.lo[e] + v.n[e]; }
                    if (side == 0) sl.hi[d] = v.lo[d] + v.ng[d];
                    else sl.lo[d] = v.lo[d] + v.n[d] - v.ng[d];
                    sl.at = total;
                    total += (size_t)(sl.hi[0] - sl.lo[0]) * (size_t)(sl.hi[1] - sl.lo[1]) * (size_t)(sl.hi[2] - sl.lo[2]);
                    slabs.push_back(sl);
                }
            }
        }
        if (total) {
            m_pec_rho_keep.be = m_be;
            m_pec_rho_keep.reserve(sizeof(double) * total);
            for (const Slab& sl : slabs)
                check(m_be->pack_box(&v, sl.lo, sl.hi, static_cast<double*>(m_pec_rho_keep.p) + sl.at, m_ctx.stream), "pack_box");
        }
        if (m_be->apply_pec_rho(&v, m_dom_lo, m_dom_hi, m_pec_lo, m_pec_hi, m_ctx.stream) != 0)
            throw std::runtime_error("apply_pec_rho failed");
        for (const Slab& sl : slabs)
            check(m_be->unpack_box(&v, sl.lo, sl.hi, static_cast<const double*>(m_pec_rho_keep.p) + sl.at, /*overwrite*/ 0,
                                   m_ctx.stream), "unpack_box");
    }

    // WarpXComm.cpp:1357-1374: filter into a temporary with the same guards, then "copy back":
    // here the two arrays simply exchange their storage (no copy)
    void ApplyFilterJ(const ablastr::fields::VectorField& current, int /*lev*/, int idim) {
        amrex::MultiFab& J = *current[idim];
        amrex::MultiFab& tmp = *m_filter_tmp[idim];
        check(m_be->filter_bilinear(&J.view(), &tmp.view(), m_ctx.stream), "filter_bilinear");
        J.swap_storage(tmp);
    }

    // WarpXComm.cpp:1386-1424 -> WarpXSumGuardCells (WarpXSumGuardCells.cpp:17-24)
    void SumBoundaryJ(const ablastr::fields::VectorField& current, int lev) { SumBoundaryJ(current, lev, m_ctx.stream); }
    void SumBoundaryJ(const ablastr::fields::VectorField& current, int /*lev*/, void* stream) {
        amrex::IntVect ng_depos_J = guard_cells.ng_depos_J;
        if (use_filter) ng_depos_J = ng_depos_J + amrex::IntVect(2) - amrex::IntVect(1);  // :1413-1416
        ng_depos_J = amrex::min(ng_depos_J, current[0]->nGrowVect());                     // :1417-1420
        m_comm->SumBoundary({current[0], current[1], current[2]}, ng_depos_J, /*refresh_guards=*/safe_guard_cells,
                            stream);
    }

    // Only the step's neighbour exchanges, with the run's real arrays and message sizes and nothing computed in between
    // (bench.py --dry-comm): [0] FillBoundary of E and B with the gather's guard depth (WarpXEvolve.cpp:515-516),
    // [1] SumBoundary of J (WarpXComm.cpp:1386-1424), [2] Redistribute of every species (one count round + the data
    // exchange of the particles that have left since the last call: none on a repeated call), [3] the three together.
    // Milliseconds per call, host clock around a stream sync: what a step pays for communication if nothing overlaps.
    // E and B are unchanged by it (FillBoundary is idempotent); J's guard values are added into its valid cells, which
    // the next step's deposition starts from zero anyway.
    void DryComm(int reps, double ms[4]) {
        using warpx::fields::FieldType;
        reps = std::min(std::max(reps, 1), 64);
        // every sum writes the total back into all images of a point: repeated on the live J the shared points double per
        // call (2^40 after bench.py's 20 + 20 calls) and a plotfile, a BTD slice or a J query taken before the next
        // deposition would read that.  The sums run on J as it is (real sizes, real strides); J is put back afterwards.
        auto J = m_fields.get_alldirs(FieldType::current_fp, 0);
        DeviceBuffer saved[3];
        for (int c = 0; c < 3; ++c) {
            const wxa_field_view v = J[c]->view();
            const size_t bytes = sizeof(double) * (size_t)v.kstride * (size_t)v.n[2];
            saved[c].be = m_be;
            saved[c].reserve(bytes);
            m_be->memcpy_async(saved[c].p, v.p, bytes, m_ctx.stream);
        }
        auto timed = [&](auto&& f) {
            sync_stream();
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) f();
            sync_stream();
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
        };
        auto fill = [&]() { FillBoundaryEB(guard_cells.ng_FieldGather); };
        auto sum = [&]() { SumBoundaryJ(J, 0); };
        auto redist = [&]() { for (int i = 0; i < mypc->nContainers(); ++i) mypc->GetParticleContainer(i).Redistribute(*m_comm); };
        ms[0] = timed(fill);
        ms[1] = timed(sum);
        ms[2] = timed(redist);
        ms[3] = timed([&]() { fill(); sum(); redist(); });
        for (int c = 0; c < 3; ++c) {
            const wxa_field_view v = J[c]->view();
            m_be->memcpy_async(v.p, saved[c].p, sizeof(double) * (size_t)v.kstride * (size_t)v.n[2], m_ctx.stream);
        }
        sync_stream();   // the copies are done before `saved` goes away
    }

    // Source/FieldSolver/WarpXPushFieldsEM.cpp:877-927
    void EvolveB(amrex::Real a_dt, DtType a_dt_type) {
        {
            PhaseTimer t(&m_ctx, kEvolveB);  // "WarpX::EvolveB()"
            m_fdtd_solver_fp->EvolveB(m_fields, 0, PatchType::fine, a_dt);
        }
        if (a_dt_type == DtType::FirstHalf && m_grown_b) {   // instead of the FillBoundaryB that would follow
            PhaseTimer t(&m_ctx, kFillBoundary);
            const int32_t grow[3] = {1, 1, 1};
            m_fdtd_solver_fp->EvolveBGuardLayer(m_fields, 0, a_dt, grow);
        }
        ApplyBfieldBoundary(0, PatchType::fine);                        // :926
    }
    // :930-1011
    void EvolveE(amrex::Real a_dt) {
        PhaseTimer t(&m_ctx, kEvolveE);  // "WarpX::EvolveE()"
        using warpx::fields::FieldType;
        m_fdtd_solver_fp->EvolveE(m_fields, 0, PatchType::fine, m_fields.get_alldirs(FieldType::Efield_fp, 0), a_dt);
        ApplyEfieldBoundary(0, PatchType::fine);                        // :990
    }

    // Source/BoundaryConditions/WarpXFieldBoundaries.cpp:51-106 -> PEC::ApplyPECtoEfield
    // (WarpX_PEC.cpp:457-538) with get_ng_fieldgather(); periodic faces: nothing to do
    void ApplyEfieldBoundary(int /*lev*/, PatchType /*patch_type*/) {
        if (!m_pec_here) return;
        auto E = m_fields.get_alldirs(warpx::fields::FieldType::Efield_fp, 0);
        const wxa_field_view Ev[3] = {E[0]->view(), E[1]->view(), E[2]->view()};
        const int32_t ng[3] = {guard_cells.ng_FieldGather[0], guard_cells.ng_FieldGather[1], guard_cells.ng_FieldGather[2]};
        if (m_be->apply_pec_e(Ev, m_dom_lo, m_dom_hi, m_pec_lo, m_pec_hi, ng, m_ctx.stream) != 0)
            throw std::runtime_error("apply_pec_e failed");
    }
    // :108-135 -> PEC::ApplyPECtoBfield (WarpX_PEC.cpp:540-626)
    void ApplyBfieldBoundary(int /*lev*/, PatchType /*patch_type*/) {
        if (!m_pec_here) return;
        auto B = m_fields.get_alldirs(warpx::fields::FieldType::Bfield_fp, 0);
        const wxa_field_view Bv[3] = {B[0]->view(), B[1]->view(), B[2]->view()};
        const int32_t ng[3] = {guard_cells.ng_FieldGather[0], guard_cells.ng_FieldGather[1], guard_cells.ng_FieldGather[2]};
        if (m_be->apply_pec_b(Bv, m_dom_lo, m_dom_hi, m_pec_lo, m_pec_hi, ng, m_ctx.stream) != 0)
            throw std::runtime_error("apply_pec_b failed");
    }

    // Source/Parallelization/WarpXComm.cpp:644-660,699-827 -> ablastr FillBoundary (Communication.cpp:71-115)
    void FillBoundaryE(const amrex::IntVect& ng, std::optional<bool> nodal_sync = std::nullopt) {
        FillBoundaryVector(warpx::fields::FieldType::Efield_fp, ng, nodal_sync.value_or(false));
    }
    void FillBoundaryB(const amrex::IntVect& ng, std::optional<bool> nodal_sync = std::nullopt) {
        FillBoundaryVector(warpx::fields::FieldType::Bfield_fp, ng, nodal_sync.value_or(false));
    }
    void FillBoundaryAux(const amrex::IntVect& /*ng*/) {}
    // WarpXComm.cpp:387-419: aux aliases fp at level 0 -> nothing to copy
    void UpdateAuxilaryData() {}

    // WarpXEvolve.cpp:473-531
    void ExplicitFillBoundaryEBUpdateAux() {
        using warpx::fields::FieldType;
        if (is_synchronized) {
            FillBoundaryEB(guard_cells.ng_alloc_EB);   // FillBoundaryE + FillBoundaryB (:487-488)
            UpdateAuxilaryData();
            FillBoundaryAux(guard_cells.ng_UpdateAux);
            auto E = m_fields.get_alldirs(FieldType::Efield_aux, 0);
            auto B = m_fields.get_alldirs(FieldType::Bfield_aux, 0);
            mypc->PushP(0, -0.5 * dt[0], *E[0], *E[1], *E[2], *B[0], *B[1], *B[2]);
            is_synchronized = false;
        } else {
            if (m_overlap) {   // on the exchange stream; PushParticlesandDeposit pushes the interior tiles meanwhile
                order_streams(m_comm_stream, 2, m_ctx.stream);
                FillBoundaryEB(guard_cells.ng_FieldGather, m_comm_stream);
                m_eb_fill_in_flight = true;
            } else {
                FillBoundaryEB(guard_cells.ng_FieldGather);   // FillBoundaryE + FillBoundaryB (:515-516)
            }
            UpdateAuxilaryData();
            FillBoundaryAux(guard_cells.ng_UpdateAux);
        }
    }

    // WarpXEvolve.cpp:65-93
    void Synchronize() {
        using warpx::fields::FieldType;
        FillBoundaryEB(guard_cells.ng_FieldGather);   // FillBoundaryE + FillBoundaryB (:68-69)
        UpdateAuxilaryData();
        FillBoundaryAux(guard_cells.ng_UpdateAux);
        auto E = m_fields.get_alldirs(FieldType::Efield_aux, 0);
        auto B = m_fields.get_alldirs(FieldType::Bfield_aux, 0);
        mypc->PushP(0, 0.5 * dt[0], *E[0], *E[1], *E[2], *B[0], *B[1], *B[2]);
        is_synchronized = true;
    }

    // WarpXEvolve.cpp:533-581
    void HandleParticlesAtBoundaries(int step, amrex::Real /*cur_time*/, int num_moved) {
        PhaseTimer t(&m_ctx, kRedistribute);
        mypc->ApplyBoundaryConditions();                                      // :537
        mypc->RedistributeLocal(num_moved + 1, *m_comm);                      // :559
        (void)step;  // :575-580 SortParticlesByBin: done between push and deposition (see Evolve)
    }

    // ---- accessors used by the C API ----
    ablastr::fields::MultiFabRegister& fields() { return m_fields; }
    MultiParticleContainer& GetPartContainer() { return *mypc; }
    WarpXContext& context() { return m_ctx; }
    amrex::Real getdt(int lev) const { return dt[lev]; }
    int64_t getistep() const { return istep; }
    amrex::Real gett_new() const { return cur_time; }
    BrickComm& comm() { return *m_comm; }
    bool any_reflecting_wall() const { return m_any_reflecting_wall; }

    guardCellManager guard_cells;
    bool use_filter = true;          // Source/WarpX.cpp:158
    bool safe_guard_cells = false;   // warpx.safe_guard_cells
    bool is_synchronized = true;     // Source/WarpX.H:1520
    // The reference synchronises the velocities with the positions at the end of the last step of every Evolve call
    // (:222-226).  A caller that advances one run through several calls (a benchmark timing steps in the middle of a run,
    // a driver that looks at fields only) can switch that off and ask for the synchronisation when it needs it: the
    // steps in between are then exactly the steps of one long Evolve call.
    bool synchronize_at_end = true;
    void SynchronizeNow() {
        if (!is_synchronized) Synchronize();
        m_be->stream_sync(m_ctx.stream);
    }
    int sort_intervals = -1;         // Source/WarpX.cpp:1335 (GPU default 4; set by the config)
    // WarpX::field_boundary_lo / field_boundary_hi restricted to periodic | PEC
    int32_t m_pec_lo[3] = {0, 0, 0}, m_pec_hi[3] = {0, 0, 0}, m_dom_lo[3] = {0, 0, 0}, m_dom_hi[3] = {0, 0, 0};
    bool m_any_pec = false;    // the domain has a PEC wall (schedule decisions: identical on every brick)
    bool m_pec_here = false;   // ... and this brick touches one
    bool m_any_reflecting_wall = false;
    // WarpX::do_moving_window, moving_window_dir, moving_window_v (m/s), moving_window_x
    bool do_moving_window = false;
    int moving_window_dir = 2;
    amrex::Real moving_window_v = 0.0, moving_window_x = 0.0;
    DeviceBuffer m_shift_tmp;

private:
    void FillBoundaryVector(warpx::fields::FieldType ft, const amrex::IntVect& ng, bool nodal_sync, void* stream) {
        auto F = m_fields.get_alldirs(ft, 0);
        m_comm->FillBoundary({F[0], F[1], F[2]}, ng, nodal_sync, stream);
    }
    void FillBoundaryVector(warpx::fields::FieldType ft, const amrex::IntVect& ng, bool nodal_sync) {
        PhaseTimer t(&m_ctx, kFillBoundary);
        auto F = m_fields.get_alldirs(ft, 0);
        for (int d = 0; d < 3; ++d)
            if (!ng.allLE(F[d]->nGrowVect()))  // WarpXComm.cpp:755-759
                throw std::runtime_error("Error: in FillBoundary, requested more guard cells than allocated");
        m_comm->FillBoundary({F[0], F[1], F[2]}, ng, nodal_sync, m_ctx.stream);
    }
    // FillBoundaryE(ng) followed by FillBoundaryB(ng): the six components share the messages
    void FillBoundaryEB(const amrex::IntVect& ng) {
        PhaseTimer t(&m_ctx, kFillBoundary);
        FillBoundaryEB(ng, m_ctx.stream);
    }
    void FillBoundaryEB(const amrex::IntVect& ng, void* stream) {
        using warpx::fields::FieldType;
        auto E = m_fields.get_alldirs(FieldType::Efield_fp, 0);
        auto B = m_fields.get_alldirs(FieldType::Bfield_fp, 0);
        for (int d = 0; d < 3; ++d)
            if (!ng.allLE(E[d]->nGrowVect()) || !ng.allLE(B[d]->nGrowVect()))
                throw std::runtime_error("Error: in FillBoundary, requested more guard cells than allocated");
        m_comm->FillBoundary({E[0], E[1], E[2], B[0], B[1], B[2]}, ng, false, stream);
    }

    const Backend* m_be;
    wxa_sim_config m_cfg;
    WarpXContext m_ctx;
    ablastr::fields::MultiFabRegister m_fields;
    std::unique_ptr<MultiParticleContainer> mypc;
    std::unique_ptr<FiniteDifferenceSolver> m_fdtd_solver_fp;
    std::unique_ptr<BrickComm> m_comm;
    std::unique_ptr<NCIGodfreyFilter> nci_godfrey_filter_exeybz, nci_godfrey_filter_bxbyez;   // Source/WarpX.H:468-469
    std::unique_ptr<amrex::MultiFab> m_nci_E[3], m_nci_B[3];
    std::unique_ptr<amrex::MultiFab> m_filter_tmp[3];
    std::unique_ptr<amrex::MultiFab> m_rho, m_rho_tmp;   // ComputeRho (diagnostics)
    DeviceBuffer m_pec_rho_keep;                         // ApplyRhofieldBoundary: the guard columns it leaves as they are
    std::unique_ptr<BTDiagnostics> m_btd;
    bool m_btd_write_species = false;
    bool m_reduced_diags_started = false;
    // field-solve exchanges: guard layer of B computed redundantly; J's guard sum on a second stream
    bool m_grown_b = false, m_overlap = false;
    bool m_env_no_guard_layer = false, m_env_fold_rho_guard_columns = false, m_switches_verified = false;
    void* m_comm_stream = nullptr;
    void* m_halo_events[4] = {nullptr, nullptr, nullptr, nullptr};
    bool m_eb_fill_in_flight = false;
    std::vector<amrex::Real> dt;
    amrex::Real cur_time = 0.0;
    int64_t istep = 0;
};

}  // namespace wxa::host
#endif
